// gridpp::bilinear (src/api/bilinear.cpp) and the box search it rests on (Grid::get_box, src/api/grid.cpp:149-229;
// point_in_rectangle, src/api/util.cpp:561-582) for gfx950.
//
// One thread per output location: the nearest grid point comes from the bin index (gpp_nearest_device), the four
// quadrants around it are tested in the reference's order, the weights (s, t) are solved once per location (they do not
// depend on the time level) and the T levels are gathered with unit-stride writes across locations.  All float
// expressions keep the reference's association (the build has -ffp-contract=off and correctly rounded float divide);
// the quadratic of the general quadrilateral is solved in double exactly as bilinear.cpp:159-267 does.
#include "common.h"

using namespace gpp;

namespace {

__device__ __forceinline__ float edge_side(float plat, float plon, float qlat, float qlon, float mlat, float mlon) {
    const float vlon = qlon - plon;
    const float vlat = -1.0f * (qlat - plat);
    const float c = -1.0f * (vlat * plon + vlon * plat);
    return (vlat * mlon + vlon * mlat) + c;
}
// util.cpp:571-582
__device__ __forceinline__ bool in_rectangle(float alat, float alon, float blat, float blon, float clat, float clon, float dlat,
                                             float dlon, float mlat, float mlon) {
    const float d1 = edge_side(alat, alon, blat, blon, mlat, mlon);
    const float d2 = edge_side(alat, alon, dlat, dlon, mlat, mlon);
    const float d3 = edge_side(blat, blon, clat, clon, mlat, mlon);
    const float d4 = edge_side(clat, clon, dlat, dlon, mlat, mlon);
    const bool cw = 0 >= d1 && 0 >= d4 && 0 <= d2 && 0 >= d3;
    const bool ccw = 0 <= d1 && 0 <= d4 && 0 >= d2 && 0 <= d3;
    return cw || ccw;
}

// grid.cpp:149-229; box = (Y1, X1, Y2, X2), all -1 when the point is in none of the four quadrants
__device__ bool get_box(const float* __restrict__ glat, const float* __restrict__ glon, int nY, int nX, int nn, float lat,
                        float lon, int& Y1, int& X1, int& Y2, int& X2) {
    Y1 = Y2 = X1 = X2 = -1;
    if(nn < 0 || nX <= 1 || nY <= 1) return false;
    const int Y = nn / nX, X = nn - Y * nX;
    const float alat = glat[nn], alon = glon[nn];
    for(int it = 0; it < 4; ++it) {
        const int xdir = (it & 1) ? 1 : -1;
        const int ydir = (it < 2) ? 1 : -1;
        if((Y == 0 && ydir == -1) || (Y == nY - 1 && ydir == 1) || (X == 0 && xdir == -1) || (X == nX - 1 && xdir == 1)) continue;
        const int b = (Y + ydir) * nX + X, c = b + xdir, d = nn + xdir;
        if(in_rectangle(alat, alon, glat[b], glon[b], glat[c], glon[c], glat[d], glon[d], lat, lon)) {
            X1 = xdir == 1 ? X : X - 1;
            X2 = X1 + 1;
            Y1 = ydir == 1 ? Y : Y - 1;
            Y2 = Y1 + 1;
            return true;
        }
    }
    return false;
}

__device__ __forceinline__ bool in_range(float v) {   // bilinear.cpp:154-157
    const float tol = 0.01f;
    return v >= -tol && v < 1 + tol;
}

// bilinear.cpp:159-267
__device__ void weights_general(float x, float y, float x0, float x1, float x2, float x3, float y0, float y1, float y2, float y3,
                                float& t_out, float& s_out) {
    const double a = -x0 + x2, b = -x0 + x1, c = x0 - x1 - x2 + x3, d = x - x0;     // differences formed in float
    const double e = -y0 + y2, f = -y0 + y1, g = y0 - y1 - y2 + y3, h = y - y0;
    double alpha = NAN, beta = NAN;
    const double Y1 = y1, Y2 = y3, Y3 = y0, Y4 = y2, X1 = x1, X2 = x3, X3 = x0, X4 = x2;
    const double X31 = X3 - X1, X21 = X2 - X1, Y42 = Y4 - Y2, Y21 = Y2 - Y1, Y31 = Y3 - Y1, Y43 = Y4 - Y3, X42 = X4 - X2,
                 X43 = X4 - X3;
    const double qa = 2 * c * e - 2 * a * g, qb = 2 * c * f - 2 * b * g;
    const double lin1 = b * e - a * f + d * g - c * h, lin2 = b * e - a * f - d * g + c * h;
    const double root = sqrt(-4 * (c * e - a * g) * (d * f - b * h) + lin1 * lin1);
    if(qa != 0 && qb != 0) {
        alpha = -(lin1 + root) / qa;
        beta = (lin2 + root) / qb;
        if(!in_range((float)alpha)) alpha = -(lin1 - root) / qa;
        if(!in_range((float)beta)) beta = (lin2 - root) / qb;
    }
    else if(qb == 0) {
        alpha = -(lin1 + root) / qa;
        if(!in_range((float)alpha)) alpha = -(lin1 - root) / qa;
        const float s = (float)alpha;
        float t;
        if(Y3 + Y43 * s - Y1 - Y21 * s == 0) t = (float)((x - X1 - X21 * s) / (X3 + X43 * s - X1 - X21 * s));
        else t = (float)((y - Y1 - Y21 * s) / (Y3 + Y43 * s - Y1 - Y21 * s));
        beta = 1 - t;
    }
    else {   // qa == 0
        beta = (lin2 + root) / qb;
        const float t = (float)(1 - beta);
        float s;
        if(Y2 + Y42 * t - Y1 - Y31 * t == 0) s = (float)((x - X1 - X31 * t) / (X2 + X42 * t - X1 - X31 * t));
        else s = (float)((y - Y1 - Y31 * t) / (Y2 + Y42 * t - Y1 - Y31 * t));
        alpha = s;
    }
    s_out = (float)alpha;
    t_out = (float)(1 - beta);
}

// bilinear.cpp:269-313: true when (s, t) end up outside [0, 1] (the reference throws there)
__device__ bool weights(float x, float y, float x0, float x1, float x2, float x3, float y0, float y1, float y2, float y3,
                        float& s, float& t) {
    const float Y1 = y1, Y2 = y3, Y3 = y0, Y4 = y2, X1 = x1, X2 = x3, X3 = x0, X4 = x2;
    const bool vertical = (double)fabsf((X3 - X1) * (Y4 - Y2) - (X4 - X2) * (Y3 - Y1)) <= 1e-4;
    const bool horizontal = (double)fabsf((X2 - X1) * (Y4 - Y3) - (X4 - X3) * (Y2 - Y1)) <= 1e-4;
    if(vertical && horizontal) {   // bilinear.cpp:138-153
        const float A = X2 - X1, B = X3 - X1, C = Y2 - Y1, D = Y3 - Y1;
        const float det = 1 / (A * D - B * C);
        s = det * ((x - X1) * (D) + (y - Y1) * (-B));
        t = det * ((x - X1) * (-C) + (y - Y1) * (A));
    }
    else weights_general(x, y, x0, x1, x2, x3, y0, y1, y2, y3, t, s);
    if(t >= 1 && (double)t <= 1.15) t = 1;
    if(t <= 0 && (double)t >= -0.15) t = 0;
    if(s >= 1 && (double)s <= 1.15) s = 1;
    if(s <= 0 && (double)s >= -0.15) s = 0;
    return !(s >= 0 && s <= 1 && t >= 0 && t <= 1);
}

__device__ __forceinline__ bool dev_valid(float v) { return !isnan(v) && !isinf(v); }

// values [nt][nY*nX], out [nt][nq].  err[0] = 1 when a box is too distorted, err[1..2] = bits of one offending (s, t).
__global__ __launch_bounds__(256) void k_bilinear(const float* __restrict__ glat, const float* __restrict__ glon, int nY, int nX,
                                                  const int* __restrict__ nn, const float* __restrict__ qlat,
                                                  const float* __restrict__ qlon, int nq, const float* __restrict__ values, int nt,
                                                  float* __restrict__ out, int* __restrict__ err) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if(q >= nq) return;
    const int n0 = nn[q];
    const float lat = qlat[q], lon = qlon[q];
    int I1, J1, I2, J2;
    const bool inside = get_box(glat, glon, nY, nX, n0, lat, lon, I1, J1, I2, J2);
    const size_t nG = (size_t)nY * nX;
    const int i0 = I1 * nX + J1, i1 = I2 * nX + J1, i2 = I1 * nX + J2, i3 = I2 * nX + J2;
    float s = 0, t = 0;
    bool solved = false, bad = false;
    for(int k = 0; k < nt; ++k) {
        const float* v = values + (size_t)k * nG;
        float res = NAN;
        bool done = false;
        if(inside) {
            const float v0 = v[i0], v1 = v[i1], v2 = v[i2], v3 = v[i3];
            if(dev_valid(v0) && dev_valid(v1) && dev_valid(v2) && dev_valid(v3)) {
                done = true;
                if(!solved) {
                    bad = weights(lon, lat, glon[i0], glon[i1], glon[i2], glon[i3], glat[i0], glat[i1], glat[i2], glat[i3], s, t);
                    solved = true;
                    if(bad && atomicCAS(&err[0], 0, 1) == 0) { err[1] = __float_as_int(s); err[2] = __float_as_int(t); }
                }
                const float P1 = v1, P2 = v3, P3 = v0, P4 = v2;
                res = P1 * (1 - s) * (1 - t) + P2 * s * (1 - t) + P3 * (1 - s) * t + P4 * s * t;
            }
        }
        if(!done) res = v[n0];   // outside the domain or a missing corner: nearest neighbour (bilinear.cpp:346-356)
        out[(size_t)k * nq + q] = res;
    }
}

__global__ void k_get_box(const float* __restrict__ glat, const float* __restrict__ glon, int nY, int nX, const int* __restrict__ nn,
                          const float* __restrict__ qlat, const float* __restrict__ qlon, int nq, int* __restrict__ box) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if(q >= nq) return;
    int I1, J1, I2, J2;
    const bool inside = get_box(glat, glon, nY, nX, nn[q], qlat[q], qlon[q], I1, J1, I2, J2);
    box[5 * q + 0] = inside; box[5 * q + 1] = I1; box[5 * q + 2] = J1; box[5 * q + 3] = I2; box[5 * q + 4] = J2;
}

__global__ void k_fill_nan(float* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = NAN;
}

__global__ void k_in_rectangle(const float* p, int* out) {
    out[0] = in_rectangle(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9]);
}

}   // namespace

void gpp_points::latlon_to_device() {
    if(latlon_on_device) return;
    ensure_host_fields();
    d_lat.upload(lats.data(), n);
    d_lon.upload(lons.data(), n);
    GPP_HIP(hipStreamSynchronize(stream()));
    latlon_on_device = true;
}

extern "C" int gpp_bilinear(gpp_points* igrid, gpp_points* to, const float* values, int nt, float* out, int mem) {
    GPP_TRY
    ensure_device();
    if(!igrid || !to) invalid("grid / points is NULL");
    if(nt < 0) invalid("negative number of time levels");
    const int nq = to->n;
    if(nq == 0 || nt == 0) return GPP_OK;
    const size_t nout = (size_t)nt * nq;
    OutField o;
    o.bind(out, nout, mem);
    if(igrid->n == 0) {   // bilinear.cpp:37-39
        hipLaunchKernelGGL(k_fill_nan, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, stream(), o.d, nout);
        GPP_HIP(hipGetLastError());
        o.finish();
        GPP_HIP(hipStreamSynchronize(stream()));
        return GPP_OK;
    }
    if(!values) invalid("values is NULL");
    InField v;
    v.bind(values, (size_t)nt * igrid->n, mem);
    to->to_device();
    to->latlon_to_device();
    igrid->latlon_to_device();
    DevBuf<int> idx, err;
    idx.get(nq);
    err.get(4);
    GPP_HIP(hipMemsetAsync(err.p, 0, 4 * sizeof(int), stream()));
    gpp_nearest_device(igrid, to->d_x.p, to->d_y.p, to->d_z.p, nq, 1, idx.p);
    hipLaunchKernelGGL(k_bilinear, dim3((nq + 255) / 256), dim3(256), 0, stream(), igrid->d_lat.p, igrid->d_lon.p, igrid->ny, igrid->nx,
                       idx.p, to->d_lat.p, to->d_lon.p, nq, v.d, nt, o.d, err.p);
    GPP_HIP(hipGetLastError());
    int herr[4];
    GPP_HIP(hipMemcpyAsync(herr, err.p, sizeof(herr), hipMemcpyDeviceToHost, stream()));
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    if(herr[0]) {   // bilinear.cpp:309-313
        float s, t;
        memcpy(&s, &herr[1], 4); memcpy(&t, &herr[2], 4);
        char msg[256];
        snprintf(msg, sizeof msg, "Problem with bilinear interpolation. Grid is rotated/distorted in a way that is not supported. "
                 "s=%g and t=%g are outside [-0.05,1.05].", s, t);
        runtime(msg);
    }
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_grid_get_box(gpp_points* grid, const float* qlats, const float* qlons, int nq, int* inside, int* boxes) {
    GPP_TRY
    ensure_device();
    if(!grid) invalid("grid is NULL");
    if(nq < 0) invalid("nq < 0");
    if(nq == 0) return GPP_OK;
    if(grid->n == 0) {
        for(int i = 0; i < nq; i++) { inside[i] = 0; for(int k = 0; k < 4; k++) boxes[4 * i + k] = -1; }
        return GPP_OK;
    }
    std::vector<float> qx(nq), qy(nq), qz(nq);
    if(gpp_convert_coordinates(qlats, qlons, nq, grid->type, qx.data(), qy.data(), qz.data()) != GPP_OK) return GPP_EINVAL;
    DevBuf<float> dx, dy, dz, dlat, dlon;
    DevBuf<int> idx, box;
    dx.upload(qx.data(), nq); dy.upload(qy.data(), nq); dz.upload(qz.data(), nq);
    dlat.upload(qlats, nq); dlon.upload(qlons, nq);
    idx.get(nq); box.get((size_t)5 * nq);
    grid->latlon_to_device();
    gpp_nearest_device(grid, dx.p, dy.p, dz.p, nq, 1, idx.p);
    hipLaunchKernelGGL(k_get_box, dim3((nq + 255) / 256), dim3(256), 0, stream(), grid->d_lat.p, grid->d_lon.p, grid->ny, grid->nx, idx.p,
                       dlat.p, dlon.p, nq, box.p);
    GPP_HIP(hipGetLastError());
    std::vector<int> h((size_t)5 * nq);
    GPP_HIP(hipMemcpyAsync(h.data(), box.p, sizeof(int) * h.size(), hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    for(int i = 0; i < nq; i++) { inside[i] = h[5 * i]; for(int k = 0; k < 4; k++) boxes[4 * i + k] = h[5 * i + 1 + k]; }
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_point_in_rectangle(const float corners_latlon[8], float lat, float lon, int* inside) {
    GPP_TRY
    ensure_device();
    if(!corners_latlon || !inside) invalid("NULL argument");
    float h[10];
    memcpy(h, corners_latlon, 8 * sizeof(float));
    h[8] = lat; h[9] = lon;
    DevBuf<float> d;
    DevBuf<int> r;
    d.upload(h, 10);
    r.get(1);
    hipLaunchKernelGGL(k_in_rectangle, dim3(1), dim3(1), 0, stream(), d.p, r.p);
    GPP_HIP(hipGetLastError());
    GPP_HIP(hipMemcpyAsync(inside, r.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}
