// Grid operations next to the hot path, on the device:
//   fill / fill_missing            src/api/fill.cpp:6-134
//   doping_square / doping_circle  src/api/doping.cpp:5-93
//   neighbourhood_search           src/api/neighbourhood_search.cpp:7-113
//   calc_gradient                  src/api/calc_gradient.cpp:7-126
//
// The scatter operations (fill, doping) are inverted from the reference's "for every point, for every grid cell it
// reaches" into race-free kernels: one thread per point walks the grid's bin index inside its radius (or its index
// window) and records, per cell, the HIGHEST point index that reaches it (atomicMax) -- the reference's sequential loop
// lets later points overwrite earlier ones, so the highest index is exactly its result; a second pass applies the
// winners.  The window operations (neighbourhood_search, calc_gradient MinMax) are one thread per cell walking its
// window in the reference's row-major order, because their selection rules depend on that order.  calc_gradient's
// LinearRegression form is five box means (the neighbourhood kernels) between two elementwise passes.
#include "common.h"
#include "oi_common.h"
#include <algorithm>

#pragma clang fp contract(off)
using namespace gpp;

namespace {

__device__ __forceinline__ bool gv(float v) { return !isnan(v) && !isinf(v); }

struct GridIx {
    const float4* sgeo;
    const float2* smeta;
    const int* bin_start;
    int axis_a, axis_b, nbx, nby;
    float amin, bmin, inv_s;
};
GridIx grid_ix(gpp_obs_index* ix) {
    return GridIx{ix->d_sgeo.p, ix->d_smeta.p, ix->d_bin_start.p, ix->axis_a, ix->axis_b, ix->nbx, ix->nby, ix->amin, ix->bmin, ix->inv_s};
}
__device__ __forceinline__ int bin_at(float v, float lo, float inv_s, int nb) {
    return (int)fminf(fmaxf(floorf((v - lo) * inv_s), 0.0f), (float)(nb - 1));
}
// every grid cell the reference's Grid::get_neighbours(lat, lon, radius) returns (kdtree.cpp:39-60,241-260): f(cell, elev)
template <class F>
__device__ __forceinline__ void cells_in_radius(const GridIx& ix, float x, float y, float z, float radius, F f) {
    if(!(radius > 0)) return;
    const float lox = x - radius, hix = x + radius, loy = y - radius, hiy = y + radius, loz = z - radius, hiz = z + radius;
    const float alo = ix.axis_a == 0 ? lox : (ix.axis_a == 1 ? loy : loz), ahi = ix.axis_a == 0 ? hix : (ix.axis_a == 1 ? hiy : hiz);
    const float blo = ix.axis_b == 1 ? loy : (ix.axis_b == 2 ? loz : lox), bhi = ix.axis_b == 1 ? hiy : (ix.axis_b == 2 ? hiz : hix);
    const int bx0 = bin_at(alo, ix.amin, ix.inv_s, ix.nbx), bx1 = bin_at(ahi, ix.amin, ix.inv_s, ix.nbx);
    const int by0 = bin_at(blo, ix.bmin, ix.inv_s, ix.nby), by1 = bin_at(bhi, ix.bmin, ix.inv_s, ix.nby);
    for(int row = by0; row <= by1; ++row) {
        const int js = ix.bin_start[row * ix.nbx + bx0], je = ix.bin_start[row * ix.nbx + bx1 + 1];
        for(int j = js; j < je; ++j) {
            const float4 g = ix.sgeo[j];
            if(!(g.x > lox && g.x < hix && g.y > loy && g.y < hiy && g.z > loz && g.z < hiz)) continue;
            const float dx = g.x - x, dy = g.y - y, dz = g.z - z;
            if(!(sqrtf(dx * dx + dy * dy + dz * dz) <= radius)) continue;
            f(__float_as_int(ix.smeta[j].y), g.w);
        }
    }
}

__global__ void k_set_int(int* p, size_t n, int v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) p[i] = v;
}

// winner = max(winner, i).  Many points reach the same cell (the reference benchmark puts 100 000 of them on one diagonal), and
// same-address atomics serialise in the L2; the winner only grows, so a plain look first -- a stale value is a smaller one, it
// can only cause a redundant atomic, never a wrong skip -- lets all but the first few contenders of a cell leave it alone.
__device__ __forceinline__ void raise_winner(int* w, int i) {
    if(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < i) atomicMax(w, i);
}
// winner[cell] = highest point index whose circle reaches the cell (and, for doping, passes the elevation test)
__global__ __launch_bounds__(256) void k_circle_winners(GridIx ix, const float* __restrict__ px, const float* __restrict__ py,
                                                        const float* __restrict__ pz, const float* __restrict__ pelev,
                                                        const float* __restrict__ radii, int np, int check_elev, float max_elev_diff,
                                                        int* __restrict__ winner) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= np) return;
    const int i = np - 1 - t;   // high indices first: once a cell holds a high winner the lower ones only read it (see raise_winner)
    const float e = check_elev ? pelev[i] : 0.0f;
    cells_in_radius(ix, px[i], py[i], pz[i], radii[i], [&](int cell, float gelev) {
        if(check_elev && fabsf(e - gelev) > max_elev_diff) return;   // doping.cpp:83-87 (a NaN difference does not skip)
        raise_winner(&winner[cell], i);
    });
}
// doping.cpp:32-45: index window around the nearest grid point of every observation
__global__ __launch_bounds__(256) void k_square_winners(const int* __restrict__ nn, const float* __restrict__ pelev,
                                                        const int* __restrict__ halfwidth, int np, const float* __restrict__ gelev, int Y, int X,
                                                        int check_elev, float max_elev_diff, int* __restrict__ winner) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= np) return;
    const int i = np - 1 - t;
    const int iy = nn[i] / X, ix = nn[i] - iy * X, hw = halfwidth[i];
    const float e = check_elev ? pelev[i] : 0.0f;
    for(int yy = max(0, iy - hw); yy <= min(Y - 1, iy + hw); ++yy)
        for(int xx = max(0, ix - hw); xx <= min(X - 1, ix + hw); ++xx) {
            if(check_elev && fabsf(e - gelev[yy * X + xx]) > max_elev_diff) continue;
            raise_winner(&winner[yy * X + xx], i);
        }
}
// mode 0: doping (winner's observation, else background); 1: fill inside (value where reached); 2: fill outside (input where reached)
__global__ void k_apply_winners(const int* __restrict__ winner, const float* __restrict__ field, const float* __restrict__ obs, float value, int mode,
                                size_t n, float* __restrict__ out) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(c >= n) return;
    const int w = winner[c];
    float r;
    if(mode == 0) r = w >= 0 ? obs[w] : field[c];
    else if(mode == 1) r = w >= 0 ? value : field[c];
    else r = w >= 0 ? field[c] : value;
    out[c] = r;
}

// fill.cpp:43-134: one thread per row (dir 0) or column (dir 1), the reference's sequential scan with its last / next cursors
__global__ void k_fill_missing_lines(const float* __restrict__ v, int Y, int X, int dir, float* __restrict__ res) {
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
    const int nlines = dir == 0 ? Y : X, len = dir == 0 ? X : Y;
    if(line >= nlines) return;
    const size_t base = dir == 0 ? (size_t)line * X : (size_t)line, step = dir == 0 ? 1 : (size_t)X;
    int last = 0, next = -1;
    for(int k = 0; k < len; ++k) {
        const float curr = v[base + k * step];
        float r = NAN;
        if(!gv(curr)) {
            if(next < k) for(next = k; next < len; ++next) if(gv(v[base + next * step])) break;
            if(next < len) {
                const float vl = v[base + last * step], vn = v[base + next * step];
                r = (vl) + (vn - vl) * (float)(k - last) / (float)(next - last);
            }
        }
        else { last = k; r = curr; }
        res[base + k * step] = r;
    }
}
// The same scan, one 256-thread workgroup per row held in LDS: every thread owns a contiguous segment, the last valid index
// before a segment and the first valid index after it come from two 256-entry block scans, then the segment is walked
// backwards (next valid index of every element) and forwards (the reference's formula).  Rows up to FM_MAXX elements.
#define FM_MAXX 8192
__global__ __launch_bounds__(256) void k_fill_missing_rows(const float* __restrict__ v, int Y, int X, float* __restrict__ res) {
    __shared__ float s_v[FM_MAXX];
    __shared__ int s_next[FM_MAXX];
    __shared__ int s_last[256], s_first[256];
    const int y = blockIdx.x, tid = threadIdx.x;
    const float* row = v + (size_t)y * X;
    for(int i = tid; i < X; i += 256) s_v[i] = row[i];
    __syncthreads();
    const int seg = (X + 255) / 256, i0 = tid * seg, i1 = min(i0 + seg, X);
    int sl = -1, sf = 0x7fffffff;
    for(int i = i0; i < i1; ++i) if(gv(s_v[i])) { sl = i; if(sf == 0x7fffffff) sf = i; }
    s_last[tid] = sl; s_first[tid] = sf;
    __syncthreads();
    // exclusive prefix max of s_last, exclusive suffix min of s_first (256 entries: one short loop per thread)
    int before = -1, after = 0x7fffffff;
    for(int t = 0; t < tid; ++t) before = max(before, s_last[t]);
    for(int t = tid + 1; t < 256; ++t) after = min(after, s_first[t]);
    int nx = after;
    for(int i = i1 - 1; i >= i0; --i) { if(gv(s_v[i])) nx = i; s_next[i] = nx; }
    int last = before < 0 ? 0 : before;   // fill.cpp:49,86: `last` starts at 0 whether or not that element is valid
    float* orow = res + (size_t)y * X;
    for(int i = i0; i < i1; ++i) {
        const float curr = s_v[i];
        float r = NAN;
        if(!gv(curr)) {
            const int next = s_next[i];
            if(next < X) {
                const float vl = s_v[last], vn = s_v[next];
                r = (vl) + (vn - vl) * (float)(i - last) / (float)(next - last);
            }
        }
        else { last = i; r = curr; }
        orow[i] = r;
    }
}
// out[x][y] = in[y][x], 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ in, int Y, int X, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    for(int j = ty; j < 32; j += 8) { const int y = y0 + j, x = x0 + tx; if(y < Y && x < X) tile[j][tx] = in[(size_t)y * X + x]; }
    __syncthreads();
    for(int j = ty; j < 32; j += 8) { const int x = x0 + j, y = y0 + tx; if(y < Y && x < X) out[(size_t)x * Y + y] = tile[tx][j]; }
}
__global__ void k_fill_missing_merge(const float* __restrict__ ry, const float* __restrict__ rx, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    int count = 0;
    float total = 0;
    if(gv(ry[i])) { total += ry[i]; count++; }
    if(gv(rx[i])) { total += rx[i]; count++; }
    out[i] = count > 0 ? total / (float)count : NAN;
}

// neighbourhood_search.cpp:33-109 for one cell
__global__ __launch_bounds__(256) void k_neighbourhood_search(const float* __restrict__ array, const float* __restrict__ search, int nY, int nX,
                                                              int halfwidth, float tmin, float tmax, float delta, const int* __restrict__ apply,
                                                              float* __restrict__ out) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(c >= (long)nY * nX) return;
    const int y = (int)(c / nX), x = (int)(c - (long)y * nX);
    const float here = search[c];
    if(!gv(here) || (apply && apply[c] == 0)) { out[c] = array[c]; return; }
    const bool active = !apply || apply[c] == 1;
    float nearest_target = NAN, accum = 0;
    long inear = 0;
    int counter = 0;
    if(active)
        for(int yy = max(0, y - halfwidth); yy <= min(nY - 1, y + halfwidth); ++yy)
            for(int xx = max(0, x - halfwidth); xx <= min(nX - 1, x + halfwidth); ++xx) {
                const long n = (long)yy * nX + xx;
                const float s = search[n], a = array[n];
                if(!gv(s) || !gv(a)) continue;
                if(s >= tmin && s <= tmax) { counter++; accum = accum + a; }
                else if(counter > 0) continue;
                else if(fabsf(s - here) >= delta) {
                    if(!gv(nearest_target)) { nearest_target = s; inear = n; }
                    else {
                        const float cur = fminf(fabsf(s - tmin), fabsf(s - tmax));
                        const float best = fminf(fabsf(nearest_target - tmin), fabsf(nearest_target - tmax));
                        if(cur < best) { nearest_target = s; inear = n; }
                    }
                }
            }
    float r;
    if(counter > 0) r = accum / (float)counter;
    else if(gv(nearest_target)) r = array[inear];
    else r = array[c];
    out[c] = r;
}

// calc_gradient.cpp:26-74 (MinMax) for one cell
__global__ __launch_bounds__(256) void k_gradient_minmax(const float* __restrict__ base, const float* __restrict__ values, int nY, int nX, int halfwidth,
                                                         int num_min, float min_range, float default_gradient, float* __restrict__ out) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(c >= (long)nY * nX) return;
    const int y = (int)(c / nX), x = (int)(c - (long)y * nX);
    float cmax = NAN, cmin = NAN;
    long imax = 0, imin = 0;
    int count = 0;
    for(int yy = max(0, y - halfwidth); yy <= min(nY - 1, y + halfwidth); ++yy)
        for(int xx = max(0, x - halfwidth); xx <= min(nX - 1, x + halfwidth); ++xx) {
            const long n = (long)yy * nX + xx;
            const float b = base[n];
            if(!gv(b) || !gv(values[n])) continue;
            if(!gv(cmax) || b > cmax) { cmax = b; imax = n; }
            if(!gv(cmin) || b < cmin) { cmin = b; imin = n; }
            count++;
        }
    float r = default_gradient;
    if(!(count < num_min || !gv(cmax) || !gv(cmin) || fabsf(cmax - cmin) <= min_range)) r = (values[imax] - values[imin]) / (cmax - cmin);
    out[c] = r;
}
// calc_gradient.cpp:79-98: the four moment fields and the validity mask
__global__ void k_gradient_moments(const float* __restrict__ base, const float* __restrict__ values, size_t n, float* __restrict__ b0,
                                   float* __restrict__ v0, float* __restrict__ bb, float* __restrict__ bv, float* __restrict__ ok) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    const float b = base[i], v = values[i];
    const bool good = gv(b) && gv(v);
    b0[i] = good ? b : NAN;
    v0[i] = good ? v : NAN;
    bb[i] = good ? b * b : NAN;
    bv[i] = good ? b * v : NAN;
    ok[i] = good ? 1.0f : 0.0f;
}
// calc_gradient.cpp:107-124
__global__ void k_gradient_regression(const float* __restrict__ mX, const float* __restrict__ mY, const float* __restrict__ mXX,
                                      const float* __restrict__ mXY, const float* __restrict__ cnt, size_t n, int num_min, float min_range,
                                      float default_gradient, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    float r = default_gradient;
    const float var = mXX[i] - mX[i] * mX[i];
    if(cnt[i] >= (float)num_min && gv(mXX[i]) && gv(mXY[i]) && gv(mX[i]) && var != 0) {
        bool valid_range = true;
        if(gv(min_range)) {
            const float range = sqrtf(var);
            if(!gv(range) || range < min_range) valid_range = false;
        }
        if(valid_range) r = (mXY[i] - mX[i] * mY[i]) / var;
    }
    out[i] = r;
}

inline unsigned blocks(size_t n) { return (unsigned)((n + 255) / 256); }

void reset_winners(DevBuf<int>& winner, size_t n) {
    winner.get(n);
    hipLaunchKernelGGL(k_set_int, dim3(blocks(n)), dim3(256), 0, stream(), winner.p, n, -1);
}

}   // namespace

extern "C" int gpp_fill(gpp_points* igrid, const float* input, gpp_points* points, const float* radii, float value, int outside, float* out, int mem) {
    GPP_TRY
    if(!igrid || !points) invalid("grid / points is NULL");
    const int np = points->n;
    if(np > 0 && !radii) invalid("radii is NULL");
    for(int i = 0; i < np; i++) if(radii[i] < 0) invalid("All radius sizes must be 0 or greater");   // fill.cpp:11-14 (radii: host array)
    const size_t n = (size_t)igrid->n;
    if(n == 0) return GPP_OK;
    if(!input || !out) invalid("input / out is NULL");
    ensure_device();
    InField in; OutField o;
    in.bind(input, n, mem);
    o.bind(out, n, mem);
    DevBuf<int> winner;
    reset_winners(winner, n);
    if(np > 0) {
        points->to_device();
        DevBuf<float> drad;
        drad.upload(radii, np);
        gpp_obs_index* ix = gpp_build_obs_index(igrid);
        hipLaunchKernelGGL(k_circle_winners, dim3(blocks(np)), dim3(256), 0, stream(), grid_ix(ix), points->d_x.p, points->d_y.p, points->d_z.p,
                           (const float*)nullptr, drad.p, np, 0, 0.0f, winner.p);
        GPP_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_apply_winners, dim3(blocks(n)), dim3(256), 0, stream(), winner.p, in.d, (const float*)nullptr, value, outside ? 2 : 1, n, o.d);
        GPP_HIP(hipGetLastError());
        o.finish();
        GPP_HIP(hipStreamSynchronize(stream()));
        return GPP_OK;
    }
    hipLaunchKernelGGL(k_apply_winners, dim3(blocks(n)), dim3(256), 0, stream(), winner.p, in.d, (const float*)nullptr, value, outside ? 2 : 1, n, o.d);
    GPP_HIP(hipGetLastError());
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_fill_missing(const float* values, int ny, int nx, float* out, int mem) {
    GPP_TRY
    if(ny < 0 || nx < 0) invalid("negative size");
    const size_t n = (size_t)ny * nx;
    if(n == 0) return GPP_OK;
    if(!values || !out) invalid("values / out is NULL");
    ensure_device();
    InField in; OutField o;
    in.bind(values, n, mem);
    o.bind(out, n, mem);
    DevBuf<float> ry, rx;
    ry.get(n); rx.get(n);
    if(nx <= FM_MAXX && ny <= FM_MAXX && !path_env("GPP_FILL_MISSING_LINES")) {
        // rows directly; columns as the rows of the transposed field
        DevBuf<float> vt, rxt;
        vt.get(n); rxt.get(n);
        hipLaunchKernelGGL(k_fill_missing_rows, dim3(ny), dim3(256), 0, stream(), in.d, ny, nx, ry.p);
        hipLaunchKernelGGL(k_transpose, dim3((nx + 31) / 32, (ny + 31) / 32), dim3(256), 0, stream(), in.d, ny, nx, vt.p);
        hipLaunchKernelGGL(k_fill_missing_rows, dim3(nx), dim3(256), 0, stream(), vt.p, nx, ny, rxt.p);
        hipLaunchKernelGGL(k_transpose, dim3((ny + 31) / 32, (nx + 31) / 32), dim3(256), 0, stream(), rxt.p, nx, ny, rx.p);
        hipLaunchKernelGGL(k_fill_missing_merge, dim3(blocks(n)), dim3(256), 0, stream(), ry.p, rx.p, n, o.d);
        GPP_HIP(hipGetLastError());
        o.finish();
        GPP_HIP(hipStreamSynchronize(stream()));   // vt / rxt die here
        return GPP_OK;
    }
    hipLaunchKernelGGL(k_fill_missing_lines, dim3(blocks(ny)), dim3(256), 0, stream(), in.d, ny, nx, 0, ry.p);
    hipLaunchKernelGGL(k_fill_missing_lines, dim3(blocks(nx)), dim3(256), 0, stream(), in.d, ny, nx, 1, rx.p);
    hipLaunchKernelGGL(k_fill_missing_merge, dim3(blocks(n)), dim3(256), 0, stream(), ry.p, rx.p, n, o.d);
    GPP_HIP(hipGetLastError());
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

// halfwidth != NULL: doping_square; radii != NULL: doping_circle (both host arrays of one entry per observation)
extern "C" int gpp_doping(gpp_points* igrid, const float* background, gpp_points* points, const float* observations, const int* halfwidth,
                          const float* radii, float max_elev_diff, float* out, int mem) {
    GPP_TRY
    if(!igrid || !points) invalid("grid / points is NULL");
    if((halfwidth == nullptr) == (radii == nullptr)) invalid("exactly one of halfwidth / radii must be given");
    if(is_valid(max_elev_diff) && max_elev_diff < 0) invalid("max_elev_diff must be greater than or equal to 0");   // doping.cpp:12-13
    const int np = points->n;
    for(int i = 0; i < np; i++) {
        if(halfwidth && halfwidth[i] < 0) invalid("All halfwidth must be greater than or equal to 0");
        if(radii && radii[i] < 0) invalid("radii must be greater than or equal to 0");
    }
    const size_t n = (size_t)igrid->n;
    if(n == 0) return GPP_OK;
    if(!background || !out || (np > 0 && !observations)) invalid("background / observations / out is NULL");
    ensure_device();
    InField bg, obs; OutField o;
    bg.bind(background, n, mem);
    obs.bind(observations, np, mem);
    o.bind(out, n, mem);
    DevBuf<int> winner, nn, dhw;
    DevBuf<float> drad;
    reset_winners(winner, n);
    const int check = is_valid(max_elev_diff) ? 1 : 0;
    if(np > 0) {
        points->to_device();
        igrid->to_device();
        if(radii) {
            drad.upload(radii, np);
            gpp_obs_index* ix = gpp_build_obs_index(igrid);
            hipLaunchKernelGGL(k_circle_winners, dim3(blocks(np)), dim3(256), 0, stream(), grid_ix(ix), points->d_x.p, points->d_y.p, points->d_z.p,
                               points->d_elev.p, drad.p, np, check, max_elev_diff, winner.p);
        }
        else {
            dhw.upload(halfwidth, np);
            nn.get(np);
            gpp_nearest_device(igrid, points->d_x.p, points->d_y.p, points->d_z.p, np, 1, nn.p);
            hipLaunchKernelGGL(k_square_winners, dim3(blocks(np)), dim3(256), 0, stream(), nn.p, points->d_elev.p, dhw.p, np, igrid->d_elev.p,
                               igrid->ny, igrid->nx, check, max_elev_diff, winner.p);
        }
        GPP_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_apply_winners, dim3(blocks(n)), dim3(256), 0, stream(), winner.p, bg.d, obs.d, 0.0f, 0, n, o.d);
    GPP_HIP(hipGetLastError());
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_neighbourhood_search(const float* array, const float* search_array, int ny, int nx, int halfwidth, float search_target_min,
                                        float search_target_max, float search_delta, const int* apply_array, float* out, int mem) {
    GPP_TRY
    if(search_target_min > search_target_max) invalid("Search_target_min must be smaller than search_target_max");   // :10-12
    if(halfwidth < 0) invalid("halfwidth must be positive");
    if(ny < 0 || nx < 0) invalid("negative size");
    const size_t n = (size_t)ny * nx;
    if(n == 0) return GPP_OK;
    if(!array || !search_array || !out) invalid("array / search_array / out is NULL");
    ensure_device();
    InField a, s; OutField o;
    a.bind(array, n, mem);
    s.bind(search_array, n, mem);
    o.bind(out, n, mem);
    DevBuf<int> dapply;
    const int* ap = nullptr;
    if(apply_array) {
        if(mem & GPP_MEM_DEVICE) ap = apply_array;
        else { dapply.upload(apply_array, n); ap = dapply.p; }
    }
    hipLaunchKernelGGL(k_neighbourhood_search, dim3(blocks(n)), dim3(256), 0, stream(), a.d, s.d, ny, nx, halfwidth, search_target_min,
                       search_target_max, search_delta, ap, o.d);
    GPP_HIP(hipGetLastError());
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_calc_gradient(const float* base, const float* values, int ny, int nx, int gradient_type, int halfwidth, int num_min,
                                 float min_range, float default_gradient, float* out, int mem) {
    GPP_TRY
    if(halfwidth <= 0) invalid("Halwidth cannot be <= 0; must be positive integer");   // calc_gradient.cpp:10-17
    if(is_valid(min_range) && min_range < 0) invalid("min_range must be >= 0");
    if(num_min < 0) invalid("num_min must be >= 0");
    if(ny <= 0) invalid("base input has no size");
    if(gradient_type != GPP_GRADIENT_MINMAX && gradient_type != GPP_GRADIENT_LINEAR_REGRESSION) invalid("unknown gradient type");
    const size_t n = (size_t)ny * nx;
    if(n == 0) return GPP_OK;
    if(!base || !values || !out) invalid("base / values / out is NULL");
    ensure_device();
    InField b, v; OutField o;
    b.bind(base, n, mem);
    v.bind(values, n, mem);
    o.bind(out, n, mem);
    if(gradient_type == GPP_GRADIENT_MINMAX) {
        hipLaunchKernelGGL(k_gradient_minmax, dim3(blocks(n)), dim3(256), 0, stream(), b.d, v.d, ny, nx, halfwidth, num_min, min_range, default_gradient, o.d);
        GPP_HIP(hipGetLastError());
    }
    else {
        DevBuf<float> buf;
        buf.get(n * 10);
        float *b0 = buf.p, *v0 = b0 + n, *bb = b0 + 2 * n, *bv = b0 + 3 * n, *ok = b0 + 4 * n;
        float *mX = b0 + 5 * n, *mY = b0 + 6 * n, *mXX = b0 + 7 * n, *mXY = b0 + 8 * n, *cnt = b0 + 9 * n;
        hipLaunchKernelGGL(k_gradient_moments, dim3(blocks(n)), dim3(256), 0, stream(), b.d, v.d, n, b0, v0, bb, bv, ok);
        GPP_HIP(hipGetLastError());
        const float* src[5] = {b0, v0, bb, bv, ok};
        float* dst[5] = {mX, mY, mXX, mXY, cnt};
        for(int k = 0; k < 5; k++)   // calc_gradient.cpp:100-105: box means through the neighbourhood kernels
            if(gpp_neighbourhood(src[k], ny, nx, 1, 0, halfwidth, k == 4 ? GPP_SUM : GPP_MEAN, dst[k], GPP_MEM_DEVICE) != GPP_OK) return GPP_ERUNTIME;
        hipLaunchKernelGGL(k_gradient_regression, dim3(blocks(n)), dim3(256), 0, stream(), mX, mY, mXX, mXY, cnt, n, num_min, min_range, default_gradient, o.d);
        GPP_HIP(hipGetLastError());
        o.finish();
        GPP_HIP(hipStreamSynchronize(stream()));
        return GPP_OK;
    }
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}
