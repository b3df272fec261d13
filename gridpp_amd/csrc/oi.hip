// Optimal interpolation on MI355X (gfx950): hand-written HIP kernels + the C-ABI entry point.
//
// Replaces the OpenMP loop of gridpp::optimal_interpolation_full (src/api/oi.cpp:221-338):
//   radius query (src/api/kdtree.cpp:39-60,247-260) -> Barnes rho (src/api/structure.cpp:26-34,
//   185-230) -> filter -> top-max_points by rho (oi.cpp:251-273) -> K = G (P+R)^-1 in double
//   (oi.cpp:289-316) -> increment, optional clamp (oi.cpp:317-335), analysis variance (:336-337).
//
// Mapping (wave64, one tile of 64 grid cells per wavefront, 4 tiles per workgroup):
//   * the observation set is bin-sorted once per Points object on two projected axes (gpp_obs_index); a tile walks only
//     the bins that can still matter, records are fetched 64 at a time and broadcast with v_readlane;
//   * k_oi_union (oi_union.h): ONE shared factorisation per tile -- the cells of a tile select almost the same
//     observations -- plus a per-lane finish; tried first for every symmetric system with max_points <= 32 (32-column
//     form) or 33..62 (64-column form, oi_union64.hip);
//   * k_oi (this file): every lane keeps its best max_points candidates as 64-bit keys (rho bits << 32 | ~obs index)
//     in LDS; lanes with the same selection share one augmented Cholesky (or pivoted LU) on rows-in-lanes; also the
//     kernel for what k_oi_union declines, for max_points = 0 / > 62, non-symmetric and spatially varying structures;
//   * k_oi_big (this file): one workgroup per grid point with more than 62 usable observations.
// Arithmetic follows the reference: float32 coordinates/distances/rho (no FMA contraction,
// correctly rounded sqrt/div, rho through a double-precision exp), double for the solve.
#include "oi_common.h"
#include "oi_union.h"

void gpp_launch_union64(const OiArgs& a, unsigned nblocks, bool plain, bool list, hipStream_t stream);   // oi_union64.hip
void gpp_launch_union48(const OiArgs& a, unsigned nblocks, bool plain, bool list, hipStream_t stream);   // oi_union48.hip
#include <rocprim/rocprim.hpp>
#include <cfloat>
#include <algorithm>
#include <memory>

#pragma clang fp contract(off)

using namespace gpp;

void gpp_free_obs_index(gpp_obs_index* p) { delete p; }

// ---- the same index built on the device (large point sets: a 4000 x 4000 grid as the source of `nearest`) ----------
__global__ void k_ix_bins(const float* __restrict__ a, const float* __restrict__ b, int n, float amin, float bmin, float inv_s, int nbx, int nby,
                          int* __restrict__ bin, int* __restrict__ iota) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    int bx = (int)floorf((a[i] - amin) * inv_s), by = (int)floorf((b[i] - bmin) * inv_s);
    bx = max(0, min(nbx - 1, bx)); by = max(0, min(nby - 1, by));
    bin[i] = by * nbx + bx;
    iota[i] = i;
}
__global__ void k_ix_starts(const int* __restrict__ sbin, int n, int nbins, int* __restrict__ start) {   // start[b] = first sorted position with bin >= b
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if(b > nbins) return;
    int lo = 0, hi = n;
    while(lo < hi) { const int mid = (lo + hi) >> 1; if(sbin[mid] < b) lo = mid + 1; else hi = mid; }
    start[b] = lo;
}
__global__ void k_ix_gather(const int* __restrict__ order, int n, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                            const float* __restrict__ elev, const float* __restrict__ laf, float4* __restrict__ sgeo, float2* __restrict__ smeta,
                            int* __restrict__ pos, float4* __restrict__ ogeo, float* __restrict__ olaf) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if(p >= n) return;
    const int o = order[p];
    sgeo[p] = make_float4(x[o], y[o], z[o], elev[o]);
    smeta[p] = make_float2(laf[o], __int_as_float(o));
    pos[o] = p;
    ogeo[p] = make_float4(x[p], y[p], z[p], elev[p]);   // (original order: p doubles as the original index here)
    olaf[p] = laf[p];
}
static gpp_obs_index* build_obs_index_device(gpp_points* pts) {
    std::unique_ptr<gpp_obs_index> ix(new gpp_obs_index);
    const int S = pts->n;
    ix->S = S;
    pts->to_device();
    const float* dax[3] = {pts->d_x.p, pts->d_y.p, pts->d_z.p};
    // extents of the three axes
    DevBuf<float> d_mm;
    DevBuf<char> tmp;
    d_mm.get(6);
    size_t tb = 0;
    GPP_HIP(rocprim::reduce((void*)nullptr, tb, dax[0], d_mm.p, FLT_MAX, (size_t)S, rocprim::minimum<float>(), stream()));
    tmp.get(tb);
    for(int d = 0; d < 3; d++) {
        GPP_HIP(rocprim::reduce((void*)tmp.p, tb, dax[d], d_mm.p + 2 * d, FLT_MAX, (size_t)S, rocprim::minimum<float>(), stream()));
        GPP_HIP(rocprim::reduce((void*)tmp.p, tb, dax[d], d_mm.p + 2 * d + 1, -FLT_MAX, (size_t)S, rocprim::maximum<float>(), stream()));
    }
    float mm[6];
    GPP_HIP(hipMemcpyAsync(mm, d_mm.p, sizeof(mm), hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    float lo[3] = {mm[0], mm[2], mm[4]}, hi[3] = {mm[1], mm[3], mm[5]};
    int order3[3] = {0, 1, 2};
    std::sort(order3, order3 + 3, [&](int p, int q) { return (hi[p] - lo[p]) > (hi[q] - lo[q]); });
    const int a = std::min(order3[0], order3[1]), b = std::max(order3[0], order3[1]);
    ix->axis_a = a; ix->axis_b = b;
    const double ea = std::max((double)hi[a] - lo[a], 1e-3), eb = std::max((double)hi[b] - lo[b], 1e-3);
    double s = std::sqrt(4.0 * ea * eb / std::max(S, 1));   // ~4 points per bin
    s = std::max(s, std::max(ea, eb) / 2048.0);
    const int nbx = std::max(1, std::min(2048, (int)std::ceil(ea / s))), nby = std::max(1, std::min(2048, (int)std::ceil(eb / s)));
    ix->amin = lo[a]; ix->bmin = lo[b]; ix->inv_s = (float)(1.0 / s);
    ix->nbx = nbx; ix->nby = nby;
    const int nbins = nbx * nby;
    // stable sort of the point indices by bin (index order inside a bin, like the host build)
    DevBuf<int> bin, sbin, iota, order;
    bin.get(S); sbin.get(S); iota.get(S); order.get(S);
    hipLaunchKernelGGL(k_ix_bins, dim3((S + 255) / 256), dim3(256), 0, stream(), dax[a], dax[b], S, ix->amin, ix->bmin, ix->inv_s, nbx, nby, bin.p, iota.p);
    GPP_HIP(hipGetLastError());
    int bits = 1;
    while((1ll << bits) < nbins) bits++;
    size_t sb = 0;
    GPP_HIP(rocprim::radix_sort_pairs((void*)nullptr, sb, bin.p, sbin.p, iota.p, order.p, (size_t)S, 0u, (unsigned)bits, stream()));
    DevBuf<char> stmp;
    stmp.get(sb);
    GPP_HIP(rocprim::radix_sort_pairs((void*)stmp.p, sb, bin.p, sbin.p, iota.p, order.p, (size_t)S, 0u, (unsigned)bits, stream()));
    ix->d_bin_start.get(nbins + 1);
    hipLaunchKernelGGL(k_ix_starts, dim3((nbins + 1 + 255) / 256), dim3(256), 0, stream(), sbin.p, S, nbins, ix->d_bin_start.p);
    ix->d_sgeo.get(S); ix->d_smeta.get(S); ix->d_pos.get(S); ix->d_ogeo.get(S); ix->d_olaf.get(S);
    hipLaunchKernelGGL(k_ix_gather, dim3((S + 255) / 256), dim3(256), 0, stream(), order.p, S, pts->d_x.p, pts->d_y.p, pts->d_z.p, pts->d_elev.p, pts->d_laf.p,
                       ix->d_sgeo.p, ix->d_smeta.p, ix->d_pos.p, ix->d_ogeo.p, ix->d_olaf.p);
    GPP_HIP(hipGetLastError());
    GPP_HIP(hipStreamSynchronize(stream()));
    pts->obs_index = ix.release();
    return pts->obs_index;
}

gpp_obs_index* gpp_build_obs_index(gpp_points* pts) {
    if(pts->obs_index) return pts->obs_index;
    if(pts->n >= (1 << 17) && !path_env("GPP_HOST_INDEX")) return build_obs_index_device(pts);
    pts->ensure_host_xyz();
    pts->ensure_host_fields();
    std::unique_ptr<gpp_obs_index> ix(new gpp_obs_index);
    int S = pts->n;
    ix->S = S;
    const std::vector<float>* ax[3] = {&pts->x, &pts->y, &pts->z};
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for(int d = 0; d < 3; d++) {
        for(int i = 0; i < S; i++) {
            float v = (*ax[d])[i];
            if(i == 0 || v < lo[d]) lo[d] = v;
            if(i == 0 || v > hi[d]) hi[d] = v;
        }
    }
    // project on the two axes with the largest extent (correctness never depends on the bins:
    // every candidate is re-tested exactly; the bins only bound the work)
    int order3[3] = {0, 1, 2};
    std::sort(order3, order3 + 3, [&](int p, int q) { return (hi[p] - lo[p]) > (hi[q] - lo[q]); });
    int a = std::min(order3[0], order3[1]), b = std::max(order3[0], order3[1]);
    ix->axis_a = a; ix->axis_b = b;
    double ea = std::max((double)hi[a] - lo[a], 1e-3), eb = std::max((double)hi[b] - lo[b], 1e-3);
    double s = std::sqrt(4.0 * ea * eb / std::max(S, 1));   // ~4 observations per bin
    s = std::max(s, std::max(ea, eb) / 2048.0);
    int nbx = std::max(1, std::min(2048, (int)std::ceil(ea / s)));
    int nby = std::max(1, std::min(2048, (int)std::ceil(eb / s)));
    ix->amin = lo[a]; ix->bmin = lo[b]; ix->inv_s = (float)(1.0 / s);
    ix->nbx = nbx; ix->nby = nby;
    std::vector<int> bin(S), start(nbx * nby + 1, 0), order(S), pos(S);
    for(int i = 0; i < S; i++) {
        int bx = (int)std::floor(((*ax[a])[i] - ix->amin) * ix->inv_s);
        int by = (int)std::floor(((*ax[b])[i] - ix->bmin) * ix->inv_s);
        bx = std::max(0, std::min(nbx - 1, bx));
        by = std::max(0, std::min(nby - 1, by));
        bin[i] = by * nbx + bx;
        start[bin[i] + 1]++;
    }
    for(int k = 0; k < nbx * nby; k++) start[k + 1] += start[k];
    std::vector<int> cur(start.begin(), start.end() - 1);
    for(int i = 0; i < S; i++) { int p = cur[bin[i]]++; order[p] = i; pos[i] = p; }   // stable: index order inside a bin
    std::vector<float4> sgeo(S), ogeo(S);
    std::vector<float2> smeta(S);
    for(int p = 0; p < S; p++) {
        int o = order[p];
        sgeo[p] = make_float4(pts->x[o], pts->y[o], pts->z[o], pts->elevs[o]);
        float of; memcpy(&of, &o, sizeof(float));
        smeta[p] = make_float2(pts->lafs[o], of);
    }
    for(int o = 0; o < S; o++) ogeo[o] = make_float4(pts->x[o], pts->y[o], pts->z[o], pts->elevs[o]);
    ix->d_bin_start.upload(start.data(), start.size());
    ix->d_pos.upload(pos.data(), S);
    ix->d_sgeo.upload(sgeo.data(), S);
    ix->d_smeta.upload(smeta.data(), S);
    ix->d_ogeo.upload(ogeo.data(), S);
    ix->d_olaf.upload(pts->lafs.data(), S);
    GPP_HIP(hipStreamSynchronize(stream()));
    pts->obs_index = ix.release();
    return pts->obs_index;
}

int gpp_tile_wshift(gpp_points* g) {
    if(path_env("GPP_TILE_WSHIFT")) return std::max(0, std::min(6, atoi(path_env("GPP_TILE_WSHIFT"))));
    if(g->ny < 2 || g->nx < 2) return g->nx >= 64 ? 6 : 3;
    if(g->tile_wshift >= 0) return g->tile_wshift;
    // metric size of a cell from the three corner points (0,0), (0,1), (1,0), in the library's own coordinates
    const float la[3] = {g->lat_at(0), g->lat_at(1), g->lat_at(g->nx)}, lo[3] = {g->lon_at(0), g->lon_at(1), g->lon_at(g->nx)};
    float x[3], y[3], z[3];
    if(gpp_convert_coordinates(la, lo, 3, g->type, x, y, z) != GPP_OK) return 3;
    auto dist = [&](int i) { return std::sqrt((double)(x[i] - x[0]) * (x[i] - x[0]) + (double)(y[i] - y[0]) * (y[i] - y[0]) + (double)(z[i] - z[0]) * (z[i] - z[0])); };
    const double dx = std::max(dist(1), 1e-9), dy = std::max(dist(2), 1e-9);
    int best = 3; double bext = 1e300;
    for(int w = 0; w <= 6; w++) {
        if((1 << w) > 2 * g->nx && w > 0) break;           // no point in tiles much wider than the grid
        if((64 >> w) > 2 * g->ny && w < 6) continue;
        const double ext = std::max((1 << w) * dx, (64 >> w) * dy);
        if(ext < bext * 0.999) { bext = ext; best = w; }
    }
    g->tile_wshift = best;
    return best;
}

__global__ void k_structure_corr(DevStructure st, float4 p1, float l1, float4 p2, float l2, int background, float* out) {
    out[0] = d_corr(st, p1.x, p1.y, p1.z, p1.w, l1, p2.x, p2.y, p2.z, p2.w, l2, background != 0);
}

// -------------------------------------------------------------------------------------------
// the OI kernel
// -------------------------------------------------------------------------------------------
// 64-bit mixers for the order-independent signature of a selected observation set
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
// Two single-member groups at once, one per half-wave (lanes 0-31 / 32-63): rows 0..n-1 of (P+R), row 30 = obs -
// background, row 31 = G of the cell; broadcasts stay inside a half (ds_bpermute instead of v_readlane).  This is
// the common case when every cell has its own observation set (e.g. elevation-dependent structure functions).
// Shared by k_oi (selections in its LDS key area) and k_oi_pairs (selections parked in HBM by k_oi); la / lb: the lanes of the
// two cells (lb < 0: none, the upper half idles), na / nb their observation counts (<= 30), sel(i, lane of the cell): the i-th
// selected observation (original index); colbuf: 31 rows of column staging, colL: 64 doubles, res: the wave's result rows.
template <bool PLAIN, class Sel>
__device__ __forceinline__ void oi_solve_pair(const OiArgs& a, const int lane, const int la, const int lb, const int na, const int nb, const Sel sel,
                                              float (*colbuf)[64], double* const colL, float (*res)[64], const float cx, const float cy,
                                              const float cz, const float ce, const float cl, const float cbg, const float cbv, bool& bad) {   // c*: the half's cell
    const int base = lane & 32, hl = lane & 31;
    const bool live = !(base && lb < 0);                 // (an odd cell out: the second half idles)
    const int lh = (base && lb >= 0) ? lb : la;          // this half's cell
    const int nh = base ? nb : na, nmax = max(na, nb);
    const unsigned orig_i = (hl < nh) ? sel(hl, lh) : 0u;
    float4 o0 = make_float4(0, 0, 0, NAN), o1 = make_float4(NAN, 0, 0, 0);
    if(hl < nh) { o0 = a.ogeo[orig_i]; o1 = a.oaux[orig_i]; }
    const bool is_g = hl == 31;
    float maxInc = -INFINITY, minInc = INFINITY;
    if(!a.allow_extrap) {   // extremes of obs - background over the half's selection (oi.cpp:318-334)
        const float dpf = (float)((double)o1.y - (double)o1.z);
        maxInc = hl < nh ? dpf : -INFINITY; minInc = hl < nh ? dpf : INFINITY;
        for(int off = 16; off > 0; off >>= 1) { maxInc = fmaxf(maxInc, __shfl_xor(maxInc, off)); minInc = fminf(minInc, __shfl_xor(minInc, off)); }
    }
    // lower triangle of P (oi.cpp:304-312) and the G row (oi.cpp:250) of each half, one entry per lane and pass: entry
    // e = i (i + 1) / 2 + p (p <= i) for e < ntri, then G entry p = e - ntri; the records of both points come by ds_bpermute
    // (PLAIN: the three Barnes factors of corr(p, p) are exp(-0) = 1 exactly, so the diagonal is written, not evaluated, and the
    //  walk covers the strict lower triangle: row i + 1 of it has the i + 1 entries of row i of the full one)
    constexpr int DG = PLAIN ? 1 : 0;
    const int ntri = nh * (nh + 1 - 2 * DG) / 2, nent = ntri + nh, nentmax = nmax * (nmax + 1 - 2 * DG) / 2 + nmax;
    if(DG && hl < nh) colbuf[hl][base + hl] = 1.0f;
    for(int e0 = 0; e0 < nentmax; e0 += 32) {
        const int e = e0 + hl;
        const bool gent = e >= ntri;
        const unsigned ip = d_tritab()[min(e, 511)];
        int i = (int)(ip >> 8) + DG, pc = (int)(ip & 255u);
        if(gent) { pc = min(e - ntri, 29); i = 31; }
        const int si = base + min(i, 29), sp = base + pc;
        float xi = __shfl(o0.x, si), yi = __shfl(o0.y, si), zi = __shfl(o0.z, si), ei = __shfl(o0.w, si), li = __shfl(o1.x, si);
        const float xp = __shfl(o0.x, sp), yp = __shfl(o0.y, sp), zp = __shfl(o0.z, sp), ep = __shfl(o0.w, sp), lp = __shfl(o1.x, sp);
        xi = gent ? cx : xi; yi = gent ? cy : yi; zi = gent ? cz : zi; ei = gent ? ce : ei; li = gent ? cl : li;
        const float c = d_corr_t<PLAIN, true>(a.s.st, xi, yi, zi, ei, li, xp, yp, zp, ep, lp, gent);
        if(e < nent) colbuf[pc][base + i] = c;
    }
    // obs and background at the observations for the obs - background row (lane 30 of the half): obs rides in column
    // position 30 of every matrix column, the background in row 30 of the staging area
    if(hl < nh) { colbuf[hl][base + 30] = o1.y; colbuf[30][base + hl] = o1.z; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double row[30];
    const bool used = hl < nh || hl >= 30;
#pragma unroll
    for(int p = 0; p < 30; ++p) {
        double v = 0.0;
        if(p < nmax) {
            v = (double)colbuf[p][lane];
            if(hl == p) v += (double)o1.w;                                       // lP + lR
            if(hl == 30) v = v - (double)colbuf[30][base + p];                   // lObs - lY
            if(!used || p >= nh || (hl < p)) v = 0.0;                            // (only the lower triangle is read)
        }
        row[p] = v;
    }
    __builtin_amdgcn_wave_barrier();
    // right-looking Cholesky; column j goes to LDS once (s_col) and comes back as broadcast reads inside the half
    #pragma unroll
    for(int j = 0; j < 30; ++j) {
        if(j < nmax) {
            // (the diagonal of each half from its lane j by v_readlane -- j is a constant here --: through the LDS crossbar the
            //  column started with a round trip of its own)
            const double ajj0 = readlane_d(row[j], j), ajj1 = readlane_d(row[j], 32 + j);
            const double ajj = base ? ajj1 : ajj0;
            const bool colok = j < nh;
            if(colok && !(ajj > 0.0)) bad = true;
            double rs = __builtin_amdgcn_rsq(colok ? ajj : 1.0);
            const double aj = colok ? ajj : 1.0;
            rs = rs * (1.5 - 0.5 * aj * rs * rs);
            rs = rs * (1.5 - 0.5 * aj * rs * rs);
            const double cj = colok ? row[j] * rs : 0.0;
            row[j] = cj;
            colL[lane] = cj;
#pragma unroll
            for(int p = j + 1; p < 30; ++p) row[p] = __builtin_fma(-cj, colL[base + p], row[p]);
        }
    }
    double inc = 0.0, a00 = 0.0;
#pragma unroll
    for(int p = 0; p < 30; ++p) {
        if(hl == 30) colL[base + p] = row[p];           // L^-1 (obs - background)
    }
    // (one lane writes, the others read: without the fences the compiler orders the two sides per thread -- readers first)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for(int p = 0; p < 30; ++p) {
        const double tp = colL[base + p];
        inc = __builtin_fma(row[p], tp, inc);
        a00 = __builtin_fma(row[p], row[p], a00);
    }
    if(is_g) {
        float increment = (float)inc;
        if(!a.allow_extrap) {
            if(maxInc > 0 && increment > maxInc) increment = maxInc;
            else if(maxInc < 0 && increment > 0) increment = maxInc;
            else if(minInc < 0 && increment < minInc) increment = minInc;
            else if(minInc > 0 && increment < 0) increment = minInc;
        }
        if(live) res[0][lh] = cbg + increment;
        if(live) res[1][lh] = (float)((double)cbv * (1.0 - a00));
    }
    __builtin_amdgcn_wave_barrier();
}

template <int N, bool LU, bool PLAIN, bool SPATIAL>
// (the 62-row tile holds its rows in 124 + 124 registers: one wave per SIMD is what it gets, and what it asks for)
__global__ __launch_bounds__(256, (N <= 32 ? 2 : 1)) void k_oi(OiArgs a) {
    __shared__ unsigned long long s_keys[4][N][64];
    __shared__ float s_res[4][2][64];
    __shared__ double s_col[4][64];   // column staging of the half-wave solves
    d_exptab_fill();                                              // 2^(j/128) for d_exp_core (every structure function of this kernel goes through it)
    if constexpr(!LU && !SPATIAL && N == 32) d_tritab_fill();     // triangle index -> (row, column) for oi_solve_pair
    __syncthreads();
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // all tiles (one per wave), or -- behind k_oi_union -- a fixed-size grid striding over the list of what that kernel
    // declined (the list length is read on the device: no host round trip between the kernels).  A list entry >= 0 is
    // tile * 32 + code (code 0..15: a 4-cell item, 16..19: a 16-cell item); an entry < 0 is ~tile, a whole tile.
    const int nrun = a.in_list ? *a.in_count : a.nrun;
    for(int trun = blockIdx.x * 4 + wid; trun < nrun; trun += gridDim.x * 4) {
    int tile = trun, sub = -1, shift = 0;
    if(a.in_list) {
        const int entry = a.in_list[trun];
        if(entry < 0) tile = ~entry;
        else {
            tile = entry >> 5;
            const int code = entry & 31;
            if(code < 16) { sub = code; shift = 2; } else { sub = code - 16; shift = 4; }
        }
    }

    int cell = -1;
    if(a.tiled2d) {
        int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        int y = ty * (64 >> a.wshift) + (lane >> a.wshift), x = (tx << a.wshift) + (lane & ((1 << a.wshift) - 1));
        if(y < a.ny && x < a.nx) cell = y * a.nx + x;
    }
    else {
        int c = tile * 64 + lane;
        if(c < a.C) cell = c;
    }
    if(sub >= 0 && (lane >> shift) != sub) cell = -1;
    float gx = 0, gy = 0, gz = 0, ge = NAN, gl = NAN, bg = NAN, bvar = 1.0f;
    if(cell >= 0) {
        gx = a.gx[cell]; gy = a.gy[cell]; gz = a.gz[cell]; ge = a.gelev[cell]; gl = a.glaf[cell];
        bg = a.bg[cell];
        if(a.bvar) bvar = a.bvar[cell];
    }
    const bool active = cell >= 0 && d_valid(bg);   // oi.cpp:223
    float res_out = bg, res_var = bvar;              // oi.cpp:198-199
    bool parked = false;                             // this cell's selection went to k_oi_pairs
    unsigned long long (*keys)[64] = s_keys[wid];

    int cnt = 0;
    if(wave_ballot(active) != 0ull) {
        const int K = a.s.K;
        bool overflow, truncated;
        DevStructure cst = a.s.st;   // this lane's structure: uniform, or the parameters at its grid point
        if(SPATIAL && cell >= 0) d_structure_at(cst, cst.cell_idx ? cst.cell_idx[cell] : cell);
        cnt = scan_tile<N, false, PLAIN, PLAIN && !LU && !SPATIAL, true>(a.s, cst, active, gx, gy, gz, ge, gl, keys, lane, overflow, truncated);
        if(wave_ballot(overflow) != 0ull) {   // more usable observations than the register tile holds: left to k_oi_big
            if(a.big_list) { if(overflow) a.big_list[atomicAdd(a.big_count, 1)] = cell; }
            else if(lane == 0) atomicOr(a.err, ERR_OVERFLOW);
            cnt = overflow ? 0 : cnt;
        }

        // ---- order-independent signature of every lane's selected set; the 64-bit keys are compacted in place
        //      to a u32 observation-index list (lower 8 KB), freeing the upper half for column staging ----------
        unsigned long long h1 = 0, h2s = 0;
        unsigned (*origs)[64] = reinterpret_cast<unsigned (*)[64]>(&keys[0][0]);          // [N][64] u32
        float (*colbuf)[64] = reinterpret_cast<float (*)[64]>(&keys[N / 2][0]);           // [N][64] f32
        for(int s = 0; s < K; ++s) {
            const unsigned long long kk = keys[s][lane];     // all lanes read slot s ...
            const unsigned o = ~(unsigned)(kk & 0xffffffffull);
            __builtin_amdgcn_wave_barrier();
            origs[s][lane] = o;                              // ... before anything of slots <= s/2 is overwritten
            if(s < cnt) {
                h1 += mix64((unsigned long long)o + 0x9e3779b97f4a7c15ull);
                h2s ^= mix64(((unsigned long long)o << 1) ^ 0xd6e8feb86659fd93ull);
            }
        }

        // ---- dense solves: one augmented Cholesky per DISTINCT observation set -----------------------------------
        unsigned long long todo = wave_ballot(cnt > 0);
        if(GPP_DBG(a, 1)) todo = 0ull;   // GPP_OI_DEBUG bit0: skip the solves (timing experiments only)
        const int nupd = __popcll(todo);
        int nsolve = 0;
        bool bad = false;
        constexpr int MEMB = 63 - N;   // lanes N+1..63 carry one member cell each
        // one group = one distinct observation set: leader lane l (n observations), member cells `members` (nm of them)
        // `extra`: further member cells of the group beyond the G rows that fit beside the matrix (62-row tile: one) -- they
        // reuse the factor through a forward substitution on rows-in-lanes instead of a factorisation of their own
        auto solve_group = [&](const int l, const int n, const unsigned long long members, const int nm, const unsigned long long extra) {
            nsolve++;
            // (the lane number as an opaque value: the 62 masks `lane == p` of the row loads below are otherwise invariants of the tile
            //  loop that the compiler computes once per kernel and keeps -- 124 scalar registers spilled into lanes of vector registers,
            //  which pushed the 62-row forms into the accumulation registers)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            // lane i < n takes the i-th selected observation of the leader; lane N+1+m takes member cell m
            const unsigned orig_i = (lane < n) ? origs[lane][l] : 0u;
            float4 o0 = make_float4(0, 0, 0, NAN), o1 = make_float4(NAN, 0, 0, 0);
            if(lane < n) { o0 = a.ogeo[orig_i]; o1 = a.oaux[orig_i]; }
            const int mi = lane - (N + 1);
            const bool is_g = mi >= 0 && mi < nm;
            const int src = is_g ? nth_set_bit(members, mi) : lane;
            // own point of this lane: observation i (matrix rows) or the member cell (G rows)
            float px = __shfl(gx, src), py = __shfl(gy, src), pz = __shfl(gz, src), pe = __shfl(ge, src), pl = __shfl(gl, src);
            const float cbg = __shfl(bg, src), cbv = __shfl(bvar, src);
            if(lane < n) { px = o0.x; py = o0.y; pz = o0.z; pe = o0.w; pl = o1.x; }
            DevStructure pst = a.s.st;   // structure parameters of this lane's own point (p1 of corr)
            if(SPATIAL) {
                pst.h = __shfl(cst.h, src); pst.v = __shfl(cst.v, src); pst.w = __shfl(cst.w, src); pst.R = __shfl(cst.R, src);
                if(lane < n) d_structure_at(pst, pst.obs_idx[orig_i]);
            }
            float maxInc = -INFINITY, minInc = INFINITY;
            // column p of [P ; G] across the lanes (rolled: one copy of the exp code), staged through LDS
            for(int p = 0; p < n; ++p) {
                const float xp = readlane_f(o0.x, p), yp = readlane_f(o0.y, p), zp = readlane_f(o0.z, p);
                const float ep = readlane_f(o0.w, p), lp = readlane_f(o1.x, p);
                // matrix rows: corr(obs_i, obs_p) (oi.cpp:304-312); G rows: corr(cell, obs_p) (oi.cpp:250)
                const float c = d_corr_t<PLAIN, true>(pst, px, py, pz, pe, pl, xp, yp, zp, ep, lp, is_g);
                colbuf[p][lane] = c;
                const float dpf = (float)((double)readlane_f(o1.y, p) - (double)readlane_f(o1.z, p));
                maxInc = fmaxf(maxInc, dpf); minInc = fminf(minInc, dpf);
            }
            if constexpr(LU) {
                // General (possibly non-symmetric or indefinite) system: the reference inverts P+R with LAPACK's pivoted
                // LU (oi.cpp:315).  K = G (P+R)^-1  <=>  (P+R)^T k = g, so lane i holds row i of A^T = column i of A,
                // the factorisation (partial pivoting, multipliers stored in place) is done once per observation set
                // and every member cell costs one forward + one backward substitution.
                // Without a variance output only the increments G (P+R)^-1 d are needed: then lane i holds row i of A = P+R itself,
                // ONE pair of substitutions gives z = A^-1 d for the whole group and every member cell is left with the dot product
                // g . z in its own lane (instead of a pair of substitutions per member cell: 600 instructions each).
                const bool zsolve = a.out_var == nullptr;
                double rowT[N];
                // A[lane][p] (this lane computed it for observation p) or A[p][lane]: two loops with compile-time strides -- with a
                // run-time stride the 62 LDS addresses of the large tile were loop invariants that the compiler kept in (accumulation)
                // registers for the whole kernel
                if(zsolve) {
#pragma unroll
                    for(int p = 0; p < N; ++p) {
                        double v = 0.0;
                        if(p < n && lane < n) {
                            v = (double)colbuf[p][lane];
                            if(ln == p) v += (double)o1.w;
                        }
                        rowT[p] = v;
                    }
                }
                else {
                    const float* const mine = &colbuf[0][0] + 64 * (lane < N ? lane : 0);
#pragma unroll
                    for(int p = 0; p < N; ++p) {
                        double v = 0.0;
                        if(p < n && lane < n) {
                            v = (double)mine[p];
                            if(ln == p) v += (double)o1.w;
                        }
                        rowT[p] = v;
                    }
                }
                const double dmine = (double)o1.y - (double)o1.z;         // d of observation `lane`
                int mystep = 64;                                           // column this row became the pivot of
                double mypinv = 0.0;                                       // 1 / pivot of that column
#pragma unroll
                for(int j = 0; j < N; ++j) {
                    if(j < n) {
                        // the largest |entry| of column j among the rows not yet used (lowest lane on a tie): a DPP maximum and a ballot
                        // (as a butterfly of ds_bpermute pairs the search was 18 trips through the LDS crossbar per column)
                        const double av = (lane < n && mystep == 64) ? fabs(rowT[j]) : -1.0;
                        const double amax = wave_max_d(av);
                        const int piv = (int)__builtin_ctzll(wave_ballot(av == amax) | (1ull << 63));
                        if(!(amax > 0.0)) bad = true;                      // exactly singular (arma::inv throws)
                        const double pjj = readlane_d(rowT[j], piv);
                        const bool elim = lane < n && mystep == 64 && lane != piv;
                        // one reciprocal per column (LAPACK's dgetf2 scales by it too): v_rcp_f64 and two Newton steps instead of the ~30
                        // instructions of an IEEE division (as the Cholesky path does with v_rsq_f64)
                        double rpj = __builtin_amdgcn_rcp(pjj);
                        rpj = __builtin_fma(rpj, __builtin_fma(-pjj, rpj, 1.0), rpj);
                        rpj = __builtin_fma(rpj, __builtin_fma(-pjj, rpj, 1.0), rpj);
                        const double f = elim ? rowT[j] * rpj : 0.0;
                        if(lane == piv) { mystep = j; mypinv = rpj; }
#pragma unroll
                        for(int p = j + 1; p < N; ++p) {
                            const double pp = readlane_d(rowT[p], piv);
                            rowT[p] = __builtin_fma(-f, pp, rowT[p]);
                        }
                        if(elim) rowT[j] = f;                              // multiplier in place of the eliminated entry
                    }
                }
                if(zsolve) {
                    double bvec = dmine, incv = 0.0;
#pragma unroll
                    for(int j = 0; j < N; ++j) {                           // forward: L y = (permuted) d
                        if(j < n) {
                            const int piv = __builtin_ctzll(wave_ballot(mystep == j));
                            const double bp = readlane_d(bvec, piv);
                            if(lane < n && mystep > j) bvec = __builtin_fma(-rowT[j], bp, bvec);
                        }
                    }
#pragma unroll
                    for(int j = N - 1; j >= 0; --j) {                      // backward: U z = y; every member cell adds g_j z_j
                        if(j < n) {
                            const int piv = __builtin_ctzll(wave_ballot(mystep == j));
                            const double zj = readlane_d(bvec, piv) * readlane_d(mypinv, piv);
                            if(lane < n && mystep < j) bvec = __builtin_fma(-rowT[j], zj, bvec);
                            incv = __builtin_fma(zj, (double)colbuf[j][lane], incv);   // corr_background(cell, obs j) * z_j   (oi.cpp:296,316)
                            if(extra != 0ull && lane == 0) s_col[wid][j] = zj;
                        }
                    }
                    if(is_g) {
                        float increment = (float)incv;
                        if(!a.allow_extrap) {
                            if(maxInc > 0 && increment > maxInc) increment = maxInc;
                            else if(maxInc < 0 && increment > 0) increment = maxInc;
                            else if(minInc < 0 && increment < minInc) increment = minInc;
                            else if(minInc > 0 && increment < 0) increment = minInc;
                        }
                        s_res[wid][0][src] = cbg + increment;
                        s_res[wid][1][src] = cbv;
                    }
                    if(extra != 0ull) {
                        // the cells of the group beyond the G rows that fit beside the matrix: z is theirs as well -- every such cell, in its
                        // own lane, adds corr_background(itself, obs j) z_j over the n observations (one pass for all of them instead of
                        // another factorisation per 31 cells)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        double incx = 0.0;
                        for(int j = 0; j < n; ++j) {
                            const float xj = readlane_f(o0.x, j), yj = readlane_f(o0.y, j), zj_ = readlane_f(o0.z, j);
                            const float ej = readlane_f(o0.w, j), lj = readlane_f(o1.x, j);
                            const float cg = d_corr_t<PLAIN, true>(cst, gx, gy, gz, ge, gl, xj, yj, zj_, ej, lj, true);
                            incx = __builtin_fma(s_col[wid][j], (double)cg, incx);
                        }
                        if((extra >> lane) & 1ull) {
                            float increment = (float)incx;
                            if(!a.allow_extrap) {
                                if(maxInc > 0 && increment > maxInc) increment = maxInc;
                                else if(maxInc < 0 && increment > 0) increment = maxInc;
                                else if(minInc < 0 && increment < minInc) increment = minInc;
                                else if(minInc > 0 && increment < 0) increment = minInc;
                            }
                            s_res[wid][0][lane] = bg + increment;
                            s_res[wid][1][lane] = bvar;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                unsigned long long mm = zsolve ? 0ull : members;
                // (the 62-row tile has ONE member row: written as a loop, the 62 pivot lanes, reciprocal pivots and right-hand-side entries
                //  were loop invariants the compiler formed up front and kept -- some 70 vector registers at the kernel's peak)
                for(bool once = true; mm != 0ull && (MEMB > 1 || once); once = false) {
                    const int ml = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    const int mi2 = __popcll(members & ((1ull << ml) - 1ull));   // index of this member among the G rows
                    const double g0 = (lane < n) ? (double)colbuf[lane][N + 1 + mi2] : 0.0;   // corr_background(cell, obs `lane`)
                    double bvec = g0;
#pragma unroll
                    for(int j = 0; j < N; ++j) {                           // forward: L y = P g
                        if(j < n) {
                            const int piv = __builtin_ctzll(wave_ballot(mystep == j));
                            const double bp = readlane_d(bvec, piv);
                            if(lane < n && mystep > j) bvec = __builtin_fma(-rowT[j], bp, bvec);
                        }
                    }
                    double inc = 0.0, a00 = 0.0;
#pragma unroll
                    for(int j = N - 1; j >= 0; --j) {                      // backward: U k = y
                        if(j < n) {
                            const int piv = __builtin_ctzll(wave_ballot(mystep == j));
                            const double xj = readlane_d(bvec, piv) * readlane_d(mypinv, piv);
                            if(lane < n && mystep < j) bvec = __builtin_fma(-rowT[j], xj, bvec);
                            inc = __builtin_fma(xj, readlane_d(dmine, j), inc);    // k . (lObs - lY)   (oi.cpp:316)
                            a00 = __builtin_fma(xj, readlane_d(g0, j), a00);       // k . G             (oi.cpp:336)
                        }
                    }
                    float increment = (float)inc;
                    if(!a.allow_extrap) {
                        if(maxInc > 0 && increment > maxInc) increment = maxInc;
                        else if(maxInc < 0 && increment > 0) increment = maxInc;
                        else if(minInc < 0 && increment < minInc) increment = minInc;
                        else if(minInc > 0 && increment < 0) increment = minInc;
                    }
                    const float bgm = readlane_f(bg, ml), bvm = readlane_f(bvar, ml);
                    if(lane == 0) {
                        s_res[wid][0][ml] = bgm + increment;
                        s_res[wid][1][ml] = (float)((double)bvm * (1.0 - a00));
                    }
                }
            }
            else {
                double row[N];
                double myrs = 0.0;   // 1 / L_jj of this lane's own row (for the extra members)
                const bool used = lane < n || lane == N || is_g;
    #pragma unroll
                for(int p = 0; p < N; ++p) {
                    double v = 0.0;
                    if(p < n) {
                        v = (double)colbuf[p][lane];
                        if(ln == p) v += (double)o1.w;                                                     // lP + lR
                        const double dp = (double)readlane_f(o1.y, p) - (double)readlane_f(o1.z, p);       // lObs - lY
                        if(lane == N) v = dp;
                        if(!used) v = 0.0;
                    }
                    row[p] = v;
                }
                // right-looking Cholesky on rows-in-lanes; row N becomes L^-1 d, rows N+1.. become L^-1 g_m
    #pragma unroll
                for(int j = 0; j < N; ++j) {
                    if(j < n) {
                        const double ajj = readlane_d(row[j], j);
                        if(!(ajj > 0.0)) bad = true;
                        double rs = __builtin_amdgcn_rsq(ajj);
                        rs = rs * (1.5 - 0.5 * ajj * rs * rs);
                        rs = rs * (1.5 - 0.5 * ajj * rs * rs);
                        const double cj = row[j] * rs;
                        row[j] = cj;
                        if(lane == j) myrs = rs;
    #pragma unroll
                        for(int p = j + 1; p < N; ++p) {
                            const double lpj = readlane_d(cj, p);
                            row[p] = __builtin_fma(-cj, lpj, row[p]);
                        }
                    }
                }
                double inc = 0.0, a00 = 0.0;
    #pragma unroll
                for(int p = 0; p < N; ++p) {
                    const double tp = readlane_d(row[p], N);
                    inc = __builtin_fma(row[p], tp, inc);      // lGSR * (lObs - lY)   (oi.cpp:316)
                    a00 = __builtin_fma(row[p], row[p], a00);  // lGSR * lG^T          (oi.cpp:336)
                }
                if(is_g) {
                    float increment = (float)inc;   // oi.cpp:317
                    if(!a.allow_extrap) {           // oi.cpp:318-334
                        if(maxInc > 0 && increment > maxInc) increment = maxInc;
                        else if(maxInc < 0 && increment > 0) increment = maxInc;
                        else if(minInc < 0 && increment < minInc) increment = minInc;
                        else if(minInc > 0 && increment < 0) increment = minInc;
                    }
                    s_res[wid][0][src] = cbg + increment;                      // oi.cpp:335
                    s_res[wid][1][src] = (float)((double)cbv * (1.0 - a00));   // oi.cpp:337
                }
                // the other cells of the group: z = L^-1 g by a column-oriented forward substitution, lane i holds g_i
                for(unsigned long long mm = extra; mm != 0ull; mm &= mm - 1ull) {
                    const int ml = __builtin_ctzll(mm);
                    const float cx = readlane_f(gx, ml), cy = readlane_f(gy, ml), cz = readlane_f(gz, ml), ce = readlane_f(ge, ml), cl = readlane_f(gl, ml);
                    double gi = (lane < n) ? (double)d_corr_t<PLAIN, true>(pst, cx, cy, cz, ce, cl, px, py, pz, pe, pl, true) : 0.0;   // lG (oi.cpp:250,296)
                    double inc2 = 0.0, a002 = 0.0;
    #pragma unroll
                    for(int j = 0; j < N; ++j) {
                        if(j < n) {
                            const double zj = readlane_d(gi * myrs, j);
                            gi = __builtin_fma(-row[j], zj, gi);
                            inc2 = __builtin_fma(zj, readlane_d(row[j], N), inc2);
                            a002 = __builtin_fma(zj, zj, a002);
                        }
                    }
                    float increment = (float)inc2;
                    if(!a.allow_extrap) {
                        if(maxInc > 0 && increment > maxInc) increment = maxInc;
                        else if(maxInc < 0 && increment > 0) increment = maxInc;
                        else if(minInc < 0 && increment < minInc) increment = minInc;
                        else if(minInc > 0 && increment < 0) increment = minInc;
                    }
                    const float bgm = readlane_f(bg, ml), bvm = readlane_f(bvar, ml);
                    if(lane == 0) {
                        s_res[wid][0][ml] = bgm + increment;
                        s_res[wid][1][ml] = (float)((double)bvm * (1.0 - a002));
                    }
                }
                    }
        };

        // two single-member groups at once, one per half-wave: oi_solve_pair
        auto solve_pair = [&](const int la, const int lb) {
            nsolve += 2;
            const int lh = (lane & 32) ? lb : la;
            oi_solve_pair<PLAIN>(a, lane, la, lb, __builtin_amdgcn_readlane(cnt, la), __builtin_amdgcn_readlane(cnt, lb),
                                 [&](const int i, const int l) { return origs[i][l]; }, colbuf, s_col[wid], s_res[wid], __shfl(gx, lh), __shfl(gy, lh),
                                 __shfl(gz, lh), __shfl(ge, lh), __shfl(gl, lh), __shfl(bg, lh), __shfl(bvar, lh), bad);
        };

        constexpr bool PAIRS = !LU && !SPATIAL && N == 32;
        unsigned long long singles = 0ull;
        while(todo) {
            const int l = __builtin_ctzll(todo);
            const int n = __builtin_amdgcn_readlane(cnt, l);
            const unsigned long long l1 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(h1 >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)h1, l);
            const unsigned long long l2 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(h2s >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)h2s, l);
            unsigned long long members = wave_ballot(cnt == n && h1 == l1 && h2s == l2) & todo;
            // at most MEMB members ride along as G rows; with the 62-row tile (MEMB = 1) the other cells of the group reuse the
            // factor through forward substitutions (Cholesky path), otherwise they form the next pass
            int nm = __popcll(members);
            unsigned long long extra = 0ull;
            if(nm > MEMB) {
                const int cut = nth_set_bit(members, MEMB);
                const unsigned long long first = members & ((1ull << cut) - 1ull);
                if((!LU && N == 62) || (LU && a.out_var == nullptr)) extra = members & ~first;   // (LU without a variance output: one solve serves them all)
                members = first;
                nm = MEMB;
            }
            todo &= ~(members | extra);
            if(PAIRS && nm == 1 && n <= 30) singles |= 1ull << l;
            else solve_group(l, n, members, nm, extra);
        }
        if constexpr(PAIRS) {
            if(a.pair_sel) {   // the selections of the single-member groups go to HBM: k_oi_pairs solves them at twice the occupancy
                parked = (singles >> lane) & 1ull;
                if(parked) {
                    // (16 B per store: a lane's record is its own 128 B line, and 30 dword stores per lane were 4x the write transactions)
                    uint4* const dst = reinterpret_cast<uint4*>(a.pair_sel + (size_t)cell * 32);
                    for(int s = 0; s < cnt; s += 4) dst[s >> 2] = make_uint4(origs[s][lane], origs[s + 1][lane], origs[s + 2][lane], origs[s + 3][lane]);
                }
                singles = 0ull;
            }
            while(singles) {
                const int la = __builtin_ctzll(singles);
                singles &= singles - 1;
                if(singles) {
                    const int lb = __builtin_ctzll(singles);
                    singles &= singles - 1;
                    solve_pair(la, lb);
                }
                else solve_group(la, __builtin_amdgcn_readlane(cnt, la), 1ull << la, 1, 0ull);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (results were written by single lanes: see solve_pair)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if(cnt > 0) { res_out = s_res[wid][0][lane]; res_var = s_res[wid][1][lane]; }
        if(wave_ballot(bad) != 0ull && lane == 0) atomicOr(a.err, ERR_SINGULAR);
        if(lane == 0 && a.counters) {
            unsigned long long* cs = a.counters + 80 + 2 * (blockIdx.x % GPP_NSLOT);
            atomicAdd(&cs[0], (unsigned long long)nupd);
            atomicAdd(&cs[1], (unsigned long long)nsolve);
        }
    }
    if(cell >= 0) {
        if(!parked) {
            a.out[cell] = res_out;
            if(a.out_var) a.out_var[cell] = res_var;
        }
        if(a.pair_n) a.pair_n[cell] = parked ? cnt : 0;
    }
    __builtin_amdgcn_wave_barrier();
    }
}

// The single-member groups k_oi parked (a.pair_n[cell] = observation count, a.pair_sel[cell][0..n-1] = the selection): one tile per
// wave as in k_oi, two cells per factorisation pass (oi_solve_pair).  Without the 16 KB of candidate keys per wave that the scan
// needs, four waves per SIMD instead of two -- the solves are chains of dependent LDS round trips and double-precision operations.
template <bool PLAIN>
__global__ __launch_bounds__(256, PLAIN ? 4 : 3) void k_oi_pairs(OiArgs a) {   // (the generic structure functions need more registers)
    __shared__ float s_cb[4][31][64];
    __shared__ double s_col[4][64];
    __shared__ float s_res[4][2][64];
    d_exptab_fill();
    d_tritab_fill();
    __syncthreads();
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + wid;
    if(tile >= a.ntiles) return;
    int cell = -1;
    if(a.tiled2d) {
        int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        int y = ty * (64 >> a.wshift) + (lane >> a.wshift), x = (tx << a.wshift) + (lane & ((1 << a.wshift) - 1));
        if(y < a.ny && x < a.nx) cell = y * a.nx + x;
    }
    else {
        int c = tile * 64 + lane;
        if(c < a.C) cell = c;
    }
    const int n = cell >= 0 ? a.pair_n[cell] : 0;
    unsigned long long singles = wave_ballot(n > 0);
    if(singles == 0ull) return;
    bool bad = false;
    const int nsolve = __popcll(singles);
    while(singles) {
        const int la = __builtin_ctzll(singles);
        singles &= singles - 1;
        int lb = -1;
        if(singles) { lb = __builtin_ctzll(singles); singles &= singles - 1; }
        const int ca = __builtin_amdgcn_readlane(cell, la), cb = __builtin_amdgcn_readlane(cell, lb < 0 ? la : lb);
        const int ch = (lane & 32) ? cb : ca;      // the half's cell: its record comes straight from HBM (two addresses per load)
        const unsigned* const mine = a.pair_sel + (size_t)ch * 32;
        oi_solve_pair<PLAIN>(a, lane, la, lb, __builtin_amdgcn_readlane(n, la), lb < 0 ? 0 : __builtin_amdgcn_readlane(n, lb),
                             [&](const int i, const int) { return mine[i]; }, s_cb[wid], s_col[wid], s_res[wid], a.gx[ch], a.gy[ch], a.gz[ch],
                             a.gelev[ch], a.glaf[ch], a.bg[ch], a.bvar ? a.bvar[ch] : 1.0f, bad);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (results were written by single lanes: see oi_solve_pair)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if(n > 0) {
        a.out[cell] = s_res[wid][0][lane];
        if(a.out_var) a.out_var[cell] = s_res[wid][1][lane];
    }
    if(wave_ballot(bad) != 0ull && lane == 0) atomicOr(a.err, ERR_SINGULAR);
    if(lane == 0 && a.counters) atomicAdd(&a.counters[80 + 2 * (blockIdx.x % GPP_NSLOT) + 1], (unsigned long long)nsolve);
}

// -------------------------------------------------------------------------------------------
// k_oi_big: grid points with more usable observations than the 62-row register tile of k_oi holds (max_points == 0 with
// many observations inside the localization radius, or max_points > 62).  One workgroup per listed cell: radius query
// over the bins (all usable observations, oi.cpp:229-260), bitonic sort of the keys in LDS (rho descending, then lower
// index: the order of oi.cpp:262-273), (P+R | d | g) in HBM scratch, right-looking Cholesky with the two right-hand
// sides riding along as extra rows.  Slow next to the tile kernels (the reference is O(n^3) per grid point here too), but
// it removes the cliff: any max_points up to BIG_N observations per grid point, symmetric structure functions.
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_oi_big(OiArgs a) {
    __shared__ unsigned long long s_key[BIG_CAND];   // 64 KB; after the sort: the per-observation tables below
    __shared__ int s_n;
    __shared__ double s_red[2][256];
    __shared__ float s_mm[2][256];
    const int tid = threadIdx.x;
    const ScanArgs& sa = a.s;
    const DevStructure& st = sa.st;
    const int nlist = *a.big_count;
    unsigned long long* const gkeys = a.big_keys + (size_t)blockIdx.x * BIG_CAND;
    // The augmented matrix ((n + 2) x n doubles) stays in LDS up to BIG_NL observations: the right-looking factorisation is a chain of
    // n column steps with a read-modify-write of the trailing matrix each -- through HBM scratch every step paid a few memory latencies
    // (1.0 us per grid point at max_points 100 with one workgroup per CU); larger systems keep the scratch of round 1.
    __shared__ double s_mat[(BIG_NL + 2) * BIG_NL];
    double* const gA = a.big_mat + (size_t)blockIdx.x * (BIG_N + 2) * BIG_N;
    for(int li = blockIdx.x; li < nlist; li += gridDim.x) {
        const int cell = a.big_list[li];
        const float gx = a.gx[cell], gy = a.gy[cell], gz = a.gz[cell], ge = a.gelev[cell], gl = a.glaf[cell];
        const float bg = a.bg[cell], bvar = a.bvar ? a.bvar[cell] : 1.0f;
        if(tid == 0) s_n = 0;
        __syncthreads();
        // ---- radius query: every bin the box [p - R, p + R] touches on the two binned axes ------------------------------------
        const float R = st.R;
        const float pa = sa.axis_a == 0 ? gx : (sa.axis_a == 1 ? gy : gz), pb = sa.axis_b == 1 ? gy : (sa.axis_b == 2 ? gz : gx);
        const int bx0 = min(max((int)floorf((pa - R - sa.amin) * sa.inv_s) - 1, 0), sa.nbx - 1), bx1 = min(max((int)floorf((pa + R - sa.amin) * sa.inv_s) + 1, 0), sa.nbx - 1);
        const int by0 = min(max((int)floorf((pb - R - sa.bmin) * sa.inv_s) - 1, 0), sa.nby - 1), by1 = min(max((int)floorf((pb + R - sa.bmin) * sa.inv_s) + 1, 0), sa.nby - 1);
        const float lox = gx - R, hix = gx + R, loy = gy - R, hiy = gy + R, loz = gz - R, hiz = gz + R;
        for(int by = by0; by <= by1; ++by) {
            const int js = sa.bin_start[by * sa.nbx + bx0], je = sa.bin_start[by * sa.nbx + bx1 + 1];
            for(int j = js + tid; j < je; j += 256) {
                const float4 rec = sa.pgeo[j];
                if(!(rec.x > lox && rec.x < hix && rec.y > loy && rec.y < hiy && rec.z > loz && rec.z < hiz)) continue;   // kdtree.cpp:46,53
                const float2 met = sa.smeta[j];
                const float dist = d_chord(rec.x, rec.y, rec.z, gx, gy, gz);
                if(!(dist <= R)) continue;                                                                             // kdtree.cpp:255
                const float rho = d_corr(st, gx, gy, gz, ge, gl, rec.x, rec.y, rec.z, rec.w, met.x, true);
                if(!(rho > 0.0f)) continue;                                                                            // oi.cpp:253
                const int k = atomicAdd(&s_n, 1);
                if(k < BIG_CAND) s_key[k] = ((unsigned long long)__float_as_uint(rho) << 32) | (unsigned)(~__float_as_int(met.y));
            }
        }
        __syncthreads();
        const int ncand = s_n;
        int n = (a.s.max_points > 0) ? min(ncand, a.s.max_points) : ncand;
        if(ncand > BIG_CAND || n > BIG_N) {   // beyond what this kernel holds: the general kernel takes the cell
            if(tid == 0) { if(a.huge_list) a.huge_list[atomicAdd(a.huge_count, 1)] = cell; else atomicOr(a.err, ERR_OVERFLOW); }
            __syncthreads();
            continue;
        }
        double* const A = n <= BIG_NL ? s_mat : gA;
        // ---- the max_points largest keys first (radix select, oi_common.h), then the bitonic sort of those alone, descending: rho
        //      descending, ties -> lower observation index (oi.cpp:262-273) ----------------------------------------------------
        int nsort = ncand;
        if(n < ncand) { block_select_largest(s_key, ncand, n, gkeys, reinterpret_cast<int*>(&s_red[0][0]), &s_n, tid); nsort = n; }
        int np2 = 1;
        while(np2 < nsort) np2 <<= 1;
        for(int i = nsort + tid; i < np2; i += 256) s_key[i] = 0ull;
        __syncthreads();
        for(int k = 2; k <= np2; k <<= 1) {
            for(int j = k >> 1; j > 0; j >>= 1) {
                for(int i = tid; i < np2; i += 256) {
                    const int ixj = i ^ j;
                    if(ixj > i) {
                        const unsigned long long x = s_key[i], y = s_key[ixj];
                        const bool desc = (i & k) == 0;
                        if(desc ? (x < y) : (x > y)) { s_key[i] = y; s_key[ixj] = x; }
                    }
                }
                __syncthreads();
            }
        }
        // ---- the selected observations: keys out to HBM, per-observation tables into the LDS the keys occupied ---------------
        for(int i = tid; i < n; i += 256) gkeys[i] = s_key[i];
        __syncthreads();
        float4* const ogeo = reinterpret_cast<float4*>(s_key);                       // x, y, z, elev      [BIG_N]
        float* const olaf = reinterpret_cast<float*>(ogeo + BIG_N);                   // laf                [BIG_N]
        float* const odp = olaf + BIG_N;                                              // obs - background   [BIG_N] (float, for the clamp)
        float maxInc = -INFINITY, minInc = INFINITY;
        for(int i = tid; i < n; i += 256) {
            const unsigned long long key = gkeys[i];
            const unsigned orig = ~(unsigned)(key & 0xffffffffull);
            const float4 g4 = a.ogeo[orig], x4 = a.oaux[orig];                        // oaux: laf, obs, pbg, ratio
            ogeo[i] = g4; olaf[i] = x4.x;
            const double d = (double)x4.y - (double)x4.z;
            odp[i] = (float)d;
            maxInc = fmaxf(maxInc, (float)d); minInc = fminf(minInc, (float)d);
            A[(size_t)n * n + i] = d;                                                 // row n: lObs - lY (oi.cpp:293)
            A[(size_t)(n + 1) * n + i] = (double)__uint_as_float((unsigned)(key >> 32));   // row n + 1: lG = rho (oi.cpp:296)
            A[(size_t)i * n + i] = (double)x4.w;                                      // lR on the diagonal, P added below
        }
        __syncthreads();
        // ---- lower triangle of P (oi.cpp:304-312) ------------------------------------------------------------------------------
        const long ntri = (long)n * (n + 1) / 2;
        for(long e = tid; e < ntri; e += 256) {
            int i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
            if((long)i * (i + 1) / 2 > e) i--;
            if((long)(i + 1) * (i + 2) / 2 <= e) i++;
            const int j = (int)(e - (long)i * (i + 1) / 2);
            const float4 pi = ogeo[i], pj = ogeo[j];
            const double c = (double)d_corr(st, pi.x, pi.y, pi.z, pi.w, olaf[i], pj.x, pj.y, pj.z, pj.w, olaf[j], false);
            if(i == j) A[(size_t)i * n + i] += c; else A[(size_t)i * n + j] = c;
        }
        __syncthreads();
        // ---- right-looking Cholesky, rows n and n + 1 ride along and end as L^-1 d and L^-1 g ---------------------------------
        double* const col = reinterpret_cast<double*>(odp + BIG_N);                   // column j of the factor [BIG_N + 2]
        bool bad = false;
        const int ti = tid >> 4, tk = tid & 15;
        for(int j = 0; j < n; ++j) {
            const double ajj = A[(size_t)j * n + j];
            if(!(ajj > 0.0)) bad = true;
            const double rs = 1.0 / sqrt(ajj);
            for(int i = j + tid; i < n + 2; i += 256) { const double v = A[(size_t)i * n + j] * rs; A[(size_t)i * n + j] = v; col[i] = v; }
            __syncthreads();
            for(int i = j + 1 + ti; i < n + 2; i += 16) {
                const double lij = col[i];
                const int kend = min(i, n - 1);
                for(int k = j + 1 + tk; k <= kend; k += 16) A[(size_t)i * n + k] -= lij * col[k];
            }
            __syncthreads();
        }
        // ---- increment = (L^-1 g) . (L^-1 d), 1 - K G^T = 1 - |L^-1 g|^2 (oi.cpp:315-316,336) --------------------------------
        double inc = 0.0, a00 = 0.0;
        for(int k = tid; k < n; k += 256) { const double zg = A[(size_t)(n + 1) * n + k]; inc += zg * A[(size_t)n * n + k]; a00 += zg * zg; }
        s_red[0][tid] = inc; s_red[1][tid] = a00; s_mm[0][tid] = maxInc; s_mm[1][tid] = minInc;
        __syncthreads();
        for(int off = 128; off > 0; off >>= 1) {
            if(tid < off) {
                s_red[0][tid] += s_red[0][tid + off]; s_red[1][tid] += s_red[1][tid + off];
                s_mm[0][tid] = fmaxf(s_mm[0][tid], s_mm[0][tid + off]); s_mm[1][tid] = fminf(s_mm[1][tid], s_mm[1][tid + off]);
            }
            __syncthreads();
        }
        if(tid == 0) {
            if(bad) atomicOr(a.err, ERR_SINGULAR);
            if(n > 0) {
                float increment = (float)s_red[0][0];   // oi.cpp:317
                const float mx = s_mm[0][0], mn = s_mm[1][0];
                if(!a.allow_extrap) {                   // oi.cpp:318-334
                    if(mx > 0 && increment > mx) increment = mx;
                    else if(mx < 0 && increment > 0) increment = mx;
                    else if(mn < 0 && increment < mn) increment = mn;
                    else if(mn > 0 && increment < 0) increment = mn;
                }
                a.out[cell] = bg + increment;                                                // oi.cpp:335
                if(a.out_var) a.out_var[cell] = (float)((double)bvar * (1.0 - s_red[1][0]));   // oi.cpp:337
            }
        }
        __syncthreads();
    }
}


// -------------------------------------------------------------------------------------------
// k_oi_huge: the general form of the local analysis (oi.cpp:229-337) for the grid points the faster kernels cannot hold: more
// than BIG_N selected observations, or more than 62 with a non-symmetric (Cressman / SOAR / TOAR vertical factors) or
// spatially varying structure function.  No capacity of its own beyond the scratch the host sized for the call: the
// candidate keys and the augmented matrix (P + R | obs - background | rho) live in HBM, the candidates are sorted by a
// bitonic network over global memory, the system is solved by Gaussian elimination with partial pivoting (what
// arma::inv = LAPACK does, so indefinite truncated kernels behave as in the reference), the pivot row staged in LDS.
// One 256-thread workgroup per listed grid point; O(n^3) like the reference.
// -------------------------------------------------------------------------------------------
#define HUGE_ROW 4096
template <bool SPATIAL>
__global__ __launch_bounds__(256) void k_oi_huge(OiArgs a, const int* __restrict__ list, const int* __restrict__ count) {
    __shared__ double s_row[HUGE_ROW];      // pivot row (the first HUGE_ROW entries are staged; longer rows are read from HBM)
    __shared__ double s_red[2][256];
    __shared__ float s_mm[2][256];
    __shared__ int s_idx[256];
    __shared__ int s_n, s_piv;
    const int tid = threadIdx.x;
    const ScanArgs& sa = a.s;
    const int nlist = *count;
    unsigned long long* const keys = a.huge_keys + (size_t)blockIdx.x * a.huge_kcap;
    double* const M = a.huge_mat + (size_t)blockIdx.x * a.huge_ncap * (a.huge_ncap + 2);
    for(int li = blockIdx.x; li < nlist; li += gridDim.x) {
        const int cell = list[li];
        const float gx = a.gx[cell], gy = a.gy[cell], gz = a.gz[cell], ge = a.gelev[cell], gl = a.glaf[cell];
        const float bg = a.bg[cell], bvar = a.bvar ? a.bvar[cell] : 1.0f;
        DevStructure st = sa.st;
        if(SPATIAL) d_structure_at(st, st.cell_idx ? st.cell_idx[cell] : cell);
        if(tid == 0) s_n = 0;
        __syncthreads();
        const float R = st.R;
        const float pa = sa.axis_a == 0 ? gx : (sa.axis_a == 1 ? gy : gz), pb = sa.axis_b == 1 ? gy : (sa.axis_b == 2 ? gz : gx);
        const int bx0 = min(max((int)floorf((pa - R - sa.amin) * sa.inv_s) - 1, 0), sa.nbx - 1), bx1 = min(max((int)floorf((pa + R - sa.amin) * sa.inv_s) + 1, 0), sa.nbx - 1);
        const int by0 = min(max((int)floorf((pb - R - sa.bmin) * sa.inv_s) - 1, 0), sa.nby - 1), by1 = min(max((int)floorf((pb + R - sa.bmin) * sa.inv_s) + 1, 0), sa.nby - 1);
        const float lox = gx - R, hix = gx + R, loy = gy - R, hiy = gy + R, loz = gz - R, hiz = gz + R;
        for(int by = by0; by <= by1; ++by) {
            const int js = sa.bin_start[by * sa.nbx + bx0], je = sa.bin_start[by * sa.nbx + bx1 + 1];
            for(int j = js + tid; j < je; j += 256) {
                const float4 rec = sa.pgeo[j];
                if(!(rec.x > lox && rec.x < hix && rec.y > loy && rec.y < hiy && rec.z > loz && rec.z < hiz)) continue;   // kdtree.cpp:46,53
                const float2 met = sa.smeta[j];
                if(!(d_chord(rec.x, rec.y, rec.z, gx, gy, gz) <= R)) continue;                                          // kdtree.cpp:255
                const float rho = d_corr(st, gx, gy, gz, ge, gl, rec.x, rec.y, rec.z, rec.w, met.x, true);
                if(!(rho > 0.0f)) continue;                                                                            // oi.cpp:253
                const int k = atomicAdd(&s_n, 1);
                if(k < a.huge_kcap) keys[k] = ((unsigned long long)__float_as_uint(rho) << 32) | (unsigned)(~__float_as_int(met.y));
            }
        }
        __syncthreads();
        const int ncand = s_n;
        const bool truncated = a.s.max_points > 0 && ncand > a.s.max_points;
        const int n = truncated ? a.s.max_points : ncand;
        int np2 = 1;
        while(np2 < ncand) np2 <<= 1;
        if(np2 > a.huge_kcap || n > a.huge_ncap) { if(tid == 0) atomicOr(a.err, ERR_OVERFLOW); __syncthreads(); continue; }
        if(n == 0) continue;
        // ---- order: rho descending, ties -> lower index, when the reference sorts (oi.cpp:262-273); candidate (= index) order otherwise
        for(int i = ncand + tid; i < np2; i += 256) keys[i] = 0ull;
        __threadfence_block();
        __syncthreads();
        for(int k = 2; k <= np2; k <<= 1) {
            for(int j = k >> 1; j > 0; j >>= 1) {
                for(int i = tid; i < np2; i += 256) {
                    const int ixj = i ^ j;
                    if(ixj > i) {
                        const unsigned long long x = keys[i], y = keys[ixj];
                        const unsigned long long kx = truncated ? x : (x & 0xffffffffull), ky = truncated ? y : (y & 0xffffffffull);
                        const bool desc = (i & k) == 0;
                        if(desc ? (kx < ky) : (kx > ky)) { keys[i] = y; keys[ixj] = x; }
                    }
                }
                __threadfence_block();
                __syncthreads();
            }
        }
        // ---- augmented matrix: row i = (corr(p_i, p_j) + [i == j] ratio_i | obs_i - pbg_i | rho_i), oi.cpp:289-314 ----------------------
        const int W = n + 2;
        float maxInc = -INFINITY, minInc = INFINITY;
        for(long e = tid; e < (long)n * n; e += 256) {
            const int i = (int)(e / n), j = (int)(e - (long)i * n);
            const unsigned oi = ~(unsigned)(keys[i] & 0xffffffffull), oj = ~(unsigned)(keys[j] & 0xffffffffull);
            const float4 gi = a.ogeo[oi], gj = a.ogeo[oj];
            const float4 xi = a.oaux[oi], xj = a.oaux[oj];
            DevStructure si = sa.st;
            if(SPATIAL) d_structure_at(si, si.obs_idx[oi]);   // corr(p1, p2) takes the scales at its FIRST point (structure.cpp:188-214)
            const double c = (double)d_corr(si, gi.x, gi.y, gi.z, gi.w, xi.x, gj.x, gj.y, gj.z, gj.w, xj.x, false);
            M[(size_t)i * W + j] = c + (i == j ? (double)xi.w : 0.0);
        }
        for(int i = tid; i < n; i += 256) {
            const unsigned long long key = keys[i];
            const float4 x4 = a.oaux[~(unsigned)(key & 0xffffffffull)];
            const double d = (double)x4.y - (double)x4.z;
            M[(size_t)i * W + n] = d;
            M[(size_t)i * W + n + 1] = (double)__uint_as_float((unsigned)(key >> 32));
            maxInc = fmaxf(maxInc, (float)d); minInc = fminf(minInc, (float)d);
        }
        __threadfence_block();
        __syncthreads();
        // ---- Gaussian elimination with partial pivoting on (A | d | g^T) -------------------------------------------------------------------
        // inc = G A^-1 d and a00 = G A^-1 G^T need u = A^-1 d and v = A^-1 G^T: both right-hand sides ride along
        bool singular = false;
        for(int k = 0; k < n; ++k) {
            double best = -1.0; int bi = -1;
            for(int i = k + tid; i < n; i += 256) { const double v = fabs(M[(size_t)i * W + k]); if(v > best) { best = v; bi = i; } }
            s_red[0][tid] = best; s_idx[tid] = bi;
            __syncthreads();
            for(int off = 128; off > 0; off >>= 1) {
                if(tid < off && (s_red[0][tid + off] > s_red[0][tid] || (s_red[0][tid + off] == s_red[0][tid] && s_idx[tid + off] >= 0 && (s_idx[tid] < 0 || s_idx[tid + off] < s_idx[tid])))) {
                    s_red[0][tid] = s_red[0][tid + off]; s_idx[tid] = s_idx[tid + off];
                }
                __syncthreads();
            }
            if(tid == 0) s_piv = (s_red[0][0] > 0.0 && s_red[0][0] < INFINITY) ? s_idx[0] : -1;
            __syncthreads();
            const int p = s_piv;
            if(p < 0) { singular = true; break; }
            // swap rows k and p (columns k .. n + 1), stage the pivot row
            for(int j = k + tid; j < W; j += 256) {
                const double vk = M[(size_t)k * W + j], vp = M[(size_t)p * W + j];
                M[(size_t)k * W + j] = vp; M[(size_t)p * W + j] = vk;
                if(j - k < HUGE_ROW) s_row[j - k] = vp;
            }
            __threadfence_block();
            __syncthreads();
            const double pinv = 1.0 / s_row[0];
            const int ti = tid >> 4, tk = tid & 15;
            for(int i = k + 1 + ti; i < n; i += 16) {
                const double f = M[(size_t)i * W + k] * pinv;
                if(f != 0.0)
                    for(int j = k + 1 + tk; j < W; j += 16) {
                        const double pr = (j - k < HUGE_ROW) ? s_row[j - k] : M[(size_t)k * W + j];
                        M[(size_t)i * W + j] -= f * pr;
                    }
            }
            __threadfence_block();
            __syncthreads();
        }
        if(singular) { if(tid == 0) atomicOr(a.err, ERR_SINGULAR); __syncthreads(); continue; }
        // ---- back substitution for the two right-hand sides (columns n, n + 1 become u, v) ----------------------------------------------
        for(int r = n - 1; r >= 0; --r) {
            double su = 0.0, sv = 0.0;
            for(int j = r + 1 + tid; j < n; j += 256) { const double m = M[(size_t)r * W + j]; su += m * M[(size_t)j * W + n]; sv += m * M[(size_t)j * W + n + 1]; }
            s_red[0][tid] = su; s_red[1][tid] = sv;
            __syncthreads();
            for(int off = 128; off > 0; off >>= 1) { if(tid < off) { s_red[0][tid] += s_red[0][tid + off]; s_red[1][tid] += s_red[1][tid + off]; } __syncthreads(); }
            if(tid == 0) {
                const double dinv = 1.0 / M[(size_t)r * W + r];
                M[(size_t)r * W + n] = (M[(size_t)r * W + n] - s_red[0][0]) * dinv;
                M[(size_t)r * W + n + 1] = (M[(size_t)r * W + n + 1] - s_red[1][0]) * dinv;
            }
            __threadfence_block();
            __syncthreads();
        }
        // increment = G u, K G^T = G v (oi.cpp:315-316,336); G in the ORIGINAL row order is rho of keys[i]: the row swaps permuted
        // equations, not unknowns, so u_i and v_i still belong to observation i
        double inc = 0.0, a00 = 0.0;
        for(int i = tid; i < n; i += 256) {
            const double g = (double)__uint_as_float((unsigned)(keys[i] >> 32));
            inc += g * M[(size_t)i * W + n]; a00 += g * M[(size_t)i * W + n + 1];
        }
        s_red[0][tid] = inc; s_red[1][tid] = a00; s_mm[0][tid] = maxInc; s_mm[1][tid] = minInc;
        __syncthreads();
        for(int off = 128; off > 0; off >>= 1) {
            if(tid < off) {
                s_red[0][tid] += s_red[0][tid + off]; s_red[1][tid] += s_red[1][tid + off];
                s_mm[0][tid] = fmaxf(s_mm[0][tid], s_mm[0][tid + off]); s_mm[1][tid] = fminf(s_mm[1][tid], s_mm[1][tid + off]);
            }
            __syncthreads();
        }
        if(tid == 0) {
            float increment = (float)s_red[0][0];   // oi.cpp:317
            const float mx = s_mm[0][0], mn = s_mm[1][0];
            if(!a.allow_extrap) {                   // oi.cpp:318-334
                if(mx > 0 && increment > mx) increment = mx;
                else if(mx < 0 && increment > 0) increment = mx;
                else if(mn < 0 && increment < mn) increment = mn;
                else if(mn > 0 && increment < 0) increment = mn;
            }
            a.out[cell] = bg + increment;                                                // oi.cpp:335
            if(a.out_var) a.out_var[cell] = (float)((double)bvar * (1.0 - s_red[1][0]));   // oi.cpp:337
        }
        __syncthreads();
    }
}
// -------------------------------------------------------------------------------------------
// host entry point
// -------------------------------------------------------------------------------------------
// one flag byte per tile of a remembered work list (union_memo): the first pass of the next call leaves these tiles to the list passes
__global__ void k_flag_tiles(const int* __restrict__ list, int n, int ntiles, unsigned char* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) { const int t = list[i]; if(t >= 0 && t < ntiles) flags[t] = 1; }
}
namespace {
struct OiWorkspace {
    // per-call observation block and status block, TWO of each, used alternately: a deferred call (GPP_ASYNC) packs the block of call k + 1 on the
    // second stream while the first pass of call k still reads its own (oi_full_impl, `steady`)
    DevBuf<float4> pgeo_s[2], oaux_s[2], saux_s[2];
    DevBuf<unsigned long long> status_s[2];
    hipEvent_t slot_e1[2] = {nullptr, nullptr}, slot_ec[2] = {nullptr, nullptr};   // last deferred call on the slot: end of its first pass / its status copy
    hipEvent_t ev_pack[2] = {nullptr, nullptr}, ev_join2[2] = {nullptr, nullptr};
    int slot = 0;
    // status block of a call, one memset and one read-back: ints [0] err, [1..3] work-list lengths, [4] large-n cell count;
    // statistics counters from byte 64 on
    DevBuf<unsigned long long> status_snap;
    unsigned long long* h_status = nullptr;   // pinned host mirror
    DevBuf<int> cell_idx, obs_idx, fb_list, fb_list2, fb_list3, big_list;
    DevBuf<unsigned long long> big_keys, huge_keys;
    DevBuf<double> big_mat, huge_mat;
    DevBuf<int> huge_list;
    DevBuf<unsigned> pair_sel;   // k_oi -> k_oi_pairs: 128 B per cell
    DevBuf<int> pair_n;
    hipEvent_t e0 = nullptr, e1 = nullptr, eu = nullptr, ev_fork = nullptr, ev_join = nullptr;
    // banded host path (round 6): upload / first pass / download per band of tile rows, and the patch of the tiles the first pass left to the list passes
    static constexpr int MAXB = 8;
    hipEvent_t ev_up[MAXB] = {}, ev_k[MAXB] = {}, ev_band0 = nullptr;
    DevBuf<float> patch;          // [tiles][64] analysis (+ [tiles][64] variance behind it)
    float* h_patch = nullptr;     // pinned mirror, then the tile ids
    size_t h_patch_cap = 0;
};
thread_local OiWorkspace g_ws;

// banded host path: the cells of the listed tiles, tile-major (lane order of the tile kernels), for the host to put in place
__global__ void k_gather_tiles(const int* __restrict__ l0, int n0, const int* __restrict__ l1, int n1, int ny, int nx, int tiles_x, int wshift,
                               const float* __restrict__ out, const float* __restrict__ var, float* __restrict__ p_out, float* __restrict__ p_var, int* __restrict__ p_ids) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if(i >= n0 + n1) return;
    int tile = i < n0 ? l0[i] : l1[i - n0];
    if(tile < 0) tile = ~tile;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y = ty * (64 >> wshift) + (lane >> wshift), x = (tx << wshift) + (lane & ((1 << wshift) - 1));
    float v = 0.0f, w = 0.0f;
    if(y < ny && x < nx) { v = out[(size_t)y * nx + x]; if(var) w = var[(size_t)y * nx + x]; }
    p_out[(size_t)i * 64 + lane] = v;
    if(p_var) p_var[(size_t)i * 64 + lane] = w;
    if(lane == 0) p_ids[i] = tile;
}
thread_local gpp_oi_stats g_stats;

// ---- asynchronous calls (GPP_MEM_DEVICE | GPP_ASYNC; round 5) ----------------------------------------------------------------------
// A call in the steady state of a repeated analysis -- same Grid / Points handles as the call before, the work list remembered -- needs the
// host for nothing: every kernel of it is launched from the geometry's memory.  With GPP_ASYNC such a call ENQUEUES its kernels, a copy of
// its status block into a page-locked slot and an event, and returns; gpp_wait() completes the oldest pending call of the thread.  What the
// status block would have triggered in the synchronous call (a tile the memory did not hold, an item left to k_oi, an error flag) is rare
// and handled the simple way: gpp_wait() runs the call again synchronously with the arguments it kept (every path writes the same bits, so
// running twice is only slower).  Calls that are not in the steady state run synchronously at once and are queued as already complete, so
// that every GPP_ASYNC call pairs with one gpp_wait().  The caller keeps the inputs, the output and both handles alive and unchanged until
// the wait returns; up to ASYNC_SLOTS calls may be pending.
constexpr int ASYNC_SLOTS = 4;
struct PendingOi {
    bool done = false;                // completed synchronously (status = rc)
    int rc = GPP_OK;
    std::string msg;
    gpp_oi_stats stats;
    // arguments (for the second run)
    gpp_points* bgrid; const float* background; const float* bvariance; gpp_points* points; const float* obs; const float* obs_variance;
    const float* background_at_points; const float* bvariance_at_points; gpp_structure st; int max_points, allow_extrapolation;
    float* out; float* out_variance; int mem;
    int slot = -1; int n_remembered = 0; bool skip_k_oi = false;
};
struct AsyncSlot { unsigned long long* h = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr, ec = nullptr; };   // page-locked copy of the status block; start / end of the first pass, status copy arrived
thread_local AsyncSlot g_aslots[ASYNC_SLOTS];
thread_local std::vector<PendingOi> g_pending;     // FIFO
thread_local unsigned g_async_seq = 0;
}

void gpp_release_oi_workspace() {   // the parked selections (128 B per grid cell of the largest call so far)
    g_ws.pair_sel.release(); g_ws.pair_n.release();
}

// MultipleStructure with spatially varying parts: the per-point scales of a call (gpp_bind_field below)
namespace { struct MultiFieldWs { DevBuf<float> mh, mv, mw, mR; DevBuf<int> tmpi; }; thread_local MultiFieldWs g_mfw; }

#ifdef GPP_POISON
// Diagnostic build only (tools/hostile/build.sh, tools/*_hostile_soak.py): every byte of the call-to-call workspaces of the OI path is
// set to `byte` (0xFF: NaNs, huge counts, negative list entries), so that a kernel reading something this call did not write meets
// hostile data instead of the remains of the previous call.
void gpp_oi_drain_pending();
extern "C" int gpp_debug_poison_oi_workspace(int byte) {
    GPP_TRY
    ensure_device();
    gpp_oi_drain_pending();
    OiWorkspace& w = g_ws;
    for(int k = 0; k < 2; k++) { w.pgeo_s[k].poison(byte); w.oaux_s[k].poison(byte); w.saux_s[k].poison(byte); w.status_s[k].poison(byte); }
    w.status_snap.poison(byte);
    w.cell_idx.poison(byte); w.obs_idx.poison(byte);
    w.fb_list.poison(byte); w.fb_list2.poison(byte); w.fb_list3.poison(byte); w.big_list.poison(byte); w.huge_list.poison(byte);
    w.big_keys.poison(byte); w.huge_keys.poison(byte); w.big_mat.poison(byte); w.huge_mat.poison(byte);
    w.pair_sel.poison(byte); w.pair_n.poison(byte); w.patch.poison(byte);
    g_mfw.mh.poison(byte); g_mfw.mv.poison(byte); g_mfw.mw.poison(byte); g_mfw.mR.poison(byte); g_mfw.tmpi.poison(byte);
    GPP_HIP(hipStreamSynchronize(stream()));
    if(w.h_status) memset(w.h_status, byte, (8 + 80 + 2 * GPP_NSLOT) * sizeof(unsigned long long));
    return GPP_OK;
    GPP_CATCH
}
#endif

extern "C" int gpp_oi_last_stats(gpp_oi_stats* s) {
    GPP_TRY
    if(!s) invalid("stats is NULL");
    *s = g_stats;
    return GPP_OK;
    GPP_CATCH
}

// ---- structure functions, host side (float semantics of the C++ overloads the reference gets) -------------------
static float st_localization(int kind, float h, float min_rho) {
    switch(kind) {
        case GPP_SK_BARNES: return sqrtf(-2 * logf(min_rho)) * h;                                  // structure.cpp:280-282
        case GPP_SK_CRESSMAN: return h;                                                            // :7-12,87-89
        case GPP_SK_SOAR: { float lm = logf(min_rho); return (-lm + logf(-lm)) * h; }              // :454-459
        case GPP_SK_TOAR: { float lm = logf(min_rho); float ll = logf(-logf(min_rho)); return (float)(((double)(-lm + ll) + 0.5 * ll) * h); }   // :604-610
        case GPP_SK_POWERLAW: return sqrtf(2 * (1 - min_rho) / min_rho) * h;                       // :755-757
        default: return 0;                                                                         // Linear :902-904
    }
}
static bool st_kind_ok(int k) { return k >= GPP_SK_BARNES && k <= GPP_SK_LINEAR; }
DevStructure gpp_resolve_structure(const gpp_structure* s) {
    if(!s) invalid("structure is NULL");
    if(!st_kind_ok(s->kind)) runtime("unknown structure function kind");
    const int kv = s->kind_v ? s->kind_v - 1 : s->kind, kw = s->kind_w ? s->kind_w - 1 : s->kind;
    if(!st_kind_ok(kv) || !st_kind_ok(kw)) runtime("unknown structure function kind");
    if(!is_valid(s->h) || s->h < 0) invalid("h must be >= 0");
    if(!is_valid(s->v) || s->v < 0) invalid("v must be >= 0");
    if(!is_valid(s->w) || s->w < 0) invalid("w must be >= 0");
    DevStructure d;
    d.kh = s->kind; d.kv = kv; d.kw = kw;
    d.h = s->h; d.v = s->v; d.w = s->w;
    d.R = (s->flags & GPP_ST_HAS_LOC) ? s->loc : st_localization(s->kind, s->h, s->min_rho);
    d.cv = (s->flags & GPP_ST_CV) ? 1 : 0;
    d.cv_dist = s->cv_dist;
    if(d.cv && (!is_valid(d.cv_dist) || d.cv_dist < 0)) invalid("Invalid 'dist' in CrossValidation structure");   // :912-913
    d.fh = d.fv = d.fw = d.fR = nullptr; d.cell_idx = d.obs_idx = nullptr;
    if(s->field && !s->kind_v && !s->kind_w && !s->field_v && !s->field_w) {   // one spatially varying structure: its own fields
        const gpp_field* f = (const gpp_field*)s->field;
        if(f->kind != s->kind) runtime("structure kind and field kind differ");
        if(f->uniform) {   // the same scales at every point: corr(p1, p2) is the scalar structure's (and symmetric): every fast path applies
            d.h = f->h[0]; d.v = f->v[0]; d.w = f->w[0]; d.R = f->R[0];
            return d;
        }
        d.fh = f->d_h.p; d.fv = f->d_v.p; d.fw = f->d_w.p; d.fR = f->d_R.p;
    }
    else if(s->field || s->field_v || s->field_w) {
        // MultipleStructure with spatially varying parts: gpp_bind_field materialises per-point scales; until then the marker below
        // tells the callers that the structure is spatial
        static float dummy = 0;
        d.fh = d.fv = d.fw = d.fR = &dummy;
    }
    return d;
}
// field index (nearest neighbour in the field's grid, src/api/structure.cpp:190) of every point of `pts`; NULL = identity
static const int* field_indices(const gpp_field* f, gpp_points* pts, DevBuf<int>& buf) {
    if(f->grid == pts) return nullptr;
    if(f->grid->n == pts->n && f->grid->type == pts->type) {
        f->grid->ensure_host_fields(); pts->ensure_host_fields();
        if(f->grid->lats == pts->lats && f->grid->lons == pts->lons) return nullptr;
    }
    pts->to_device();
    buf.get(pts->n);
    gpp_nearest_device(f->grid, pts->d_x.p, pts->d_y.p, pts->d_z.p, pts->n, 1, buf.p);
    return buf.p;
}
// per-point scale out[i] = src[idx[i]] (idx NULL: identity) or the constant `value` (src NULL)
__global__ void k_gather_scale(const float* __restrict__ src, const int* __restrict__ idx, float value, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = src ? src[idx ? idx[i] : i] : value;
}
void gpp_bind_field(DevStructure& d, const gpp_structure* s, gpp_points* bgrid, gpp_points* points, DevBuf<int>& cbuf, DevBuf<int>& obuf) {
    if(!s->field && !s->field_v && !s->field_w) return;
    if(!d.fh) return;   // a uniform field: gpp_resolve_structure made it a scalar structure
    if(s->kind_v || s->kind_w || s->field_v || s->field_w) {
        // MultipleStructure(sh, sv, sw) with spatially varying parts (structure.cpp:90-138): corr_h comes from sh with the elevation /
        // laf of p1 on both sides (its vertical factors are 1), corr_v from sv at zero horizontal distance, corr_w from sw likewise:
        // per point, h and the localization distance are sh's, v is sv's, w is sw's -- each looked up at the nearest point of ITS
        // grid.  Materialised once per call for [background points | observations]; the kernels index it like one field.
        DevBuf<float>&mh = g_mfw.mh, &mv = g_mfw.mv, &mw = g_mfw.mw, &mR = g_mfw.mR;
        DevBuf<int>& tmpi = g_mfw.tmpi;
        const int C = bgrid->n, S = points->n, n = C + S;
        mh.get(n); mv.get(n); mw.get(n); mR.get(n);
        const float Rconst = (s->flags & GPP_ST_HAS_LOC) ? s->loc : st_localization(s->kind, s->h, s->min_rho);
        auto fill = [&](const gpp_field* f, const float* src_c, float value, float* out) {
            // background points, then observations
            for(int part = 0; part < 2; ++part) {
                gpp_points* pts = part ? points : bgrid;
                const int m = part ? S : C;
                if(m == 0) continue;
                const int* idx = nullptr;
                if(f) {
                    if(f->grid->type != pts->type) invalid("the structure function's grid and the background must have the same coordinate type");
                    idx = field_indices(f, pts, tmpi);
                }
                hipLaunchKernelGGL(k_gather_scale, dim3((m + 255) / 256), dim3(256), 0, stream(), f ? src_c : (const float*)nullptr, idx, value, m, out + (part ? C : 0));
                GPP_HIP(hipGetLastError());
                GPP_HIP(hipStreamSynchronize(stream()));   // tmpi is reused by the next lookup
            }
        };
        const gpp_field* fh = (const gpp_field*)s->field;
        const gpp_field* fv = (const gpp_field*)s->field_v;
        const gpp_field* fw = (const gpp_field*)s->field_w;
        fill(fh, fh ? fh->d_h.p : nullptr, s->h, mh.p);
        fill(fh, fh ? fh->d_R.p : nullptr, Rconst, mR.p);
        fill(fv, fv ? fv->d_v.p : nullptr, s->v, mv.p);
        fill(fw, fw ? fw->d_w.p : nullptr, s->w, mw.p);
        d.fh = mh.p; d.fv = mv.p; d.fw = mw.p; d.fR = mR.p;
        d.cell_idx = nullptr;                         // identity: entry i belongs to background point i
        std::vector<int> id(S);
        for(int i = 0; i < S; i++) id[i] = C + i;     // observation i lives behind the background points
        obuf.upload(id.data(), id.size());
        d.obs_idx = obuf.p;
        return;
    }
    const gpp_field* f = (const gpp_field*)s->field;
    if(f->grid->type != bgrid->type) invalid("the structure function's grid and the background must have the same coordinate type");
    d.cell_idx = field_indices(f, bgrid, cbuf);
    const int* oi = field_indices(f, points, obuf);
    if(!oi) {   // identity for the observations too: materialise it (the kernel indexes obs_idx unconditionally)
        std::vector<int> id(points->n);
        for(int i = 0; i < points->n; i++) id[i] = i;
        obuf.upload(id.data(), id.size());
        oi = obuf.p;
    }
    d.obs_idx = oi;
}

extern "C" int gpp_field_create(gpp_points* grid, const float* h, const float* v, const float* w, int kind, float min_rho, gpp_field** out) {
    GPP_TRY
    if(!grid || !out || !h || !v || !w) invalid("NULL argument");
    if(!st_kind_ok(kind) || kind == GPP_SK_CRESSMAN) runtime("this structure function has no spatially varying form");
    ensure_device();
    std::unique_ptr<gpp_field> f(new gpp_field);
    f->grid = grid; f->n = grid->n; f->kind = kind; f->min_rho = min_rho;
    f->h.assign(h, h + grid->n); f->v.assign(v, v + grid->n); f->w.assign(w, w + grid->n);
    f->R.resize(grid->n);
    for(int i = 0; i < grid->n; i++) f->R[i] = st_localization(kind, f->h[i], min_rho);   // localization_distance(h), e.g. structure.cpp:280-282
    f->uniform = grid->n > 0 && is_valid(f->h[0]) && is_valid(f->v[0]) && is_valid(f->w[0]) && f->h[0] >= 0 && f->v[0] >= 0 && f->w[0] >= 0;
    for(int i = 1; i < grid->n && f->uniform; i++) f->uniform = f->h[i] == f->h[0] && f->v[i] == f->v[0] && f->w[i] == f->w[0];
    f->d_h.upload(f->h.data(), grid->n); f->d_v.upload(f->v.data(), grid->n); f->d_w.upload(f->w.data(), grid->n); f->d_R.upload(f->R.data(), grid->n);
    GPP_HIP(hipStreamSynchronize(stream()));
    *out = f.release();
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_field_destroy(gpp_field* f) {
    GPP_TRY
    delete f;
    return GPP_OK;
    GPP_CATCH
}
// scalar structure at one location of a spatially varying one (nearest neighbour of (lat, lon) in the field's grid)
static gpp_structure structure_at(const gpp_structure* s, float lat, float lon) {
    gpp_structure t = *s;
    if(s->field_v || s->field_w || (s->field && (s->kind_v || s->kind_w))) {   // MultipleStructure with spatially varying parts
        auto at = [&](const gpp_field* f) {
            int idx = -1;
            if(gpp_points_nearest_neighbour(f->grid, &lat, &lon, 1, 1, &idx) != GPP_OK || idx < 0) runtime("structure function grid is empty");
            return idx;
        };
        if(s->field) { const gpp_field* f = (const gpp_field*)s->field; const int k = at(f); t.h = f->h[k]; t.min_rho = f->min_rho; t.loc = f->R[k]; t.flags |= GPP_ST_HAS_LOC; }
        if(s->field_v) { const gpp_field* f = (const gpp_field*)s->field_v; t.v = f->v[at(f)]; }
        if(s->field_w) { const gpp_field* f = (const gpp_field*)s->field_w; t.w = f->w[at(f)]; }
        t.field = t.field_v = t.field_w = nullptr;
        return t;
    }
    if(!s->field) return t;
    const gpp_field* f = (const gpp_field*)s->field;
    int idx = -1;
    if(gpp_points_nearest_neighbour(f->grid, &lat, &lon, 1, 1, &idx) != GPP_OK || idx < 0) runtime("structure function grid is empty");
    t.field = nullptr;
    t.h = f->h[idx]; t.v = f->v[idx]; t.w = f->w[idx]; t.min_rho = f->min_rho;
    t.loc = f->R[idx]; t.flags |= GPP_ST_HAS_LOC;
    return t;
}
extern "C" int gpp_structure_min_rho(int kind, float h, float hmax, float* min_rho) {
    GPP_TRY
    if(!min_rho) invalid("NULL");
    if(!st_kind_ok(kind)) runtime("unknown structure function kind");
    if(kind != GPP_SK_CRESSMAN && is_valid(hmax) && hmax < 0) invalid("hmax must be >= 0");
    if(!is_valid(h) || h < 0) invalid("h must be >= 0");
    float r = 0.0013f;   // structure.cpp:5
    if(is_valid(hmax)) switch(kind) {
        case GPP_SK_BARNES: r = (float)std::exp(std::pow((double)(hmax / h), 2) / -2); break;                       // :154-155
        case GPP_SK_SOAR: r = (1 + hmax / h) * expf(-hmax / h); break;                                                // :328-329
        case GPP_SK_TOAR: r = (float)((1 + hmax / h + std::pow((double)(hmax / h), 2) / 3) * expf(-hmax / h)); break; // :478-479
        case GPP_SK_POWERLAW: r = (float)(1 / (1 + 0.5 * std::pow((double)(hmax / h), 2))); break;                    // :629-630
        default: break;
    }
    *min_rho = r;
    return GPP_OK;
    GPP_CATCH
}
static float loc_dist(const gpp_structure* s) { return gpp_resolve_structure(s).R; }
extern "C" int gpp_structure_localization_distance(const gpp_structure* s, float lat, float lon, float* dist) {
    GPP_TRY
    if(!s) invalid("structure is NULL");
    gpp_structure t = structure_at(s, lat, lon);
    *dist = loc_dist(&t);
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_structure_corr(const gpp_structure* s, const float p1[7], const float p2[7], int background, float* rho) {
    GPP_TRY
    if(!s) invalid("structure is NULL");
    gpp_structure t = structure_at(s, p1[5], p1[6]);
    DevStructure d = gpp_resolve_structure(&t);
    DevBuf<float> out;
    out.get(1);
    hipLaunchKernelGGL(k_structure_corr, dim3(1), dim3(1), 0, stream(), d, make_float4(p1[0], p1[1], p1[2], p1[3]), p1[4],
                       make_float4(p2[0], p2[1], p2[2], p2[3]), p2[4], background, out.p);
    GPP_HIP(hipGetLastError());
    GPP_HIP(hipMemcpyAsync(rho, out.p, sizeof(float), hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

void gpp_oi_drain_pending();
static int oi_full_impl(gpp_points* bgrid, const float* background, const float* bvariance,
                        gpp_points* points, const float* obs, const float* obs_variance,
                        const float* background_at_points, const float* bvariance_at_points,
                        const gpp_structure* st, int max_points, int allow_extrapolation,
                        float* out, float* out_variance, int mem) {
    GPP_TRY
    // argument checks of oi.cpp:152-186 (sizes are implied by the handles; pointers must be present)
    if(max_points < 0) invalid("max_points must be >= 0");
    if(!bgrid || !points) invalid("grid/points handle is NULL");
    if(bgrid->type != points->type)
        invalid("Both background and observations points must be of same coordinate type (lat/lon or x/y)");
    const int C = bgrid->n, S = points->n;
    if(C > 0 && (!background || !out)) invalid("background/out is NULL");
    if(S > 0 && (!obs || !obs_variance || !background_at_points)) invalid("observation arrays are NULL");
    ensure_device();
    g_stats = gpp_oi_stats();
    g_stats.cells = C;
    if(C == 0) return GPP_OK;

    OiWorkspace& ws = g_ws;
    InField f_bg, f_bvar, f_obs, f_ov, f_pbg, f_bvp;
    OutField f_out, f_var;
    // Host arrays of a large grid (what a gridpp script passes, swig/vector.i:42-55,172-180): the background goes up, and the analysis comes
    // down, in bands of tile rows beside the first pass when the call takes the tile kernel (`banded` below) -- 64 MB each way over PCIe ran
    // strictly before and behind a 4.3 ms kernel.  Until that is known the upload is only prepared.
    const bool band_candidate = !(mem & GPP_MEM_DEVICE) && C >= (1 << 20) && S > 0 && !path_env("GPP_OI_NO_BANDS");
    const bool band_f64 = (mem & GPP_HOST_F64) != 0;     // float64 host arrays (numpy's default): a band goes up as doubles and is cast on the upload stream
    Staged<double> wide_bg, wide_bv;                     // (returned to the pool at the end of the call: everything that reads them is behind the call's last synchronisation)
    if(band_candidate) {
        f_bg.d = f_bg.staged.get(C);
        if(bvariance) f_bvar.d = f_bvar.staged.get(C);
    }
    else {
        f_bg.bind(background, C, mem);
        f_bvar.bind(bvariance, C, mem);
    }
    f_out.bind(out, C, mem);
    f_var.bind(out_variance, C, mem);

    if(S == 0) {   // oi.cpp:189-190: return the background
        GPP_HIP(hipMemcpyAsync(f_out.d, f_bg.d, sizeof(float) * C, hipMemcpyDeviceToDevice, stream()));
        if(f_var.d) {
            if(f_bvar.d) GPP_HIP(hipMemcpyAsync(f_var.d, f_bvar.d, sizeof(float) * C, hipMemcpyDeviceToDevice, stream()));
            else invalid("bvariance is required when the variance is requested and there are no observations");
        }
        f_out.finish(); f_var.finish();
        GPP_HIP(hipStreamSynchronize(stream()));
        return GPP_OK;
    }
    f_obs.bind(obs, S, mem);
    f_ov.bind(obs_variance, S, mem);
    f_pbg.bind(background_at_points, S, mem);
    f_bvp.bind(bvariance_at_points, S, mem);

    bgrid->to_device();
    gpp_obs_index* ix = gpp_build_obs_index(points);

    if(!ws.e0) { GPP_HIP(hipEventCreate(&ws.e0)); GPP_HIP(hipEventCreate(&ws.e1)); GPP_HIP(hipEventCreate(&ws.eu)); }
    constexpr size_t SB = 8 + 80 + 2 * GPP_NSLOT;   // status block, in 8-byte words
    if((mem & GPP_ASYNC) && !g_aslots[0].h) {
        // everything a deferred call needs, on the FIRST call that asks for one (which itself runs synchronously: no memory of its geometry yet):
        // page-locked status slots, events, the two streams beside the library stream -- milliseconds that must not fall into a later call
        for(int k = 0; k < ASYNC_SLOTS; k++) {
            AsyncSlot& sl = g_aslots[k];
            GPP_HIP(hipHostMalloc((void**)&sl.h, SB * sizeof(unsigned long long), hipHostMallocDefault));
            GPP_HIP(hipEventCreate(&sl.e0)); GPP_HIP(hipEventCreate(&sl.e1)); GPP_HIP(hipEventCreateWithFlags(&sl.ec, hipEventDisableTiming));
        }
        for(int k = 0; k < 2; k++) { GPP_HIP(hipEventCreateWithFlags(&ws.ev_pack[k], hipEventDisableTiming)); GPP_HIP(hipEventCreateWithFlags(&ws.ev_join2[k], hipEventDisableTiming)); }
        (void)stream2(); (void)stream3();
    }
    // The workspace of this call -- the slot's observation block and status block -- is bound BELOW, behind the decision whether the call is deferred:
    // a call that is not completes the deferred calls of the thread first, and one of those may need its synchronous re-run, i.e. a nested
    // oi_full_impl that flips the slot, reallocates the slot buffers for its own observation count and overwrites the statistics.
    // Nothing of the workspace is touched, and no pointer into it is held, before that drain.
    int wsl = 0;
    DevBuf<float4>* pgeo_b = nullptr; DevBuf<float4>* oaux_b = nullptr; DevBuf<float4>* saux_b = nullptr;
    DevBuf<unsigned long long>* status_b = nullptr;
    int* d_ints = nullptr, *d_err = nullptr, *d_fb_count = nullptr, *d_big_count = nullptr, *d_huge_count = nullptr;
    unsigned long long* d_counters = nullptr;
    auto launch_pack = [&](hipStream_t on) {   // the observation block of this call (also clears the status block)
        hipLaunchKernelGGL(k_pack_obs, dim3((S + 255) / 256), dim3(256), 0, on, S, ix->d_sgeo.p, ix->d_pos.p, ix->d_olaf.p,
                           f_obs.d, f_ov.d, f_pbg.d, f_bvp.d, 1, pgeo_b->p, oaux_b->p, saux_b->p, status_b->p, (int)SB);
        GPP_HIP(hipGetLastError());
    };

    // register-tile size of the solve: 32 rows (max_points <= 32, the common case) or 62 rows (everything up to 62
    // usable observations per grid point; one member cell per factorisation)
    const int N = (max_points > 0 && max_points <= 32) ? 32 : 62;
    OiArgs a = OiArgs();
    a.gx = bgrid->d_x.p; a.gy = bgrid->d_y.p; a.gz = bgrid->d_z.p; a.gelev = bgrid->d_elev.p; a.glaf = bgrid->d_laf.p;
    a.bg = f_bg.d; a.bvar = f_bvar.d; a.out = f_out.d; a.out_var = f_var.d;
    a.C = C; a.ny = bgrid->ny; a.nx = bgrid->nx;
    a.tiled2d = (bgrid->nx > 0 && (long)bgrid->ny * bgrid->nx == C) ? 1 : 0;
    a.wshift = 3;
    if(a.tiled2d) {
        a.wshift = gpp_tile_wshift(bgrid);
        const int tw = 1 << a.wshift, th = 64 >> a.wshift;
        a.tiles_x = (a.nx + tw - 1) / tw; a.ntiles = a.tiles_x * ((a.ny + th - 1) / th);
    }
    else { a.tiles_x = 0; a.ntiles = (C + 63) / 64; }
    a.s.smeta = ix->d_smeta.p; a.s.bin_start = ix->d_bin_start.p;
    a.ogeo = ix->d_ogeo.p;
    a.S = S; a.s.axis_a = ix->axis_a; a.s.axis_b = ix->axis_b; a.s.nbx = ix->nbx; a.s.nby = ix->nby;
    a.s.amin = ix->amin; a.s.bmin = ix->bmin; a.s.inv_s = ix->inv_s;
    a.s.st = gpp_resolve_structure(st);
    gpp_bind_field(a.s.st, st, bgrid, points, ws.cell_idx, ws.obs_idx);
    a.s.max_points = max_points;
    { const double occ = (double)S / ((double)ix->nbx * ix->nby);
      const int kk = (max_points > 0 && max_points <= N) ? max_points : N;
      a.s.q0 = std::max(1, std::min(8, (int)std::ceil(0.5 * (std::sqrt(1.6 * kk / std::max(occ, 1e-3)) - 1.0))));
      if(path_env("GPP_Q0")) a.s.q0 = atoi(path_env("GPP_Q0"));
      // expected distance of the kk-th nearest observation: k_oi_union looks for its bulk disc below 1.5x that and then
      // visits rings 0.15x wide
      const double r_k = std::sqrt(kk / (3.14159265358979 * std::max(occ, 1e-3))) / ix->inv_s;
      a.s.ring_r0 = (float)(1.5 * r_k); a.s.ring_dr = (float)(0.15 * r_k); }
    a.allow_extrap = allow_extrapolation ? 1 : 0;
    a.s.K = (max_points > 0 && max_points <= N) ? max_points : N;
    a.debug = timing_env("GPP_OI_DEBUG") ? atoi(timing_env("GPP_OI_DEBUG")) : 0;

    // Cholesky needs a symmetric positive definite P+R: true for every kernel on distances and for the even vertical / laf
    // kernels (Barnes, Powerlaw, Linear); Cressman / SOAR / TOAR factors on SIGNED elevation / laf differences make P
    // non-symmetric (structure.cpp:35-64), and a truncated kernel can be indefinite -> pivoted LU like the reference.
    auto odd = [](int k) { return k == GPP_SK_CRESSMAN || k == GPP_SK_SOAR || k == GPP_SK_TOAR; };
    const bool spatial = a.s.st.fh != nullptr;   // per-point length scales: P is not symmetric (corr(p1, p2) uses p1's scales)
    bool use_lu = spatial || (a.s.st.v != 0 && odd(a.s.st.kv)) || (a.s.st.w != 0 && odd(a.s.st.kw)) || path_env("GPP_OI_FORCE_LU");
    int err = 0;
    unsigned long long counters[80 + 2 * GPP_NSLOT];
    const bool plain = a.s.st.kh == GPP_SK_BARNES && a.s.st.kv == GPP_SK_BARNES && a.s.st.kw == GPP_SK_BARNES && !a.s.st.cv;
    hipStream_t cur = stream();   // the stream the launch helpers below use: the library stream, or stream2() for the list passes that run beside the first pass
    auto launch_k_oi = [&](const bool lu) {   // k_oi over a.nrun tiles (all, or the fallback list of k_oi_union)
        const dim3 grid((a.nrun + 3) / 4), block(256);
        if(spatial) {   // (plain: three Barnes factors with per-point scales -- the straight-line correlation code takes them per lane)
            if(N == 32) { if(plain) hipLaunchKernelGGL((k_oi<32, true, true, true>), grid, block, 0, cur, a); else hipLaunchKernelGGL((k_oi<32, true, false, true>), grid, block, 0, cur, a); }
            else hipLaunchKernelGGL((k_oi<62, true, false, true>), grid, block, 0, cur, a);
        }
        else if(N == 32) {
            if(lu) hipLaunchKernelGGL((k_oi<32, true, false, false>), grid, block, 0, cur, a);
            else if(plain) hipLaunchKernelGGL((k_oi<32, false, true, false>), grid, block, 0, cur, a);
            else hipLaunchKernelGGL((k_oi<32, false, false, false>), grid, block, 0, cur, a);
        }
        else {
            if(lu) hipLaunchKernelGGL((k_oi<62, true, false, false>), grid, block, 0, cur, a);
            else hipLaunchKernelGGL((k_oi<62, false, false, false>), grid, block, 0, cur, a);
        }
        GPP_HIP(hipGetLastError());
    };
    // Cholesky path with the 32-row tile: the cells whose observation set no other cell of their work item shares (all of them on
    // rough terrain with elevation-dependent rho) are parked by k_oi and solved by k_oi_pairs at twice the occupancy (128 B of
    // selection per cell; beyond PAIR_PARK_MAX bytes of it k_oi solves them itself as before).  k_oi over a work list leaves
    // the counts of the cells it does not visit alone: they are cleared first.
    constexpr size_t PAIR_PARK_MAX = (size_t)8 << 30;
    const bool pairs_ok = N == 32 && !spatial && (size_t)C * 128 <= PAIR_PARK_MAX && !path_env("GPP_OI_NO_PAIRS");
    auto park_on = [&](const bool clear) {
        a.pair_sel = ws.pair_sel.get((size_t)C * 32); a.pair_n = ws.pair_n.get((size_t)C);
        if(clear) GPP_HIP(hipMemsetAsync(a.pair_n, 0, (size_t)C * sizeof(int), cur));
    };
    auto launch_pairs = [&]() {
        const dim3 grid((a.ntiles + 3) / 4), block(256);
        if(plain) hipLaunchKernelGGL(k_oi_pairs<true>, grid, block, 0, cur, a);
        else hipLaunchKernelGGL(k_oi_pairs<false>, grid, block, 0, cur, a);
        GPP_HIP(hipGetLastError());
        a.pair_sel = nullptr; a.pair_n = nullptr;
    };
    const int* h_ints = nullptr;   // (the page-locked mirror of the status block: bound with the workspace below)
    auto fetch = [&]() {   // the whole status block in one copy
        GPP_HIP(hipMemcpyAsync(ws.h_status, status_b->p, SB * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream()));
        GPP_HIP(hipStreamSynchronize(stream()));
        err = h_ints[0];
        memcpy(counters, ws.h_status + 8, sizeof(counters));
    };
    // one factorisation per tile (k_oi_union) when the system is symmetric and max_points fits the 32-column tile;
    // the tiles it declines, and every other configuration, run on k_oi (one factorisation per distinct selection)
    // (With elevation / laf dependent rho the cells of a tile agree less; on smooth terrain most tiles still fit, on
    //  white-noise elevations none does and the first pass costs a few per cent before the lists hand everything to k_oi.)
    const bool want_union = path_env("GPP_OI_UNION") ? atoi(path_env("GPP_OI_UNION")) != 0 : true;
    auto& memo = bgrid->union_memo;
    const bool memo_hit = memo.points_id == points->serial && memo.h == a.s.st.h && memo.v == a.s.st.v && memo.w == a.s.st.w && memo.kh == a.s.st.kh &&
                          memo.kv == a.s.st.kv && memo.kw == a.s.st.kw && memo.cv == a.s.st.cv && memo.max_points == max_points;
    const bool memo_says_no = memo_hit && memo.declined > 0.5f;   // more than half of the tiles went to k_oi last time: skip the first pass
    // (the 62-row form only for 33 <= max_points <= 62: with max_points = 0 a cell may hold more than any tile kernel can)
    const bool use_union = !use_lu && (N == 32 || (max_points > 32 && max_points <= 62)) && want_union && !memo_says_no && !path_env("GPP_OI_NO_UNION");
    const int WPB = N == 32 ? UnionCfg<32>::WPB : UnionCfg<64>::WPB;   // work items (waves) per workgroup of k_oi_union
    // ---- the banded host path (see band_candidate above) --------------------------------------------------------------------------------
    a.tile0 = 0; a.tile_n = a.ntiles;
    const bool banded = band_candidate && a.tiled2d && use_union && N == 32 && !path_env("GPP_OI_PAIR_TILES") && a.ntiles / std::max(a.tiles_x, 1) >= 2 * OiWorkspace::MAXB;
    if(band_candidate && !banded) {   // whole arrays after all
        f_bg.bind(background, C, mem);
        f_bvar.bind(bvariance, C, mem);
    }
    if(banded && band_f64) { wide_bg.get(C); if(bvariance) wide_bv.get(C); }
    // The analysis comes down per band only into page-locked memory (the Python mirror's result arrays, gpp_host_alloc): a copy into pageable
    // memory blocks the host until it is done, i.e. until the band's kernel has run -- nothing behind it would be enqueued in time.
    bool band_down = false, zero_copy_out = false;
    if(banded) {
        hipPointerAttribute_t at;
        band_down = hipPointerGetAttributes(&at, out) == hipSuccess && at.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        if(band_down && out_variance) { band_down = hipPointerGetAttributes(&at, out_variance) == hipSuccess && at.type == hipMemoryTypeHost; (void)hipGetLastError(); }
        // Round 6, second step: a page-locked result array is WRITTEN BY THE KERNELS THEMSELVES (it is mapped into the device's address space: the
        // stores of a tile travel over PCIe as posted writes while the kernel goes on) -- no download at all.  The per-band device-to-host copies
        // of the first step turned out to be copy KERNELS of the runtime (`__amd_rocclr_copyBuffer`) that do not run beside the first pass but
        // between its workgroups: every band's launch grew by the duration of the copy it overlapped with (tools/host_path_trace.py,
        // profiles/r06_host_path_trace.txt).  GPP_OI_BANDS_COPY_DOWN keeps that form for the comparison.
        if(band_down && !path_env("GPP_OI_BANDS_COPY_DOWN")) {
            void* dp = nullptr; void* dv = nullptr;
            bool okp = hipHostGetDevicePointer(&dp, out, 0) == hipSuccess && dp != nullptr;
            if(okp && out_variance) okp = hipHostGetDevicePointer(&dv, out_variance, 0) == hipSuccess && dv != nullptr;
            (void)hipGetLastError();
            if(okp) {
                zero_copy_out = true;
                band_down = false;
                a.out = static_cast<float*>(dp);
                if(out_variance) a.out_var = static_cast<float*>(dv);
            }
        }
        if(!ws.ev_band0) GPP_HIP(hipEventCreateWithFlags(&ws.ev_band0, hipEventDisableTiming));
        for(int b = 0; b < OiWorkspace::MAXB; b++) if(!ws.ev_up[b]) { GPP_HIP(hipEventCreateWithFlags(&ws.ev_up[b], hipEventDisableTiming)); GPP_HIP(hipEventCreateWithFlags(&ws.ev_k[b], hipEventDisableTiming)); }
    }
    struct BandGuard {   // whatever ends the call: no copy of it stays in flight on the side streams (into arrays the caller may free)
        bool on;
        ~BandGuard() { if(on) { (void)hipStreamSynchronize(stream2()); (void)hipStreamSynchronize(stream3()); (void)hipStreamSynchronize(stream4()); } }
    } band_guard{banded};
    bool bands_down_done = false;   // the band downloads of this call are enqueued: the end of the call patches the declined tiles instead of copying everything
    // cells with more usable observations than the 62-row tile holds are listed: symmetric systems go to k_oi_big (Cholesky, up
    // to BIG_N observations), what that kernel cannot hold and every listed cell of a non-symmetric or spatially varying
    // structure to k_oi_huge (pivoted elimination in HBM scratch, no capacity of its own)
    const bool big_ok = N == 62;
    auto run_huge = [&](const int* d_list, const int* d_count, const int ncells) {
        // scratch for the worst case of this call: every observation a candidate, max_points (or all of them) selected
        size_t kcap = 1; while(kcap < (size_t)S) kcap <<= 1;
        const size_t ncap = (max_points > 0) ? (size_t)std::min(max_points, S) : (size_t)S;
        const size_t per_wg = ncap * (ncap + 2) * sizeof(double) + kcap * sizeof(unsigned long long);
        size_t budget = (size_t)16 << 30;   // 16 GB of the 288 GB for this rarely used path
        if(path_env("GPP_OI_HUGE_BUDGET_MB")) budget = (size_t)atol(path_env("GPP_OI_HUGE_BUDGET_MB")) << 20;
        if(per_wg > budget) runtime("optimal_interpolation: a grid point may select more observations than the scratch budget of the general kernel holds (set max_points, or raise GPP_OI_HUGE_BUDGET_MB)");
        const int nwg = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)ncells, 512), budget / per_wg));
        a.huge_kcap = (int)kcap; a.huge_ncap = (int)ncap;
        a.huge_keys = ws.huge_keys.get((size_t)nwg * kcap);
        a.huge_mat = ws.huge_mat.get((size_t)nwg * ncap * (ncap + 2));
        if(spatial) hipLaunchKernelGGL(k_oi_huge<true>, dim3(nwg), dim3(256), 0, stream(), a, d_list, d_count);
        else hipLaunchKernelGGL(k_oi_huge<false>, dim3(nwg), dim3(256), 0, stream(), a, d_list, d_count);
        GPP_HIP(hipGetLastError());
    };
    bool ran_union = false, ran_overlap = false;
    // spatially varying Barnes structure on the tile path (round 6; no variance output: that needs a second substitution per cell)
    const bool sp_union = spatial && plain && N == 32 && !f_var.d && !path_env("GPP_OI_NO_SP_UNION");
    bool sp_failed = false, ran_sp = false;
    const bool async_req = (mem & GPP_ASYNC) && (mem & GPP_MEM_DEVICE) && !path_env("GPP_OI_NO_ASYNC");
    int async_slot = -1;
    // `steady`: a GPP_ASYNC call that will take the overlapped branch below (the conditions of `overlap`): it is deferred, and its observation
    // block is packed on the second stream, AHEAD of the first pass of the call before it.  Every other call first completes the deferred calls
    // of this thread (they share the workspace) and packs on the library stream.
    // (either the geometry remembers a list of declined tiles, or its last call declined none and left nothing to k_oi)
    bool steady = async_req && N == 32 && !f_out.host && !f_var.host && use_union && !use_lu && memo_hit && !path_env("GPP_OI_NO_OVERLAP") &&
                  ((memo.nlist > 0 && memo.list_ntiles == a.ntiles) || (memo.nlist == 0 && memo.declined == 0.0f && memo.leftover == 0));
    if(steady) {
        async_slot = (int)(g_async_seq++ % ASYNC_SLOTS);
        for(const PendingOi& q : g_pending) if(!q.done && q.slot == async_slot) { async_slot = -1; break; }   // (all slots in flight: this call runs synchronously)
        if(async_slot < 0) steady = false;
    }
    if(!steady) {
        gpp_oi_drain_pending();
        g_stats = gpp_oi_stats();      // (a re-run inside the drain left its own)
        g_stats.cells = C;
    }
    ws.slot ^= 1;
    wsl = ws.slot;
    pgeo_b = &ws.pgeo_s[wsl]; oaux_b = &ws.oaux_s[wsl]; saux_b = &ws.saux_s[wsl]; status_b = &ws.status_s[wsl];
    pgeo_b->get(S); oaux_b->get(S); saux_b->get(S);
    status_b->get(SB);
    if(!ws.h_status) GPP_HIP(hipHostMalloc((void**)&ws.h_status, (SB + 1) * sizeof(unsigned long long), hipHostMallocDefault));
    h_ints = reinterpret_cast<const int*>(ws.h_status);
    d_ints = reinterpret_cast<int*>(status_b->p);
    d_err = d_ints; d_fb_count = d_ints + 1; d_big_count = d_ints + 4; d_huge_count = d_ints + 5;
    d_counters = status_b->p + 8;
    a.s.pgeo = pgeo_b->p; a.oaux = oaux_b->p; a.saux = saux_b->p;
    a.s.scan_stats = timing_env("GPP_SCAN_STATS") ? d_counters + 2 : nullptr;
    a.err = d_err; a.counters = d_counters; a.tail_count = d_ints + 6;
    if(big_ok) {
        ws.big_list.get((size_t)C);
        a.big_list = ws.big_list.p; a.big_count = d_big_count;
    }
    if(!steady) { launch_pack(stream()); GPP_HIP(hipEventRecord(ws.e0, stream())); }
    int overlap_left = 0, overlap_new = 0, overlap_remembered = 0;
    const int* patch_l0 = nullptr; const int* patch_l1 = nullptr;
    int patch_n0 = 0, patch_n1 = 0;
    for(int attempt = 0; attempt < 2; ++attempt) {
        a.in_list = nullptr; a.in_count = nullptr; a.out_list = nullptr; a.out_count = nullptr; a.nrun = a.ntiles;
        ran_union = false; ran_overlap = false; ran_sp = false;
        if(use_union && !use_lu) {
            // worst case per list: every tile declined and split into 4 (level 1) resp. 16 (level 2) items
            const int SHORT_ITEMS = 3072;   // as many work items as the chip holds waves of this kernel
            // (the short-list pass writes up to 16 entries per declined tile, SHORT_ITEMS at most, whatever the number of tiles)
            ws.fb_list.get((size_t)a.ntiles); ws.fb_list2.get(2 * (size_t)a.ntiles + 64);
            ws.fb_list3.get(std::max<size_t>(4 * (size_t)a.ntiles, (size_t)SHORT_ITEMS) + 64);
            const dim3 block(64 * UnionCfg<32>::WPB);
            const bool pair_tiles = path_env("GPP_OI_PAIR_TILES") != nullptr;   // (off: measured 3 % slower than one tile per wave -- the waiting wave of a pair)
            a.pair_solo = path_env("GPP_OI_PAIR_SOLO") ? 1 : 0;
            auto launch_union = [&](const long items, const bool list) {   // one wave per work item
                const long nb = (items + WPB - 1) / WPB;
                const dim3 grid((unsigned)std::min<long>(nb, 0x7fffffffL));
                // (max_points 33 .. 48: the 48-column form at two waves per SIMD; 49 .. 62: the 64-column form)
                if(N != 32 && max_points <= 48 && !path_env("GPP_OI_NO_UNION48")) gpp_launch_union48(a, grid.x, plain, list, cur);
                else if(N != 32) gpp_launch_union64(a, grid.x, plain, list, cur);
                // (round 6: the first pass of the 32-column form shares one factorisation between the two tiles of a pair)
                else if(!list && pair_tiles) {
                    const dim3 pgrid((unsigned)std::min<long>(union_pair_count(a), 0x7fffffffL)), pblock(128);
                    if(plain) hipLaunchKernelGGL(k_oi_union_pair<true>, pgrid, pblock, 0, cur, a); else hipLaunchKernelGGL(k_oi_union_pair<false>, pgrid, pblock, 0, cur, a);
                }
                // (the first pass is persistent: a grid that fills the chip, every wave strides over the tiles)
                else if(plain) { if(list) hipLaunchKernelGGL((k_oi_union<true, true, 32>), grid, block, 0, cur, a); else hipLaunchKernelGGL((k_oi_union<true, false, 32>), UnionCfg<32>::persistent<true>() ? dim3(union_persist_grid<k_oi_union<true, false, 32>>(block.x, nb)) : grid, block, 0, cur, a); }
                else { if(list) hipLaunchKernelGGL((k_oi_union<false, true, 32>), grid, block, 0, cur, a); else hipLaunchKernelGGL((k_oi_union<false, false, 32>), grid, block, 0, cur, a); }
                GPP_HIP(hipGetLastError());
            };
            // The first pass over all tiles -- on the banded host path as one launch per band of tile rows: the band's rows of the background
            // (and of its variance) go up on the second stream, the band's launch waits for them, the band's rows of the analysis come down on
            // the third behind it.  Bands: 1 : 2 : 3 : 3 : 2 : 1 of the tile rows -- a short first upload and a short last download are what is
            // not hidden; every launch costs its last, partly filled round of workgroups (~45 us).  The tiles the first pass leaves to the
            // list passes are final only behind those: their cells are patched at the end of the call (k_gather_tiles).
            auto first_pass = [&]() {
                if(!banded) { launch_union(a.ntiles, false); return; }
                // (GPP_OI_BAND_SHARES="a,b,c,...": another split for A/B runs, at most MAXB bands)
                int share[OiWorkspace::MAXB] = {1, 2, 3, 3, 2, 1, 0, 0};
                int nband = 6, total_share = 12;
                if(const char* e = path_env("GPP_OI_BAND_SHARES")) {
                    nband = 0; total_share = 0;
                    for(const char* q = e; *q && nband < OiWorkspace::MAXB; ) { const int v = atoi(q); if(v > 0) { share[nband++] = v; total_share += v; } while(*q && *q != ',') q++; if(*q == ',') q++; }
                    if(nband == 0) { nband = 1; share[0] = 1; total_share = 1; }
                }
                const int th = 64 >> a.wshift, trows = a.ntiles / a.tiles_x;
                const hipStream_t sUp = stream2(), sDown = stream3(), sOdd = stream4();
                const bool two_streams = path_env("GPP_OI_BANDS_TWO_STREAMS") != nullptr;
                // (the bands alternate between the library stream and a fourth one: the last round of a band's workgroups fills a fraction of the
                //  chip, and in one stream the next band waits behind it -- six times ~45 us)
                GPP_HIP(hipEventRecord(ws.ev_band0, stream()));
                GPP_HIP(hipStreamWaitEvent(sUp, ws.ev_band0, 0));    // (the staging buffers' previous readers are behind the library stream)
                GPP_HIP(hipStreamWaitEvent(sOdd, ws.ev_band0, 0));   // (and so is the status block the kernels count in, cleared by k_pack_obs)
                int ty0 = 0, acc = 0;
                for(int b = 0; b < nband; b++) {
                    acc += share[b];
                    const int ty1 = b == nband - 1 ? trows : std::max(ty0 + 1, (int)((long)trows * acc / total_share));
                    const size_t r0 = (size_t)ty0 * th, r1 = std::min<size_t>((size_t)a.ny, (size_t)ty1 * th);
                    const size_t off = r0 * a.nx, cnt = (r1 - r0) * a.nx;
                    auto up = [&](const float* src, const float* dst, Staged<double>& wide) {
                        if(!band_f64) { GPP_HIP(hipMemcpyAsync(const_cast<float*>(dst) + off, src + off, cnt * sizeof(float), hipMemcpyHostToDevice, sUp)); return; }
                        // (the rounding of the reference's PyArray_CastToType, swig/vector.i:42-55, on the device: k_stage_f64)
                        GPP_HIP(hipMemcpyAsync(wide.p + off, reinterpret_cast<const double*>(src) + off, cnt * sizeof(double), hipMemcpyHostToDevice, sUp));
                        hipLaunchKernelGGL(k_stage_f64, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, sUp, (const double*)(wide.p + off), cnt, const_cast<float*>(dst) + off);
                        GPP_HIP(hipGetLastError());
                    };
                    up(background, f_bg.d, wide_bg);
                    if(bvariance) up(bvariance, f_bvar.d, wide_bv);
                    GPP_HIP(hipEventRecord(ws.ev_up[b], sUp));
                    const hipStream_t sK = ((b & 1) && two_streams) ? sOdd : stream();
                    GPP_HIP(hipStreamWaitEvent(sK, ws.ev_up[b], 0));
                    a.tile0 = ty0 * a.tiles_x; a.tile_n = (ty1 - ty0) * a.tiles_x;
                    cur = sK;
                    launch_union(a.tile_n, false);
                    cur = stream();
                    GPP_HIP(hipEventRecord(ws.ev_k[b], sK));
                    if(band_down) {
                        GPP_HIP(hipStreamWaitEvent(sDown, ws.ev_k[b], 0));
                        GPP_HIP(hipMemcpyAsync(out + off, f_out.d + off, cnt * sizeof(float), hipMemcpyDeviceToHost, sDown));
                        if(out_variance) GPP_HIP(hipMemcpyAsync(out_variance + off, f_var.d + off, cnt * sizeof(float), hipMemcpyDeviceToHost, sDown));
                    }
                    ty0 = ty1;
                }
                for(int b = std::max(0, nband - 2); b < nband; b++) GPP_HIP(hipStreamWaitEvent(stream(), ws.ev_k[b], 0));   // (what follows on the library stream -- the list passes, the read-back -- is behind every band, whichever stream ran the last ones)
                a.tile0 = 0; a.tile_n = a.ntiles;
                bands_down_done = band_down;
            };
            // The tiles it declined.  The usual case is a short list (or none): it is taken WITHOUT asking the host how long it
            // is -- the short-list pass and the k_oi pass behind it are launched with fixed small grids and read the lengths
            // on the device; a list too long for that grid is left untouched by both and handled after the one read-back of
            // the call.  (A host round trip here cost ~25 us per call.)  When the last call with this geometry had a long
            // list, the host asks first, as the two-level passes need its length for their grids anyway.
            // (a geometry whose last call left nothing to k_oi: not even its empty launch; the read-back says if that was wrong)
            const bool skip_k_oi = memo_hit && memo.leftover == 0;
            auto short_passes = [&](const int nitems) {
                // every declined tile straight to its sixteen 4-cell items (one pass; the latency of a pass, one lone work
                // item, is what a short list costs)
                a.in_list = ws.fb_list.p; a.in_count = d_fb_count; a.out_list = ws.fb_list3.p; a.out_count = d_fb_count + 2; a.level = 3;
                launch_union(nitems, true);
                // what is still left, one factorisation per distinct selection (grid-stride over the list)
                a.in_list = ws.fb_list3.p; a.in_count = d_fb_count + 2; a.out_list = nullptr; a.out_count = nullptr; a.nrun = 4 * 128;
                if(!skip_k_oi) launch_k_oi(false);   // (usually nothing is left: a small grid keeps the empty launch cheap)
            };
            auto long_passes = [&](const int n1) {
                // pass 2: the declined tiles as 4 items of 16 cells (smaller unions); a list too long for the split to pay is
                // forwarded whole by the kernel
                a.in_list = ws.fb_list.p; a.in_count = d_fb_count; a.out_list = ws.fb_list2.p; a.out_count = d_fb_count + 1; a.level = 1;
                launch_union(4 * (long)n1, true);
                // pass 3: the declined 16-cell items as 4 items of 4 cells (at most 4 n1 of them)
                a.in_list = ws.fb_list2.p; a.in_count = d_fb_count + 1; a.out_list = ws.fb_list3.p; a.out_count = d_fb_count + 2; a.level = 2;
                a.parent_count = d_fb_count;
                launch_union(16 * (long)n1, true);
                // pass 4: what is still left, one factorisation per distinct selection (grid-stride over the list)
                a.in_list = ws.fb_list3.p; a.in_count = d_fb_count + 2; a.out_list = nullptr; a.out_count = nullptr; a.nrun = 4 * 512;
                const bool pairs = pairs_ok && !skip_k_oi && 4 * (long)n1 > a.ntiles;   // (a short list: not worth the pass over every cell's count)
                if(pairs) park_on(true);
                if(!skip_k_oi) launch_k_oi(false);
                if(pairs) launch_pairs();
            };
            const bool expect_long = memo_hit && 16.0 * (double)memo.declined * (double)a.ntiles > (double)SHORT_ITEMS;
            // Round 5: the list passes BESIDE the first pass.  The tiles a geometry declines do not depend on the values (only on the
            // coordinates, the structure and which observations are usable), so the list of the last call is this call's list: it is kept
            // with the geometry (memo.list, one flag byte per tile), the first pass leaves the flagged tiles alone, and the list passes over
            // the REMEMBERED list are launched on a second stream at the same time -- their latency (one lone work item per pass: 55 us of a
            // 0.62 ms step at 500 rows per rank, 146 us of the 4.46 ms headline) disappears behind the first pass.  Nothing depends on the
            // memory being right: a flagged tile is solved by the list passes whatever the first pass would have made of it (all paths give
            // the same bits), and a tile the first pass declines that was NOT flagged arrives in its out_list as before, is taken by the serial
            // passes after the read-back and joins the remembered list.
            const bool overlap = memo_hit && memo.nlist > 0 && memo.list_ntiles == a.ntiles && !path_env("GPP_OI_NO_OVERLAP");
            const int n_remembered = overlap ? memo.nlist : 0;
            const int* const d_mcount = memo.count.p;
            auto remembered_passes = [&]() {   // the list passes over the REMEMBERED list, on `cur`
                if(16 * (long)n_remembered <= SHORT_ITEMS) {
                    a.in_list = memo.list.p; a.in_count = d_mcount; a.out_list = ws.fb_list3.p; a.out_count = d_fb_count + 2; a.level = 3;
                    launch_union(16 * (long)n_remembered, true);
                    a.in_list = ws.fb_list3.p; a.in_count = d_fb_count + 2; a.out_list = nullptr; a.out_count = nullptr; a.nrun = 4 * 128;
                    if(!skip_k_oi) launch_k_oi(false);
                }
                else {
                    a.in_list = memo.list.p; a.in_count = d_mcount; a.out_list = ws.fb_list2.p; a.out_count = d_fb_count + 1; a.level = 1;
                    launch_union(4 * (long)n_remembered, true);
                    a.in_list = ws.fb_list2.p; a.in_count = d_fb_count + 1; a.out_list = ws.fb_list3.p; a.out_count = d_fb_count + 2; a.level = 2;
                    a.parent_count = d_mcount;
                    launch_union(16 * (long)n_remembered, true);
                    a.in_list = ws.fb_list3.p; a.in_count = d_fb_count + 2; a.out_list = nullptr; a.out_count = nullptr; a.nrun = 4 * 512;
                    const bool pairs = pairs_ok && !skip_k_oi && 4 * (long)n_remembered > a.ntiles;
                    if(pairs) park_on(true);
                    if(!skip_k_oi) launch_k_oi(false);
                    if(pairs) launch_pairs();
                }
            };
            auto defer = [&]() -> int {
                // GPP_ASYNC in the steady state.  Three streams: B packs this call's observation block and runs its list passes -- AHEAD of the
                // first pass of the call before it, which still runs on A (the blocks and status blocks alternate between two workspace
                // slots) --, A runs the first passes back to back, C copies each call's status block into its page-locked slot once its first
                // pass (A) and its list passes (B) are done.  Nothing here waits for the host, and the host waits for nothing here.
                AsyncSlot& sl = g_aslots[async_slot];
                const hipStream_t sA = stream(), sB = stream2(), sC = stream3();
                // (the buffers of this workspace slot: the deferred call that used them last must have finished its first pass and its status copy)
                if(ws.slot_e1[wsl]) GPP_HIP(hipStreamWaitEvent(sB, ws.slot_e1[wsl], 0));
                if(ws.slot_ec[wsl]) GPP_HIP(hipStreamWaitEvent(sB, ws.slot_ec[wsl], 0));
                launch_pack(sB);
                GPP_HIP(hipEventRecord(ws.ev_pack[wsl], sB));
                cur = sB;
                if(n_remembered > 0) remembered_passes();
                GPP_HIP(hipEventRecord(ws.ev_join2[wsl], sB));
                cur = sA;
                GPP_HIP(hipStreamWaitEvent(sA, ws.ev_pack[wsl], 0));
                GPP_HIP(hipEventRecord(sl.e0, sA));
                a.skip_flags = n_remembered > 0 ? memo.flags.p : nullptr;
                a.out_list = ws.fb_list.p; a.out_count = d_fb_count;
                launch_union(a.ntiles, false);
                a.skip_flags = nullptr;
                GPP_HIP(hipEventRecord(sl.e1, sA));      // (one timestamp behind the first pass: every command between two first passes is a gap on the GPU)
                GPP_HIP(hipStreamWaitEvent(sC, sl.e1, 0));
                GPP_HIP(hipStreamWaitEvent(sC, ws.ev_join2[wsl], 0));
                GPP_HIP(hipMemcpyAsync(sl.h, status_b->p, SB * sizeof(unsigned long long), hipMemcpyDeviceToHost, sC));
                GPP_HIP(hipEventRecord(sl.ec, sC));
                ws.slot_e1[wsl] = sl.e1; ws.slot_ec[wsl] = sl.ec;
                PendingOi pc;
                pc.bgrid = bgrid; pc.background = background; pc.bvariance = bvariance; pc.points = points; pc.obs = obs; pc.obs_variance = obs_variance;
                pc.background_at_points = background_at_points; pc.bvariance_at_points = bvariance_at_points; pc.st = *st; pc.max_points = max_points;
                pc.allow_extrapolation = allow_extrapolation; pc.out = out; pc.out_variance = out_variance; pc.mem = mem & ~GPP_ASYNC;
                pc.slot = async_slot; pc.n_remembered = n_remembered; pc.skip_k_oi = skip_k_oi;
                pc.stats = g_stats;
                g_pending.push_back(pc);
                return GPP_OK;
            };
            if(overlap) {
                if(!ws.ev_fork) { GPP_HIP(hipEventCreateWithFlags(&ws.ev_fork, hipEventDisableTiming)); GPP_HIP(hipEventCreateWithFlags(&ws.ev_join, hipEventDisableTiming)); }
                if(steady) return defer();
                GPP_HIP(hipEventRecord(ws.ev_fork, stream()));             // (behind k_pack_obs, which also cleared the status block)
                a.skip_flags = memo.flags.p;
                a.out_list = ws.fb_list.p; a.out_count = d_fb_count;
                first_pass();                                              // pass 1 on the library stream
                GPP_HIP(hipEventRecord(ws.eu, stream()));
                a.skip_flags = nullptr;
                cur = stream2();
                GPP_HIP(hipStreamWaitEvent(cur, ws.ev_fork, 0));
                remembered_passes();
                GPP_HIP(hipEventRecord(ws.ev_join, cur));
                cur = stream();
                GPP_HIP(hipStreamWaitEvent(cur, ws.ev_join, 0));
                GPP_HIP(hipEventRecord(ws.e1, stream()));
                fetch();
                if(skip_k_oi && h_ints[3] > 0) {   // 4-cell items for k_oi after all (before the counts below are cleared)
                    a.in_list = ws.fb_list3.p; a.in_count = d_fb_count + 2; a.out_list = nullptr; a.out_count = nullptr; a.nrun = 4 * 512;
                    launch_k_oi(false);
                }
                const int n_new = h_ints[1];
                const int left_remembered = h_ints[3];
                if(n_new > 0) {   // tiles the first pass declined that the memory did not hold: the serial passes, as without the overlap
                    GPP_HIP(hipMemsetAsync(d_fb_count + 1, 0, 2 * sizeof(int), stream()));
                    if(16 * (long)n_new > SHORT_ITEMS) { long_passes(n_new); if(skip_k_oi) launch_k_oi(false); }
                    else {
                        a.in_list = ws.fb_list.p; a.in_count = d_fb_count; a.out_list = ws.fb_list3.p; a.out_count = d_fb_count + 2; a.level = 3;
                        launch_union(16 * (long)n_new, true);
                        a.in_list = ws.fb_list3.p; a.in_count = d_fb_count + 2; a.out_list = nullptr; a.out_count = nullptr; a.nrun = 4 * 512;
                        launch_k_oi(false);
                    }
                }
                if(n_new > 0 || (skip_k_oi && left_remembered > 0)) {
                    GPP_HIP(hipEventRecord(ws.e1, stream()));
                    fetch();
                }
                overlap_left = left_remembered + (n_new > 0 ? h_ints[3] : 0);
                overlap_new = n_new; overlap_remembered = n_remembered; ran_overlap = true;
            }
            else if(!expect_long) {
                if(steady) return defer();      // (nothing declined the last time: the first pass alone, deferred)
                a.out_list = ws.fb_list.p; a.out_count = d_fb_count;
                first_pass();                                              // pass 1: every tile
                GPP_HIP(hipEventRecord(ws.eu, stream()));
                // (a geometry that declined no tile the last time: not even the two empty launches; the read-back says if that was wrong)
                const bool expect_none = memo_hit && memo.declined == 0.0f;
                if(!expect_none) short_passes(SHORT_ITEMS);
                GPP_HIP(hipEventRecord(ws.e1, stream()));
                fetch();
                const int n1 = h_ints[1];
                if(16 * (long)n1 > SHORT_ITEMS || (expect_none && n1 > 0)) {   // a long list after all (or an unexpected one): nothing has touched it yet
                    if(16 * (long)n1 > SHORT_ITEMS) long_passes(n1); else short_passes(16 * n1);
                    GPP_HIP(hipEventRecord(ws.e1, stream()));
                    fetch();
                }
            }
            else {
                a.out_list = ws.fb_list.p; a.out_count = d_fb_count;
                first_pass();                                              // pass 1: every tile
                GPP_HIP(hipEventRecord(ws.eu, stream()));
                // A long list is expected, and the same geometry declines the same tiles: the two-level passes are launched for the length
                // of the last call (+ 1/8) without asking the host first -- they read the true length on the device.  Only if MORE tiles were
                // declined than those grids hold (other observations turned invalid) are the passes run again with the true length: every
                // pass writes the same values for the cells it handles, so that is merely slower.
                const int cap = path_env("GPP_OI_LONG_CAP") ? atoi(path_env("GPP_OI_LONG_CAP")) : memo.n1 + memo.n1 / 8 + 64;   // (the override: tests force the second run)
                // (the statistics counters as the first pass left them: a second run of the list passes must not count its cells twice)
                ws.status_snap.get(SB);
                GPP_HIP(hipMemcpyAsync(ws.status_snap.p, d_counters, (SB - 8) * sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream()));
                long_passes(cap);
                GPP_HIP(hipEventRecord(ws.e1, stream()));
                fetch();
                const int n1 = h_ints[1];
                if(n1 > cap) {
                    GPP_HIP(hipMemsetAsync(d_fb_count + 1, 0, 2 * sizeof(int), stream()));
                    GPP_HIP(hipMemcpyAsync(d_counters, ws.status_snap.p, (SB - 8) * sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream()));
                    long_passes(n1);
                    GPP_HIP(hipEventRecord(ws.e1, stream()));
                    fetch();
                }
            }
            if(!overlap && skip_k_oi && h_ints[3] > 0) {   // 4-cell items for k_oi after all
                a.in_list = ws.fb_list3.p; a.in_count = d_fb_count + 2; a.out_list = nullptr; a.out_count = nullptr; a.nrun = 4 * 512;
                launch_k_oi(false);
                GPP_HIP(hipEventRecord(ws.e1, stream()));
                fetch();
            }
            ran_union = true;
        }
        else if(sp_union && !sp_failed) {
            // spatially varying Barnes structure: one LU per tile (k_oi_union_sp), the tiles it declines to k_oi's pivoted LU as whole tiles
            ws.fb_list.get((size_t)a.ntiles);
            a.out_list = ws.fb_list.p; a.out_count = d_fb_count;
            a.tile0 = 0; a.tile_n = a.ntiles;
            hipLaunchKernelGGL(k_oi_union_sp<0>, dim3((unsigned)((a.ntiles + 1) / 2)), dim3(128), 0, stream(), a);
            GPP_HIP(hipGetLastError());
            GPP_HIP(hipEventRecord(ws.eu, stream()));
            GPP_HIP(hipEventRecord(ws.e1, stream()));
            fetch();
            const int n1 = h_ints[1];
            if(n1 > 0) {
                // the declined tiles as smaller items (smaller unions), as on the scalar path: a short list straight to its 4-cell items, a long one
                // through the 16-cell level; what is still left to k_oi's pivoted LU
                ws.fb_list2.get(2 * (size_t)a.ntiles + 64);
                ws.fb_list3.get(std::max<size_t>(4 * (size_t)a.ntiles, (size_t)3072) + 64);
                auto launch_sp_list = [&](const long items) {
                    const long nb = (items + UnionCfg<32>::WPB - 1) / UnionCfg<32>::WPB;
                    hipLaunchKernelGGL((k_oi_union<true, true, 32, true>), dim3((unsigned)std::min<long>(nb, 0x7fffffffL)), dim3(64 * UnionCfg<32>::WPB), 0, stream(), a);
                    GPP_HIP(hipGetLastError());
                };
                if(16 * (long)n1 <= 3072) {
                    a.in_list = ws.fb_list.p; a.in_count = d_fb_count; a.out_list = ws.fb_list3.p; a.out_count = d_fb_count + 2; a.level = 3;
                    launch_sp_list(16 * (long)n1);
                }
                else {
                    a.in_list = ws.fb_list.p; a.in_count = d_fb_count; a.out_list = ws.fb_list2.p; a.out_count = d_fb_count + 1; a.level = 1;
                    launch_sp_list(4 * (long)n1);
                    a.in_list = ws.fb_list2.p; a.in_count = d_fb_count + 1; a.out_list = ws.fb_list3.p; a.out_count = d_fb_count + 2; a.level = 2;
                    a.parent_count = d_fb_count;
                    launch_sp_list(16 * (long)n1);
                }
                a.in_list = ws.fb_list3.p; a.in_count = d_fb_count + 2; a.out_list = nullptr; a.out_count = nullptr; a.nrun = 4 * 512;
                launch_k_oi(true);
                GPP_HIP(hipEventRecord(ws.e1, stream()));
                fetch();
            }
            ran_sp = true;
        }
        else {
            const bool pairs = pairs_ok && !use_lu;
            if(pairs) park_on(false);    // (k_oi visits every cell)
            launch_k_oi(use_lu);
            if(pairs) launch_pairs();
            GPP_HIP(hipEventRecord(ws.e1, stream()));
            fetch();
        }
        // (with the list passes beside the first pass its out_list holds only the tiles the memory did not know: overlap_new of them)
        const int n_listed = ran_union ? h_ints[1] : 0;             // entries of ws.fb_list written by this call's first pass
        const int nfb[3] = {ran_overlap ? overlap_remembered + overlap_new : n_listed, ran_union ? h_ints[2] : 0, ran_overlap ? overlap_left : (ran_union ? h_ints[3] : 0)};
        g_stats.fallback_tiles = ran_sp ? h_ints[1] : nfb[0];
        g_stats.fallback_subtiles = nfb[2];
        // (banded host path: the tiles whose cells the band downloads took too early -- everything the first pass did not finish itself)
        if(ran_overlap) { patch_l0 = memo.list.p; patch_n0 = overlap_remembered; patch_l1 = ws.fb_list.p; patch_n1 = overlap_new; }
        else { patch_l0 = ws.fb_list.p; patch_n0 = ran_union ? std::min(n_listed, a.ntiles) : 0; patch_l1 = nullptr; patch_n1 = 0; }
        if(ran_union) {
            if(!memo_hit) { memo.nlist = 0; memo.list_ntiles = 0; }
            memo.points_id = points->serial; memo.h = a.s.st.h; memo.v = a.s.st.v; memo.w = a.s.st.w; memo.max_points = max_points;
            memo.kh = a.s.st.kh; memo.kv = a.s.st.kv; memo.kw = a.s.st.kw; memo.cv = a.s.st.cv;
            memo.declined = (float)nfb[0] / (float)a.ntiles;
            memo.leftover = nfb[2];
            memo.n1 = nfb[0];
            // the declined tiles themselves, for the next call with this geometry (see `overlap` above).  The serial path lists EVERY declined
            // tile in ws.fb_list: the memory is rebuilt from it; the overlapped path lists only the tiles the memory did not hold: appended.
            // (room for every tile; nothing is launched when nothing changes -- the steady state of a repeated call)
            const bool keep_list = !path_env("GPP_OI_NO_OVERLAP");
            int n_add = 0;
            if(!ran_overlap) {
                memo.nlist = 0;
                if(keep_list && n_listed > 0) {
                    memo.list.get((size_t)a.ntiles); memo.flags.get((size_t)a.ntiles); memo.count.get(1);
                    GPP_HIP(hipMemsetAsync(memo.flags.p, 0, (size_t)a.ntiles, stream()));
                    memo.list_ntiles = a.ntiles;
                    n_add = std::min(n_listed, a.ntiles);
                }
            }
            else n_add = std::min(overlap_new, a.ntiles - memo.nlist);
            if(n_add > 0) {
                GPP_HIP(hipMemcpyAsync(memo.list.p + memo.nlist, ws.fb_list.p, (size_t)n_add * sizeof(int), hipMemcpyDeviceToDevice, stream()));
                hipLaunchKernelGGL(k_flag_tiles, dim3((n_add + 255) / 256), dim3(256), 0, stream(), (const int*)(memo.list.p + memo.nlist), n_add, a.ntiles, memo.flags.p);
                GPP_HIP(hipGetLastError());
                memo.nlist += n_add;
                ws.h_status[SB] = (unsigned long long)memo.nlist;      // (pinned: one word behind the mirror of the status block)
                GPP_HIP(hipMemcpyAsync(memo.count.p, ws.h_status + SB, sizeof(int), hipMemcpyHostToDevice, stream()));
                GPP_HIP(hipStreamSynchronize(stream()));
            }
        }
        if(big_ok) {
            const int nbig = h_ints[4];
            if(nbig > 0 && !use_lu && !path_env("GPP_OI_NO_BIG")) {
                const int nwg = std::min(nbig, 256);
                a.big_keys = ws.big_keys.get((size_t)nwg * BIG_CAND);
                a.big_mat = ws.big_mat.get((size_t)nwg * (BIG_N + 2) * BIG_N);
                ws.huge_list.get((size_t)nbig);
                a.huge_list = ws.huge_list.p; a.huge_count = d_huge_count;
                hipLaunchKernelGGL(k_oi_big, dim3(nwg), dim3(256), 0, stream(), a);
                GPP_HIP(hipGetLastError());
                GPP_HIP(hipEventRecord(ws.e1, stream()));
                fetch();
                g_stats.big_cells = nbig;
                const int nhuge = h_ints[5];
                if(nhuge > 0) {   // beyond BIG_N observations / BIG_CAND candidates
                    run_huge(ws.huge_list.p, d_huge_count, nhuge);
                    GPP_HIP(hipEventRecord(ws.e1, stream()));
                    fetch();
                }
            }
            else if(nbig > 0) {   // non-symmetric / spatially varying structure (or the retry after a non-positive pivot)
                run_huge(ws.big_list.p, d_big_count, nbig);
                GPP_HIP(hipEventRecord(ws.e1, stream()));
                fetch();
                g_stats.big_cells = nbig;
            }
        }
        if((err & ERR_SINGULAR) && ran_sp && !sp_failed) {   // a pivot of the unpivoted tile LU vanished: the whole call on the pivoted LU of k_oi
            sp_failed = true;
            g_stats.fallback_tiles = a.ntiles;
            GPP_HIP(hipMemsetAsync(status_b->p, 0, SB * sizeof(unsigned long long), stream()));
            continue;
        }
        if((err & ERR_SINGULAR) && !use_lu) {   // a pivot was not positive: redo the call with the pivoted LU, as LAPACK would
            use_lu = true;
            g_stats.fallback_tiles = a.ntiles;
            GPP_HIP(hipMemsetAsync(status_b->p, 0, SB * sizeof(unsigned long long), stream()));
            continue;
        }
        break;
    }
    if(bands_down_done && !use_lu && !(err & (ERR_SINGULAR | ERR_OVERFLOW))) {
        // the bands of the analysis are on their way down (third stream); what the list passes finished after the band of its tile left:
        // those tiles' cells, tile-major, through a page-locked block, put in place by the host
        const int np = patch_n0 + patch_n1;
        if(np > 0) {
            const size_t words = (size_t)np * 64 * (f_var.d ? 2 : 1) + (size_t)np;
            float* const d_patch = ws.patch.get(words);
            if(words > ws.h_patch_cap) {
                if(ws.h_patch) GPP_HIP(hipHostFree(ws.h_patch));
                ws.h_patch = nullptr; ws.h_patch_cap = 0;
                GPP_HIP(hipHostMalloc((void**)&ws.h_patch, words * sizeof(float), hipHostMallocDefault));
                ws.h_patch_cap = words;
            }
            float* const d_pv = f_var.d ? d_patch + (size_t)np * 64 : nullptr;
            int* const d_ids = reinterpret_cast<int*>(d_patch + (size_t)np * 64 * (f_var.d ? 2 : 1));
            hipLaunchKernelGGL(k_gather_tiles, dim3(np), dim3(64), 0, stream(), patch_l0, patch_n0, patch_l1, patch_n1, a.ny, a.nx, a.tiles_x, a.wshift,
                               (const float*)f_out.d, (const float*)f_var.d, d_patch, d_pv, d_ids);
            GPP_HIP(hipGetLastError());
            GPP_HIP(hipMemcpyAsync(ws.h_patch, d_patch, words * sizeof(float), hipMemcpyDeviceToHost, stream()));
        }
        GPP_HIP(hipStreamSynchronize(stream()));
        GPP_HIP(hipStreamSynchronize(stream3()));      // the bands are in `out`
        if(np > 0) {
            const float* const hp = ws.h_patch;
            const float* const hv = f_var.d ? hp + (size_t)np * 64 : nullptr;
            const int* const ids = reinterpret_cast<const int*>(hp + (size_t)np * 64 * (f_var.d ? 2 : 1));
            const int th = 64 >> a.wshift, tw = 1 << a.wshift;
            for(int i = 0; i < np; i++) {
                const int ty = ids[i] / a.tiles_x, tx = ids[i] - ty * a.tiles_x;
                for(int l = 0; l < 64; l++) {
                    const int y = ty * th + (l >> a.wshift), x = tx * tw + (l & (tw - 1));
                    if(y < a.ny && x < a.nx) { out[(size_t)y * a.nx + x] = hp[(size_t)i * 64 + l]; if(hv) out_variance[(size_t)y * a.nx + x] = hv[(size_t)i * 64 + l]; }
                }
            }
        }
    }
    else if(zero_copy_out) GPP_HIP(hipStreamSynchronize(stream()));   // (the kernels wrote the caller's page-locked arrays themselves)
    else if(f_out.host || f_var.host) {   // (device outputs: the stream is already idle after the status read-back)
        if(bands_down_done) GPP_HIP(hipStreamSynchronize(stream3()));   // (a second run of the call -- pivoted LU -- wrote everything again: the early bands must not land on top of the full copy)
        f_out.finish(); f_var.finish();
        GPP_HIP(hipStreamSynchronize(stream()));
    }
    float ms = 0;
    GPP_HIP(hipEventElapsedTime(&ms, ws.e0, ws.e1));
    g_stats.kernel_ms = ms;
    g_stats.union_kernel_ms = 0;
    if(ran_union || (ran_sp && !sp_failed)) GPP_HIP(hipEventElapsedTime(&g_stats.union_kernel_ms, ws.e0, ws.eu));
    g_stats.cells_updated = 0; g_stats.solves = 0;
    for(int k = 0; k < GPP_NSLOT; k++) { g_stats.cells_updated += (long long)counters[80 + 2 * k]; g_stats.solves += (long long)counters[81 + 2 * k]; }
    if(timing_env("GPP_SCAN_STATS")) fprintf(stderr, "[gpp] scan: %llu candidates iterated, %llu survivor-branch executions, %d tiles\n", counters[2], counters[3], a.ntiles);
#ifdef GPP_UNION_PROFILE
    { const char* nm[24] = {"cell loads", "bbox+init", "phase-1 loads", "rings", "phase 2", "classify", "union records", "P build", "eliminate", "export", "per-lane",
                            "own classify", "B1 wait", "merge", "B2 wait", "B3 wait", "bisection", "bulk evaluations", "end_bulk", "-", "-", "-", "-", "-"};
      const bool two = counters[20 + 24] != 0;
      for(int w = 0; w < 2; w++) {
          double tot = 0; for(int i = 0; i < 24; i++) tot += (double)counters[20 + 24 * w + i];
          if(tot == 0) continue;
          fprintf(stderr, "[gpp] union phases, wave %d (shader clocks per tile, %%):", w); for(int i = 0; i < 19; i++) fprintf(stderr, " %s %.0f (%.1f%%);", nm[i], counters[20 + 24 * w + i] / (double)a.ntiles * (two ? 2 : 1), 100.0 * counters[20 + 24 * w + i] / tot); fprintf(stderr, " total %.0f\n", tot / (double)a.ntiles * (two ? 2 : 1)); } }
#endif
#ifdef GPP_UNION_STATS
    fprintf(stderr, "[gpp] union: fallback reasons: slots %llu, union>40 %llu, extras>12 %llu, layout %llu, per-cell extras>6 %llu; per tile: insertions %.1f, evictions %.1f, candidates evaluated outside the bulk disc %.1f, records loaded in phase 2 %.1f\n",
                            counters[4], counters[5], counters[6], counters[7], counters[8], counters[9] / (double)a.ntiles, counters[10] / (double)a.ntiles, counters[11] / (double)a.ntiles, counters[12] / (double)a.ntiles);
#endif
    if(timing_env("GPP_SCAN_STATS")) { fprintf(stderr, "[gpp] wave-level insertions per tile histogram:"); for(int i = 0; i < 70; i++) fprintf(stderr, " %d:%llu", i, counters[4 + i]); fprintf(stderr, "\n"); }
    if(err & ERR_SINGULAR) runtime("optimal_interpolation: local (P+R) matrix is singular");
    if(err & ERR_OVERFLOW) runtime("optimal_interpolation: more usable observations at a grid point than the scratch of the general kernel was sized for");
    return GPP_OK;
    GPP_CATCH
}

// gridpp::optimal_interpolation / optimal_interpolation_full (src/api/oi.cpp:26-412).  With GPP_MEM_DEVICE | GPP_ASYNC: see PendingOi above.
extern "C" int gpp_optimal_interpolation_full(gpp_points* bgrid, const float* background, const float* bvariance,
                                              gpp_points* points, const float* obs, const float* obs_variance,
                                              const float* background_at_points, const float* bvariance_at_points,
                                              const gpp_structure* st, int max_points, int allow_extrapolation,
                                              float* out, float* out_variance, int mem) {
    const bool async_req = (mem & GPP_ASYNC) && (mem & GPP_MEM_DEVICE);
    const size_t before = g_pending.size();
    const int rc = oi_full_impl(bgrid, background, bvariance, points, obs, obs_variance, background_at_points, bvariance_at_points, st, max_points,
                                allow_extrapolation, out, out_variance, mem);
    if(async_req && g_pending.size() == before) {   // it ran synchronously -- or failed before anything was enqueued: queued as complete, so that EVERY
        PendingOi pc;                               // GPP_ASYNC call pairs with one gpp_wait(), which reports the same code and message again
        pc.done = true; pc.rc = rc; pc.stats = g_stats;
        if(rc != GPP_OK) pc.msg = gpp_last_error();
        g_pending.push_back(pc);
    }
    return rc;
}

// completes a deferred call in place: its status, statistics and message are kept with the queue entry
static int complete_pending(PendingOi& pc) {
    GPP_TRY
    if(pc.done) return pc.rc;
    AsyncSlot& sl = g_aslots[pc.slot];
    GPP_HIP(hipEventSynchronize(sl.ec));
    const int* const hi = reinterpret_cast<const int*>(sl.h);
    const int err = hi[0], n_new = hi[1], left = hi[3];
    if(err == 0 && n_new == 0 && !(pc.skip_k_oi && left > 0)) {
        pc.stats.fallback_tiles = pc.n_remembered; pc.stats.fallback_subtiles = left;
        GPP_HIP(hipEventElapsedTime(&pc.stats.kernel_ms, sl.e0, sl.e1));
        pc.stats.union_kernel_ms = pc.stats.kernel_ms;   // (the first pass is what stream A ran between the two timestamps; the list passes ran beside the call before)
        pc.stats.cells_updated = 0; pc.stats.solves = 0;
        const unsigned long long* const c = sl.h + 8;
        for(int k = 0; k < GPP_NSLOT; k++) { pc.stats.cells_updated += (long long)c[80 + 2 * k]; pc.stats.solves += (long long)c[81 + 2 * k]; }
        pc.done = true; pc.rc = GPP_OK;
        return GPP_OK;
    }
    // the status block asks for something the enqueued kernels did not do (a tile the memory did not hold, items for k_oi, an error flag):
    // the synchronous call does it -- and raises what is to be raised
    pc.done = true;    // (first: the blocking call below completes the deferred calls of the thread before it starts, and must not come back to this one)
    pc.rc = oi_full_impl(pc.bgrid, pc.background, pc.bvariance, pc.points, pc.obs, pc.obs_variance, pc.background_at_points, pc.bvariance_at_points, &pc.st,
                         pc.max_points, pc.allow_extrapolation, pc.out, pc.out_variance, pc.mem);
    pc.stats = g_stats;
    if(pc.rc != GPP_OK) pc.msg = gpp_last_error();
    return pc.rc;
    GPP_CATCH
}
// Completes the OLDEST pending GPP_ASYNC call of the calling thread: its results are in `out` when this returns GPP_OK; errors of that call
// (a singular local system, ...) are reported here, with gpp_last_error().  gpp_oi_last_stats() then describes that call.
extern "C" int gpp_wait(void) {
    GPP_TRY
    if(g_pending.empty()) invalid("gpp_wait: no asynchronous call is pending");
    PendingOi pc = g_pending.front();
    g_pending.erase(g_pending.begin());
    const int rc = complete_pending(pc);
    g_stats = pc.stats;
    if(rc != GPP_OK) gpp::set_error(pc.msg.c_str());
    return rc;
    GPP_CATCH
}
// number of GPP_ASYNC calls of the calling thread that gpp_wait() has not completed yet
extern "C" int gpp_pending(int* count) {
    if(!count) return gpp::fail(GPP_EINVAL, "count is NULL");
    *count = (int)g_pending.size();
    return GPP_OK;
}
// (library-internal: before a handle or a workspace of the thread goes away -- the calls are completed, their results and status wait for gpp_wait)
void gpp_oi_drain_pending() {
    for(PendingOi& pc : g_pending) (void)complete_pending(pc);     // (the entries stay queued, complete: every GPP_ASYNC call still pairs with its gpp_wait)
}
