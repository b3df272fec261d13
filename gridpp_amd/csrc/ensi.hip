// Ensemble optimal interpolation (EnSI) on MI355X (gfx950).
//
// Replaces the serial loop of gridpp::optimal_interpolation_ensi (src/api/oi_ensi.cpp:114-568).
//
// Per grid cell the reference forms the E x E matrix Pinv = Y^T R^-1 Y + (E-1) I, inverts it, takes the
// symmetric square root of (E-1) Pinv^-1 by eig_sym, and applies W = sqrt + w 1^T to the ensemble
// perturbations (:379-553).  With A = R^-1/2 Y (n x E, n <= max_points selected observations) all of that is a
// function of the n x n matrix B = A A^T = U S U^T  (c = E - 1):
//     P      = I/c + A^T [-(1/c)(cI + B)^-1] A          (Woodbury)
//     W_sym  = I   + A^T [U diag(-1 / (sqrt(c+S)(sqrt(c+S)+sqrt(c)))) U^T] A
//     w      = P Y^T R^-1 d = A^T (cI + B)^-1 R^-1/2 d
// so only an n x n (<= 32 x 32) symmetric eigenproblem is solved per cell instead of an E x E inverse plus an
// E x E eigenproblem.  One wavefront per tile of 64 cells: the candidate scan is shared with the OI kernel
// (oi_common.h), then the wave walks its cells; B, U live in LDS, the Jacobi sweeps use the round-robin
// ordering (n/2 disjoint rotations per step, applied to rows, then columns of B and U by all 64 lanes).
// The final member update reproduces the reference's float accumulation over k (:508-511) term by term.
#include <mutex>
#include "oi_common.h"
#include <algorithm>
#include <type_traits>

#pragma clang fp contract(off)
using namespace gpp;

#define EN 32          // max selected observations per cell
#define EMAXV 64       // max valid ensemble members (one lane per member)
#define BP (EN + 1)    // pitch of B / U (doubles)

#define ENSI_PARK_D 2208   // doubles parked per cell: U 1024 | U^T B U 1024 | sD 32 | sD (obs - yhat) 32 | rho 32 | (obs index, obs) 32 | (yhat, -) 32

struct EnsiArgs {
    const float *gx, *gy, *gz, *gelev, *glaf;
    const float* bg;          // [C][E]
    float* out;               // [C][E]
    int C, E, ny, nx, tiles_x, ntiles, tiled2d, wshift;
    ScanArgs s;
    const float4* ogeo;       // original order: x,y,z,elev
    const float4* oaux;       // original order: laf, obs, gYhat, sigma
    const float* gY;          // [S][nV] perturbations of the valid members (float)
    const int* validIdx;      // [nV]
    int valid_identity;       // every member is valid: validIdx[k] == k (no load needed to address a member)
    unsigned* sel;            // [ntiles][EN][64] scratch: the selections of every tile
    double* cpark;            // [tiles of the batch][64][ENSI_PARK_D] k_ensi_pair -> k_ensi_members: U, U^T B U, sD, r, rho of every cell
    int tile0;                // first tile of the batch
    unsigned* meta;           // [ntiles][64] k_ensi_scan -> k_ensi_pair: selection length | 0x100 if the reference sorted
    unsigned long long* hsigs;   // [ntiles][64] order-independent signature of every selection
    double* gram;             // [ntiles][2][EN*EN] scratch: Y Y^T of the current run(s) of equal selections
    int* big_list;            // cells with more usable observations than the 32-row tile holds (k_ensi_big), or NULL
    int* big_count;
    unsigned long long* big_keys;   // per workgroup of k_ensi_big: EBIG_CAND sorted candidate keys
    int* huge_list;           // cells k_ensi_big cannot hold (more than EBIG_CAND candidates): k_ensi_huge; count in big_count[1]
    unsigned long long* huge_keys;  // per workgroup of k_ensi_huge: huge_kcap candidate keys
    int huge_kcap;
    double* huge_mat;         // per workgroup of k_ensi_huge: Pinv (nV x nV) | eigenvectors (nV x nV) | five vectors of nV
    double jtol2;             // k_ensi_pair: the Jacobi sweeps stop at (off-diagonal norm)^2 <= jtol2 * c^2, c = nV - 1
    int debug;                // GPP_ENSI_DEBUG (timing experiments only): 1 no Jacobi, 2 no member update, 4 no B build, 8 no M_W
    int nV;
    int allow_extrap;
    int* err;
    unsigned long long* counters;
};

// member validity over the whole field (oi_ensi.cpp:187-201): flags[e] = 0 if any cell is invalid
__global__ void k_ensi_member_flags(const float* __restrict__ bg, long n, int E, int* __restrict__ flags) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n && !d_valid(bg[i])) flags[(int)(i % E)] = 0;
}
// gYhat / gY (oi_ensi.cpp:166-178): row mean with calc_statistic(Mean) semantics over ALL members
__global__ void k_ensi_obs_prep(const float* __restrict__ pbg, int S, int E, const int* __restrict__ validIdx, int nV,
                                float* __restrict__ gYhat, float* __restrict__ gY) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if(s >= S) return;
    const float* row = pbg + (long)s * E;
    float total = 0; int count = 0;
    for(int e = 0; e < E; e++) { float v = row[e]; if(d_valid(v)) { total += v; count++; } }
    float mean = count > 0 ? total / (float)count : NAN;
    gYhat[s] = mean;
    for(int k = 0; k < nV; k++) {
        float v = row[validIdx[k]];
        gY[(long)s * nV + k] = (d_valid(v) && d_valid(mean)) ? v - mean : v;
    }
}
// cells that found at least one usable observation (k_ensi_scan's meta: the count in the low byte): what the reference counts as
// "condition number error" when fewer than two members are valid (Pinv is the zero matrix: rcond <= 0, oi_ensi.cpp:386-390)
__global__ void k_ensi_count_cells(const unsigned* __restrict__ meta, long n, unsigned long long* __restrict__ out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for(; i < n; i += (long)gridDim.x * blockDim.x) c += (meta[i] & 0xffu) ? 1ull : 0ull;
    for(int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
__global__ void k_copy(const float* __restrict__ in, long n, float* __restrict__ out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = in[i];
}

__device__ __forceinline__ double d_rsq_nr(const double a) {   // 1/sqrt(a), a > 0, ~1 ulp
    double r = __builtin_amdgcn_rsq(a);
    r = r * (1.5 - 0.5 * a * r * r);
    r = r * (1.5 - 0.5 * a * r * r);
    return r;
}
__device__ __forceinline__ double d_rcp_nr(const double a) {   // 1/a, ~1 ulp
    double r = __builtin_amdgcn_rcp(a);
    r = r * (2.0 - a * r);
    r = r * (2.0 - a * r);
    return r;
}
// ---- 32 x 32 (K <= 32) double-precision products on the matrix cores ---------------------------------------------------
// v_mfma_f64_16x16x4_f64: A operand = A[row = lane & 15][k = lane >> 4], B operand = B[k = lane >> 4][col = lane & 15],
// C/D: col = lane & 15, row = (lane >> 4) + 4 * reg.  Four 16 x 16 accumulator tiles per wave cover the 32 x 32 result;
// the operands come straight out of LDS through the accessors (which return 0 outside the n x n matrices).  One MFMA
// replaces 16 scalar multiply-adds per lane: the n x n x n products of a cell shrink from ~1.8 k to ~0.1 k instructions.
typedef double v4d __attribute__((ext_vector_type(4)));
struct Acc32 { v4d t[2][2]; };
template <class FA, class FB>
__device__ __forceinline__ Acc32 mfma_32x32(const int lane, const int K, FA a_at, FB b_at) {
    Acc32 c;
    c.t[0][0] = c.t[0][1] = c.t[1][0] = c.t[1][1] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int r = lane & 15, kq = lane >> 4;
    for(int kk = 0; kk < K; kk += 4) {
        const int k = kk + kq;
        const double a0 = a_at(r, k), a1 = a_at(r + 16, k), b0 = b_at(k, r), b1 = b_at(k, r + 16);
        c.t[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c.t[0][0], 0, 0, 0);
        c.t[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c.t[0][1], 0, 0, 0);
        c.t[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c.t[1][0], 0, 0, 0);
        c.t[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c.t[1][1], 0, 0, 0);
    }
    return c;
}
// store the n x n part of an accumulator set into an LDS matrix of pitch BP
__device__ __forceinline__ void acc32_store(const Acc32& c, const int lane, const int n, double* M) {
#pragma unroll
    for(int ti = 0; ti < 2; ++ti)
#pragma unroll
        for(int tj = 0; tj < 2; ++tj)
#pragma unroll
            for(int r = 0; r < 4; ++r) {
                const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tj + (lane & 15);
                if(row < n && col < n) M[row * BP + col] = c.t[ti][tj][r];
            }
}

__device__ __forceinline__ double wave_sum_d(double v) {
    for(int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}


#include "ensi_pair.h"
#include "ensi_members3.h"

// ---------------------------------------------------------------------------------------------------------------------
// Grid points with more than 32 usable observations (max_points == 0 or > 32): one 256-thread workgroup per cell, the E x E formulation
// the reference itself uses (oi_ensi.cpp:379-553), E <= 64 valid members.  (Round 1 diagonalised Pinv with a cyclic Jacobi in LDS --
// `k_ensi_big`, like the LDS-resident tile kernel `k_ensi` taken out of the library in round 4; k_ensi_big_ns below replaced it.)
#define EBIG_CAND 8192
#define EP 65            // pitch of the 64 x 64 matrices (doubles)

// ---------------------------------------------------------------------------------------------------------------------
// k_ensi_big_ns: k_ensi_big with the eigen-decomposition replaced by a matrix-core iteration (round 3).  The cyclic Jacobi of
// k_ensi_big is a chain of ~400 rotation steps, three workgroup barriers and a handful of dependent LDS round trips each: 1.2 ms per
// grid point, fifty times the tile path.  What the path needs of Pinv = Y^T R^-1 Y + c I (SPD, eigenvalues >= c) is its inverse
// and its inverse square root (oi_ensi.cpp:398-421: P = inv(Pinv), sqrt((E - 1) P) by eig_sym) -- functions the coupled
// Newton-Schulz iteration delivers with nothing but 64 x 64 x 64 products:
//     Y_0 = Pinv / s,  Z_0 = I;   T = (3 I - Z Y) / 2,  Y <- Y T,  Z <- T Z;     Y -> (Pinv / s)^1/2,  Z -> (Pinv / s)^-1/2,
// s = (|Pinv|_inf + c) / 2, so that the spectrum of Pinv / s lies in (0, 2).  It converges quadratically once the smallest
// eigenvalue 2 c / (|Pinv|_inf + c) has grown to order one (a factor 2.25 per step before that): 10-16 steps of three products on
// v_mfma_f64_16x16x4, four waves sharing each product by rows.  The symmetric square root is unique, so the result is the one
// eig_sym gives, to the ~1e-13 the iteration is stopped at.  A grid point that does not converge (non-finite input) is listed for
// k_ensi_huge, whose Jacobi reproduces the reference's rcond <= 0 passthrough.
#ifndef GPP_ENSI_JTOL2
#define GPP_ENSI_JTOL2 1.6e-3   // stopping threshold of the Jacobi sweeps of k_ensi_pair, relative to c^2: |E| <= 0.040 c.  Round 3: 0.010 c with a series of
                                // three products (tools/ensi_soak.py: 1 value in 0.9 M outside the plain 1e-5 measure, worst 1.19e-5).  Round 4: five products
                                // and 0.020 c (NO value of the soak outside the plain measure), then seven float32 products in the tile layout and 0.040 c
                                // (ensi_pair.h, k_ensi_members: the same worst deviation as the converged sweeps, 2.5e-6; 0.050 c: 4.1e-6)
#endif
#define NSP 65           // pitch of the three 64 x 64 matrices (doubles)
#ifndef GPP_NS_TOL
#define GPP_NS_TOL 1e-25
#endif
#ifndef GPP_NS_EXTRA
#define GPP_NS_EXTRA 0
#endif
template <bool SPATIAL, bool FULL>   // FULL: 49..64 valid members (all four tile rows: strips); else the tiles on and above the diagonal
__global__ __launch_bounds__(256) void k_ensi_big_ns(EnsiArgs a) {
    extern __shared__ double ns_lds[];
    double* const M0 = ns_lds;                        // Y (first: candidate keys, then the Y chunk of the Pinv build); at the end W
    double* const M1 = ns_lds + 64 * NSP;             // Z
    double* const M2 = ns_lds + 2 * 64 * NSP;         // T
    unsigned long long* const s_key = reinterpret_cast<unsigned long long*>(ns_lds);   // [EBIG_CAND] = 64 KB <= the first two matrices
    __shared__ double s_t[64], s_w[64], s_X[64];
    __shared__ double s_off[256];
    __shared__ int s_n;
    __shared__ float s_val[64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const ScanArgs& sa = a.s;
    const int nV = a.nV, E = a.E;
    const int nlist = *a.big_count;
    unsigned long long* const gkeys = a.big_keys + (size_t)blockIdx.x * EBIG_CAND;
    const double c = (double)((float)(nV - 1));       // oi_ensi.cpp:383 (float product, delta = 1)
    auto block_sum = [&](double v) {
        s_off[tid] = v;
        __syncthreads();
        for(int off = 128; off > 0; off >>= 1) { if(tid < off) s_off[tid] += s_off[tid + off]; __syncthreads(); }
        const double r = s_off[0];
        __syncthreads();
        return r;
    };
    for(int li = blockIdx.x; li < nlist; li += gridDim.x) {
        const int cell = a.big_list[li];
        const float gx = a.gx[cell], gy = a.gy[cell], gz = a.gz[cell], ge = a.gelev[cell], gl = a.glaf[cell];
        DevStructure st = sa.st;
        if(SPATIAL) d_structure_at(st, st.cell_idx ? st.cell_idx[cell] : cell);
        __syncthreads();
        if(tid == 0) s_n = 0;
        __syncthreads();
        // ---- radius query + filter (valid observation, rho > 0: oi_ensi.cpp:213-237) -----------------------------------------
        const float R = st.R;
        const float pa = sa.axis_a == 0 ? gx : (sa.axis_a == 1 ? gy : gz), pb = sa.axis_b == 1 ? gy : (sa.axis_b == 2 ? gz : gx);
        const int bx0 = min(max((int)floorf((pa - R - sa.amin) * sa.inv_s) - 1, 0), sa.nbx - 1), bx1 = min(max((int)floorf((pa + R - sa.amin) * sa.inv_s) + 1, 0), sa.nbx - 1);
        const int by0 = min(max((int)floorf((pb - R - sa.bmin) * sa.inv_s) - 1, 0), sa.nby - 1), by1 = min(max((int)floorf((pb + R - sa.bmin) * sa.inv_s) + 1, 0), sa.nby - 1);
        const float lox = gx - R, hix = gx + R, loy = gy - R, hiy = gy + R, loz = gz - R, hiz = gz + R;
        for(int by = by0; by <= by1; ++by) {
            const int js = sa.bin_start[by * sa.nbx + bx0], je = sa.bin_start[by * sa.nbx + bx1 + 1];
            for(int j = js + tid; j < je; j += 256) {
                const float4 rec = sa.pgeo[j];   // x = NaN for an unusable observation: fails the box test
                if(!(rec.x > lox && rec.x < hix && rec.y > loy && rec.y < hiy && rec.z > loz && rec.z < hiz)) continue;
                const float2 met = sa.smeta[j];
                if(!(d_chord(rec.x, rec.y, rec.z, gx, gy, gz) <= R)) continue;
                const float rho = d_corr(st, gx, gy, gz, ge, gl, rec.x, rec.y, rec.z, rec.w, met.x, true);
                if(!(rho > 0.0f)) continue;
                const int k = atomicAdd(&s_n, 1);
                if(k < EBIG_CAND) s_key[k] = ((unsigned long long)__float_as_uint(rho) << 32) | (unsigned)(~__float_as_int(met.y));
            }
        }
        __syncthreads();
        const int ncand = s_n;
        const bool truncated = a.s.max_points > 0 && ncand > a.s.max_points;
        const int n = truncated ? a.s.max_points : ncand;
        if(ncand > EBIG_CAND) {   // more candidates than the LDS sort holds: k_ensi_huge takes the cell
            if(tid == 0) a.huge_list[atomicAdd(a.big_count + 1, 1)] = cell;
            continue;
        }
        if(n == 0 || nV <= 1) continue;
        // ---- the max_points largest keys by a radix select (oi_common.h) instead of sorting every candidate; only they are sorted
        int nsort = ncand;
        if(truncated) { block_select_largest(s_key, ncand, n, gkeys, reinterpret_cast<int*>(s_off), &s_n, tid); nsort = n; }
        // ---- order: rho descending (ties -> lower index) when the reference sorts, candidate (= index) order otherwise (:243-269)
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
                        const unsigned long long kx = truncated ? x : (x & 0xffffffffull), ky = truncated ? y : (y & 0xffffffffull);
                        const bool desc = (i & k) == 0;
                        if(desc ? (kx < ky) : (kx > ky)) { s_key[i] = y; s_key[ixj] = x; }
                    }
                }
                __syncthreads();
            }
        }
        for(int i = tid; i < n; i += 256) gkeys[i] = s_key[i];
        __threadfence_block();
        __syncthreads();
        // ---- Pinv = Y^T Rinv Y + c I on the matrix cores, t = Y^T Rinv d; chunks of 64 observations staged in LDS.  Wave wv accumulates
        //      rows [16 wv, 16 wv + 16) of Pinv: A(row a, k = i) = Y(i, a) rinv_i, B(k = i, col b) = Y(i, b); lane (kq, r16) ends with
        //      Pinv(16 wv + kq + 4 r, 16 tb + r16) in accT[tb][r].  (Round 1 summed them on the vector unit, sixteen entries a thread:
        //      49 of the 203 ms of the 500 x 500 x 50 case, 96 of 159 ms with 400 observations per grid point.)
        constexpr int YCP = 80;                                          // pitch of the Y chunk: the four k rows of an operand fall on different banks
        float* const yc = reinterpret_cast<float*>(s_key);               // [64][YCP] Y chunk
        double* const rinv = reinterpret_cast<double*>(yc + 64 * YCP);   // [64]
        double* const dvec = rinv + 64;                                  // [64]
        const int r16 = lane & 15, kq = lane >> 4;
        v4d accT[4];
#pragma unroll
        for(int tb = 0; tb < 4; ++tb) accT[tb] = (v4d){0.0, 0.0, 0.0, 0.0};
        double acct = 0.0;
        for(int i0 = 0; i0 < n; i0 += 64) {
            const int m = min(64, n - i0), m4 = (m + 3) & ~3;
            __syncthreads();
            for(int e = tid; e < m4 * 64; e += 256) {
                const int i = e >> 6, k = e & 63;
                float v = 0.0f;
                if(i < m && k < nV) {
                    const unsigned orig = ~(unsigned)(gkeys[i0 + i] & 0xffffffffull);
                    v = a.gY[(long)orig * nV + k];
                }
                yc[i * YCP + k] = v;
            }
            if(tid < 64) {
                double ri = 0.0, dv = 0.0;
                if(tid < m) {
                    const unsigned long long key = gkeys[i0 + tid];
                    const unsigned orig = ~(unsigned)(key & 0xffffffffull);
                    const float4 x4 = a.oaux[orig];                       // laf, obs, gYhat, sigma
                    const float s2 = x4.w * x4.w;                         // float product (oi_ensi.cpp:300)
                    ri = (double)__uint_as_float((unsigned)(key >> 32)) / (double)s2;
                    dv = (double)x4.y - (double)x4.z;
                }
                rinv[tid] = ri; dvec[tid] = dv;
            }
            __syncthreads();
            for(int ks = 0; 4 * ks < m4; ++ks) {
                const int io = 4 * ks + kq;
                const double aop = (double)yc[io * YCP + 16 * wv + r16] * rinv[io];
#pragma unroll
                for(int tb = 0; tb < 4; ++tb) accT[tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, (double)yc[io * YCP + 16 * tb + r16], accT[tb], 0, 0, 0);
            }
            if(tid < nV) for(int i = 0; i < m; ++i) acct = __builtin_fma((double)yc[i * YCP + tid] * rinv[i], dvec[i], acct);
        }
        __syncthreads();
        // ---- scaling: s = (|Pinv|_inf + c) / 2; Y_0 = Pinv / s (rows / columns beyond nV: a decoupled identity block -- with the scaled steps it does not stay the identity, it converges to it again), Z_0 = I
        double rowabs = 0.0;
#pragma unroll
        for(int tb = 0; tb < 4; ++tb)
#pragma unroll
            for(int r = 0; r < 4; ++r) {
                const int ai = 16 * wv + kq + 4 * r, bi = 16 * tb + r16;
                const double v = (ai < nV && bi < nV) ? accT[tb][r] + (ai == bi ? c : 0.0) : 0.0;
                accT[tb][r] = v;
                M2[ai * NSP + bi] = fabs(v);
            }
        if(tid < 64) s_t[tid] = (tid < nV) ? acct : 0.0;
        __syncthreads();
        if(tid < 64) { for(int k = 0; k < 64; ++k) rowabs += M2[tid * NSP + k]; }
        s_off[tid] = tid < 64 ? rowabs : 0.0;
        __syncthreads();
        for(int off = 128; off > 0; off >>= 1) { if(tid < off) s_off[tid] = fmax(s_off[tid], s_off[tid + off]); __syncthreads(); }
        const double sc = 0.5 * (s_off[0] + c), rsc = 1.0 / sc;
        __syncthreads();
#pragma unroll
        for(int tb = 0; tb < 4; ++tb)
#pragma unroll
            for(int r = 0; r < 4; ++r) {
                const int ai = 16 * wv + kq + 4 * r, bi = 16 * tb + r16;
                M0[ai * NSP + bi] = (ai < nV && bi < nV) ? accT[tb][r] * rsc : (ai == bi ? 1.0 : 0.0);
                M1[ai * NSP + bi] = ai == bi ? 1.0 : 0.0;
            }
        __syncthreads();
        // ---- coupled Newton-Schulz on the matrix cores.  Y, Z and T are symmetric (polynomials of one matrix), so only the 16 x 16
        //      tiles on and above the diagonal of the nt x nt tile grid (nt = ceil(nV / 16)) are computed for up to 48 members -- 3 of 16
        //      for 17..32 -- and stored twice, tile p of the list on wave p mod 4; with 49..64 members (nt = 4) every wave takes a strip of
        //      16 rows of the full product instead (one A operand per four MFMAs: measured faster than 10 tiles of two operands each).
        const int nt = (nV + 15) >> 4, npair = nt * (nt + 1) / 2, K4 = nt * 16;
        auto tile_of = [&](const int p, int& ti, int& tj) {   // p-th pair (ti <= tj), row by row
            int q = p; ti = 0;
            while(q >= nt - ti) { q -= nt - ti; ti++; }
            tj = ti + q;
        };
        auto tile_product = [&](const double* L, const double* Rm, const int ti, const int tj) {   // tile (ti, tj) of L Rm
            v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
            for(int kk = 0; kk < K4; kk += 4) {
                const int k = kk + kq;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(L[(16 * ti + r16) * NSP + k], Rm[k * NSP + 16 * tj + r16], acc, 0, 0, 0);
            }
            return acc;
        };
        bool converged = false;
        const int row0 = 16 * wv;
        // Scaled steps: the spectrum of M = Z Y lies in [lo, hi] (at the start [c / s, 2 - c / s], both bounds known), and with
        // T = sqrt(mu) (3 I - mu M) / 2 the next M is f(mu M), f(x) = x (3 - x)^2 / 4 -- Y Z^-1 = Pinv / s is untouched by the factor, so
        // the limit is the same.  mu = 3 / (lo + sqrt(lo hi) + hi) makes f(mu lo) = f(mu hi), the largest lower bound one step can
        // reach: it grows by 6.75 per step instead of 2.25 while it is small, and mu -> 1 as the bounds close on 1.
        double lo = c * rsc, hi = 2.0 - lo;
        int extra_left = GPP_NS_EXTRA;
        for(int it = 0; it < 64; ++it) {
            double res2 = 0.0;
            const double mu = 3.0 / (lo + sqrt(lo * hi) + hi), smu = sqrt(mu), hmu = 0.5 * mu;
            { const double x = mu * lo; lo = fmin(1.0, 0.25 * x * (3.0 - x) * (3.0 - x)); hi = 1.0; }
            if constexpr (FULL) {   // 49..64 members: wave wv owns rows [16 wv, 16 wv + 16) of every product (one A operand per four MFMAs)
                v4d t4[4];
#pragma unroll
                for(int j = 0; j < 4; ++j) t4[j] = (v4d){0.0, 0.0, 0.0, 0.0};
                for(int kk = 0; kk < 64; kk += 4) {   // Z Y
                    const int k = kk + kq;
                    const double av = M1[(row0 + r16) * NSP + k];
#pragma unroll
                    for(int j = 0; j < 4; ++j) t4[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, M0[k * NSP + 16 * j + r16], t4[j], 0, 0, 0);
                }
#pragma unroll
                for(int j = 0; j < 4; ++j)
#pragma unroll
                    for(int r = 0; r < 4; ++r) {
                        const int row = row0 + kq + 4 * r, col = 16 * j + r16;
                        const double d = (row == col ? 1.0 : 0.0) - t4[j][r];     // I - Z Y
                        res2 += d * d;
                        M2[row * NSP + col] = smu * ((row == col ? 1.5 : 0.0) - hmu * t4[j][r]);   // T = sqrt(mu) (3 I - mu Z Y) / 2
                    }
            }
            else {
                v4d t3[3];
                int tis[3], tjs[3];
#pragma unroll
                for(int u = 0; u < 3; ++u) {
                    const int p = wv + 4 * u;
                    tis[u] = tjs[u] = 0;
                    if(p < npair) { tile_of(p, tis[u], tjs[u]); t3[u] = tile_product(M1, M0, tis[u], tjs[u]); }   // Z Y
                }
#pragma unroll
                for(int u = 0; u < 3; ++u) {
                    if(wv + 4 * u < npair) {
#pragma unroll
                        for(int r = 0; r < 4; ++r) {
                            const int row = 16 * tis[u] + kq + 4 * r, col = 16 * tjs[u] + r16;
                            const double d = (row == col ? 1.0 : 0.0) - t3[u][r];     // I - Z Y
                            res2 += (tis[u] == tjs[u]) ? d * d : 2.0 * d * d;
                            const double tv = smu * ((row == col ? 1.5 : 0.0) - hmu * t3[u][r]);    // T = sqrt(mu) (3 I - mu Z Y) / 2
                            M2[row * NSP + col] = tv;
                            M2[col * NSP + row] = tv;
                        }
                    }
                }
            }
            res2 = block_sum(res2);    // (its barriers also publish T)
            if(res2 < GPP_NS_TOL) { if(extra_left-- <= 0) { converged = true; break; } }
            if(!(res2 == res2)) break;   // NaN: a non-finite matrix
            if constexpr (FULL) {
                v4d y4[4], z4[4];
#pragma unroll
                for(int j = 0; j < 4; ++j) { y4[j] = (v4d){0.0, 0.0, 0.0, 0.0}; z4[j] = (v4d){0.0, 0.0, 0.0, 0.0}; }
                for(int kk = 0; kk < 64; kk += 4) {   // Y T and T Z
                    const int k = kk + kq;
                    const double ay = M0[(row0 + r16) * NSP + k], at = M2[(row0 + r16) * NSP + k];
#pragma unroll
                    for(int j = 0; j < 4; ++j) {
                        y4[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, M2[k * NSP + 16 * j + r16], y4[j], 0, 0, 0);
                        z4[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(at, M1[k * NSP + 16 * j + r16], z4[j], 0, 0, 0);
                    }
                }
                __syncthreads();   // every wave has read Y and Z
#pragma unroll
                for(int j = 0; j < 4; ++j)
#pragma unroll
                    for(int r = 0; r < 4; ++r) {
                        const int row = row0 + kq + 4 * r, col = 16 * j + r16;
                        M0[row * NSP + col] = y4[j][r];
                        M1[row * NSP + col] = z4[j][r];
                    }
                __syncthreads();
            }
            else {
            // Y T and T Z as FULL products over the nt x nt tile grid (tile p = wv + 4 u, row by row; nt <= 3): no mirroring here.  The coupled
            // iteration is only stable with Y multiplied from the right and Z from the left by the SAME T (Higham, "Stable iterations for the
            // matrix square root"): until round 5 this path computed the tiles on and above the diagonal and stored their transposes below it,
            // i.e. (Y T) above and (T Y) below.  Equal in exact arithmetic -- in double the non-commuting part of the rounding errors then
            // grows by a factor of several hundred per step: with cond(Pinv) ~ 1e4 (observation sigmas x 0.01) the residual bottomed out at
            // |I - Z Y|_F ~ 2e-8, never reached the tolerance, the cell went to k_ensi_huge -- and converged cells carried errors of 1e-10
            // instead of 1e-13 (tools/ensi_hostile_soak.py found it; profiles/r05_ensi_illcond.txt).  T itself stays exactly symmetric
            // (Z Y is computed on the upper tiles and mirrored above): a perturbation of T that is the same in both products is harmless.
            v4d y3[3], z3[3];
            int tis[3], tjs[3];
            const int nfull = nt * nt;
#pragma unroll
            for(int u = 0; u < 3; ++u) {
                const int p = wv + 4 * u;
                tis[u] = tjs[u] = 0;
                if(p < nfull) {
                    tis[u] = p / nt; tjs[u] = p - tis[u] * nt;
                    y3[u] = tile_product(M0, M2, tis[u], tjs[u]);   // Y T
                    z3[u] = tile_product(M2, M1, tis[u], tjs[u]);   // T Z
                }
            }
            __syncthreads();   // every wave has read Y and Z
#pragma unroll
            for(int u = 0; u < 3; ++u) {
                if(wv + 4 * u < nfull) {
#pragma unroll
                    for(int r = 0; r < 4; ++r) {
                        const int row = 16 * tis[u] + kq + 4 * r, col = 16 * tjs[u] + r16;
                        M0[row * NSP + col] = y3[u][r];
                        M1[row * NSP + col] = z3[u][r];
                    }
                }
            }
            __syncthreads();
            }
        }
        if(!converged) {   // non-finite input (or no convergence): the general kernel decides (its Jacobi has the reference's rcond <= 0 passthrough)
            if(tid == 0) a.huge_list[atomicAdd(a.big_count + 1, 1)] = cell;
            continue;
        }
        // ---- w = P t = Z (Z t) / s ; W = sqrt(c P) + w 1^T = sqrt(c / s) Z + w 1^T  (oi_ensi.cpp:401-444) -> M0 --------------------
        if(tid < 64) {
            double u = 0.0;
            for(int k = 0; k < 64; ++k) u = __builtin_fma(M1[tid * NSP + k], s_t[k], u);
            s_X[tid] = u;
        }
        __syncthreads();
        if(tid < 64) {
            double wv2 = 0.0;
            for(int k = 0; k < 64; ++k) wv2 = __builtin_fma(M1[tid * NSP + k], s_X[k], wv2);
            s_w[tid] = wv2 * rsc;
        }
        __syncthreads();
        const double sq = sqrt(c * rsc);
        for(int e = tid; e < 64 * 64; e += 256) {
            const int ai = e >> 6, bi = e & 63;
            M0[ai * NSP + bi] = sq * M1[ai * NSP + bi] + s_w[ai];
        }
        // ---- ensemble side (oi_ensi.cpp:447-553): thread e < nV owns member e ---------------------------------------------------
        const int ek = (tid < nV) ? a.validIdx[tid] : 0;
        const float value = (tid < nV) ? a.bg[(long)cell * E + ek] : 0.0f;
        if(tid < 64) s_val[tid] = value;
        __syncthreads();
        float total = 0; int count = 0;
        for(int k = 0; k < nV; ++k) { const float v = s_val[k]; if(d_valid(v)) { total += v; count++; } }
        const float ensMean = total / (float)count;
        if(tid < nV) s_X[tid] = (double)value - (double)ensMean;
        __syncthreads();
        if(tid < nV) {
            float acc = 0.0f;
            for(int k = 0; k < nV; ++k) acc = (float)((double)acc + s_X[k] * M0[k * NSP + tid]);   // float += double product (:508-511)
            float currIncrement = acc;
            if(!a.allow_extrap) {   // :520-552; lY[e] is a LINEAR index into the n x nV column-major matrix
                const int li_ = tid % n, lk_ = tid / n;
                const unsigned oo = ~(unsigned)(gkeys[li_] & 0xffffffffull);
                const double lYe = (double)a.gY[(long)oo * nV + lk_];
                float maxInc = 0, minInc = 0;
                for(int i = 0; i < n; ++i) {
                    const unsigned oi_ = ~(unsigned)(gkeys[i] & 0xffffffffull);
                    const float4 x4 = a.oaux[oi_];
                    const float dv = (float)((double)x4.y - (lYe + (double)x4.z));
                    if(i == 0 || dv > maxInc) maxInc = dv;
                    if(i == 0 || dv < minInc) minInc = dv;
                }
                const double Xe = s_X[tid];
                const float memberIncrement = (float)((double)currIncrement - Xe);
                if(maxInc > 0 && memberIncrement > maxInc) currIncrement = (float)((double)maxInc + Xe);
                else if(maxInc < 0 && memberIncrement > 0) currIncrement = (float)(0.0 + Xe);
                else if(minInc < 0 && memberIncrement < minInc) currIncrement = (float)((double)minInc + Xe);
                else if(minInc > 0 && memberIncrement < 0) currIncrement = (float)(0.0 + Xe);
            }
            a.out[(long)cell * E + ek] = ensMean + currIncrement;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_ensi_huge: the same E x E formulation with no capacity of its own -- any number of candidates, selected observations
// and valid members (oi_ensi.cpp:187-201,244-261,379-421 have no limit either).  Candidate keys, Pinv, the eigenvectors and
// the per-member vectors live in HBM scratch sized by the host for the call; the sort is a bitonic network over global memory,
// the Jacobi sweeps take their rotations 32 pairs at a time.  One 256-thread workgroup per listed cell: the reference's O(E^3)
// per grid point, slowly -- its point is that such a call returns a result instead of an exception.
template <bool SPATIAL>
__global__ __launch_bounds__(256) void k_ensi_huge(EnsiArgs a, const int* __restrict__ list, const int* __restrict__ count) {
    __shared__ float s_yc[16384];                    // Y chunk: rows of nV floats
    __shared__ double s_rinv[64], s_dvec[64];
    __shared__ double s_cs[32], s_sn[32];
    __shared__ int s_p[32], s_q[32];
    __shared__ double s_off[256];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const ScanArgs& sa = a.s;
    const int nV = a.nV, E = a.E;
    const int nlist = *count;
    unsigned long long* const keys = a.huge_keys + (size_t)blockIdx.x * a.huge_kcap;
    double* const B = a.huge_mat + (size_t)blockIdx.x * (2 * (size_t)nV * nV + 5 * (size_t)nV);
    double* const V = B + (size_t)nV * nV;
    double* const v_t = V + (size_t)nV * nV;
    double* const v_w = v_t + nV;
    double* const v_X = v_w + nV;
    double* const v_sq = v_X + nV;
    double* const v_val = v_sq + nV;
    const double c = (double)((float)(nV - 1));       // oi_ensi.cpp:383 (float product, delta = 1)
    const int chunk = max(1, min(64, 16384 / max(nV, 1)));   // observations per staged chunk
    auto block_sum = [&](double v) {   // sum over the workgroup, returned to every thread
        s_off[tid] = v;
        __syncthreads();
        for(int off = 128; off > 0; off >>= 1) { if(tid < off) s_off[tid] += s_off[tid + off]; __syncthreads(); }
        const double r = s_off[0];
        __syncthreads();
        return r;
    };
    for(int li = blockIdx.x; li < nlist; li += gridDim.x) {
        const int cell = list[li];
        const float gx = a.gx[cell], gy = a.gy[cell], gz = a.gz[cell], ge = a.gelev[cell], gl = a.glaf[cell];
        DevStructure st = sa.st;
        if(SPATIAL) d_structure_at(st, st.cell_idx ? st.cell_idx[cell] : cell);
        if(tid == 0) s_n = 0;
        __syncthreads();
        // ---- radius query + filter (valid observation, rho > 0: oi_ensi.cpp:213-237) -----------------------------------------
        const float R = st.R;
        const float pa = sa.axis_a == 0 ? gx : (sa.axis_a == 1 ? gy : gz), pb = sa.axis_b == 1 ? gy : (sa.axis_b == 2 ? gz : gx);
        const int bx0 = min(max((int)floorf((pa - R - sa.amin) * sa.inv_s) - 1, 0), sa.nbx - 1), bx1 = min(max((int)floorf((pa + R - sa.amin) * sa.inv_s) + 1, 0), sa.nbx - 1);
        const int by0 = min(max((int)floorf((pb - R - sa.bmin) * sa.inv_s) - 1, 0), sa.nby - 1), by1 = min(max((int)floorf((pb + R - sa.bmin) * sa.inv_s) + 1, 0), sa.nby - 1);
        const float lox = gx - R, hix = gx + R, loy = gy - R, hiy = gy + R, loz = gz - R, hiz = gz + R;
        for(int by = by0; by <= by1; ++by) {
            const int js = sa.bin_start[by * sa.nbx + bx0], je = sa.bin_start[by * sa.nbx + bx1 + 1];
            for(int j = js + tid; j < je; j += 256) {
                const float4 rec = sa.pgeo[j];   // x = NaN for an unusable observation: fails the box test
                if(!(rec.x > lox && rec.x < hix && rec.y > loy && rec.y < hiy && rec.z > loz && rec.z < hiz)) continue;
                const float2 met = sa.smeta[j];
                if(!(d_chord(rec.x, rec.y, rec.z, gx, gy, gz) <= R)) continue;
                const float rho = d_corr(st, gx, gy, gz, ge, gl, rec.x, rec.y, rec.z, rec.w, met.x, true);
                if(!(rho > 0.0f)) continue;
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
        if(np2 > a.huge_kcap) { if(tid == 0) atomicOr(a.err, 1); __syncthreads(); continue; }   // (the host sizes the keys for every observation: cannot happen)
        if(n == 0 || nV <= 1) continue;
        // ---- order: rho descending (ties -> lower index) when the reference sorts, candidate (= index) order otherwise (:243-269)
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
        // ---- Pinv = Y^T Rinv Y + c I, t = Y^T Rinv d: chunks of observations staged in LDS, the sums kept in HBM (:380-385, :427-437)
        for(long e = tid; e < (long)nV * nV; e += 256) { const int ai = (int)(e / nV), bi = (int)(e - (long)ai * nV); B[e] = 0.0; V[e] = (ai == bi) ? 1.0 : 0.0; }
        for(int k = tid; k < nV; k += 256) v_t[k] = 0.0;
        __threadfence_block();
        __syncthreads();
        for(int i0 = 0; i0 < n; i0 += chunk) {
            const int m = min(chunk, n - i0);
            for(int e = tid; e < m * nV; e += 256) {
                const int i = e / nV, k = e - i * nV;
                const unsigned orig = ~(unsigned)(keys[i0 + i] & 0xffffffffull);
                s_yc[i * nV + k] = a.gY[(long)orig * nV + k];
            }
            if(tid < m) {
                const unsigned long long key = keys[i0 + tid];
                const unsigned orig = ~(unsigned)(key & 0xffffffffull);
                const float4 x4 = a.oaux[orig];                       // laf, obs, gYhat, sigma
                const float s2 = x4.w * x4.w;                         // float product (oi_ensi.cpp:300)
                s_rinv[tid] = (double)__uint_as_float((unsigned)(key >> 32)) / (double)s2;
                s_dvec[tid] = (double)x4.y - (double)x4.z;
            }
            __syncthreads();
            for(long e = tid; e < (long)nV * nV; e += 256) {
                const int ai = (int)(e / nV), bi = (int)(e - (long)ai * nV);
                double sacc = B[e];   // (one chain over all the observations, as in k_ensi_big)
                for(int i = 0; i < m; ++i) sacc = __builtin_fma((double)s_yc[i * nV + ai] * s_rinv[i], (double)s_yc[i * nV + bi], sacc);
                B[e] = sacc;
            }
            for(int k = tid; k < nV; k += 256) {
                double acct = v_t[k];
                for(int i = 0; i < m; ++i) acct = __builtin_fma((double)s_yc[i * nV + k] * s_rinv[i], s_dvec[i], acct);
                v_t[k] = acct;
            }
            __threadfence_block();
            __syncthreads();
        }
        for(int k = tid; k < nV; k += 256) B[(size_t)k * nV + k] += c;
        __threadfence_block();
        __syncthreads();
        // ---- cyclic Jacobi on the nV x nV matrix in HBM, round-robin pairs, 32 pairs at a time -------------------------------------
        const int mm = nV + (nV & 1), half = mm >> 1;
        for(int sweep = 0; sweep < 60; ++sweep) {
            double off2 = 0.0;
            // (the products of diagonals floored at 1e-300: a zero or denormal diagonal must read as "not converged", not as 0 / 0 = NaN = stop)
            // Stopping test in the SCALED measure sum (b_ij^2 / |b_ii b_jj|) <= 1e-26 (Demmel / Veselic: every eigenvalue and eigenvector of a positive
            // definite matrix to high RELATIVE accuracy).  Until round 5 the test was |off|_F <= 1e-11 trace: with observation sigmas x 0.01 the
            // spectrum of Pinv spans c ... 1e7 and the eigenvectors of the SMALL eigenvalues -- the ones that carry the weight in sqrt(c / D) --
            // were left with errors of 1e-5 (36 % of the float32 outputs off by an ulp, 1.5e-4 in the plain measure; tools/ensi_hostile_soak.py)
            for(long e = tid; e < (long)nV * nV; e += 256) { const int i = (int)(e / nV), j = (int)(e - (long)i * nV); if(j < i) { const double v = B[e]; off2 += v * v / fmax(fabs(B[(size_t)i * nV + i] * B[(size_t)j * nV + j]), 1e-300); } }
            off2 = block_sum(off2);
            if(!(off2 > 1e-26)) break;   // (also on NaN: a non-finite matrix)
            for(int step = 0; step < mm - 1; ++step) {
                for(int g0 = 0; g0 < half; g0 += 32) {
                    const int npair = min(32, half - g0);
                    if(tid < npair) {
                        const int t = g0 + tid;
                        int p, q;
                        if(t == 0) { p = mm - 1; q = step; }
                        else { p = (step + t) % (mm - 1); q = (step - t + (mm - 1)) % (mm - 1); }
                        if(p > q) { const int t_ = p; p = q; q = t_; }
                        double cs = 1.0, sn = 0.0;
                        if(q < nV) {
                            const double apq = B[(size_t)p * nV + q];
                            if(apq != 0.0) {
                                const double theta = (B[(size_t)q * nV + q] - B[(size_t)p * nV + p]) / (2.0 * apq);
                                const double t_ = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                                cs = 1.0 / sqrt(t_ * t_ + 1.0); sn = t_ * cs;
                                if(!(fabs(theta) < 1e150)) { cs = 1.0; sn = 0.0; }
                            }
                        }
                        else q = p;
                        s_p[tid] = p; s_q[tid] = q; s_cs[tid] = cs; s_sn[tid] = sn;
                    }
                    __syncthreads();
                    const int pk = tid >> 3;
                    const bool work = pk < npair && s_p[pk] != s_q[pk];
                    const int p = work ? s_p[pk] : 0, q = work ? s_q[pk] : 0;
                    const double cs = work ? s_cs[pk] : 1.0, sn = work ? s_sn[pk] : 0.0;
                    if(work)   // columns: B <- B J, V <- V J
                        for(int r = tid & 7; r < nV; r += 8) {
                            const double bp = B[(size_t)r * nV + p], bq = B[(size_t)r * nV + q];
                            const double vp = V[(size_t)r * nV + p], vq = V[(size_t)r * nV + q];
                            B[(size_t)r * nV + p] = cs * bp - sn * bq; B[(size_t)r * nV + q] = sn * bp + cs * bq;
                            V[(size_t)r * nV + p] = cs * vp - sn * vq; V[(size_t)r * nV + q] = sn * vp + cs * vq;
                        }
                    __threadfence_block();
                    __syncthreads();
                    if(work)   // rows: B <- J^T B
                        for(int cidx = tid & 7; cidx < nV; cidx += 8) {
                            const double bp = B[(size_t)p * nV + cidx], bq = B[(size_t)q * nV + cidx];
                            B[(size_t)p * nV + cidx] = cs * bp - sn * bq; B[(size_t)q * nV + cidx] = sn * bp + cs * bq;
                        }
                    __threadfence_block();
                    __syncthreads();
                }
            }
        }
        // eigenvalues D_k = B_kk; a non-finite or non-positive one is the reference's rcond <= 0 passthrough (oi_ensi.cpp:386-390)
        double bad = 0.0;
        for(int k = tid; k < nV; k += 256) { const double dk = B[(size_t)k * nV + k]; if(!(dk > 0.0) || isinf(dk)) bad = 1.0; }
        if(block_sum(bad) > 0.0) { if(tid == 0 && a.counters) atomicAdd(&a.counters[72], 1ull); continue; }   // ([72]: grid points left untouched, see gpp_ensi_last_stats)
        // ---- w = P t = V D^-1 V^T t ; W = V diag(sqrt(c / D)) V^T + w 1^T (:401-444) -> B --------------------------------------------
        for(int k = tid; k < nV; k += 256) {
            double u = 0.0;
            for(int r = 0; r < nV; ++r) u = __builtin_fma(V[(size_t)r * nV + k], v_t[r], u);   // (V^T t)_k
            v_sq[k] = sqrt(c / B[(size_t)k * nV + k]);
            v_X[k] = u / B[(size_t)k * nV + k];
        }
        __threadfence_block();
        __syncthreads();
        for(int k = tid; k < nV; k += 256) {
            double wv = 0.0;
            for(int r = 0; r < nV; ++r) wv = __builtin_fma(V[(size_t)k * nV + r], v_X[r], wv);
            v_w[k] = wv;
        }
        __threadfence_block();
        __syncthreads();
        for(long e = tid; e < (long)nV * nV; e += 256) {
            const int ai = (int)(e / nV), bi = (int)(e - (long)ai * nV);
            double sacc = 0.0;
            for(int k = 0; k < nV; ++k) sacc = __builtin_fma(V[(size_t)ai * nV + k] * v_sq[k], V[(size_t)bi * nV + k], sacc);
            B[e] = sacc + v_w[ai];
        }
        // ---- ensemble side (oi_ensi.cpp:447-553): thread e owns members e, e + 256, ... -------------------------------------------------
        for(int k = tid; k < nV; k += 256) v_val[k] = (double)a.bg[(long)cell * E + a.validIdx[k]];
        __threadfence_block();
        __syncthreads();
        float total = 0; int cnt = 0;
        for(int k = 0; k < nV; ++k) { const float v = (float)v_val[k]; if(d_valid(v)) { total += v; cnt++; } }
        const float ensMean = total / (float)cnt;
        for(int k = tid; k < nV; k += 256) v_X[k] = v_val[k] - (double)ensMean;
        __threadfence_block();
        __syncthreads();
        for(int e = tid; e < nV; e += 256) {
            float acc = 0.0f;
            for(int k = 0; k < nV; ++k) acc = (float)((double)acc + v_X[k] * B[(size_t)k * nV + e]);   // float += double product (:508-511)
            float currIncrement = acc;
            if(!a.allow_extrap) {   // :520-552; lY[e] is a LINEAR index into the n x nV column-major matrix
                const int li_ = e % n, lk_ = e / n;
                const unsigned oo = ~(unsigned)(keys[li_] & 0xffffffffull);
                const double lYe = (double)a.gY[(long)oo * nV + lk_];
                float maxInc = 0, minInc = 0;
                for(int i = 0; i < n; ++i) {
                    const unsigned oi_ = ~(unsigned)(keys[i] & 0xffffffffull);
                    const float4 x4 = a.oaux[oi_];
                    const float dv = (float)((double)x4.y - (lYe + (double)x4.z));
                    if(i == 0 || dv > maxInc) maxInc = dv;
                    if(i == 0 || dv < minInc) minInc = dv;
                }
                const double Xe = v_X[e];
                const float memberIncrement = (float)((double)currIncrement - Xe);
                if(maxInc > 0 && memberIncrement > maxInc) currIncrement = (float)((double)maxInc + Xe);
                else if(maxInc < 0 && memberIncrement > 0) currIncrement = (float)(0.0 + Xe);
                else if(minInc < 0 && memberIncrement < minInc) currIncrement = (float)((double)minInc + Xe);
                else if(minInc > 0 && memberIncrement < 0) currIncrement = (float)(0.0 + Xe);
            }
            a.out[(long)cell * E + a.validIdx[e]] = ensMean + currIncrement;
        }
        __syncthreads();
    }
}

#include "ensi_multi.h"

namespace {
struct EnsiWorkspace {
    DevBuf<float4> pgeo, oaux;
    DevBuf<float> gYhat, gY, gYm, obs0;
    DevBuf<int> flags, validIdx, err, cell_idx, obs_idx;
    DevBuf<unsigned> sel, meta;
    DevBuf<unsigned long long> hsigs;
    DevBuf<double> gram, cpark, huge_mat;
    DevBuf<unsigned long long> counters, big_keys, huge_keys;
    DevBuf<int> big_list, big_count;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};
thread_local EnsiWorkspace g_ews;
thread_local float g_ensi_ms = 0;
thread_local gpp_ensi_stats g_ensi_stats = {0, 0, 0, 0};
thread_local int g_ensi_converge = 0;
}

void gpp_release_ensi_workspace() {
    EnsiWorkspace& ws = g_ews;
    ws.cpark.release(); ws.gram.release(); ws.sel.release(); ws.huge_mat.release(); ws.big_keys.release(); ws.huge_keys.release();
}

#ifdef GPP_POISON
// Diagnostic build only (tools/hostile/build.sh, tools/ensi_hostile_soak.py, tools/ensi_multi_hostile_soak.py): every byte of the
// call-to-call workspaces of optimal_interpolation_ensi and optimal_interpolation_ensi_multi (they share one workspace) is set to `byte` --
// the park of k_ensi_pair (cpark), the two Gram matrices per tile, the parked selections with their metadata and signatures, the packed
// observations, the lists, counters and flags -- so that a kernel reading something THIS call did not write meets NaNs, absurd counts
// and negative indices instead of the previous call's values.
extern "C" int gpp_debug_poison_ensi_workspace(int byte) {
    GPP_TRY
    ensure_device();
    EnsiWorkspace& w = g_ews;
    w.pgeo.poison(byte); w.oaux.poison(byte);
    w.gYhat.poison(byte); w.gY.poison(byte); w.gYm.poison(byte); w.obs0.poison(byte);
    w.flags.poison(byte); w.validIdx.poison(byte); w.err.poison(byte); w.cell_idx.poison(byte); w.obs_idx.poison(byte);
    w.sel.poison(byte); w.meta.poison(byte); w.hsigs.poison(byte);
    w.gram.poison(byte); w.cpark.poison(byte); w.huge_mat.poison(byte);
    w.counters.poison(byte); w.big_keys.poison(byte); w.huge_keys.poison(byte);
    w.big_list.poison(byte); w.big_count.poison(byte);
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}
#endif

extern "C" int gpp_ensi_set_convergence(int to_convergence) {
    g_ensi_converge = to_convergence ? 1 : 0;
    return GPP_OK;
}


extern "C" int gpp_ensi_last_stats(gpp_ensi_stats* st) {
    GPP_TRY
    if(!st) invalid("NULL");
    *st = g_ensi_stats;
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_ensi_last_kernel_ms(float* ms) {
    GPP_TRY
    if(!ms) invalid("NULL");
    *ms = g_ensi_ms;
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_optimal_interpolation_ensi(gpp_points* bgrid, const float* background, int ne, gpp_points* points,
                                              const float* obs, const float* sigmas, const float* background_at_points,
                                              const gpp_structure* st, int max_points, int allow_extrapolation,
                                              float* out, int mem) {
    GPP_TRY
    if(max_points < 0) invalid("max_points must be >= 0");                                       // oi_ensi.cpp:123-124
    if(!bgrid || !points) invalid("grid/points handle is NULL");
    if(bgrid->type != points->type)
        invalid("Both background and observations points must be of same coorindate type (lat/lon or x/y)");
    if(!st) invalid("structure is NULL");
    if(ne < 0) invalid("negative ensemble size");
    const int C = bgrid->n, S = points->n, E = ne;
    ensure_device();
    g_ensi_ms = 0;
    g_ensi_stats = gpp_ensi_stats{(long long)C, 0, 0, 0.0f};
    if(C == 0 || E == 0) return GPP_OK;
    EnsiWorkspace& ws = g_ews;
    InField f_bg, f_obs, f_sig, f_pbg;
    OutField f_out;
    f_bg.bind(background, (size_t)C * E, mem);
    f_out.bind(out, (size_t)C * E, mem);
    const long nbg = (long)C * E;
    hipLaunchKernelGGL(k_copy, dim3((unsigned)((nbg + 255) / 256)), dim3(256), 0, stream(), f_bg.d, nbg, f_out.d);   // output = background (:146)
    GPP_HIP(hipGetLastError());
    if(S == 0) {   // oi_ensi.cpp:135-137
        f_out.finish();
        GPP_HIP(hipStreamSynchronize(stream()));
        return GPP_OK;
    }
    f_obs.bind(obs, S, mem);
    f_sig.bind(sigmas, S, mem);
    f_pbg.bind(background_at_points, (size_t)S * E, mem);
    bgrid->to_device();
    gpp_obs_index* ix = gpp_build_obs_index(points);

    // valid members (host list; E is small)
    std::vector<int> flags(E, 1);
    ws.flags.upload(flags.data(), E);
    hipLaunchKernelGGL(k_ensi_member_flags, dim3((unsigned)((nbg + 255) / 256)), dim3(256), 0, stream(), f_bg.d, nbg, E, ws.flags.p);
    GPP_HIP(hipMemcpyAsync(flags.data(), ws.flags.p, sizeof(int) * E, hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    std::vector<int> valid;
    for(int e = 0; e < E; e++) if(flags[e]) valid.push_back(e);
    const int nV = (int)valid.size();
    // k_ensi_pair takes any number of valid members; k_ensi_big_ns (more than 32 usable observations at a grid point) holds one member per lane
    const bool use_pair = true;
    if(nV == 0) { f_out.finish(); GPP_HIP(hipStreamSynchronize(stream())); return GPP_OK; }
    ws.validIdx.upload(valid.data(), nV);
    ws.gYhat.get(S); ws.gY.get((size_t)S * nV);
    hipLaunchKernelGGL(k_ensi_obs_prep, dim3((S + 127) / 128), dim3(128), 0, stream(), f_pbg.d, S, E, ws.validIdx.p, nV, ws.gYhat.p, ws.gY.p);
    ws.pgeo.get(S); ws.oaux.get(S);
    // oaux = (laf, obs, gYhat, sigma); only the observation itself must be valid (oi_ensi.cpp:235)
    hipLaunchKernelGGL(k_pack_obs, dim3((S + 255) / 256), dim3(256), 0, stream(), S, ix->d_sgeo.p, ix->d_pos.p, ix->d_olaf.p,
                       f_obs.d, f_sig.d, (const float*)ws.gYhat.p, (const float*)nullptr, 0, ws.pgeo.p, ws.oaux.p);
    GPP_HIP(hipGetLastError());
    ws.err.get(1); ws.counters.get(80 + 1024 * 32);
    GPP_HIP(hipMemsetAsync(ws.err.p, 0, sizeof(int), stream()));
    GPP_HIP(hipMemsetAsync(ws.counters.p, 0, sizeof(unsigned long long) * (80 + 1024 * 32), stream()));
    if(!ws.e0) { GPP_HIP(hipEventCreate(&ws.e0)); GPP_HIP(hipEventCreate(&ws.e1)); }

    EnsiArgs a = EnsiArgs();
    a.gx = bgrid->d_x.p; a.gy = bgrid->d_y.p; a.gz = bgrid->d_z.p; a.gelev = bgrid->d_elev.p; a.glaf = bgrid->d_laf.p;
    a.bg = f_bg.d; a.out = f_out.d;
    a.C = C; a.E = E; a.ny = bgrid->ny; a.nx = bgrid->nx;
    a.tiled2d = (bgrid->nx > 0 && (long)bgrid->ny * bgrid->nx == C) ? 1 : 0;
    a.wshift = 3;
    if(a.tiled2d) {
        a.wshift = gpp_tile_wshift(bgrid);
        const int tw = 1 << a.wshift, th = 64 >> a.wshift;
        a.tiles_x = (a.nx + tw - 1) / tw; a.ntiles = a.tiles_x * ((a.ny + th - 1) / th);
    }
    else { a.tiles_x = 0; a.ntiles = (C + 63) / 64; }
    a.s.pgeo = ws.pgeo.p; a.s.smeta = ix->d_smeta.p; a.s.bin_start = ix->d_bin_start.p;
    a.s.axis_a = ix->axis_a; a.s.axis_b = ix->axis_b; a.s.nbx = ix->nbx; a.s.nby = ix->nby;
    a.s.amin = ix->amin; a.s.bmin = ix->bmin; a.s.inv_s = ix->inv_s;
    a.s.st = gpp_resolve_structure(st);
    gpp_bind_field(a.s.st, st, bgrid, points, ws.cell_idx, ws.obs_idx);
    a.s.max_points = max_points;
    { const double occ = (double)S / ((double)ix->nbx * ix->nby);
      const int kk = (max_points > 0 && max_points <= 32) ? max_points : 32;
      a.s.q0 = std::max(1, std::min(8, (int)std::ceil(0.5 * (std::sqrt(1.6 * kk / std::max(occ, 1e-3)) - 1.0)))); }
    a.s.scan_stats = timing_env("GPP_SCAN_STATS") ? ws.counters.p + 2 : nullptr;
    a.s.K = (max_points > 0 && max_points <= EN) ? max_points : EN;
    a.ogeo = ix->d_ogeo.p; a.oaux = ws.oaux.p;
    a.gY = ws.gY.p; a.validIdx = ws.validIdx.p; a.nV = nV; a.valid_identity = (nV == E) ? 1 : 0;
    a.sel = ws.sel.get((size_t)a.ntiles * EN * 64);
    a.gram = ws.gram.get((size_t)a.ntiles * 2 * EN * EN);   // two matrices per tile: k_ensi_pair works on two groups at a time
    a.debug = timing_env("GPP_ENSI_DEBUG") ? atoi(timing_env("GPP_ENSI_DEBUG")) : 0;
    a.jtol2 = g_ensi_converge ? 0.0 : GPP_ENSI_JTOL2;
    if(const char* jt = timing_env("GPP_ENSI_JTOL")) { if(!g_ensi_converge) { const double v = atof(jt); a.jtol2 = v * v; } }   // (experiments: |E| <= v c)   // gpp_ensi_set_convergence(1): the Jacobi sweeps run to convergence (no perturbation series to speak of)
    a.allow_extrap = allow_extrapolation ? 1 : 0;
    a.err = ws.err.p; a.counters = ws.counters.p;
    // cells with more than 32 usable observations go to k_ensi_big (scalar structure functions; the spatially varying forms
    // fail loudly there)
    const bool big_ok = (max_points == 0 || max_points > EN) && !path_env("GPP_ENSI_NO_BIG");
    if(big_ok) {
        a.big_list = ws.big_list.get(2 * (size_t)C); a.huge_list = a.big_list + C; a.big_count = ws.big_count.get(2);
        GPP_HIP(hipMemsetAsync(ws.big_count.p, 0, 2 * sizeof(int), stream()));
    }
    GPP_HIP(hipEventRecord(ws.e0, stream()));
    if(use_pair) {
        a.meta = ws.meta.get((size_t)a.ntiles * 64);
        a.hsigs = ws.hsigs.get((size_t)a.ntiles * 64);
        if(a.s.st.fh) hipLaunchKernelGGL(k_ensi_scan<true>, dim3(a.ntiles), dim3(64), 0, stream(), a);
        else hipLaunchKernelGGL(k_ensi_scan<false>, dim3(a.ntiles), dim3(64), 0, stream(), a);
        GPP_HIP(hipGetLastError());
        if(nV <= 1) hipLaunchKernelGGL(k_ensi_count_cells, dim3(256), dim3(256), 0, stream(), (const unsigned*)a.meta, (long)a.ntiles * 64, ws.counters.p + 72);
        // spectral side (pairs of cells, warm-started along a tile) and ensemble side (one wave per cell) in batches of tiles: what
        // the second kernel needs of a cell (17 KB) waits in HBM.  The park is kept between calls (freeing and re-allocating tens of
        // GB costs seconds) and is therefore bounded: two fifths of the device memory (115 GB of 288: config 5's 110 GB go through in one
        // batch; every batch boundary costs the tails of both kernels, 1.5 ms on config 5 -- a quarter = two batches until round 6), never more
        // than half of what is free right now; gpp_release_workspaces() gives it back (GPP_ENSI_PARK_MB)
        size_t park_bytes = (size_t)72 << 30;
        {
            size_t free_b = 0, total_b = 0;
            if(hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                park_bytes = std::min(total_b / 5 * 2, std::max<size_t>((free_b + ws.cpark.cap * sizeof(double)) / 2, (size_t)64 << 20));
        }
        if(path_env("GPP_ENSI_PARK_MB")) park_bytes = (size_t)atol(path_env("GPP_ENSI_PARK_MB")) << 20;
        const size_t per_tile = (size_t)64 * ENSI_PARK_D * sizeof(double);
        int tcap = (int)std::max<size_t>(1, std::min<size_t>((size_t)a.ntiles, park_bytes / per_tile));
        for(;;) {   // (somebody else may hold the memory after all: halve the batch until the park fits)
            try { a.cpark = ws.cpark.get((size_t)tcap * 64 * ENSI_PARK_D); break; }
            catch(const Error&) { if(tcap <= 1) throw; (void)hipGetLastError(); tcap = (tcap + 1) / 2; }
        }
        for(int t0 = 0; t0 < a.ntiles; t0 += tcap) {
            const int nt = std::min(tcap, a.ntiles - t0);
            a.tile0 = t0;
            if(a.s.st.fh) hipLaunchKernelGGL(k_ensi_pair<true>, dim3(nt), dim3(64), 0, stream(), a);
            else hipLaunchKernelGGL(k_ensi_pair<false>, dim3(nt), dim3(64), 0, stream(), a);
            // (at most 64 valid members: the three-waves-per-SIMD form, ensi_members3.h; GPP_ENSI_MEMBERS2: the two-area kernel it replaced)
            if(a.nV <= 64 && !path_env("GPP_ENSI_MEMBERS2")) hipLaunchKernelGGL(k_ensi_members3, dim3((unsigned)nt * 64u), dim3(64), 0, stream(), a);
            else if(a.nV <= 64) hipLaunchKernelGGL(k_ensi_members<true>, dim3((unsigned)nt * 64u), dim3(64), 0, stream(), a);
            else hipLaunchKernelGGL(k_ensi_members<false>, dim3((unsigned)nt * 64u), dim3(64), 0, stream(), a);
            GPP_HIP(hipGetLastError());
        }
    }
    GPP_HIP(hipGetLastError());
    int nbig_cells = 0;
    if(big_ok) {
        int nbig = 0;
        GPP_HIP(hipMemcpyAsync(&nbig, ws.big_count.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
        GPP_HIP(hipStreamSynchronize(stream()));
        nbig_cells = nbig;
        auto launch_huge = [&](const int* list, const int* count, int nitems) {   // no capacity of its own: scratch sized for this call
            int kcap = 1;
            while(kcap < S) kcap <<= 1;
            // scratch per workgroup: candidate keys for every observation + the E x E matrices; within the 16 GB budget of the general
            // kernels (like optimal_interpolation_ensi_multi): fewer workgroups when a grid point needs much
            const size_t per_wg = (size_t)kcap * sizeof(unsigned long long) + (2 * (size_t)nV * nV + 5 * (size_t)nV) * sizeof(double);
            size_t budget = (size_t)16 << 30;
            if(path_env("GPP_OI_HUGE_BUDGET_MB")) budget = (size_t)atol(path_env("GPP_OI_HUGE_BUDGET_MB")) << 20;
            if(per_wg > budget) runtime("optimal_interpolation_ensi: the scratch of one grid point (" + std::to_string(per_wg >> 20) + " MB) does not fit the budget of the general kernel (GPP_OI_HUGE_BUDGET_MB)");
            const int nwg = (int)std::max<size_t>(1, std::min<size_t>({(size_t)nitems, (size_t)512, budget / per_wg}));
            a.huge_kcap = kcap;
            a.huge_keys = ws.huge_keys.get((size_t)nwg * kcap);
            a.huge_mat = ws.huge_mat.get((size_t)nwg * (2 * (size_t)nV * nV + 5 * (size_t)nV));
            if(a.s.st.fh) hipLaunchKernelGGL(k_ensi_huge<true>, dim3(nwg), dim3(256), 0, stream(), a, list, count);
            else hipLaunchKernelGGL(k_ensi_huge<false>, dim3(nwg), dim3(256), 0, stream(), a, list, count);
            GPP_HIP(hipGetLastError());
        };
        if(nbig > 0 && nV > 16384) runtime("optimal_interpolation_ensi: more than 16384 valid ensemble members at a grid point with more than 32 observations are not supported on the GPU path (the general kernel stages one row of Y in 64 KB of LDS)");
        if(nbig > 0 && nV > EMAXV) launch_huge(a.big_list, a.big_count, nbig);      // more valid members than one lane each: the general kernel
        else if(nbig > 0) {
            const int nwg = std::min(nbig, 1024);
            a.big_keys = ws.big_keys.get((size_t)nwg * EBIG_CAND);
            {
                const size_t ns_lds = (size_t)3 * 64 * NSP * sizeof(double);
                static std::once_flag ns_once;
                std::call_once(ns_once, [=] {
                    GPP_HIP(hipFuncSetAttribute((const void*)k_ensi_big_ns<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ns_lds));
                    GPP_HIP(hipFuncSetAttribute((const void*)k_ensi_big_ns<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ns_lds));
                    GPP_HIP(hipFuncSetAttribute((const void*)k_ensi_big_ns<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ns_lds));
                    GPP_HIP(hipFuncSetAttribute((const void*)k_ensi_big_ns<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ns_lds));
                });
                const bool full = nV > 48;
                if(a.s.st.fh) { if(full) hipLaunchKernelGGL((k_ensi_big_ns<true, true>), dim3(nwg), dim3(256), ns_lds, stream(), a); else hipLaunchKernelGGL((k_ensi_big_ns<true, false>), dim3(nwg), dim3(256), ns_lds, stream(), a); }
                else { if(full) hipLaunchKernelGGL((k_ensi_big_ns<false, true>), dim3(nwg), dim3(256), ns_lds, stream(), a); else hipLaunchKernelGGL((k_ensi_big_ns<false, false>), dim3(nwg), dim3(256), ns_lds, stream(), a); }
            }
            GPP_HIP(hipGetLastError());
            int nhuge = 0;   // cells with more candidates than the LDS sort of k_ensi_big holds (or no convergence of the iteration)
            GPP_HIP(hipMemcpyAsync(&nhuge, ws.big_count.p + 1, sizeof(int), hipMemcpyDeviceToHost, stream()));
            GPP_HIP(hipStreamSynchronize(stream()));
            if(nhuge > 0) launch_huge(a.huge_list, a.big_count + 1, nhuge);
        }
    }
    GPP_HIP(hipEventRecord(ws.e1, stream()));
    int err = 0;
    unsigned long long npass = 0;
    GPP_HIP(hipMemcpyAsync(&err, ws.err.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipMemcpyAsync(&npass, ws.counters.p + 72, sizeof(npass), hipMemcpyDeviceToHost, stream()));
    f_out.finish();
    static thread_local std::vector<unsigned long long> hcv(80 + 1024 * 32);
    unsigned long long* const hc = hcv.data();
    if(timing_env("GPP_ENSI_STATS")) GPP_HIP(hipMemcpyAsync(hc, ws.counters.p, sizeof(unsigned long long) * hcv.size(), hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    GPP_HIP(hipEventElapsedTime(&g_ensi_ms, ws.e0, ws.e1));
    // grid points the call left untouched because their E x E system is singular or not finite (the reference's "Condition number error in
    // N points. Using raw values in those points.", oi_ensi.cpp:386-390,557-561; the mirrors print it): with fewer than two valid members
    // that is every grid point with an observation in range, the large-n cells (all of them have observations) included
    g_ensi_stats.kernel_ms = g_ensi_ms;
    g_ensi_stats.condition_passthrough = (long long)npass + ((nV <= 1 && big_ok) ? (long long)nbig_cells : 0);
    if(timing_env("GPP_ENSI_STATS")) {
        unsigned long long sw = 0; for(int i = 0; i < 32; i++) sw += hc[4 + i];
        fprintf(stderr, "[gpp] ensi: %llu cells solved, %.2f Jacobi sweeps per cell\n", hc[1], hc[1] ? (use_pair ? GPP_ENSI_JCHUNK / 16.0 : 1.0) * (double)sw / (double)hc[1] : 0.0);
        for(int sl = 0; sl < 1024; sl++) for(int i = 0; i < 12; i++) { hc[40 + i] += hc[80 + sl * 32 + i]; hc[60 + i] += hc[80 + sl * 32 + 16 + i]; }
        unsigned long long tot = 0; for(int i = 0; i < 12; i++) tot += hc[40 + i];
        unsigned long long tot2 = 0; for(int i = 0; i < 12; i++) tot2 += hc[60 + i];
        if(tot2) { fprintf(stderr, "[gpp] ensi members phases (%% of wave cycles):"); for(int i = 0; i < 12; i++) fprintf(stderr, " %d:%.1f", i, 100.0 * (double)hc[60 + i] / (double)tot2); fprintf(stderr, "\n"); }
        if(tot) { fprintf(stderr, "[gpp] ensi phases (%% of wave cycles):"); for(int i = 0; i < 10; i++) fprintf(stderr, " %d:%.1f", i, 100.0 * (double)hc[40 + i] / (double)tot); fprintf(stderr, "\n"); }
    }
    if(err & 1) runtime(big_ok ? "Internal error. optimal_interpolation_ensi: candidate scratch of the general kernel too small"
                               : "optimal_interpolation_ensi: a grid point has more usable observations than the 32-row tile kernel holds and the large-n kernels are switched off (GPP_ENSI_NO_BIG)");
    return GPP_OK;
    GPP_CATCH
}

// optimal_interpolation_ensi_multi_{ebe (1), ebesc (2), utem (3)}: src/api/oi_ensi_multi.cpp:329-1311 (Points overloads; a Grid is
// its row-major flattening, :34-327)
extern "C" int gpp_optimal_interpolation_ensi_multi(int variant, gpp_points* bgrid, const float* bratios, const float* background,
                                                    const float* background_corr, int ne, gpp_points* points, const float* pobs,
                                                    const float* pratios, const float* pbackground, const float* pbackground_corr,
                                                    const gpp_structure* st, int max_points, int allow_extrapolation, float* out, int mem) {
    GPP_TRY
    if(variant < 1 || variant > 3) invalid("variant must be 1 (ebe), 2 (ebesc) or 3 (utem)");
    if(max_points < 0) invalid("max_points must be >= 0");                                      // :341-342
    if(!bgrid || !points) invalid("grid/points handle is NULL");
    if(bgrid->type != points->type)
        invalid("Both background and observations points must be of same coorindate type (lat/lon or x/y)");
    if(!st) invalid("structure is NULL");
    if(ne < 0) invalid("negative ensemble size");
    if(mem & GPP_HOST_F64) invalid("GPP_HOST_F64 is not supported by optimal_interpolation_ensi_multi");
    const bool corr = variant != 2;
    const int C = bgrid->n, S = points->n, E = ne;
    ensure_device();
    g_ensi_ms = 0;
    if(C == 0 || E == 0) return GPP_OK;
    EnsiWorkspace& ws = g_ews;
    InField f_bg, f_bgc, f_br, f_obs, f_pr, f_pbg, f_pbgc;
    OutField f_out;
    f_bg.bind(background, (size_t)C * E, mem);
    f_out.bind(out, (size_t)C * E, mem);
    const long nbg = (long)C * E;
    hipLaunchKernelGGL(k_copy, dim3((unsigned)((nbg + 255) / 256)), dim3(256), 0, stream(), f_bg.d, nbg, f_out.d);   // output = background (:375)
    GPP_HIP(hipGetLastError());
    if(S == 0) { f_out.finish(); GPP_HIP(hipStreamSynchronize(stream())); return GPP_OK; }        // :361-363
    if(corr) { f_bgc.bind(background_corr, (size_t)C * E, mem); f_pbgc.bind(pbackground_corr, (size_t)S * E, mem); }
    f_br.bind(bratios, C, mem);
    f_obs.bind(pobs, variant == 3 ? (size_t)S : (size_t)S * E, mem);
    f_pr.bind(pratios, S, mem);
    f_pbg.bind(pbackground, (size_t)S * E, mem);
    bgrid->to_device();
    gpp_obs_index* ix = gpp_build_obs_index(points);
    // members valid in every field (:395-418)
    std::vector<int> flags(E, 1);
    ws.flags.upload(flags.data(), E);
    const long npb = (long)S * E;
    hipLaunchKernelGGL(k_ensi_member_flags, dim3((unsigned)((nbg + 255) / 256)), dim3(256), 0, stream(), f_bg.d, nbg, E, ws.flags.p);
    hipLaunchKernelGGL(k_ensi_member_flags, dim3((unsigned)((npb + 255) / 256)), dim3(256), 0, stream(), f_pbg.d, npb, E, ws.flags.p);
    if(corr) {
        hipLaunchKernelGGL(k_ensi_member_flags, dim3((unsigned)((nbg + 255) / 256)), dim3(256), 0, stream(), f_bgc.d, nbg, E, ws.flags.p);
        hipLaunchKernelGGL(k_ensi_member_flags, dim3((unsigned)((npb + 255) / 256)), dim3(256), 0, stream(), f_pbgc.d, npb, E, ws.flags.p);
    }
    GPP_HIP(hipMemcpyAsync(flags.data(), ws.flags.p, sizeof(int) * E, hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    std::vector<int> valid;
    for(int e = 0; e < E; e++) if(flags[e]) valid.push_back(e);
    const int nV = (int)valid.size();
    if(nV == 0) { f_out.finish(); GPP_HIP(hipStreamSynchronize(stream())); return GPP_OK; }      // :419-420
    if(nV > 16384) runtime("optimal_interpolation_ensi_multi: more than 16384 valid ensemble members are not supported on the GPU path (the general kernel stages one row of Y in 64 KB of LDS)");
    if(variant == 1 && nV > 4096) runtime("optimal_interpolation_ensi_multi_ebe: more than 4096 valid ensemble members are not supported on the GPU path");
    ws.validIdx.upload(valid.data(), nV);
    ws.gYhat.get(S); ws.gY.get((size_t)S * nV); ws.gYm.get((size_t)S * nV); ws.obs0.get(S);
    hipLaunchKernelGGL(k_multi_obs_prep, dim3((S + 127) / 128), dim3(128), 0, stream(), variant, f_pbg.d, corr ? f_pbgc.d : f_pbg.d, f_obs.d, S, E,
                       (const int*)ws.validIdx.p, nV, ws.gY.p, ws.gYm.p, ws.gYhat.p, ws.obs0.p);
    ws.pgeo.get(S); ws.oaux.get(S);
    // oaux = (laf, obs[.][0] / obs, gYhat, pratio); an observation is usable when that first value is valid (:480)
    hipLaunchKernelGGL(k_pack_obs, dim3((S + 255) / 256), dim3(256), 0, stream(), S, ix->d_sgeo.p, ix->d_pos.p, ix->d_olaf.p,
                       (const float*)ws.obs0.p, f_pr.d, (const float*)ws.gYhat.p, (const float*)nullptr, 0, ws.pgeo.p, ws.oaux.p);
    GPP_HIP(hipGetLastError());
    ws.err.get(1);
    GPP_HIP(hipMemsetAsync(ws.err.p, 0, sizeof(int), stream()));
    if(!ws.e0) { GPP_HIP(hipEventCreate(&ws.e0)); GPP_HIP(hipEventCreate(&ws.e1)); }
    MultiArgs ma = MultiArgs();
    EnsiArgs& a = ma.e;
    a.gx = bgrid->d_x.p; a.gy = bgrid->d_y.p; a.gz = bgrid->d_z.p; a.gelev = bgrid->d_elev.p; a.glaf = bgrid->d_laf.p;
    a.bg = f_bg.d; a.out = f_out.d;
    a.C = C; a.E = E;
    a.s.pgeo = ws.pgeo.p; a.s.smeta = ix->d_smeta.p; a.s.bin_start = ix->d_bin_start.p;
    a.s.axis_a = ix->axis_a; a.s.axis_b = ix->axis_b; a.s.nbx = ix->nbx; a.s.nby = ix->nby;
    a.s.amin = ix->amin; a.s.bmin = ix->bmin; a.s.inv_s = ix->inv_s;
    a.s.st = gpp_resolve_structure(st);
    gpp_bind_field(a.s.st, st, bgrid, points, ws.cell_idx, ws.obs_idx);
    if(a.s.st.fh) runtime("optimal_interpolation_ensi_multi: spatially varying structure functions are not supported on the GPU path");
    a.s.max_points = max_points;
    a.ogeo = ix->d_ogeo.p; a.oaux = ws.oaux.p;
    a.gY = ws.gY.p; a.validIdx = ws.validIdx.p; a.nV = nV; a.valid_identity = (nV == E) ? 1 : 0;
    a.allow_extrap = allow_extrapolation ? 1 : 0;
    a.err = ws.err.p;
    ma.gYm = ws.gYm.p; ma.bgc = corr ? f_bgc.d : f_bg.d; ma.bratios = f_br.d;
    ma.pobs2 = f_obs.d; ma.pbg2 = f_pbg.d;
    ma.oob = (variant != 3 && valid[nV - 1] != nV - 1) ? 1 : 0;
    const int nwg = std::min(C, 2048);
    a.big_keys = ws.big_keys.get((size_t)nwg * EBIG_CAND);
    a.big_list = ws.big_list.get(2 * (size_t)C); a.huge_list = a.big_list + C; a.big_count = ws.big_count.get(2);
    GPP_HIP(hipMemsetAsync(ws.big_count.p, 0, 2 * sizeof(int), stream()));
    GPP_HIP(hipEventRecord(ws.e0, stream()));
    const bool all_huge = variant == 3 && nV > EMAXV;   // utem keeps one member per lane in k_ensi_multi: more go to the general kernel
    if(!all_huge) {
        if(variant == 1) hipLaunchKernelGGL(k_ensi_multi<1>, dim3(nwg), dim3(256), 0, stream(), ma);
        else if(variant == 2) hipLaunchKernelGGL(k_ensi_multi<2>, dim3(nwg), dim3(256), 0, stream(), ma);
        else hipLaunchKernelGGL(k_ensi_multi<3>, dim3(nwg), dim3(256), 0, stream(), ma);
        GPP_HIP(hipGetLastError());
    }
    int nhuge = C;
    if(!all_huge) {
        GPP_HIP(hipMemcpyAsync(&nhuge, ws.big_count.p + 1, sizeof(int), hipMemcpyDeviceToHost, stream()));
        GPP_HIP(hipStreamSynchronize(stream()));
    }
    if(nhuge > 0) {   // the grid points beyond the LDS areas of k_ensi_multi: scratch sized for this call, within a budget
        int kcap = 1;
        while(kcap < S) kcap <<= 1;
        const size_t ncap = (variant == 3) ? 0 : (size_t)((max_points > 0) ? std::min(max_points, S) : S);
        const size_t stride = std::max(ncap * (ncap + 1) + (size_t)nV, 2 * (size_t)nV * nV + 6 * (size_t)nV);
        size_t budget = (size_t)16 << 30;
        if(path_env("GPP_OI_HUGE_BUDGET_MB")) budget = (size_t)atol(path_env("GPP_OI_HUGE_BUDGET_MB")) << 20;
        const size_t per_wg = stride * sizeof(double) + (size_t)kcap * sizeof(unsigned long long);
        if(per_wg > budget) runtime("optimal_interpolation_ensi_multi: a grid point may select " + std::to_string(ncap) + " observations: its system does not fit the scratch budget of the GPU path");
        const int hwg = (int)std::max<size_t>(1, std::min<size_t>({(size_t)nhuge, (size_t)512, budget / per_wg}));
        ma.e.huge_kcap = kcap; ma.huge_ncap = (int)ncap; ma.huge_stride = stride;
        ma.e.huge_keys = ws.huge_keys.get((size_t)hwg * kcap);
        ma.e.huge_mat = ws.huge_mat.get((size_t)hwg * stride);
        const int* list = all_huge ? nullptr : a.huge_list;
        const int* cnt = all_huge ? nullptr : a.big_count + 1;
        if(variant == 1) hipLaunchKernelGGL(k_ensi_multi_huge<1>, dim3(hwg), dim3(256), 0, stream(), ma, list, cnt);
        else if(variant == 2) hipLaunchKernelGGL(k_ensi_multi_huge<2>, dim3(hwg), dim3(256), 0, stream(), ma, list, cnt);
        else hipLaunchKernelGGL(k_ensi_multi_huge<3>, dim3(hwg), dim3(256), 0, stream(), ma, list, cnt);
        GPP_HIP(hipGetLastError());
    }
    GPP_HIP(hipEventRecord(ws.e1, stream()));
    int err = 0;
    GPP_HIP(hipMemcpyAsync(&err, ws.err.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
    f_out.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    GPP_HIP(hipEventElapsedTime(&g_ensi_ms, ws.e0, ws.e1));
    if(err & 4) runtime("optimal_interpolation_ensi_multi: an ensemble member is invalid in front of a valid one: the reference indexes its innovation matrix with the original member index (oi_ensi_multi.cpp:565), which is out of bounds");
    if(err & 2) runtime("optimal_interpolation_ensi_multi: singular matrix at a grid point (arma::inv fails in the reference)");
    if(err & 1) runtime("Internal error. optimal_interpolation_ensi_multi: scratch of the general kernel too small");
    return GPP_OK;
    GPP_CATCH
}
