// Neighbourhood filters on MI355X (gfx950).
//
// Replaces src/api/neighbourhood.cpp:
//   neighbourhood(vec2|vec3, halfwidth, statistic)            :12-242
//   neighbourhood_quantile_fast(vec2|vec3, q|q-field, hw, thresholds)  :296-527
//   neighbourhood_quantile / neighbourhood_brute_force        :528-654
// and the per-cell statistics of src/api/util.cpp:19-178,339-414.
//
// Design (HBM-bound: the (Y,X,E) cube is read exactly once):
//   * member pass: one wavefront per 64 consecutive cells; the 64*E contiguous floats are
//     streamed with 16-byte coalesced loads into an odd-pitch LDS tile, then lane l walks row l
//     in member order -- this keeps the reference's sequential float accumulation
//     (util.cpp:22-38) bit-for-bit while the HBM side stays fully coalesced;
//   * quantile_fast: the same pass emits all T threshold fractions per cell (T/8 register
//     batches), so the cube is not re-read T times as in the reference (:453-472);
//   * box statistics: separable row pass + column pass (double sums + int counts) instead of the
//     reference's serial summed-area table; sums differ from the SAT only in double rounding;
//   * min/max: separable running min/max ignoring non-finite values;
//   * exact quantile / median / brute force: one wavefront per cell, k-th order statistic by
//     bisection on the monotone uint32 image of the float values (no sort, no scratch).
#include <mutex>
#include <atomic>
#include "common.h"
#include <algorithm>
#include <rocprim/rocprim.hpp>

#pragma clang fp contract(off)
using namespace gpp;

#include "row_stats.h"
#include "qf_box.h"

// -------------------------------------------------------------------------------------------
// member pass
// -------------------------------------------------------------------------------------------
#define MEMBER_EC 160   // rows up to this many members are staged through LDS
#define TB 12           // thresholds per register batch (multiple of 4)

// Mean / Sum / Count of one LDS row with 16-byte reads issued ahead of the (sequential, reference-order) float adds
__device__ __forceinline__ float row_mean_sum_count_v4(const float* row, int E, int statistic) {
    float total = 0; int count = 0;
    const float4* r4 = reinterpret_cast<const float4*>(row);
    const int n4 = E >> 2;
#pragma unroll 5
    for(int i = 0; i < n4; i++) {
        const float4 q = r4[i];
        if(nv(q.x)) { total += q.x; count++; }
        if(nv(q.y)) { total += q.y; count++; }
        if(nv(q.z)) { total += q.z; count++; }
        if(nv(q.w)) { total += q.w; count++; }
    }
    for(int i = n4 * 4; i < E; i++) { const float v = row[i]; if(nv(v)) { total += v; count++; } }
    if(statistic == GPP_COUNT) return (float)count;
    if(count == 0) return NAN;
    return (statistic == GPP_MEAN) ? total / (float)count : total;
}

// sum_k += (v <= th_k) for four thresholds.  Four compares into four SGPR pairs, then four add-with-carry: two
// instructions per (member, threshold) and no wait states between a compare and its consumer (the compiler's own
// sequence spends a third of the issue slots on s_nop between v_cmp and the instruction that reads the mask).
// NaN compares false, like `v <= th`.
__device__ __forceinline__ void count_le4(const float v, const float t0, const float t1, const float t2, const float t3,
                                          int& s0, int& s1, int& s2, int& s3) {
    unsigned long long m0, m1, m2, m3;
    asm volatile("v_cmp_ge_f32_e64 %4, %9, %8\n\t"
                 "v_cmp_ge_f32_e64 %5, %10, %8\n\t"
                 "v_cmp_ge_f32_e64 %6, %11, %8\n\t"
                 "v_cmp_ge_f32_e64 %7, %12, %8\n\t"
                 "v_addc_co_u32_e64 %0, %4, %0, 0, %4\n\t"
                 "v_addc_co_u32_e64 %1, %5, %1, 0, %5\n\t"
                 "v_addc_co_u32_e64 %2, %6, %2, 0, %6\n\t"
                 "v_addc_co_u32_e64 %3, %7, %3, 0, %7"
                 : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
                 : "v"(v), "s"(t0), "s"(t1), "s"(t2), "s"(t3));
}

// per-lane work on one row (members of one cell); forceinlined separately for LDS rows and global rows so that the
// LDS path keeps address space 3 (a pointer that may be either becomes a slow flat access)
template <int MODE>
__device__ __forceinline__ void member_row_work(const float* row, const int E, const bool vec_ok, const int statistic,
                                                const float* __restrict__ thr, const int T, float* __restrict__ out, const long C, const long cell,
                                                const QfGeom& g) {
    if(MODE == 0) {
        if(vec_ok && (statistic == GPP_MEAN || statistic == GPP_SUM || statistic == GPP_COUNT))
            out[cell] = row_mean_sum_count_v4(row, E, statistic);
        else out[cell] = row_statistic(row, E, statistic);
    }
    else {
        int count = 0;
        for(int t0 = 0; t0 < T; t0 += TB) {
            float th[TB]; int sum[TB];
#pragma unroll
            for(int k = 0; k < TB; k++) { th[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((t0 + k < T) ? thr[t0 + k] : 0.0f))); sum[k] = 0; }
            int cnt = 0;
            if(vec_ok) {   // 16-byte LDS reads, 4 members per read, issued ahead of the compares
                const float4* r4 = reinterpret_cast<const float4*>(row);
                const int n4 = E >> 2;
                for(int i = 0; i < n4; i++) {
                    const float4 q = r4[i];
                    const float vv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for(int j = 0; j < 4; j++) {
                        const bool ok = nv(vv[j]);
                        const float v = ok ? vv[j] : NAN;   // NaN compares false against every threshold
                        cnt += ok ? 1 : 0;
#pragma unroll
                        for(int k = 0; k < TB; k += 4) count_le4(v, th[k], th[k + 1], th[k + 2], th[k + 3], sum[k], sum[k + 1], sum[k + 2], sum[k + 3]);
                    }
                }
            }
            else {
#pragma unroll 4
                for(int e = 0; e < E; e++) {
                    const bool ok = nv(row[e]);
                    const float v = ok ? row[e] : NAN;
                    cnt += ok ? 1 : 0;
#pragma unroll
                    for(int k = 0; k < TB; k++) sum[k] += (v <= th[k]) ? 1 : 0;
                }
            }
            count = cnt;
            if(MODE == 2) {   // raw counts, one byte each (E <= 254) in the padded planes of qf_box.h
                unsigned char* out8 = reinterpret_cast<unsigned char*>(out) + qf_cell_offset(g, cell);
#pragma unroll
                for(int k = 0; k < TB; k++)
                    if(t0 + k < T) out8[(long)(t0 + k) * g.Pp] = (unsigned char)sum[k];
                if(t0 + TB >= T) {
                    out8[(long)T * g.Pp] = (unsigned char)count;
                    if(count != E) { g.rowflag[cell / g.X] = 1; g.rowflag[g.Y] = 1; }
                }
            }
            else {
#pragma unroll
                for(int k = 0; k < TB; k++)
                    if(t0 + k < T) out[(long)(t0 + k) * C + cell] = count > 0 ? (float)sum[k] / (float)count : NAN;
            }
        }
    }
}

// mode 0: out[c] = calc_statistic(members of c)                       (neighbourhood.cpp:21-25)
// mode 1: out[t*C + c] = #(valid members <= thr[t]) / #valid members  (neighbourhood.cpp:456-471)
// One wavefront per workgroup, grid-stride over tiles of 64 consecutive cells.  The 64*E contiguous floats (16*E
// float4) of a tile go HBM -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave-instruction, linear layout, no
// VGPR round trip) into one half of a double buffer while lane l walks row l of the previous tile in member order
// out of the other half.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int MODE>
__global__ __launch_bounds__(64) void k_member_pass(const float* __restrict__ in, long C, int E, int statistic,
                                                    const float* __restrict__ thr, int T, float* __restrict__ out, int use_dma, const QfGeom g) {
    extern __shared__ float4 lds4[];
    const int lane = threadIdx.x;
    const long ntiles = (C + 63) / 64;
    const long nfull = C / 64;                       // tiles with 64 cells
    const int nf4 = 16 * E;                          // float4 per full tile
    const int nchunk = (nf4 + 63) / 64;              // 1 KiB wave-loads per tile
    const int bufstride = nchunk * 64;               // float4 per buffer (padded to whole chunks)
    if(use_dma) {
        const float4* in4 = reinterpret_cast<const float4*>(in);
        auto issue = [&](long tile, int b) {
            const float4* src = in4 + tile * nf4;
            float4* dst = lds4 + b * bufstride;
            for(int k = 0; k < nchunk; k++) {
                const int idx = k * 64 + lane;
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + (idx < nf4 ? idx : nf4 - 1)), (lds_void_t*)(dst + k * 64), 16, 0, 0);
            }
        };
        if(MODE != 0) {
            // threshold counting is VALU-bound (2*T*E compare/adds per cell): a single buffer doubles the resident waves
            for(long tile = blockIdx.x; tile < nfull; tile += gridDim.x) {
                __syncthreads();                                    // previous tile fully consumed
                issue(tile, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                member_row_work<MODE>(reinterpret_cast<const float*>(lds4) + lane * E, E, (E & 3) == 0, statistic, thr, T, out, C, tile * 64 + lane, g);
            }
        }
        else {
            int b = 0;
            if((long)blockIdx.x < nfull) issue(blockIdx.x, 0);
            for(long tile = blockIdx.x; tile < nfull; tile += gridDim.x) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile has landed in buffer b
                __syncthreads();
                const long nxt = tile + gridDim.x;
                if(nxt < nfull) issue(nxt, b ^ 1);                   // next tile flies during the row walk
                const float* row = reinterpret_cast<const float*>(lds4 + b * bufstride) + lane * E;
                member_row_work<MODE>(row, E, (E & 3) == 0, statistic, thr, T, out, C, tile * 64 + lane, g);
                b ^= 1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the last, partial tile (if any): lane-private walk straight from memory, on block 0
        if(nfull < ntiles && blockIdx.x == 0) {
            const long cell = nfull * 64 + lane;
            if(cell < C) member_row_work<MODE>(in + cell * E, E, false, statistic, thr, T, out, C, cell, g);
        }
        return;
    }
    // generic path (long rows or unaligned input): lane-private walk straight from memory
    for(long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long cell = tile * 64 + lane;
        if(cell < C) member_row_work<MODE>(in + cell * E, E, false, statistic, thr, T, out, C, cell, g);
    }
}

// -------------------------------------------------------------------------------------------
// quantile_fast member pass, round 3: k_qf_lut + k_qf_count<NL>  (neighbourhood.cpp:453-472)
//
// plane t of the result = #(valid members <= thr[t]) per cell, plane T = #valid members, one byte each (E <= 255) -- what
// k_member_pass<2> writes with 2 * T instructions per member.  Here a member costs ~8:
//   * rank instead of T compares.  With u[0 .. U) the distinct finite thresholds in ascending order and
//     idx(v) = #{k : u[k] < v}:   v <= u[k]  <=>  idx(v) <= k.  idx comes from a 256-entry bucket table (LDS): the bucket of v
//     is b(v) = sat_u8(fma(v, scale, off)) -- monotone in v whatever the rounding, so every threshold in a lower bucket is
//     below v and every threshold in a higher bucket is not: only the (at most one) threshold INSIDE the bucket needs the
//     exact float compare, idx = idx0[b] + (v > thr_in[b]).  k_qf_lut computes the buckets of the thresholds with the very
//     instructions of the member pass (so table and pass cannot disagree by a rounding) and raises a flag -- the call then
//     takes k_member_pass<2> -- when two distinct thresholds share a bucket or a threshold is not finite.
//     Invalid members cost nothing: NaN / -inf saturate into bucket 0, whose entry (-inf, 255) turns a finite value into
//     rank 0 (255 + 1, byte arithmetic) and leaves 255 for the others; +inf lands in bucket 255 = (FLT_MAX, U) -> U + 1.
//     Any rank above U means "not counted".
//   * sums of absolute differences instead of counters.  The ranks of four members are the bytes of one dword, and with
//     f(k) = sum_m |idx_m - k| (v_sad_u8: four members per instruction)   #(idx <= k) = (f(k + 1) - f(k) + E) / 2,
//     so all counts of a cell come from U + 2 accumulators: (U + 2) / 4 instructions per member.  #(idx <= U) = #valid.
//   * lanes = consecutive float4 of the cube while the ranks are formed (coalesced global_load_dwordx4, no staging of the
//     floats), lanes = cells for the sums: only the rank bytes (E bytes per cell) cross the LDS.
// -------------------------------------------------------------------------------------------
#define QF_NB 256
#ifndef QC_G
#define QC_G 5     // float4 loads in flight per lane and group
#endif
struct QfLut {
    float scale, off;
    int U;          // distinct finite thresholds
    int flag;       // 1: this table cannot serve the thresholds (two in one bucket, bucket 0 / 255 hit, non-finite threshold)
    int ident;      // rank[t] == t for every t (thresholds strictly ascending): planes leave in accumulator order
    int rank[16];   // thr[t] = u[rank[t]]
    int pad[3];
    uint2 lut[QF_NB];   // (threshold inside the bucket or a sentinel, rank below the bucket)
};
__device__ __forceinline__ unsigned qf_bucket_into(const float v, const float scale, const float off, const unsigned sel, const unsigned old) {
    return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v, scale, off), sel, old);
}
__global__ __launch_bounds__(QF_NB) void k_qf_lut(const float* __restrict__ thr, int T, QfLut* __restrict__ L, int* __restrict__ rowflag, int nflags) {
    // (round 6: also clears the row flags of the count planes -- [Y] "some row is flagged", [Y + 1] "the count pass was launched for another
    //  number of distinct thresholds than this table has" -- which were two fill launches of their own in front of the count pass)
    for(int k = threadIdx.x; k < nflags; k += QF_NB) rowflag[k] = 0;
    __shared__ float u[16], st[16];
    __shared__ int ub[16], srank[16];
    __shared__ int s_U, s_flag;
    __shared__ float s_scale, s_off;
    const int j = threadIdx.x;
    if(j < 16) st[j] = j < T ? thr[j] : 0.0f;    // (one round of loads: thread 0 works on the LDS copy)
    __syncthreads();
    if(j == 0) {
        int flag = 0, U = 0, ident = 1;
        float scale = 1.0f, off = 0.0f;
        for(int t = 0; t < T; t++) if(!nv(st[t]) || fabsf(st[t]) > 1e37f) flag = 1;
        if(!flag) {   // distinct values, ascending (T <= 16: insertion)
            for(int t = 0; t < T; t++) {
                const float v = st[t];
                int k = 0; bool dup = false;
                while(k < U && u[k] < v) k++;
                if(k < U && u[k] == v) dup = true;
                if(!dup) { for(int m = U; m > k; m--) u[m] = u[m - 1]; u[k] = v; U++; }
            }
            for(int t = 0; t < T; t++) { int k = 0; while(u[k] < st[t]) k++; srank[t] = k; if(k != t) ident = 0; }
            // lowest threshold -> bucket 1.5, highest -> bucket 253.5
            const float lo = u[0], hi = u[U - 1];
            scale = (hi > lo) ? 252.0f / (hi - lo) : 1.0f;
            if(!nv(scale) || scale <= 0) scale = 1.0f;
            off = (hi > lo ? 1.5f : 127.5f) - lo * scale;
            for(int k = 0; k < U; k++) {
                ub[k] = (int)(qf_bucket_into(u[k], scale, off, 0, 0) & 0xffu);
                if(ub[k] < 1 || ub[k] > 254 || (k > 0 && ub[k] == ub[k - 1])) flag = 1;
            }
            if(!nv(off)) flag = 1;
        }
        s_scale = scale; s_off = off;
        L->scale = scale; L->off = off; L->ident = ident;
        s_U = U; s_flag = flag;
        L->U = U; L->flag = flag;
    }
    __syncthreads();
    if(j < T) L->rank[j] = srank[j];
    if(s_flag) return;
    const int U = s_U;
    int below = 0; float inside = INFINITY;   // `v > +inf` is never true: an empty bucket keeps the rank below it
    for(int k = 0; k < U; k++) { if(ub[k] < j) below++; else if(ub[k] == j) inside = u[k]; }
    if(j == 0) { inside = -INFINITY; below = 255; }
    if(j == QF_NB - 1) inside = 3.402823466e38f;
    L->lut[j] = make_uint2(__float_as_uint(inside), (unsigned)below);
}

template <int NL>
__global__ __launch_bounds__(64) void k_qf_count(const float* __restrict__ in, long C, int E, const float* __restrict__ thr, int T,
                                                 const QfLut* __restrict__ L, unsigned char* __restrict__ out8, const QfGeom g, const int Uexp) {
    // Uexp >= 0: the host chose this instantiation from the LAST call's number of distinct thresholds without waiting for this call's table
    // (one host round trip less per call).  A table that says otherwise stops the pass and, through rowflag[Y + 1], the box pass behind it;
    // the host sees it in the one read-back at the end of the call and runs the call again the slow way.
    if(Uexp >= 0 && (L->U != Uexp || L->flag != 0)) {
        if(blockIdx.x == 0 && threadIdx.x == 0) g.rowflag[g.Y + 1] = 1;
        return;
    }
    extern __shared__ unsigned qc_lds[];
    uint2* const lut = reinterpret_cast<uint2*>(qc_lds);          // [QF_NB]
    unsigned* const ranks = qc_lds + 2 * QF_NB;                    // [16 E] rank bytes of a tile, four members per dword, cell-major
    const int lane = threadIdx.x;
    const int E4 = E >> 2;                                        // float4 per cell = dwords of rank bytes per cell = 64-lane loads per tile
    if((unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)qc_lds != 0u) __builtin_trap();   // see quad()
    for(int k = lane; k < QF_NB; k += 64) lut[k] = L->lut[k];
    const float scale = L->scale, off = L->off;
    const int ident = L->ident;
    const long nfull = C / 64;
    const float4* in4 = reinterpret_cast<const float4*>(in);
    const unsigned zero = 0;
    __syncthreads();
    const int ng = (E4 + QC_G - 1) / QC_G;   // groups of QC_G loads (the last one may be shorter)
    float4 cur[QC_G], nxt[QC_G];
    // A workgroup takes super-tiles of four consecutive tiles (256 cells): the counts of a super-tile leave through an LDS
    // buffer as one dword store per lane and plane (256 contiguous bytes per instruction instead of four times 64)
    unsigned char* const obuf = reinterpret_cast<unsigned char*>(ranks + max(16 * E, NL * 64));   // [T + 1][256] (behind the rank area, which also serves as the [NL][64] count exchange)
    const long nsuper = (nfull + 3) / 4;
    if(ng > 0 && (long)blockIdx.x < nsuper) {
        const float4* src0 = in4 + (long)blockIdx.x * 4 * 16 * E + lane;
#pragma unroll
        for(int i = 0; i < QC_G; i++) if(i < E4) cur[i] = src0[i * 64];
    }
    for(long st = blockIdx.x; st < nsuper; st += gridDim.x) {
      const int nsub = (int)min(4L, nfull - st * 4);
      const bool buffered = nsub == 4 && (g.X & 3) == 0;
      for(int sub = 0; sub < nsub; sub++) {
        const long tile = st * 4 + sub;
        const long next_tile = sub + 1 < nsub ? tile + 1 : (st + gridDim.x < nsuper ? (st + gridDim.x) * 4 : -1);
        const float4* src = in4 + tile * 16 * E + lane;
        // rank bytes of four members (one float4) -> ranks[k * 64 + lane]
        auto quad = [&](const float4 q, const int k) {
            unsigned bk = 0;
            bk = qf_bucket_into(q.x, scale, off, 0, bk);
            bk = qf_bucket_into(q.y, scale, off, 1, bk);
            bk = qf_bucket_into(q.z, scale, off, 2, bk);
            bk = qf_bucket_into(q.w, scale, off, 3, bk);
            unsigned a0, a1, a2, a3;   // byte offsets of the four table entries (one instruction each: byte select + shift)
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a0) : "v"(3u), "v"(bk));
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a1) : "v"(3u), "v"(bk));
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a2) : "v"(3u), "v"(bk));
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a3) : "v"(3u), "v"(bk));
            // (the table sits at LDS address 0 -- checked at kernel entry -- so the byte offsets ARE the addresses: no base add)
            typedef unsigned qc_u2 __attribute__((ext_vector_type(2)));
            typedef const __attribute__((address_space(3))) qc_u2* lds_u2;
#if defined(QC_ABL) && (QC_ABL & 1)   // timing experiment: no table gathers
            const qc_u2 e0 = {a0, a1}, e1 = {a1, a2}, e2 = {a2, a3}, e3 = {a3, a0};
#else
            const qc_u2 e0 = *(lds_u2)(size_t)a0, e1 = *(lds_u2)(size_t)a1, e2 = *(lds_u2)(size_t)a2, e3 = *(lds_u2)(size_t)a3;
#endif
            unsigned pk;   // byte m = rank of member m = rank below its bucket + (value > threshold inside the bucket)
            asm volatile("v_cmp_gt_f32_e32 vcc, %1, %2\n\t"
                         "v_addc_co_u32_sdwa %0, vcc, %3, %4, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"
                         : "=v"(pk) : "v"(q.x), "v"(e0.x), "v"(e0.y), "v"(zero) : "vcc");
            asm volatile("v_cmp_gt_f32_e32 vcc, %1, %2\n\t"
                         "v_addc_co_u32_sdwa %0, vcc, %3, %4, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                         : "+v"(pk) : "v"(q.y), "v"(e1.x), "v"(e1.y), "v"(zero) : "vcc");
            asm volatile("v_cmp_gt_f32_e32 vcc, %1, %2\n\t"
                         "v_addc_co_u32_sdwa %0, vcc, %3, %4, vcc dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                         : "+v"(pk) : "v"(q.z), "v"(e2.x), "v"(e2.y), "v"(zero) : "vcc");
            asm volatile("v_cmp_gt_f32_e32 vcc, %1, %2\n\t"
                         "v_addc_co_u32_sdwa %0, vcc, %3, %4, vcc dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                         : "+v"(pk) : "v"(q.w), "v"(e3.x), "v"(e3.y), "v"(zero) : "vcc");
            ranks[k * 64 + lane] = pk;
        };
        // whole groups of QC_G loads in flight: the next group is asked for before the current one is ranked, and the first
        // group of the NEXT tile before the sums of this one (`cur` holds it: loaded before the loop for the first tile)
        for(int g0 = 0; g0 < ng; g0++) {
            if(g0 + 1 < ng) {
#pragma unroll
                for(int i = 0; i < QC_G; i++) if((g0 + 1) * QC_G + i < E4) nxt[i] = src[((g0 + 1) * QC_G + i) * 64];
            }
            else if(next_tile >= 0) {
                const float4* nsrc = in4 + next_tile * 16 * E + lane;
#pragma unroll
                for(int i = 0; i < QC_G; i++) if(i < E4) nxt[i] = nsrc[i * 64];
            }
#pragma unroll
            for(int i = 0; i < QC_G; i++) if(g0 * QC_G + i < E4) quad(cur[i], g0 * QC_G + i);
#pragma unroll
            for(int i = 0; i < QC_G; i++) cur[i] = nxt[i];
        }
        __syncthreads();
        unsigned f[NL];
#pragma unroll
        for(int l = 0; l < NL; l++) f[l] = 0;
        const unsigned* row = ranks + lane * E4;
#if defined(QC_ABL) && (QC_ABL & 2)   // timing experiment: a fifth of the sums
        const int E4c = E4 / 5;
#else
        const int E4c = E4;
#endif
#pragma unroll 5
        for(int k = 0; k < E4c; k++) {
            const unsigned w = row[k];
#pragma unroll
            for(int l = 0; l < NL; l++) f[l] = __builtin_amdgcn_sad_u8(w, 0x01010101u * (unsigned)l, f[l]);
        }
        const long cell = tile * 64 + lane;
        unsigned cum[NL - 1];
#pragma unroll
        for(int l = 0; l < NL - 1; l++) cum[l] = (f[l + 1] - f[l] + (unsigned)E) >> 1;   // #(rank <= l); l = U: #valid members
        // (row, column) of the cell in the padded planes: one scalar division per tile, the lanes step on from there
        int y = (int)((tile * 64) / g.X), x = (int)(tile * 64 - (long)y * g.X) + lane;
        while(x >= g.X) { x -= g.X; y++; }
        unsigned char* o8 = out8 + (long)(y + QF_PADY) * g.Xp + x + QF_PADX;
        if(cum[NL - 2] != (unsigned)E) { g.rowflag[y] = 1; g.rowflag[g.Y] = 1; }
        if(!ident) {   // thresholds not in ascending order: the counts change places through the (consumed) rank area
            __syncthreads();
#pragma unroll
            for(int l = 0; l < NL - 1; l++) ranks[l * 64 + lane] = cum[l];
        }
        if(buffered) {
            if(ident) {   // U == T == NL - 2
#pragma unroll
                for(int l = 0; l < NL - 1; l++) obuf[l * 256 + sub * 64 + lane] = (unsigned char)cum[l];
            }
            else {
                for(int t = 0; t < T; t++) obuf[t * 256 + sub * 64 + lane] = (unsigned char)ranks[L->rank[t] * 64 + lane];
                obuf[T * 256 + sub * 64 + lane] = (unsigned char)cum[NL - 2];
            }
        }
        else if(ident) {
#pragma unroll
            for(int l = 0; l < NL - 1; l++) o8[(long)l * g.Pp] = (unsigned char)cum[l];
        }
        else {
            for(int t = 0; t < T; t++) o8[(long)t * g.Pp] = (unsigned char)ranks[L->rank[t] * 64 + lane];
            o8[(long)T * g.Pp] = (unsigned char)cum[NL - 2];
        }
        __syncthreads();   // the rank bytes are consumed
      }
      if(buffered) {   // lane l: cells 4 l .. 4 l + 3 of the super-tile (one row: X is a multiple of 4), every plane
          int y = (int)((st * 256) / g.X), x = (int)(st * 256 - (long)y * g.X) + 4 * lane;
          while(x >= g.X) { x -= g.X; y++; }
          unsigned char* o8 = out8 + (long)(y + QF_PADY) * g.Xp + x + QF_PADX;
          const unsigned* ob4 = reinterpret_cast<const unsigned*>(obuf);
          for(int t = 0; t <= T; t++) *reinterpret_cast<unsigned*>(o8 + (long)t * g.Pp) = ob4[t * 64 + lane];
          __syncthreads();
      }
    }
    // the last, partial tile (if any): lane-private walk straight from memory, on block 0
    if(nfull * 64 < C && blockIdx.x == 0) {
        const long cell = nfull * 64 + lane;
        if(cell < C) member_row_work<2>(in + cell * E, E, false, 0, thr, T, reinterpret_cast<float*>(out8), C, cell, g);
    }
}

// Median / exact quantile over the members of each cell (rare path; one lane per cell from memory)
__global__ void k_member_quantile(const float* __restrict__ in, long C, int E, float q, float* __restrict__ out) {
    long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(c < C) out[c] = row_quantile(in + c * E, E, q);
}

// -------------------------------------------------------------------------------------------
// separable box statistics (planes: blockIdx.z selects a [Y][X] plane)
// -------------------------------------------------------------------------------------------
// row pass: for each cell the sum (double) and count (int) of the valid values in [x-hw, x+hw]
__global__ __launch_bounds__(256) void k_box_rows(const float* __restrict__ in, int Y, int X, int hw, double* __restrict__ rs, int* __restrict__ rc,
                                                  int* __restrict__ plane_has_invalid, int ybase) {
    extern __shared__ float lds[];   // 256 + 2*hwc floats
    const long plane = (long)blockIdx.z * Y * X;
    const int y = ybase + blockIdx.y;   // (the host launches row chunks of at most 65535: gridDim.y is 16 bits)
    const int x0 = blockIdx.x * 256;
    const int hwc = min(hw, X);   // a window wider than the row is the whole row
    const float* row = in + plane + (long)y * X;
    int bad = 0;
    for(int i = threadIdx.x; i < 256 + 2 * hwc; i += 256) {
        int x = x0 - hwc + i;
        const bool inrow = x >= 0 && x < X;
        const float v = inrow ? row[x] : 0.0f;
        bad |= (inrow && !nv(v)) ? 1 : 0;
        lds[i] = inrow ? v : NAN;
    }
    const int any_bad = __syncthreads_or(bad);   // (also the barrier after the tile load)
    int x = x0 + threadIdx.x;
    double s = 0; int c = 0;
    if(any_bad) {   // some value of this tile is missing: validity per tap
        if(threadIdx.x == 0) atomicOr(&plane_has_invalid[blockIdx.z], 1);
        if(x < X) for(int k = 0; k <= 2 * hwc; k++) { float v = lds[threadIdx.x + k]; if(nv(v)) { s += (double)v; c++; } }
    }
    else {          // all valid (the common case): two instructions per tap, the count is the clipped window width
        for(int i = threadIdx.x; i < 256 + 2 * hwc; i += 256) { const int xx = x0 - hwc + i; if(!(xx >= 0 && xx < X)) lds[i] = 0.0f; }
        __syncthreads();
        if(x < X) {
#pragma unroll 8
            for(int k = 0; k <= 2 * hwc; k++) s += (double)lds[threadIdx.x + k];
            c = min(x + hwc, X - 1) - max(x - hwc, 0) + 1;
        }
    }
    if(x >= X) return;
    rs[plane + (long)y * X + x] = s;
    rc[plane + (long)y * X + x] = c;
}
// The same pass with four consecutive outputs per thread: the first one is the sum of its 2 hw + 1 taps, the next three slide the
// window (add the entering tap, subtract the leaving one) -- (2 hw + 7) / 4 taps per output instead of 2 hw + 1 when nothing is
// missing in the tile.  LDS index i lives at i + i / 4, which makes the lanes' stride odd (5 floats): no bank conflicts.
#define ROWS4_TILE 1024
__device__ __forceinline__ int pad4(int i) { return i + (i >> 2); }
__global__ __launch_bounds__(256) void k_box_rows4(const float* __restrict__ in, int Y, int X, int hw, double* __restrict__ rs, int* __restrict__ rc,
                                                   int* __restrict__ plane_has_invalid, int ybase) {
    extern __shared__ float lds[];   // pad4(ROWS4_TILE + 2*hwc) floats
    const long plane = (long)blockIdx.z * Y * X;
    const int y = ybase + blockIdx.y;
    const int x0 = blockIdx.x * ROWS4_TILE;
    const int hwc = min(hw, X);
    const int ntap = ROWS4_TILE + 2 * hwc;
    const float* row = in + plane + (long)y * X;
    int bad = 0;
    for(int i = threadIdx.x; i < ntap; i += 256) {
        const int x = x0 - hwc + i;
        const bool inrow = x >= 0 && x < X;
        const float v = inrow ? row[x] : 0.0f;
        bad |= (inrow && !nv(v)) ? 1 : 0;
        lds[pad4(i)] = inrow ? v : NAN;
    }
    const int any_bad = __syncthreads_or(bad);
    const int base = 4 * threadIdx.x;
    double s4[4]; int c4[4];
    if(any_bad) {   // some value of this tile is missing: validity per tap, every output on its own
        if(threadIdx.x == 0) atomicOr(&plane_has_invalid[blockIdx.z], 1);
#pragma unroll
        for(int j = 0; j < 4; ++j) {
            double s = 0; int c = 0;
            if(x0 + base + j < X) for(int k = 0; k <= 2 * hwc; k++) { const float v = lds[pad4(base + j + k)]; if(nv(v)) { s += (double)v; c++; } }
            s4[j] = s; c4[j] = c;
        }
    }
    else {
        for(int i = threadIdx.x; i < ntap; i += 256) { const int xx = x0 - hwc + i; if(!(xx >= 0 && xx < X)) lds[pad4(i)] = 0.0f; }
        __syncthreads();
        double s = 0;
#pragma unroll 8
        for(int k = 0; k <= 2 * hwc; k++) s += (double)lds[pad4(base + k)];
        s4[0] = s;
#pragma unroll
        for(int j = 1; j < 4; ++j) {
            s += (double)lds[pad4(base + j + 2 * hwc)];
            s -= (double)lds[pad4(base + j - 1)];
            s4[j] = s;
        }
#pragma unroll
        for(int j = 0; j < 4; ++j) { const int x = x0 + base + j; c4[j] = min(x + hwc, X - 1) - max(x - hwc, 0) + 1; }
    }
#pragma unroll
    for(int j = 0; j < 4; ++j) {
        const int x = x0 + base + j;
        if(x < X) { rs[plane + (long)y * X + x] = s4[j]; rc[plane + (long)y * X + x] = c4[j]; }
    }
}
// Row pass for windows too wide for an LDS tile (min(hw, X) beyond ~20 000 columns): one workgroup per row, every thread owns a
// contiguous segment of the row, sums the window of its first cell directly and slides it along the segment (valid values only).
__global__ __launch_bounds__(256) void k_box_rows_wide(const float* __restrict__ in, int Y, int X, int hw, double* __restrict__ rs, int* __restrict__ rc,
                                                       int* __restrict__ plane_has_invalid, int ybase) {
    const long plane = (long)blockIdx.z * Y * X;
    const int y = ybase + blockIdx.y;
    const float* row = in + plane + (long)y * X;
    const int seg = (X + 255) / 256, xa = threadIdx.x * seg, xb = min(X, xa + seg);
    if(xa >= xb) return;
    double s = 0; int c = 0, bad = 0;
    const int lo = max(xa - hw, 0), hi = (int)min((long)X - 1, (long)xa + hw);
    for(int k = lo; k <= hi; ++k) { const float v = row[k]; if(nv(v)) { s += (double)v; c++; } else bad = 1; }
    for(int x = xa; x < xb; ++x) {
        if(x > xa) {
            const long in_k = (long)x + hw; const int out_k = x - hw - 1;
            if(in_k < X) { const float v = row[in_k]; if(nv(v)) { s += (double)v; c++; } else bad = 1; }
            if(out_k >= 0) { const float v = row[out_k]; if(nv(v)) { s -= (double)v; c--; } }
        }
        rs[plane + (long)y * X + x] = s;
        rc[plane + (long)y * X + x] = c;
    }
    if(bad) atomicOr(&plane_has_invalid[blockIdx.z], 1);
}
// column pass + finish (neighbourhood.cpp:132-142): Mean / Sum / Count.  Each thread owns one column of a strip of
// COL_STRIP rows and slides the window down it (2 reads per cell instead of 2*hw+1).
#ifndef COL_STRIP
#define COL_STRIP 32   // rows per thread-strip: 32 beats 64 by 4-6 % (more strips in flight; the pass is latency-bound)
#endif
__global__ __launch_bounds__(256) void k_box_cols(const double* __restrict__ rs, const int* __restrict__ rc, int Y, int X, int hw, int statistic, float* __restrict__ out, int qf_reps,
                                                  const int* __restrict__ plane_has_invalid) {
    const long plane = (long)blockIdx.z * Y * X;
    const bool counted = plane_has_invalid[blockIdx.z] != 0;   // no missing value in the plane: counts are window sizes, rc is not read
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y0 = blockIdx.y * COL_STRIP;
    if(x >= X || y0 >= Y) return;
    const int y1 = min(Y, y0 + COL_STRIP);
    const double* cs = rs + plane + x;
    const int* cc = rc + plane + x;
    double s = 0; int c = 0;
    int lo = max(0, y0 - hw), hi = (int)min((long)Y - 1, (long)y0 + hw);   // current window [lo, hi]
    const int hwc = min(hw, X);
    const int cx = min(x + hwc, X - 1) - max(x - hwc, 0) + 1;              // row count of every row when nothing is missing
    for(int yy = lo; yy <= hi; yy++) { s += cs[(long)yy * X]; if(counted) c += cc[(long)yy * X]; }
    if(!counted) c = cx * (hi - lo + 1);
    for(int y = y0; y < y1; y++) {
        if(y > y0) {
            const long add = (long)y + hw;
            const int sub = y - hw - 1;
            if(add <= (long)Y - 1) { s += cs[add * X]; c += counted ? cc[add * X] : cx; }
            if(sub >= 0) { s -= cs[(long)sub * X]; c -= counted ? cc[(long)sub * X] : cx; }
        }
        float o = NAN;
        if(statistic == GPP_COUNT) o = (float)c;
        else if(c > 0) o = (statistic == GPP_MEAN) ? (float)(s / (double)c) : (float)s;
        if(qf_reps > 0 && nv(o)) {   // quantile_fast epilogue (neighbourhood.cpp:378-389 / 494-506): E-fold float sum / E, clamp
            float yv = o;
            if(qf_reps > 1) { float sum = 0; for(int e = 0; e < qf_reps; e++) sum += o; yv = sum / (float)qf_reps; }
            o = yv > 1 ? 1.0f : (yv < 0 ? 0.0f : yv);
        }
        out[plane + (long)y * X + x] = o;
    }
}
// Row pass + column pass in one kernel for halfwidths up to BM_MAXHW (round 5).  A workgroup marches down a strip of BM_W output columns in
// chunks of BM_C rows: the (BM_W + 2 hw) values of every new row go through LDS once, their row-window sums (doubles, sliding along segments of
// eight columns as k_box_rows4 does) and the numbers of valid values in those windows go into a ring of BM_RING rows, and the outputs whose
// 2 hw + 1 rows are in the ring are finished from it (sliding down segments of eight rows as k_box_cols does).  The values of the next chunk are
// asked for before the sums of the current one are formed, so the trip to memory hides behind them.  The plane is read 1 + 2 hw / BM_W times and
// written once; the two-kernel form writes and re-reads a plane of doubles and one of counts in between (config 4 Mean: 0.09 + 0.05 ms for the
// two passes, this kernel: see DESIGN 11.5).  (A first version without the march -- tiles of 32 x 64 outputs, each loading its halo rows again --
// took 0.22 ms: 60 KB of LDS per workgroup leave two of them on a CU, and a workgroup that loads, sums and stores once is mostly latency.)
// a workgroup barrier that orders LDS accesses only (__syncthreads also waits for every global load and store of the wave: here that would be a
// wait for the next chunk's values in the middle of the current chunk)
__device__ __forceinline__ void bm_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#define BM_W 64
#define BM_C 32
#define BM_RING 64
#define BM_MAXHW 16          // BM_C + 2 BM_MAXHW <= BM_RING
#define BM_RP (BM_W + 1)     // pitch of the ring rows (doubles)
#define BM_P (BM_W + 2 * BM_MAXHW + 1)   // pitch of the chunk's rows (floats; odd, and wide enough for every thread's unconditional store)
__host__ __device__ inline size_t bm_tin_bytes(int) { return (((size_t)BM_C * BM_P * sizeof(float)) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t bm_lds_bytes(int hw) { return bm_tin_bytes(hw) + (size_t)BM_RING * (BM_RP * sizeof(double) + BM_W + 1) + 2 * sizeof(int); }
template <int HW>
__global__ __launch_bounds__(256) void k_box_march(const float* __restrict__ in, int Y, int X, int statistic, float* __restrict__ out, int qf_reps, int SH, int seg0) {
    constexpr int hw = HW;   // (a template parameter: every window loop unrolls, its LDS reads are issued together and waited for once)
    extern __shared__ __attribute__((aligned(16))) unsigned char bm_lds[];
    constexpr int Wt = BM_W + 2 * hw, P = BM_P;
    float* const tin = reinterpret_cast<float*>(bm_lds);                                       // [BM_C][P]: the rows of the chunk, 0 outside the field
    double* const ring = reinterpret_cast<double*>(bm_lds + bm_tin_bytes(hw));                 // [BM_RING][BM_RP]: row-window sums of the strip's columns
    unsigned char* const rcnt = reinterpret_cast<unsigned char*>(ring + (size_t)BM_RING * BM_RP);   // [BM_RING][BM_W]: valid values in those windows (rows of counted chunks only)
    unsigned char* const rflag = rcnt + BM_RING * BM_W;                                        // [BM_RING]: the ring row comes from a counted chunk
    int* const bflag = reinterpret_cast<int*>(rflag + BM_RING);                                // [2]: chunk k (slot k % 2) holds a missing value
    const long plane = (long)blockIdx.z * Y * X;
    const int x0 = blockIdx.x * BM_W;
    const int ya = (seg0 + (int)blockIdx.y) * SH, yb = min(Y, ya + SH);          // this workgroup's output rows (seg0: the launch covers a band of the row segments, banded host path)
    if(ya >= yb) return;
    const int tid = threadIdx.x;
    const int yl0 = ya - hw;                                        // first row it loads; ring slot of field row y: (y - yl0) % BM_RING
    const int nchunk = (yb - 1 + hw - yl0) / BM_C + 1;
    const int clo = max(0, hw - x0), chi = min(Wt, X - x0 + hw);    // loaded columns [clo, chi) lie inside the field
    constexpr int NR = BM_C / 8, NCc = (BM_W + 2 * BM_MAXHW + 31) / 32;
    float v[NR][NCc];
    int xo[NCc]; bool colok[NCc];                                   // this thread's columns: clamped field column, inside the field
#pragma unroll
    for(int j = 0; j < NCc; j++) { const int c = (tid & 31) + 32 * j; xo[j] = min(max(x0 - hw + c, 0), X - 1); colok[j] = c >= clo && c < chi; }
    auto fetch = [&](const int k) {   // thread: rows tid / 32 + 8 i, columns tid % 32 + 32 j of chunk k (every load is made, from a clamped address: no branch per value)
#pragma unroll
        for(int i = 0; i < NR; i++) {
            const int y = yl0 + k * BM_C + (tid >> 5) + 8 * i;
            const bool rowok = y >= 0 && y < Y;
            const float* const row = in + plane + (long)min(max(y, 0), Y - 1) * X;
#pragma unroll
            for(int j = 0; j < NCc; j++) {
#if defined(BM_ABL) && (BM_ABL & 1)     // timing experiment: no loads
                const float t = (float)(y + j);
#else
                const float t = row[xo[j]];
#endif
                v[i][j] = (rowok && colok[j]) ? t : 0.0f;
            }
        }
    };
    fetch(0);
    if(tid < 2) bflag[tid] = 0;
    bm_lds_barrier();
    int ynext = ya;                                                 // first output row not yet written
    // The outputs of a chunk wait in registers and are stored one chunk later, in front of the next fetch: stores and loads complete in the order
    // they were issued, so the wait for a chunk's values would otherwise be a wait for the stores issued behind its fetch as well (measured: 0.080 ms
    // with the stores straight from the sums, whatever the arithmetic cost)
    float po[8];
    int py8 = 0, pnrow = 0;                                         // first row and number of rows of the waiting outputs (0: none)
    const int pc = tid & 63, px = x0 + pc;
    auto flush = [&]() {
#if defined(BM_ABL) && (BM_ABL & 2)     // timing experiment: no stores (but for values that do not occur)
#pragma unroll
        for(int j = 0; j < 8; j++) if(j < pnrow && po[j] == -12345.0f) out[plane + (long)(py8 + j) * X + px] = po[j];
#else
        if(pnrow == 8) {   // (the usual case under one branch)
            float* const o8 = out + plane + (long)py8 * X + px;
#pragma unroll
            for(int j = 0; j < 8; j++) o8[(long)j * X] = po[j];
        }
        else {
#pragma unroll
            for(int j = 0; j < 8; j++) if(j < pnrow) out[plane + (long)(py8 + j) * X + px] = po[j];
        }
#endif
        pnrow = 0;
    };
    unsigned hist = 0u;                                             // bit i: chunk k - i held a missing value (a window reaches back two chunks at most)
    for(int k = 0; k < nchunk; k++) {
        int bad = 0;
#pragma unroll
        for(int i = 0; i < NR; i++) {
            const int r = (tid >> 5) + 8 * i;
#pragma unroll
            for(int j = 0; j < NCc; j++) {
                const int c = (tid & 31) + 32 * j;
                tin[r * P + c] = v[i][j];                           // (c < 96 <= P)
                bad |= nv(v[i][j]) ? 0 : 1;                         // (outside the field: 0)
            }
        }
        if(bad) bflag[k & 1] = 1;
        bm_lds_barrier();                                           // the chunk is in LDS, and the previous chunk's outputs are finished
        const bool counted = bflag[k & 1] != 0;
        if(tid == 0) bflag[(k + 1) & 1] = 0;                        // (read by everybody one chunk ago, before the barrier below; written again behind it)
        hist = (hist << 1) | (counted ? 1u : 0u);
        flush();                                                    // the previous chunk's outputs
        if(k + 1 < nchunk) fetch(k + 1);                            // in flight while this chunk is summed
        {   // row-window sums of row tid / 8 of the chunk, output columns 8 (tid % 8) .. + 7
            const int r = tid >> 3, sg = tid & 7;
            const int slot = (k * BM_C + r) & (BM_RING - 1);
            const float* const t = tin + r * P + 8 * sg;            // output column 8 sg + j: loaded columns 8 sg + j .. 8 sg + j + 2 hw
            double* const rs = ring + (size_t)slot * BM_RP + 8 * sg;
            if(sg == 0) rflag[slot] = counted ? 1 : 0;
            if(!counted) {   // nothing missing in the chunk: the windows are counted in closed form when they are needed (below)
                float tv[8 + 2 * HW];
#pragma unroll
                for(int q = 0; q < 8 + 2 * HW; q++) tv[q] = t[q];
                double sc[4] = {0.0, 0.0, 0.0, 0.0};                // (four chains: the adds of one wait for each other)
#pragma unroll
                for(int q = 0; q <= 2 * HW; q++) sc[q & 3] += (double)tv[q];
                double s = (sc[0] + sc[1]) + (sc[2] + sc[3]);
                rs[0] = s;
#pragma unroll
                for(int j = 1; j < 8; j++) { s += (double)tv[j + 2 * HW]; s -= (double)tv[j - 1]; rs[j] = s; }
            }
            else {
                const int y = yl0 + k * BM_C + r;
                const bool rowok = y >= 0 && y < Y;
                unsigned char* const rc = rcnt + slot * BM_W + 8 * sg;
                auto ok = [&](const int q) { const int c = 8 * sg + q; return rowok && c >= clo && c < chi && nv(t[q]); };
                double s = 0.0; int n = 0;
                for(int q = 0; q <= 2 * hw; q++) if(ok(q)) { s += (double)t[q]; n++; }
                rs[0] = s; rc[0] = (unsigned char)n;
#pragma unroll
                for(int j = 1; j < 8; j++) {
                    if(ok(j + 2 * hw)) { s += (double)t[j + 2 * hw]; n++; }
                    if(ok(j - 1)) { s -= (double)t[j - 1]; n--; }
                    rs[j] = s; rc[j] = (unsigned char)n;
                }
            }
        }
        bm_lds_barrier();                                           // the ring holds the rows up to ytop; the chunk's LDS rows are free again
        const int ytop = yl0 + (k + 1) * BM_C - 1;
        const int ylim = min(yb, ytop - hw + 1);                    // outputs [ynext, ylim) have their 2 hw + 1 rows in the ring (at most BM_C of them)
        {
            const int c = pc, x = px;
            const int y8 = ynext + (tid >> 6) * 8;                  // this thread's eight output rows
            if(x < X && y8 < ylim) {
                py8 = y8; pnrow = min(8, ylim - y8);
                const bool slow = (hist & 7u) != 0u;                // some row of the windows may lack values: counts from the ring
                const int cx = min(x + hw, X - 1) - max(x - hw, 0) + 1;   // field columns in a row window of this column
                // valid values in the row window of field row y: counted by its chunk, or all of its field columns
                auto rown = [&](const int y) {
                    const int sl = (y - yl0) & (BM_RING - 1);
                    return rflag[sl] ? (int)rcnt[sl * BM_W + c] : ((y >= 0 && y < Y) ? cx : 0);
                };
                const double* const rp = ring + c;
                const int stop = (y8 - hw - yl0) & (BM_RING - 1);   // ring slot of the first row of the first window
                int n = 0;
                double rv[2 * HW + 8];                              // the row-window sums of the 2 hw + 8 rows under this thread's eight windows
#pragma unroll
                for(int q = 0; q < 2 * HW + 8; q++) rv[q] = rp[((stop + q) & (BM_RING - 1)) * BM_RP];   // (rows behind ylim + hw: whatever the ring holds there, not used)
                double sc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for(int q = 0; q <= 2 * HW; q++) sc[q & 3] += rv[q];
                double s = (sc[0] + sc[1]) + (sc[2] + sc[3]);
                if(slow) for(int q = -hw; q <= hw; q++) n += rown(y8 + q);
                if(!slow && statistic == GPP_MEAN && qf_reps == 0 && ynext - hw >= 0 && ylim - 1 + hw <= Y - 1) {
                    // the usual chunk: every window has its 2 hw + 1 rows inside the field and nothing is missing -- one count, one reciprocal
                    // (s / n through the reciprocal and one correction by the exact residual: see below)
                    const double nd = (double)(cx * (2 * HW + 1)), rinv = 1.0 / nd;
#pragma unroll
                    for(int j = 0; j < 8; j++) {
                        if(j > 0) { s += rv[j + 2 * HW]; s -= rv[j - 1]; }
                        const double q0 = s * rinv;
                        po[j] = (float)__builtin_fma(__builtin_fma(-q0, nd, s), rinv, q0);   // (rows from ylim on: computed, not stored)
                    }
                }
                else {
                int nprev = -1;
                double rinv = 0.0;
#pragma unroll
                for(int j = 0; j < 8; j++) {
                    const int y = y8 + j;
                    if(y >= ylim) break;
                    if(j > 0) {
                        s += rv[j + 2 * HW]; s -= rv[j - 1];
                        if(slow) n += rown(y + hw) - rown(y - hw - 1);
                    }
                    if(!slow) n = cx * (min(y + hw, Y - 1) - max(y - hw, 0) + 1);
                    float o = NAN;
                    if(statistic == GPP_COUNT) o = (float)n;
                    else if(n > 0) {
                        if(statistic == GPP_MEAN) {
                            // s / n through the reciprocal of n (the same for almost every row) and one correction by the exact residual: within
                            // an ulp(double) of the quotient -- the float it rounds to is the division's except on a rounding boundary
                            const double nd = (double)n;
                            if(n != nprev) { rinv = 1.0 / nd; nprev = n; }
                            const double q0 = s * rinv;
                            o = (float)__builtin_fma(__builtin_fma(-q0, nd, s), rinv, q0);
                        }
                        else o = (float)s;
                    }
                    if(qf_reps > 0 && nv(o)) {   // quantile_fast epilogue (neighbourhood.cpp:378-389 / 494-506): E-fold float sum / E, clamp
                        float yv = o;
                        if(qf_reps > 1) { float sum = 0; for(int e = 0; e < qf_reps; e++) sum += o; yv = sum / (float)qf_reps; }
                        o = yv > 1 ? 1.0f : (yv < 0 ? 0.0f : yv);
                    }
                    po[j] = o;
                }
                }
            }
        }
        ynext = max(ynext, ylim);
    }
    flush();
}
// Min / Max for halfwidths up to BM_MM_MAXHW in the same marching form (round 5): the row-window extrema of every new row go into a ring of floats,
// the outputs are the extrema of 2 hw + 1 ring rows.  Values that do not count (missing, infinite, outside the field) enter as the identity of
// the operation (+inf for Min, -inf for Max -- no valid value is infinite), a window that holds nothing else gives NaN (neighbourhood.cpp:144-196).
// The eight windows of a segment share the values [7, 2 hw]: their extremum once, then the running extrema of the values below and above it.
template <int HW, bool IS_MAX>
__device__ __forceinline__ void bm_window_extrema(const float (&t)[8 + 2 * HW], float (&o)[8]) {
    auto op = [](const float a, const float b) { return IS_MAX ? fmaxf(a, b) : fminf(a, b); };
    if constexpr (2 * HW >= 7) {
        float core = t[7];
#pragma unroll
        for(int q = 8; q <= 2 * HW; q++) core = op(core, t[q]);
        float lo[8], hi[8];                       // lo[j]: extremum of t[j .. 6], hi[j]: of t[2 hw + 1 .. 2 hw + j]
        lo[7] = core;
#pragma unroll
        for(int j = 6; j >= 0; j--) lo[j] = op(lo[j + 1], t[j]);
        hi[0] = core;
#pragma unroll
        for(int j = 1; j < 8; j++) hi[j] = op(hi[j - 1], t[2 * HW + j]);
#pragma unroll
        for(int j = 0; j < 8; j++) o[j] = op(lo[j], hi[j]);
    }
    else {
#pragma unroll
        for(int j = 0; j < 8; j++) {
            float m = t[j];
#pragma unroll
            for(int q = 1; q <= 2 * HW; q++) m = op(m, t[j + q]);
            o[j] = m;
        }
    }
}
// Halfwidths 17 .. 32 run the same kernel over strips of 32 columns with a ring of 128 rows (the loaded row is 32 + 2 hw <= 96 values wide either way).
#define BM_MM_MAXHW 32
template <int HW> struct MinMaxGeom { static constexpr int W = HW <= BM_MAXHW ? BM_W : BM_W / 2, RING = HW <= BM_MAXHW ? BM_RING : 2 * BM_RING, FP = W + 1; };
template <int HW> __host__ __device__ inline size_t bm_minmax_lds_bytes() { return bm_tin_bytes(0) + (size_t)MinMaxGeom<HW>::RING * MinMaxGeom<HW>::FP * sizeof(float); }
template <int HW, bool IS_MAX>
__global__ __launch_bounds__(256) void k_minmax_march(const float* __restrict__ in, int Y, int X, float* __restrict__ out, int SH, int seg0) {
    constexpr int hw = HW;
    constexpr int BM_W_ = MinMaxGeom<HW>::W, BM_RING_ = MinMaxGeom<HW>::RING, BM_FP = MinMaxGeom<HW>::FP;
    static_assert(BM_W_ + 2 * HW <= BM_P - 1 && BM_C + 2 * HW <= BM_RING_, "the chunk's rows and the ring hold the windows");
    extern __shared__ __attribute__((aligned(16))) unsigned char bm_lds[];
    constexpr int P = BM_P;
    const float ident = IS_MAX ? -INFINITY : INFINITY;
    float* const tin = reinterpret_cast<float*>(bm_lds);                               // [BM_C][P]: the rows of the chunk, `ident` where nothing counts
    float* const ring = reinterpret_cast<float*>(bm_lds + bm_tin_bytes(0));            // [BM_RING][BM_FP]: row-window extrema of the strip's columns
    const int x0 = blockIdx.x * BM_W_;
    const int ya = (seg0 + (int)blockIdx.y) * SH, yb = min(Y, ya + SH);
    if(ya >= yb) return;
    const int tid = threadIdx.x;
    const int yl0 = ya - hw;
    const int nchunk = (yb - 1 + hw - yl0) / BM_C + 1;
    constexpr int Wt = BM_W_ + 2 * hw;
    const int clo = max(0, hw - x0), chi = min(Wt, X - x0 + hw);
    constexpr int NR = BM_C / 8, NCc = (BM_W + 2 * BM_MAXHW + 31) / 32;
    float v[NR][NCc];
    int xo[NCc]; bool colok[NCc];
#pragma unroll
    for(int j = 0; j < NCc; j++) { const int c = (tid & 31) + 32 * j; xo[j] = min(max(x0 - hw + c, 0), X - 1); colok[j] = c >= clo && c < chi; }
    auto fetch = [&](const int k) {
#pragma unroll
        for(int i = 0; i < NR; i++) {
            const int y = yl0 + k * BM_C + (tid >> 5) + 8 * i;
            const bool rowok = y >= 0 && y < Y;
            const float* const row = in + (long)min(max(y, 0), Y - 1) * X;
#pragma unroll
            for(int j = 0; j < NCc; j++) {
                const float t = row[xo[j]];
                v[i][j] = (rowok && colok[j] && nv(t)) ? t : ident;
            }
        }
    };
    fetch(0);
    int ynext = ya;
    float po[8];
    int py8 = 0, pnrow = 0;
    const int pc = tid % BM_W_, px = x0 + pc;
    auto flush = [&]() {
        if(pnrow == 8) {
            float* const o8 = out + (long)py8 * X + px;
#pragma unroll
            for(int j = 0; j < 8; j++) o8[(long)j * X] = po[j];
        }
        else {
#pragma unroll
            for(int j = 0; j < 8; j++) if(j < pnrow) out[(long)(py8 + j) * X + px] = po[j];
        }
        pnrow = 0;
    };
    for(int k = 0; k < nchunk; k++) {
#pragma unroll
        for(int i = 0; i < NR; i++)
#pragma unroll
            for(int j = 0; j < NCc; j++) tin[((tid >> 5) + 8 * i) * P + (tid & 31) + 32 * j] = v[i][j];
        bm_lds_barrier();                                           // the chunk is in LDS, and the previous chunk's outputs are finished
        flush();
        if(k + 1 < nchunk) fetch(k + 1);
        {
            const int r = tid / (BM_W_ / 8), sg = tid % (BM_W_ / 8);
            if(r < BM_C) {
                const int slot = (k * BM_C + r) & (BM_RING_ - 1);
                const float* const t = tin + r * P + 8 * sg;
                float tv[8 + 2 * HW], o[8];
#pragma unroll
                for(int q = 0; q < 8 + 2 * HW; q++) tv[q] = t[q];
                bm_window_extrema<HW, IS_MAX>(tv, o);
                float* const rs = ring + slot * BM_FP + 8 * sg;
#pragma unroll
                for(int j = 0; j < 8; j++) rs[j] = o[j];
            }
        }
        bm_lds_barrier();
        const int ytop = yl0 + (k + 1) * BM_C - 1;
        const int ylim = min(yb, ytop - hw + 1);
        {
            const int y8 = ynext + (tid / BM_W_) * 8;
            if(px < X && y8 < ylim) {
                py8 = y8; pnrow = min(8, ylim - y8);
                const float* const rp = ring + pc;
                const int stop = (y8 - hw - yl0) & (BM_RING_ - 1);
                float rv[8 + 2 * HW], o[8];
#pragma unroll
                for(int q = 0; q < 8 + 2 * HW; q++) rv[q] = rp[((stop + q) & (BM_RING_ - 1)) * BM_FP];   // (rows behind ylim + hw: not used)
                bm_window_extrema<HW, IS_MAX>(rv, o);
#pragma unroll
                for(int j = 0; j < 8; j++) po[j] = (o[j] == ident) ? NAN : o[j];
            }
        }
        ynext = max(ynext, ylim);
    }
    flush();
}
// separable min / max ignoring non-finite values (dir 0: along x, dir 1: along y)
__global__ __launch_bounds__(256) void k_minmax_pass(const float* __restrict__ in, int Y, int X, int hw, int is_max, int dir, float* __restrict__ out, int ybase) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = ybase + blockIdx.y * 4 + (threadIdx.x >> 6);
    if(x >= X || y >= Y) return;
    float m = NAN;
    if(dir == 0) {
        const int a = max(0, x - hw), b = (int)min((long)X - 1, (long)x + hw);
        for(int k = a; k <= b; k++) { float v = in[(long)y * X + k]; if(nv(v) && (!nv(m) || (is_max ? v > m : v < m))) m = v; }
    }
    else {
        const int a = max(0, y - hw), b = (int)min((long)Y - 1, (long)y + hw);
        for(int k = a; k <= b; k++) { float v = in[(long)k * X + x]; if(nv(v) && (!nv(m) || (is_max ? v > m : v < m))) m = v; }
    }
    out[(long)y * X + x] = m;
}
__global__ void k_fill_nan(float* out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = NAN;
}
__global__ void k_square(const float* __restrict__ in, long n, float* __restrict__ out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) { float v = in[i]; out[i] = v * v; }
}
// neighbourhood.cpp:222-233
__global__ void k_std_finish(const float* __restrict__ mean, const float* __restrict__ mean2, long n, int is_std, float* __restrict__ out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) { float v = mean2[i] - mean[i] * mean[i]; out[i] = is_std ? sqrtf(v) : v; }
}

// -------------------------------------------------------------------------------------------
// brute force: one wavefront per cell, window (x members) gathered on the fly
// -------------------------------------------------------------------------------------------
struct Window {
    const float* in; int X, E, ya, yb, xa, xb;
    __device__ int count() const { return (yb - ya + 1) * (xb - xa + 1) * E; }
    // element k of the window in the reference's gather order (neighbourhood.cpp:579-587,631-640)
    __device__ float at(int k) const {
        int rowlen = (xb - xa + 1) * E;
        int r = k / rowlen, o = k - r * rowlen;
        return in[((long)(ya + r) * X + xa) * E + o];
    }
};
__device__ __forceinline__ int wave_sum_i(int v) { for(int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off); return v; }

__device__ float wave_kth(const Window& w, int n, int k, int lane) {
    unsigned lo = 0, hi = 0xffffffffu;
    while(lo < hi) {
        unsigned mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for(int i = lane; i < n; i += 64) { float v = w.at(i); if(nv(v) && f2ord(v) <= mid) c++; }
        c = wave_sum_i(c);
        if(c >= k + 1) hi = mid; else lo = mid + 1;
    }
    return ord2f(lo);
}
// statistic over the window with the reference's semantics (calc_statistic / calc_quantile)
__global__ __launch_bounds__(256) void k_brute(const float* __restrict__ in, int Y, int X, int E, int hw, int statistic, float q, unsigned seed, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long cell = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if(cell >= (long)Y * X) return;
    const int y = (int)(cell / X), x = (int)(cell - (long)y * X);
    Window w{in, X, E, max(0, y - hw), (int)min((long)Y - 1, (long)y + hw), max(0, x - hw), (int)min((long)X - 1, (long)x + hw)};
    const int n = w.count();
    float value = NAN;
    if(statistic == GPP_MEAN || statistic == GPP_SUM || statistic == GPP_COUNT || statistic == GPP_STD || statistic == GPP_VARIANCE) {
        // sequential float accumulation in gather order (util.cpp:22-38,41-72): one lane walks the window
        if(lane == 0) {
            float total = 0, total2 = 0, K = NAN; int count = 0;
            const bool sv = (statistic == GPP_STD || statistic == GPP_VARIANCE);
            for(int i = 0; i < n; i++) {
                float v = w.at(i);
                if(!nv(v)) continue;
                if(sv) { if(!nv(K)) K = v; float d = v - K; total += d; total2 += d * d; }
                else total += v;
                count++;
            }
            if(statistic == GPP_COUNT) value = (float)count;
            else if(count > 0) {
                if(statistic == GPP_MEAN) value = total / (float)count;
                else if(statistic == GPP_SUM) value = total;
                else { float mean = total / (float)count, mean2 = total2 / (float)count; float var = mean2 - mean * mean; if(var < 0) var = 0; value = (statistic == GPP_STD) ? sqrtf(var) : var; }
            }
        }
    }
    else {
        int N = 0;
        for(int i = lane; i < n; i += 64) if(nv(w.at(i))) N++;
        N = wave_sum_i(N);
        if(statistic == GPP_RANDOMCHOICE) {   // util.cpp:74-95 uses rand(); any valid element is a correct draw
            if(N > 0) {
                unsigned h = (unsigned)cell * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                value = wave_kth(w, n, (int)(h % (unsigned)N), lane);   // rank-th smallest valid value
            }
        }
        else {
            float qq = (statistic == GPP_MIN) ? 0.0f : (statistic == GPP_MEDIAN) ? 0.5f : (statistic == GPP_MAX) ? 1.0f : q;
            if(N > 0 && nv(qq)) {
                if(qq == 0) value = wave_kth(w, n, 0, lane);
                else if(qq == 1) value = wave_kth(w, n, N - 1, lane);
                else {
                    float pos = qq * (float)(N - 1);
                    int li = (int)floorf(pos), ui = (int)ceilf(pos);
                    float lv = wave_kth(w, n, li, lane);
                    float uv = (ui == li) ? lv : wave_kth(w, n, ui, lane);
                    value = quantile_from_order(qq, N, lv, uv, li, ui);
                }
            }
        }
    }
    if(lane == 0) out[cell] = value;
}

// -------------------------------------------------------------------------------------------
// quantile_fast finish: neighbourhood.cpp:483-522 + util.cpp:339-414
// -------------------------------------------------------------------------------------------
// stats plane -> yarray plane, in place: E-fold float accumulation then / count (3-D form :494-499), clamp to [0,1]
__global__ void k_qf_yarray(float* __restrict__ planes, long n, int reps) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    float s = planes[i];
    if(!nv(s)) { planes[i] = NAN; return; }
    float yv;
    if(reps <= 1) yv = s;   // 2-D form (:378-384): sum = s, count = 1
    else { float sum = 0; for(int e = 0; e < reps; e++) sum += s; yv = sum / (float)reps; }
    if(yv > 1) yv = 1; else if(yv < 0) yv = 0;
    planes[i] = yv;
}
__global__ void k_qf_interp(const float* __restrict__ ya, long C, int T, const float* __restrict__ thr, const float* __restrict__ q, int qfield, float* __restrict__ out) {
    long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(c >= C) return;
    const float x = qfield ? q[c] : q[0];
    bool missing = false;
    for(int t = 0; t < T; t++) if(!nv(ya[(long)t * C + c])) missing = true;
    float o = NAN;
    if(!missing) {
        const float y0a = ya[c], yLa = ya[(long)(T - 1) * C + c];
        if(x == 1 && y0a == 1) o = thr[0];
        else if(x == 0 && yLa == 0) o = thr[T - 1];
        else if(!nv(x)) o = NAN;                       // interpolate(): util.cpp:378-379
        else if(x > yLa) o = thr[T - 1];               // util.cpp:386-389
        else if(x < y0a) o = thr[0];
        else {
            int i0 = -1, i1 = -1;                      // get_lower_index / get_upper_index (util.cpp:339-376)
            for(int i = 0; i < T; i++) { float cv = ya[(long)i * C + c]; if(cv < x) i0 = i; else if(cv == x) { i0 = i; break; } else break; }
            for(int i = T - 1; i >= 0; i--) { float cv = ya[(long)i * C + c]; if(cv > x) i1 = i; else if(cv == x) { i1 = i; break; } else break; }
            if(i0 < 0) i0 = 0;
            if(i1 < 0) i1 = T - 1;
            const float x0 = ya[(long)i0 * C + c], x1 = ya[(long)i1 * C + c], y0 = thr[i0], y1 = thr[i1];
            if(x0 == x1) {
                if(i0 == 0 && i1 == T - 1) o = (y0 + y1) / 2;
                else if(i0 == 0) o = y1;
                else if(i1 == T - 1) o = y0;
                else o = (y0 + y1) / 2;
            }
            else o = y0 + (y1 - y0) * (x - x0) / (x1 - x0);
        }
    }
    out[c] = o;
}

// -------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------
namespace {
struct NbWorkspace {
    DevBuf<float> flat, tmp, tmp2, planes, thr, qf;
    DevBuf<double> rs;
    DevBuf<int> rc, plane_flags;
    const void* pad_ptr = nullptr;   // the padding of the quantile_fast count planes is in place for this buffer (this ALLOCATION of it) and shape
    unsigned long long pad_gen = 0;
    int pad_y = 0, pad_x = 0, pad_t = 0, pad_e = 0;
    hipEvent_t ev_band0 = nullptr, ev_up[6] = {};   // banded host path of gpp_neighbourhood
    int spec_nt = -1, spec_U = -1;   // quantile_fast: the number of distinct thresholds the last call with spec_nt thresholds had (its table was usable)
    int* h_pin = nullptr;            // a few page-locked words for the read-back at the end of such a call
};
thread_local NbWorkspace g_nb;

#ifdef GPP_POISON
// Diagnostic build only (tools/hostile/build.sh, tools/nbh_hostile_soak.py): the call-to-call workspaces of the neighbourhood family set to
// `byte`.  The byte planes of the fused quantile_fast path carry state across calls BY DESIGN (their padding is written once per layout,
// gpp_neighbourhood_quantile_fast below): keep_padding = 0 poisons the whole buffer and forgets the layout (the next call must lay the
// padding out again); keep_padding = 1 poisons everything EXCEPT the padding of the remembered layout -- every cell byte of every plane
// -- so that the cache is exercised while a count pass that leaves a cell unwritten is still caught.
__global__ void k_poison_plane_cells(unsigned char* __restrict__ cnt8, QfGeom g, int nplanes, int byte) {
    const long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(cell >= (long)g.Y * g.X) return;
    const long o = qf_cell_offset(g, cell);
    for(int t = 0; t < nplanes; t++) cnt8[(size_t)t * g.Pp + o] = (unsigned char)byte;
}
extern "C" int gpp_debug_poison_nbh_workspace(int byte, int keep_padding) {
    GPP_TRY
    ensure_device();
    NbWorkspace& w = g_nb;
    w.flat.poison(byte); w.tmp.poison(byte); w.tmp2.poison(byte); w.thr.poison(byte); w.qf.poison(byte);
    w.rs.poison(byte); w.rc.poison(byte); w.plane_flags.poison(byte);
    if(keep_padding && w.pad_ptr && w.pad_ptr == (const void*)w.planes.p && w.pad_gen == w.planes.gen) {
        const QfGeom g = qf_geom(w.pad_y, w.pad_x);
        const long C = (long)g.Y * g.X;
        hipLaunchKernelGGL(k_poison_plane_cells, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream(), reinterpret_cast<unsigned char*>(w.planes.p), g, w.pad_t + 1, byte);
        GPP_HIP(hipGetLastError());
    }
    else { w.planes.poison(byte); w.pad_ptr = nullptr; }
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}
#endif

template <int MODE>
void member_pass_launch(const float* d_in, long C, int E, int statistic, const float* d_thr, int T, float* d_out, const QfGeom& g) {
    const long tiles = (C + 63) / 64;
    const int nchunk = (16 * E + 63) / 64;
    const bool dma = E <= MEMBER_EC && (reinterpret_cast<size_t>(d_in) & 15) == 0;
    const size_t lds = dma ? (size_t)(MODE != 0 ? 1 : 2) * nchunk * 1024 : 16;   // (double) buffer of whole 1 KiB chunks
    static std::once_flag attr_once;
    std::call_once(attr_once, [] { GPP_HIP(hipFuncSetAttribute((const void*)k_member_pass<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ((16 * MEMBER_EC + 63) / 64) * 1024)); });
    const int waves_per_cu = (int)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / std::max<size_t>(lds, 1)));
    const long grid = std::min<long>(tiles, (long)256 * waves_per_cu);
    hipLaunchKernelGGL((k_member_pass<MODE>), dim3((unsigned)grid), dim3(64), lds, stream(), d_in, C, E, statistic, d_thr, T, d_out, dma ? 1 : 0, g);
    GPP_HIP(hipGetLastError());
}
void member_pass(const float* d_in, long C, int E, int mode, int statistic, const float* d_thr, int T, float* d_out, const QfGeom& g = QfGeom()) {
    if(mode == 0) member_pass_launch<0>(d_in, C, E, statistic, d_thr, T, d_out, g);
    else if(mode == 2) member_pass_launch<2>(d_in, C, E, statistic, d_thr, T, d_out, g);
    else member_pass_launch<1>(d_in, C, E, statistic, d_thr, T, d_out, g);
}
// byte counts of quantile_fast by ranks (k_qf_count<U + 2>)
template <int NL>
void qf_count_launch_nl(const float* d_in, long C, int E, const float* d_thr, int T, const QfLut* lut, unsigned char* cnt8, const QfGeom& g, const int Uexp) {
    const size_t lds = (size_t)(2 * QF_NB + std::max(16 * E, NL * 64)) * sizeof(unsigned) + (size_t)(T + 1) * 256;
    int waves_per_cu = (int)std::max<size_t>(1, std::min<size_t>(24, (160 * 1024) / lds));
    if(path_env("GPP_QF_WAVES")) waves_per_cu = std::max(1, std::min(waves_per_cu, atoi(path_env("GPP_QF_WAVES"))));   // (A/B: fewer, longer streams)
    const long grid = std::max<long>(1, std::min<long>((C / 64 + 3) / 4, (long)256 * waves_per_cu));
    hipLaunchKernelGGL((k_qf_count<NL>), dim3((unsigned)grid), dim3(64), lds, stream(), d_in, C, E, d_thr, T, lut, cnt8, g, Uexp);
    GPP_HIP(hipGetLastError());
}
void qf_count_launch(const float* d_in, long C, int E, const float* d_thr, int T, const QfLut* lut, int U, unsigned char* cnt8, const QfGeom& g, const int Uexp) {
    switch(U + 2) {
#define QF_NL_CASE(n) case n: qf_count_launch_nl<n>(d_in, C, E, d_thr, T, lut, cnt8, g, Uexp); break;
        QF_NL_CASE(3) QF_NL_CASE(4) QF_NL_CASE(5) QF_NL_CASE(6) QF_NL_CASE(7) QF_NL_CASE(8) QF_NL_CASE(9) QF_NL_CASE(10)
        QF_NL_CASE(11) QF_NL_CASE(12) QF_NL_CASE(13) QF_NL_CASE(14) QF_NL_CASE(15) QF_NL_CASE(16) QF_NL_CASE(17) QF_NL_CASE(18)
#undef QF_NL_CASE
        default: runtime("Internal error. quantile_fast: number of distinct thresholds");
    }
}
// Mean / Sum / Count of `nplanes` [Y][X] planes
// The launch geometry of the marching kernels for a [Y][X] plane: row segments of SH rows, `segs` of them.  (seg0, nseg): the band of segments a launch covers --
// all of them, or one band of the banded host path of gpp_neighbourhood.
struct MarchGeom { int SH = 0, segs = 0; bool ok = false; };
MarchGeom march_geom(int Y, int X, int hw, int statistic, int nplanes = 1) {
    MarchGeom g;
    if(path_env("GPP_BOX_TWO_PASS")) return g;
    if(statistic == GPP_MEAN || statistic == GPP_SUM || statistic == GPP_COUNT) {
        if(hw > BM_MAXHW || nplanes > 65535) return g;
        const int strips = (X + BM_W - 1) / BM_W;
        const long fill = path_env("GPP_BM_FILL") ? std::max(1, atoi(path_env("GPP_BM_FILL"))) : 768;   // (A/B: workgroups the launch aims for)
        const long want = std::max<long>(1, fill / std::max<long>(1, (long)strips * nplanes));
        const int segs = (int)std::min<long>(want, (Y + BM_C - 1) / BM_C);
        g.SH = ((Y + segs - 1) / segs + BM_C - 1) / BM_C * BM_C;
    }
    else if(statistic == GPP_MIN || statistic == GPP_MAX) {
        if(hw > BM_MM_MAXHW) return g;
        const int W = hw <= BM_MAXHW ? BM_W : BM_W / 2;
        const int strips = (X + W - 1) / W;
        const long fill = path_env("GPP_BM_FILL") ? std::max(1, atoi(path_env("GPP_BM_FILL"))) : 1280;
        const long want = std::max<long>(1, fill / strips);
        const int segs = (int)std::min<long>(want, (Y + BM_C - 1) / BM_C);
        g.SH = ((Y + segs - 1) / segs + BM_C - 1) / BM_C * BM_C;
    }
    else return g;
    while((Y + g.SH - 1) / g.SH > 65535) g.SH += BM_C;
    g.segs = (Y + g.SH - 1) / g.SH;
    g.ok = true;
    return g;
}
void box_stat(const float* d_in, int Y, int X, int nplanes, int hw, int statistic, float* d_out, int qf_reps = 0, int seg0 = 0, int nseg = -1) {
    if(hw <= BM_MAXHW && nplanes <= 65535 && !path_env("GPP_BOX_TWO_PASS")) {   // both passes in one kernel (k_box_march)
        // about three workgroups per CU: strips x row segments x planes; a segment is a whole number of chunks (its first chunk is run-in: 2 hw rows)
        const int strips = (X + BM_W - 1) / BM_W;
        const MarchGeom mg = march_geom(Y, X, hw, GPP_MEAN, nplanes);
        const int SH = mg.SH;
        const dim3 grid(strips, nseg < 0 ? mg.segs : nseg, nplanes);
        switch(hw) {
#define BM_CASE(n) case n: hipLaunchKernelGGL(k_box_march<n>, grid, dim3(256), bm_lds_bytes(n), stream(), d_in, Y, X, statistic, d_out, qf_reps, SH, seg0); break;
            BM_CASE(0) BM_CASE(1) BM_CASE(2) BM_CASE(3) BM_CASE(4) BM_CASE(5) BM_CASE(6) BM_CASE(7) BM_CASE(8)
            BM_CASE(9) BM_CASE(10) BM_CASE(11) BM_CASE(12) BM_CASE(13) BM_CASE(14) BM_CASE(15) BM_CASE(16)
#undef BM_CASE
        }
        GPP_HIP(hipGetLastError());
        return;
    }
    long n = (long)Y * X * nplanes;
    double* rs = g_nb.rs.get(n);
    int* rc = g_nb.rc.get(n);
    int hwc = std::min(hw, X);
    const bool wide = (256 + 2 * (size_t)hwc) * sizeof(float) > 160 * 1024 || path_env("GPP_BOX_ROWS_WIDE");   // no LDS tile holds the window
    static std::once_flag lds_once;
    if(!wide && (256 + 2 * (size_t)hwc) * sizeof(float) > 64 * 1024)
        std::call_once(lds_once, [] { GPP_HIP(hipFuncSetAttribute((const void*)k_box_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); });
    int* flags = g_nb.plane_flags.get(nplanes);
    GPP_HIP(hipMemsetAsync(flags, 0, sizeof(int) * nplanes, stream()));
    for(int ybase = 0; ybase < Y; ybase += 65535) {   // gridDim.y is a 16-bit quantity
        const int ny = std::min(65535, Y - ybase);
        if(wide)
            hipLaunchKernelGGL(k_box_rows_wide, dim3(1, ny, nplanes), dim3(256), 0, stream(), d_in, Y, X, hw, rs, rc, flags, ybase);
        else if(hwc <= 1024 && hwc > 0 && !path_env("GPP_BOX_ROWS_DIRECT")) {
            const int ntap = ROWS4_TILE + 2 * hwc;
            hipLaunchKernelGGL(k_box_rows4, dim3((X + ROWS4_TILE - 1) / ROWS4_TILE, ny, nplanes), dim3(256), (size_t)(ntap + (ntap >> 2) + 4) * sizeof(float), stream(),
                               d_in, Y, X, hw, rs, rc, flags, ybase);
        }
        else
            hipLaunchKernelGGL(k_box_rows, dim3((X + 255) / 256, ny, nplanes), dim3(256), (256 + 2 * hwc) * sizeof(float), stream(), d_in, Y, X, hw, rs, rc, flags, ybase);
        GPP_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_box_cols, dim3((X + 255) / 256, (Y + COL_STRIP - 1) / COL_STRIP, nplanes), dim3(256), 0, stream(), rs, rc, Y, X, hw, statistic, d_out, qf_reps, flags);
    GPP_HIP(hipGetLastError());
}
void brute(const float* d_in, int Y, int X, int E, int hw, int statistic, float q, float* d_out) {
    long C = (long)Y * X;
    static std::atomic<unsigned> seed_state{12345u};
    const unsigned seed = seed_state.fetch_add(1013904223u) * 1664525u + 1013904223u;
    hipLaunchKernelGGL(k_brute, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, stream(), d_in, Y, X, E, hw, statistic, q, seed, d_out);
    GPP_HIP(hipGetLastError());
}
// neighbourhood(vec2, hw, stat) on a device-resident plane (neighbourhood.cpp:28-242)
void neighbourhood2d(const float* d_in, int Y, int X, int hw, int statistic, float* d_out, int seg0 = 0, int nseg = -1) {
    long C = (long)Y * X;
    if(statistic == GPP_MEAN || statistic == GPP_SUM || statistic == GPP_COUNT) box_stat(d_in, Y, X, 1, hw, statistic, d_out, 0, seg0, nseg);
    else if((statistic == GPP_MIN || statistic == GPP_MAX) && hw <= BM_MM_MAXHW && !path_env("GPP_BOX_TWO_PASS")) {   // both passes in one kernel (k_minmax_march)
        const int W = hw <= BM_MAXHW ? BM_W : BM_W / 2;
        const int strips = (X + W - 1) / W;
        const MarchGeom mg = march_geom(Y, X, hw, statistic);
        const int SH = mg.SH;
        const dim3 grid(strips, nseg < 0 ? mg.segs : nseg);
        const bool mx = statistic == GPP_MAX;
        switch(hw) {
#define BM_CASE(n) case n: if(mx) hipLaunchKernelGGL((k_minmax_march<n, true>), grid, dim3(256), bm_minmax_lds_bytes<n>(), stream(), d_in, Y, X, d_out, SH, seg0); \
                           else hipLaunchKernelGGL((k_minmax_march<n, false>), grid, dim3(256), bm_minmax_lds_bytes<n>(), stream(), d_in, Y, X, d_out, SH, seg0); break;
            BM_CASE(0) BM_CASE(1) BM_CASE(2) BM_CASE(3) BM_CASE(4) BM_CASE(5) BM_CASE(6) BM_CASE(7) BM_CASE(8)
            BM_CASE(9) BM_CASE(10) BM_CASE(11) BM_CASE(12) BM_CASE(13) BM_CASE(14) BM_CASE(15) BM_CASE(16)
            BM_CASE(17) BM_CASE(18) BM_CASE(19) BM_CASE(20) BM_CASE(21) BM_CASE(22) BM_CASE(23) BM_CASE(24)
            BM_CASE(25) BM_CASE(26) BM_CASE(27) BM_CASE(28) BM_CASE(29) BM_CASE(30) BM_CASE(31) BM_CASE(32)
#undef BM_CASE
        }
        GPP_HIP(hipGetLastError());
    }
    else if(statistic == GPP_MIN || statistic == GPP_MAX) {
        float* t = g_nb.tmp.get(C);
        for(int dir = 1; dir >= 0; --dir)
            for(int ybase = 0; ybase < Y; ybase += 4 * 65535) {
                const dim3 grid((X + 63) / 64, (std::min(4 * 65535, Y - ybase) + 3) / 4);
                hipLaunchKernelGGL(k_minmax_pass, grid, dim3(256), 0, stream(), dir ? d_in : (const float*)t, Y, X, hw, statistic == GPP_MAX, dir, dir ? t : d_out, ybase);
                GPP_HIP(hipGetLastError());
            }
    }
    else if(statistic == GPP_STD || statistic == GPP_VARIANCE) {
        float* mean = g_nb.tmp.get(C);
        float* sq = g_nb.tmp2.get(2 * C);
        float* mean2 = sq + C;
        box_stat(d_in, Y, X, 1, hw, GPP_MEAN, mean);
        hipLaunchKernelGGL(k_square, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream(), d_in, C, sq);
        box_stat(sq, Y, X, 1, hw, GPP_MEAN, mean2);
        hipLaunchKernelGGL(k_std_finish, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream(), (const float*)mean, (const float*)mean2, C, statistic == GPP_STD, d_out);
        GPP_HIP(hipGetLastError());
    }
    else brute(d_in, Y, X, 1, hw, statistic, 0.0f, d_out);   // Median, RandomChoice (:236-238)
}
}   // namespace

static void check_stat(int s) {
    switch(s) {
        case GPP_MEAN: case GPP_MIN: case GPP_MEDIAN: case GPP_MAX: case GPP_QUANTILE: case GPP_STD: case GPP_VARIANCE:
        case GPP_SUM: case GPP_COUNT: case GPP_RANDOMCHOICE: return;
        default: runtime("Internal error. Cannot compute statistic");
    }
}

extern "C" int gpp_neighbourhood(const float* input, int ny, int nx, int ne, int is3d, int halfwidth, int statistic, float* out, int mem) {
    GPP_TRY
    if(halfwidth < 0) invalid("Half width must be > 0");                                          // :29-30
    if(statistic == GPP_QUANTILE) invalid("Use neighbourhood_quantile for computing neighbourhood quantiles");   // :31-32
    check_stat(statistic);
    if(ny < 0 || nx < 0 || ne < 0) invalid("negative size");
    if(ny == 0 || nx == 0 || ne == 0) return GPP_OK;                                                // :33-34
    ensure_device();
    const long C = (long)ny * nx;
    InField in; OutField o;
    // Round 6: a large 2-D plane in host memory (numpy in, numpy out: 64 MB up, a 0.07 ms kernel, 64 MB down -- 2.4 ms for a 4000 x 4000 plane) travels in
    // bands of the marching kernels' row segments: a band's rows (+ the halfwidth rows below it) go up on the second stream, the band's launch waits for
    // them, and a page-locked result array is written by the kernels themselves (mapped host memory: optimal_interpolation's host path, oi.hip).
    if(!is3d && !(mem & GPP_MEM_DEVICE) && C >= (1L << 20) && !path_env("GPP_NBH_NO_BANDS")) {
        const MarchGeom mg = march_geom(ny, nx, halfwidth, statistic);
        hipPointerAttribute_t at;
        void* dp = nullptr;
        bool okp = mg.ok && mg.segs >= 4 && hipPointerGetAttributes(&at, out) == hipSuccess && at.type == hipMemoryTypeHost && hipHostGetDevicePointer(&dp, out, 0) == hipSuccess && dp != nullptr;
        (void)hipGetLastError();
        if(okp) {
            const bool f64 = (mem & GPP_HOST_F64) != 0;
            float* const d_in = in.staged.get((size_t)C);
            Staged<double> wide;
            if(f64) wide.get((size_t)C);
            if(!g_nb.ev_band0) {
                GPP_HIP(hipEventCreateWithFlags(&g_nb.ev_band0, hipEventDisableTiming));
                for(int b = 0; b < 6; b++) GPP_HIP(hipEventCreateWithFlags(&g_nb.ev_up[b], hipEventDisableTiming));
            }
            struct Guard { ~Guard() { (void)hipStreamSynchronize(stream2()); (void)hipStreamSynchronize(stream()); } } guard;   // (nothing of the call stays in flight, whatever ends it)
            const hipStream_t sUp = stream2();
            GPP_HIP(hipEventRecord(g_nb.ev_band0, stream()));
            GPP_HIP(hipStreamWaitEvent(sUp, g_nb.ev_band0, 0));     // (the staging buffer's previous readers are behind the library stream)
            static const int share[6] = {1, 2, 3, 3, 2, 1};
            const int nband = std::min(6, mg.segs);
            int s0 = 0, acc = 0, tot = 0;
            for(int b = 0; b < nband; b++) tot += share[b];
            long up = 0;    // rows uploaded so far
            for(int b = 0; b < nband; b++) {
                acc += share[b];
                const int s1 = b == nband - 1 ? mg.segs : std::max(s0 + 1, (int)((long)mg.segs * acc / tot));
                const long need = std::min<long>(ny, (long)s1 * mg.SH + halfwidth);       // the band's windows reach `halfwidth` rows below its last row
                if(need > up) {
                    const size_t off = (size_t)up * nx, cnt = (size_t)(need - up) * nx;
                    if(!f64) GPP_HIP(hipMemcpyAsync(d_in + off, input + off, cnt * sizeof(float), hipMemcpyHostToDevice, sUp));
                    else {
                        GPP_HIP(hipMemcpyAsync(wide.p + off, reinterpret_cast<const double*>(input) + off, cnt * sizeof(double), hipMemcpyHostToDevice, sUp));
                        hipLaunchKernelGGL(k_stage_f64, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, sUp, (const double*)(wide.p + off), cnt, d_in + off);
                        GPP_HIP(hipGetLastError());
                    }
                    up = need;
                }
                GPP_HIP(hipEventRecord(g_nb.ev_up[b], sUp));
                GPP_HIP(hipStreamWaitEvent(stream(), g_nb.ev_up[b], 0));
                neighbourhood2d(d_in, ny, nx, halfwidth, statistic, static_cast<float*>(dp), s0, s1 - s0);
                s0 = s1;
            }
            GPP_HIP(hipStreamSynchronize(stream()));
            GPP_HIP(hipStreamSynchronize(sUp));
            return GPP_OK;
        }
    }
    in.bind(input, (size_t)C * ne, mem);
    o.bind(out, C, mem);
    const float* plane = in.d;
    if(is3d) {   // 3-D form: member statistic first (:12-27)
        float* flat = g_nb.flat.get(C);
        if(statistic == GPP_MEDIAN) hipLaunchKernelGGL(k_member_quantile, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream(), in.d, C, ne, 0.5f, flat);
        else if(statistic == GPP_RANDOMCHOICE) brute(in.d, ny, nx, ne, 0, GPP_RANDOMCHOICE, 0, flat);
        else member_pass(in.d, C, ne, 0, statistic, nullptr, 0, flat);
        plane = flat;
    }
    neighbourhood2d(plane, ny, nx, halfwidth, statistic, o.d);
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_neighbourhood_brute_force(const float* input, int ny, int nx, int ne, int halfwidth, int statistic, float quantile, float* out, int mem) {
    GPP_TRY
    if(halfwidth < 0) invalid("Half width must be > 0");   // :558-559
    check_stat(statistic);
    if(statistic == GPP_QUANTILE && (quantile < 0 || quantile > 1))
        invalid("calc_quantile: Quantile must be between 0 and 1 inclusive");   // util.cpp:113-115
    if(ny <= 0 || nx <= 0 || ne <= 0) return GPP_OK;
    ensure_device();
    const long C = (long)ny * nx;
    InField in; OutField o;
    in.bind(input, (size_t)C * ne, mem);
    o.bind(out, C, mem);
    brute(in.d, ny, nx, ne, halfwidth, statistic, quantile, o.d);
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_neighbourhood_quantile_fast(const float* input, int ny, int nx, int ne, int is3d, const float* quantile, int nq,
                                               int halfwidth, const float* thresholds, int nt, float* out, int mem) {
    GPP_TRY
    if(halfwidth < 0) invalid("Half width must be > 0");                                   // :303-304,419-420
    if(ny <= 0 || nx <= 0 || ne <= 0) return GPP_OK;                                        // :306-307
    const long C = (long)ny * nx;
    if(nq != 1 && nq != C) invalid("Quantile must be the same size as input, or size (1, 1)");   // :312-313
    if(nt < 0) invalid("negative number of thresholds");
    ensure_device();
    InField in, qf, th;
    OutField o;
    in.bind(input, (size_t)C * ne, mem);
    o.bind(out, C, mem);
    // 3-D input with at most 254 members, 16 thresholds and a halfwidth of 16: byte counts in padded planes + the box pass of
    // qf_box.hip.  The counts come from the rank / sum-of-absolute-differences pass (k_qf_count) when the rows are whole float4s
    const bool fused = nt > 0 && is3d && ne <= 254 && nt <= 16 && halfwidth <= QF_MAXHW && C < (1L << 31) && !path_env("GPP_QF_NO_FUSED");
    bool ranked = fused && (ne & 3) == 0 && (reinterpret_cast<size_t>(in.d) & 15) == 0 && !path_env("GPP_QF_NO_RANKS");
    if(nt > 0) th.bind(thresholds, nt, mem & ~GPP_HOST_F64);   // GPP_HOST_F64 applies to `input` only: quantile / thresholds stay float32
    // GPP_Q_HOST: the quantile argument is host memory although the field is in HBM (the scalar quantile of a script beside a device-resident
    // cube: uploaded by the caller and read back here for its validation it cost two transfers and a host round trip per call)
    const int qmem = (mem & GPP_Q_HOST) ? GPP_MEM_HOST : (mem & ~GPP_HOST_F64);
    auto check_q = [&](const float* hq) {   // :315-321
        for(int i = 0; i < nq; i++)
            if(is_valid(hq[i]) && (hq[i] < 0 || hq[i] > 1)) invalid("All quantiles must be >= 0 and <= 1");
    };
    const bool q_on_device = (qmem & GPP_MEM_DEVICE) != 0;
    if(!q_on_device) check_q(quantile);
    if(nt == 0) {   // :330-331: all missing
        if(q_on_device) {
            std::vector<float> hq(nq);
            GPP_HIP(hipMemcpyAsync(hq.data(), quantile, sizeof(float) * nq, hipMemcpyDeviceToHost, stream())); GPP_HIP(hipStreamSynchronize(stream()));
            check_q(hq.data());
        }
        hipLaunchKernelGGL(k_fill_nan, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream(), o.d, C);
        o.finish();
        GPP_HIP(hipStreamSynchronize(stream()));
        return GPP_OK;
    }
    qf.bind(quantile, nq, qmem);
    if(!g_nb.h_pin) GPP_HIP(hipHostMalloc((void**)&g_nb.h_pin, 64, hipHostMallocDefault));
    // Two rounds at most: the first may launch the count pass for the number of distinct thresholds the LAST call had (no host round trip in
    // front of the kernels: the table's head and the quantile are read back once, behind the box pass); a table that turns out different
    // stops the passes on the device and the second round does what every call did until round 5 -- wait for the table, then launch.
    for(int round = 0; round < 2; round++) {
        QfLut* lut = nullptr;
        int lut_head[5] = {0, 0, 0, 1, 0};   // scale, off, U, flag, ident
        int* rowflag = nullptr;
        const bool spec = fused && ranked && round == 0 && g_nb.spec_nt == nt && g_nb.spec_U >= 0 && !path_env("GPP_QF_NO_SPEC");
        if(fused) rowflag = g_nb.plane_flags.get(ny + 2);
        if(ranked) {
            lut = reinterpret_cast<QfLut*>(g_nb.qf.get((sizeof(QfLut) + 3) / 4));
            hipLaunchKernelGGL(k_qf_lut, dim3(1), dim3(QF_NB), 0, stream(), th.d, nt, lut, rowflag, fused ? ny + 2 : 0);
            GPP_HIP(hipGetLastError());
            if(!spec) GPP_HIP(hipMemcpyAsync(lut_head, lut, sizeof(lut_head), hipMemcpyDeviceToHost, stream()));
        }
        else if(fused) GPP_HIP(hipMemsetAsync(rowflag, 0, sizeof(int) * (ny + 2), stream()));
        std::vector<float> hq;
        if(q_on_device && !spec) {   // quantile validation needs the values on the host
            hq.resize(nq);
            GPP_HIP(hipMemcpyAsync(hq.data(), quantile, sizeof(float) * nq, hipMemcpyDeviceToHost, stream()));
        }
        if(!spec && (ranked || q_on_device)) GPP_HIP(hipStreamSynchronize(stream()));
        if(q_on_device && !spec) check_q(hq.data());
        if(!spec && ranked) {
            if(lut_head[3]) ranked = false;   // two thresholds in one bucket / a non-finite threshold: the compare-per-threshold pass
            else { g_nb.spec_nt = nt; g_nb.spec_U = lut_head[2]; }
        }
        if(!fused) break;
        QfGeom g = qf_geom(ny, nx);
        unsigned char* cnt8 = reinterpret_cast<unsigned char*>(g_nb.planes.get(((size_t)(nt + 1) * g.Pp + 3) / 4));
        if(g_nb.pad_ptr != cnt8 || g_nb.pad_gen != g_nb.planes.gen || g_nb.pad_y != ny || g_nb.pad_x != nx || g_nb.pad_t != nt || g_nb.pad_e != ne) {
            // the padding is written when the planes are laid out (the passes below only ever write the cells of the field)
            g_nb.pad_ptr = nullptr;
            GPP_HIP(hipMemsetAsync(cnt8, 255, (size_t)nt * g.Pp, stream()));
            GPP_HIP(hipMemsetAsync(cnt8 + (size_t)nt * g.Pp, ne, (size_t)g.Pp, stream()));
            g_nb.pad_ptr = cnt8; g_nb.pad_gen = g_nb.planes.gen; g_nb.pad_y = ny; g_nb.pad_x = nx; g_nb.pad_t = nt; g_nb.pad_e = ne;
        }
        g.rowflag = rowflag;
        if(ranked) qf_count_launch(in.d, C, ne, th.d, nt, lut, spec ? g_nb.spec_U : lut_head[2], cnt8, g, spec ? g_nb.spec_U : -1);
        else member_pass(in.d, C, ne, 2, 0, th.d, nt, reinterpret_cast<float*>(cnt8), g);
#ifdef QF_SIDE_EXPERIMENT   // timing experiment only: the box pass on the second stream WITHOUT waiting for the counts (wrong results)
        if(path_env("GPP_QF_SIDE")) { qf_box_launch(cnt8, g, ne, halfwidth, nt, th.d, qf.d, nq == 1 ? 0 : 1, o.d, stream2()); GPP_HIP(hipStreamSynchronize(stream2())); }
        else
#endif
        qf_box_launch(cnt8, g, ne, halfwidth, nt, th.d, qf.d, nq == 1 ? 0 : 1, o.d);
        if(spec) {   // the one read-back of the call: did the count pass run, and the quantile(s) for the validation the slow round does first
            GPP_HIP(hipMemcpyAsync(g_nb.h_pin, rowflag + ny + 1, sizeof(int), hipMemcpyDeviceToHost, stream()));
            if(q_on_device) {
                if(nq == 1) GPP_HIP(hipMemcpyAsync(g_nb.h_pin + 1, quantile, sizeof(float), hipMemcpyDeviceToHost, stream()));
                else { hq.resize(nq); GPP_HIP(hipMemcpyAsync(hq.data(), quantile, sizeof(float) * nq, hipMemcpyDeviceToHost, stream())); }
            }
            GPP_HIP(hipStreamSynchronize(stream()));
            if(q_on_device) check_q(nq == 1 ? reinterpret_cast<const float*>(g_nb.h_pin + 1) : hq.data());
            if(g_nb.h_pin[0] != 0) { g_nb.spec_nt = -1; g_nb.spec_U = -1; continue; }   // other thresholds than the last call's: once more, the slow way
        }
        o.finish();
        GPP_HIP(hipStreamSynchronize(stream()));
        return GPP_OK;
    }
    g_nb.pad_ptr = nullptr;   // (the unfused path below uses the same buffer)
    float* planes = g_nb.planes.get((size_t)nt * C);
    float* stats = g_nb.tmp2.get((size_t)nt * C);
    member_pass(in.d, C, ne, 1, 0, th.d, nt, planes);                 // fractions per threshold (:453-472)
    // stats[t] = neighbourhood(temp, hw, Mean) (:473) with the yarray epilogue (:494-506) fused into the column pass
    box_stat(planes, ny, nx, nt, halfwidth, GPP_MEAN, stats, is3d ? ne : 1);
    hipLaunchKernelGGL(k_qf_interp, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream(), (const float*)stats, C, nt, th.d, qf.d, nq == 1 ? 0 : 1, o.d);
    GPP_HIP(hipGetLastError());
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

// -------------------------------------------------------------------------------------------
// util: calc_statistic / calc_quantile on rows, calc_even_quantiles / get_neighbourhood_thresholds
// -------------------------------------------------------------------------------------------
__global__ void k_rows_quantile(const float* __restrict__ in, long rows, int len, const float* __restrict__ q, int qfield, float* __restrict__ out) {
    long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(r < rows) out[r] = row_quantile(in + r * len, len, qfield ? q[r] : q[0]);
}
__global__ void k_rows_statistic(const float* __restrict__ in, long rows, int len, int statistic, float* __restrict__ out) {
    long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(r >= rows) return;
    if(statistic == GPP_MEDIAN) out[r] = row_quantile(in + r * len, len, 0.5f);
    else out[r] = row_statistic(in + r * len, len, statistic);
}

// gridpp::calc_statistic(vec2, statistic) (util.cpp:19-110,208-215): one value per row
extern "C" int gpp_calc_statistic(const float* array, long rows, int len, int statistic, float* out, int mem) {
    GPP_TRY
    check_stat(statistic);
    if(statistic == GPP_QUANTILE || statistic == GPP_RANDOMCHOICE) runtime("Internal error. Cannot compute statistic");
    if(rows <= 0) return GPP_OK;
    ensure_device();
    InField in; OutField o;
    in.bind(array, (size_t)rows * len, mem);
    o.bind(out, rows, mem);
    if(len <= MEMBER_EC && statistic != GPP_MEDIAN && len > 0) member_pass(in.d, rows, len, 0, statistic, nullptr, 0, o.d);
    else hipLaunchKernelGGL(k_rows_statistic, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream(), in.d, rows, len, statistic, o.d);
    GPP_HIP(hipGetLastError());
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}
// gridpp::calc_quantile(vec, q) / (vec2, q) / (vec3, vec2 q) (util.cpp:111-207): nq == 1 or nq == rows
extern "C" int gpp_calc_quantile(const float* array, long rows, int len, const float* quantile, long nq, float* out, int mem) {
    GPP_TRY
    if(nq != 1 && nq != rows) invalid("Dimension mismatch between array and quantile");
    if(rows <= 0) return GPP_OK;
    ensure_device();
    std::vector<float> hq(nq);
    if(mem & GPP_MEM_DEVICE) { GPP_HIP(hipMemcpyAsync(hq.data(), quantile, sizeof(float) * nq, hipMemcpyDeviceToHost, stream())); GPP_HIP(hipStreamSynchronize(stream())); }
    else memcpy(hq.data(), quantile, sizeof(float) * nq);
    for(long i = 0; i < nq; i++)
        if(hq[i] < 0 || hq[i] > 1) invalid("calc_quantile: Quantile must be between 0 and 1 inclusive");   // util.cpp:113-115 (NaN passes)
    InField in, q; OutField o;
    in.bind(array, (size_t)rows * len, mem);
    q.bind(quantile, nq, mem);
    o.bind(out, rows, mem);
    hipLaunchKernelGGL(k_rows_quantile, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream(), in.d, rows, len, q.d, nq == 1 ? 0 : 1, o.d);
    GPP_HIP(hipGetLastError());
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

__global__ void k_count_equal(const float* __restrict__ v, long n, float x, unsigned long long* __restrict__ cnt) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool eq = i < n && v[i] == x;
    unsigned long long m = __ballot(eq);
    if((threadIdx.x & 63) == 0 && m) atomicAdd(cnt, (unsigned long long)__popcll(m));
}
struct IsValidF { __device__ bool operator()(const float& v) const { return !isnan(v) && !isinf(v); } };

// gridpp::calc_even_quantiles (util.cpp:261-338) on the valid values of `values`; the global sort and the
// unique pass run on the device (rocPRIM radix sort / select / unique), the index picks on the host.
// out must hold `num` floats; *count is the number written.
extern "C" int gpp_calc_even_quantiles(const float* values, long n, int num, int only_valid, float* out, int* count, int mem) {
    GPP_TRY
    if(!count) invalid("count is NULL");
    *count = 0;
    if(num <= 0 || n <= 0) return GPP_OK;
    if(n > 0x7fffffffL) runtime("calc_even_quantiles: more than 2^31 values");
    ensure_device();
    InField in;
    in.bind(values, n, mem);
    DevBuf<float> valid, sorted, uniq;
    DevBuf<int> dnum;
    DevBuf<char> tmp;
    DevBuf<unsigned long long> dcnt;
    valid.get(n); sorted.get(n); uniq.get(n); dnum.get(1); dcnt.get(1);
    size_t bytes = 0;
    int nvalid = (int)n;
    const float* src = in.d;
    if(only_valid) {
        GPP_HIP(rocprim::select((void*)nullptr, bytes, in.d, valid.p, dnum.p, (size_t)n, IsValidF(), stream()));
        tmp.get(bytes);
        GPP_HIP(rocprim::select((void*)tmp.p, bytes, in.d, valid.p, dnum.p, (size_t)n, IsValidF(), stream()));
        GPP_HIP(hipMemcpyAsync(&nvalid, dnum.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
        GPP_HIP(hipStreamSynchronize(stream()));
        src = valid.p;
    }
    const int size = nvalid;
    if(size == 0) return GPP_OK;
    bytes = 0;
    GPP_HIP(rocprim::radix_sort_keys((void*)nullptr, bytes, src, sorted.p, (size_t)size, 0u, 32u, stream()));
    tmp.get(bytes);
    GPP_HIP(rocprim::radix_sort_keys((void*)tmp.p, bytes, src, sorted.p, (size_t)size, 0u, 32u, stream()));
    bytes = 0;
    GPP_HIP(rocprim::unique((void*)nullptr, bytes, sorted.p, uniq.p, dnum.p, (size_t)size, rocprim::equal_to<float>(), stream()));
    tmp.get(bytes);
    GPP_HIP(rocprim::unique((void*)tmp.p, bytes, sorted.p, uniq.p, dnum.p, (size_t)size, rocprim::equal_to<float>(), stream()));
    int nu = 0;
    GPP_HIP(hipMemcpyAsync(&nu, dnum.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    auto fetch = [&](const float* d, long i) { float v; GPP_HIP(hipMemcpyAsync(&v, d + i, sizeof(float), hipMemcpyDeviceToHost, stream())); GPP_HIP(hipStreamSynchronize(stream())); return v; };
    std::vector<float> q;
    if(num >= size) {   // util.cpp:271-281: all unique values
        q.resize(nu);
        GPP_HIP(hipMemcpyAsync(q.data(), uniq.p, sizeof(float) * nu, hipMemcpyDeviceToHost, stream()));
        GPP_HIP(hipStreamSynchronize(stream()));
    }
    else {
        const float lowest = fetch(uniq.p, 0), highest = fetch(uniq.p, nu - 1);
        GPP_HIP(hipMemsetAsync(dcnt.p, 0, sizeof(unsigned long long), stream()));
        hipLaunchKernelGGL(k_count_equal, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, stream(), (const float*)sorted.p, (long)size, lowest, dcnt.p);
        unsigned long long cl = 0;
        GPP_HIP(hipMemcpyAsync(&cl, dcnt.p, sizeof(cl), hipMemcpyDeviceToHost, stream()));
        GPP_HIP(hipStreamSynchronize(stream()));
        const long count_lower = (long)cl;
        q.push_back(lowest);
        if(num == 2) { if(lowest != highest) q.push_back(highest); }
        else {
            int first_remaining = 1;   // index into uniq of the first value > last_added
            const bool repeated = count_lower < size && count_lower > size / num;   // util.cpp:306
            if(repeated) { q.push_back(fetch(uniq.p, 1)); first_remaining = 2; }
            const long nrem = nu - first_remaining;
            if(nrem > 0) {
                const int num_left = num - (int)q.size();
                for(int i = 1; i <= num_left; i++) {
                    float f = float(i) / (num_left);
                    int index = (int)((float)nrem * f - 1);   // util.cpp:322
                    if(index >= 0) q.push_back(fetch(uniq.p, first_remaining + index));
                    else runtime("Internal error in calc_even_quantiles.");
                }
            }
        }
    }
    *count = (int)q.size();
    if(mem & GPP_MEM_DEVICE) GPP_HIP(hipMemcpyAsync(out, q.data(), sizeof(float) * q.size(), hipMemcpyHostToDevice, stream()));
    else memcpy(out, q.data(), sizeof(float) * q.size());
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}
