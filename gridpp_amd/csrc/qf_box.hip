// neighbourhood_quantile_fast, 3-D input: the box pass over the count planes (gfx950).
//
// Replaces src/api/neighbourhood.cpp:473-522 (per threshold: stats = neighbourhood(temp, halfwidth, Mean); per cell: the
// E-fold float sum of the clamped means, then interpolate()) and util.cpp:339-414, on the byte planes of qf_box.h.
//
// k_qf_box<HW, GENERAL>: one workgroup marches down a strip of 256 output columns.  Thread (p, s) owns threshold plane p and
// the 8 columns of segment s (QB_SEG) and keeps, in registers, the exact window sums V[8] of its cells (doubles: every temp
// is a float32 in [0, 1] with a resolution of 2^-31 at worst and a window holds at most 33^2 of them, so no partial sum is
// ever rounded -- the reference's summed-area table differs from them by its own rounding only).  One step = one output
// row: V += H(row y + HW) - H(row y - HW - 1), where H is the horizontal window sum of a row.  Both rows come as 40 bytes
// (five 8-byte loads, always aligned and always in bounds thanks to the padding: the 8 columns and 16 on either side),
// every byte is turned into temp = count / E through a 256-entry table of doubles in LDS (ds_read_b64; 255 = padding ->
// 0.0), and the 8 window sums of the difference row cost ONE pass over its 8 + 2 HW values: with a pivot inside all the
// windows of a block, the running suffix sums below the pivot and the running prefix sums above it are exactly the two
// parts of every window.
// No LDS tile, no barrier and no other thread is involved until the 8 clamped means (yarray) of the step are ready;
// they meet the other planes' in LDS, where thread c interpolates column c of the strip (util.cpp:377-414).
// Rows in which some cell has fewer than E valid members (rowflag) form count / valid per cell and count the valid
// cells of every window; everywhere else the window counts are known from the geometry alone.
#include "qf_box.h"
#include <algorithm>
#include <numeric>

#pragma clang fp contract(off)
using namespace gpp;

namespace {
#define QB_SEG 8        // output columns per thread
#define QB_NDW ((QB_SEG + 32) / 4)   // dwords of a row a thread reads: its columns and 16 on either side
#define QB_SW 256       // output columns per workgroup (32 segments)
#ifndef QB_SH
#define QB_SH 64        // output rows per workgroup (plus 2 HW + 1 rows of run-in)
#endif

// a workgroup barrier that orders LDS accesses only (the threads of a workgroup share nothing else; __syncthreads also waits for the global
// stores of the interpolation stage)
__device__ __forceinline__ void qb_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ bool qb_nv(float v) { return !isnan(v) && !isinf(v); }

// 8 * byte I of the dword array w (I is a constant after unrolling): one instruction, the byte offset of a table entry
__device__ __forceinline__ unsigned qb_byte_x8(const unsigned (&w)[QB_NDW], const int i) {
    unsigned a;
    const unsigned d = w[i >> 2];
    switch(i & 3) {
        case 0: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a) : "v"(3u), "v"(d)); break;
        case 1: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a) : "v"(3u), "v"(d)); break;
        case 2: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a) : "v"(3u), "v"(d)); break;
        default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a) : "v"(3u), "v"(d)); break;
    }
    return a;
}
__device__ __forceinline__ unsigned qb_byte(const unsigned (&w)[QB_NDW], const int i) { return (w[i >> 2] >> (8 * (i & 3))) & 0xffu; }
// the table of doubles sits at LDS address 0 (checked at kernel entry): the byte offset is the address
__device__ __forceinline__ double qb_tab(const unsigned a) { return *(const __attribute__((address_space(3))) double*)(size_t)a; }

// the QB_SEG window sums of one row of QB_SEG + 2 HW values val(i), i = byte index in the 40-byte row (column x0 - 16 + i):
// acc(j, sum of val over [16 - HW + j, 16 + HW + j]).  Blocks of BS <= 2 HW + 1 outputs share a pivot.
template <int HW, class Val, class Acc>
__device__ __forceinline__ void qb_windows(Val val, Acc acc) {
    constexpr int W = 2 * HW + 1;
    constexpr int BS = W >= QB_SEG ? QB_SEG : (W >= 4 ? 4 : (W >= 2 ? 2 : 1));
    constexpr int a = 16 - HW, b = 16 + HW;
#pragma unroll
    for(int j0 = 0; j0 < QB_SEG; j0 += BS) {
        const int m = a + j0 + BS - 1;   // first value of the block's last window: inside every window of the block
        auto S = val(m);
        acc(m - a, S);
#pragma unroll
        for(int i = m - 1; i >= a + j0; i--) { S += val(i); acc(i - a, S); }
        if(b + j0 + BS - 1 > m) {
            auto R = val(m + 1);
            if(m + 1 >= b + j0) acc(m + 1 - b, R);
#pragma unroll
            for(int i = m + 2; i <= b + j0 + BS - 1; i++) { R += val(i); if(i >= b + j0) acc(i - b, R); }
        }
    }
}

__device__ float qb_interp(const float* __restrict__ ya, const int stride, const int T, const float* __restrict__ thr, const float x) {
    bool missing = false;
    for(int t = 0; t < T; t++) if(!qb_nv(ya[t * stride])) missing = true;
    if(missing) return NAN;
    const float y0a = ya[0], yLa = ya[(T - 1) * stride];
    if(x == 1 && y0a == 1) return thr[0];                     // neighbourhood.cpp:508-513
    if(x == 0 && yLa == 0) return thr[T - 1];
    if(!qb_nv(x)) return NAN;                                  // interpolate(): util.cpp:378-379
    if(x > yLa) return thr[T - 1];                             // util.cpp:386-389
    if(x < y0a) return thr[0];
    int i0 = -1, i1 = -1;                                      // get_lower_index / get_upper_index (util.cpp:339-376)
    for(int i = 0; i < T; i++) { const float cv = ya[i * stride]; if(cv < x) i0 = i; else if(cv == x) { i0 = i; break; } else break; }
    for(int i = T - 1; i >= 0; i--) { const float cv = ya[i * stride]; if(cv > x) i1 = i; else if(cv == x) { i1 = i; break; } else break; }
    if(i0 < 0) i0 = 0;
    if(i1 < 0) i1 = T - 1;
    const float x0 = ya[i0 * stride], x1 = ya[i1 * stride], t0 = thr[i0], t1 = thr[i1];
    if(x0 == x1) {
        if(i0 == 0 && i1 == T - 1) return (t0 + t1) / 2;
        if(i0 == 0) return t1;
        if(i1 == T - 1) return t0;
        return (t0 + t1) / 2;
    }
    return t0 + (t1 - t0) * (x - x0) / (x1 - x0);
}

// F(o): what the reference makes of a window mean o before it interpolates -- `sum += o` E times in float, / E, clamped to [0, 1]
// (neighbourhood.cpp:494-506).  acc = the E-fold sum.
__device__ __forceinline__ float qb_fold_finish(const float acc, const float o, const int reps, const double rreps) {
    // acc / E through the double reciprocal: exact for E < 2^8 (acc / E can neither hit nor come within 2^-41 of a float32
    // rounding boundary, the double product is within 2^-52 of it)
    const float yv = reps > 1 ? (float)((double)acc * rreps) : o;
    return yv > 1 ? 1.0f : (yv < 0 ? 0.0f : yv);
}

#ifndef QB_LAZY
#define QB_LAZY 1
#endif
// Round 6: the E-fold sums where interpolate() looks.  The threads (p, s) leave the window MEANS o in LDS, not F(o); thread c of the
// interpolation stage needs F only at the two thresholds that bracket its quantile: the comparisons of interpolate() (util.cpp:339-389,
// neighbourhood.cpp:508-513) come out the same on o as on F(o) whenever o is CLEARLY on one side of x, because F is monotone and stays
// close to o:  s_E = fl(... fl(o + o) ... + o) differs from E o by at most u o (E (E + 1) / 2 - 1) (1 + 2e-5), u = 2^-24 (every addition
// rounds the running sum, which is <= k o (1 + 2e-5) after k terms), the division and the rounding to float add 2^-24 + 2^-52:
//      |F(o) - o| <= 7.7e-6 o   for E <= 255 (the byte planes hold no more);   F(0) = 0 and F(1) = 1 exactly; the clamp only moves F towards o.
// "Clearly": o (1 + 1.2e-5) < x or o (1 - 1.2e-5) > x (or o is exactly 0 or 1, or E = 1: then F(o) = o and the comparison itself is
// exact).  A column with a threshold that is NOT clearly on one side takes all its F(o) the slow way and then the unchanged interpolate().
// Per row of a strip: 256 x 2 sums instead of 256 x T (config 4: k_qf_box<15> 0.96 -> 0.86 ms; vector instructions 322 M -> 216 M -- DESIGN 4.3).
__device__ float qb_interp_lazy(float* __restrict__ ya, const int stride, const int T, const float* __restrict__ thr, const float x,
                                const int reps, const double rreps) {
    bool missing = false;
    unsigned done = 0, amb = 0;          // bit t: ya[t] holds F(o_t) / o_t is not clearly on one side of x
    for(int t = 0; t < T; t++) {
        const float v = ya[t * stride];
        if(!qb_nv(v)) missing = true;
        if(v == 0.0f || v == 1.0f || reps <= 1) done |= 1u << t;
        else if(!(v * 1.000012f < x) && !(v * 0.999988f > x)) amb |= 1u << t;   // (a NaN quantile lands here too)
    }
    if(missing) return NAN;
    // a threshold that is not clearly on one side: F itself there and at both neighbours (one of them is the other end of the bracket), three sums
    // side by side, written back over the means.  (Config 4 with the quantile ON a threshold's expected rank -- q = 0.5, thresholds 0 .. 10 over
    // uniform [0, 10) members: the window means of that plane scatter around q itself -- has such a column in every fifth wave.)
    while(__any(amb != 0u)) {
        const int a = amb ? __builtin_ctz(amb) : 0;
        const int ia = max(a - 1, 0), ib = min(a + 1, T - 1);
        const float va = ya[ia * stride], vm = ya[a * stride], vb = ya[ib * stride];
        float sa = 0.0f, sm = 0.0f, sb = 0.0f;
#pragma unroll 4
        for(int e = 0; e < reps; e++) { sa += va; sm += vm; sb += vb; }
        if(amb) {
            if(!((done >> ia) & 1u)) ya[ia * stride] = qb_fold_finish(sa, va, reps, rreps);
            if(!((done >> a) & 1u)) ya[a * stride] = qb_fold_finish(sm, vm, reps, rreps);
            if(!((done >> ib) & 1u)) ya[ib * stride] = qb_fold_finish(sb, vb, reps, rreps);
            const unsigned bits = (1u << ia) | (1u << a) | (1u << ib);
            done |= bits; amb &= ~bits;
        }
    }
    // interpolate() on the column: every comparison below has the outcome it has on F(o)
    const float y0a = ya[0], yLa = ya[(T - 1) * stride];
    if(x == 1 && y0a == 1) return thr[0];                     // neighbourhood.cpp:508-513
    if(x == 0 && yLa == 0) return thr[T - 1];
    if(!qb_nv(x)) return NAN;                                  // interpolate(): util.cpp:378-379
    if(x > yLa) return thr[T - 1];                             // util.cpp:386-389
    if(x < y0a) return thr[0];
    int i0 = -1, i1 = -1;                                      // get_lower_index / get_upper_index (util.cpp:339-376)
    for(int i = 0; i < T; i++) { const float cv = ya[i * stride]; if(cv < x) i0 = i; else if(cv == x) { i0 = i; break; } else break; }
    for(int i = T - 1; i >= 0; i--) { const float cv = ya[i * stride]; if(cv > x) i1 = i; else if(cv == x) { i1 = i; break; } else break; }
    if(i0 < 0) i0 = 0;
    if(i1 < 0) i1 = T - 1;
    // ... and F itself at the two thresholds the result is interpolated between
    const float o0 = ya[i0 * stride], o1 = ya[i1 * stride];
    const bool have0 = ((done >> i0) & 1u) != 0u, have1 = ((done >> i1) & 1u) != 0u;
    float x0 = o0, x1 = o1;
    if(__any(!have0 || !have1)) {
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll 4
        for(int e = 0; e < reps; e++) { a0 += o0; a1 += o1; }
        if(!have0) x0 = qb_fold_finish(a0, o0, reps, rreps);
        if(!have1) x1 = qb_fold_finish(a1, o1, reps, rreps);
    }
    const float t0 = thr[i0], t1 = thr[i1];
    if(x0 == x1) {
        if(i0 == 0 && i1 == T - 1) return (t0 + t1) / 2;
        if(i0 == 0) return t1;
        if(i1 == T - 1) return t0;
        return (t0 + t1) / 2;
    }
    return t0 + (t1 - t0) * (x - x0) / (x1 - x0);
}

// GENERAL = false: every cell of the field has E valid members (g.rowflag[Y] == 0, else the kernel returns at once);
// GENERAL = true: the other case (returns at once when no row is flagged).  Both are launched, one of them works.
template <int HW, bool GENERAL>
__global__ __launch_bounds__(512) void k_qf_box(const unsigned char* __restrict__ cnt8, const QfGeom g, const int reps, const int T,
                                                const float* __restrict__ thr, const float* __restrict__ q, const int qfield, float* __restrict__ out, const int SH) {
    extern __shared__ double qb_lds[];
    double* const tab = qb_lds;                                             // [256] count -> (double)(float)(count / E); 255 -> 0
    float* const sthr = reinterpret_cast<float*>(tab + 256);                // [16]
    float* const ya = sthr + 16;                                            // [2][T][QB_SW] clamped means of the current / previous step
    if((unsigned)(size_t)(__attribute__((address_space(3))) double*)qb_lds != 0u) __builtin_trap();   // see qb_tab()
    if(g.rowflag[g.Y + 1] != 0) return;      // (the count pass did not run: launched for another table than this call's, see k_qf_count)
    if((g.rowflag[g.Y] != 0) != GENERAL) return;
    const int tid = threadIdx.x;
    const int p = tid / (QB_SW / QB_SEG), s = tid % (QB_SW / QB_SEG);
    const int X = g.X, Y = g.Y;
    const int xs = blockIdx.x * QB_SW;                                      // first column of the strip
    const int x0 = xs + s * QB_SEG;                                         // first column of this thread's segment
    const int ybeg = blockIdx.y * SH, yend = min(Y, ybeg + SH);
    const bool active = p < T && x0 < X;
    for(int k = tid; k < 256; k += blockDim.x) tab[k] = (k <= reps) ? (double)((float)k / (float)reps) : 0.0;
    if(tid < 16) sthr[tid] = tid < T ? thr[tid] : 0.0f;
    __syncthreads();
    // byte i of a loaded row = padded column x0 + i = field column x0 - 16 + i (QF_PADX == 16)
    const unsigned char* const plane = cnt8 + (long)(active ? p : 0) * g.Pp + (active ? x0 : 0);
    const unsigned char* const vplane = cnt8 + (long)T * g.Pp + (active ? x0 : 0);
    auto load_row = [&](const unsigned char* base, const int y, unsigned (&w)[QB_NDW]) __attribute__((always_inline)) {   // y = field row, or < -QF_PADY..: row 0 of the plane is padding
        typedef unsigned u2 __attribute__((ext_vector_type(2)));   // (a segment starts on a multiple of 8 columns)
        const u2* rp = reinterpret_cast<const u2*>(base + (long)(y + QF_PADY) * g.Xp);
#pragma unroll
        for(int k = 0; k < QB_NDW / 2; k++) { const u2 v = rp[k]; w[2 * k] = v.x; w[2 * k + 1] = v.y; }
    };
    double V[QB_SEG];
    int Vc[QB_SEG];
#pragma unroll
    for(int j = 0; j < QB_SEG; j++) { V[j] = 0.0; Vc[j] = 0; }
    // the number of field columns in the window of column x (rows whose every cell has E valid members count their cells this way)
    auto nxw = [&](const int j) __attribute__((always_inline)) { const int x = x0 + j; return min(x + HW, X - 1) - max(x - HW, 0) + 1; };
    // one row of a flagged kind: count / valid per cell (neighbourhood.cpp:465-470), and the valid cells counted
    auto flagged_row = [&](const int y, const int sign) __attribute__((always_inline)) {
        unsigned b[QB_NDW], v[QB_NDW];
        load_row(plane, y, b);
        load_row(vplane, y, v);
        const double sg = (double)sign;
        qb_windows<HW>([&](const int i) {
                           const unsigned cb = qb_byte(b, i), cv = qb_byte(v, i);
                           const bool ok = (unsigned)(x0 - 16 + i) < (unsigned)X && cv > 0;
                           return ok ? (double)((float)cb / (float)cv) : 0.0; },
                       [&](const int j, const double S) { V[j] += sg * S; });
        qb_windows<HW>([&](const int i) { return ((unsigned)(x0 - 16 + i) < (unsigned)X && qb_byte(v, i) > 0) ? 1 : 0; },
                       [&](const int j, const int S) { Vc[j] += sign * S; });
    };
    auto plain_row = [&](const int y, const int sign) __attribute__((always_inline)) {   // every cell of the row has E valid members (or the row is outside the field: padding)
        unsigned b[QB_NDW];
        load_row(plane, y, b);
        const double sg = (double)sign;
        qb_windows<HW>([&](const int i) { return qb_tab(qb_byte_x8(b, i)); }, [&](const int j, const double S) { V[j] += sg * S; });
        if(y >= 0 && y < Y) {
#pragma unroll
            for(int j = 0; j < QB_SEG; j++) Vc[j] += sign * nxw(j);
        }
    };
    // step: the window of rows moves from [y - HW - 1, y + HW - 1] to [y - HW, y + HW]
    auto step = [&](const int yin, const int yout) __attribute__((always_inline)) {
        const bool in_field = yin >= 0 && yin < Y, out_field = yout >= 0 && yout < Y;
        bool fin = false, fout = false;
        if constexpr (GENERAL) { fin = in_field && g.rowflag[yin] != 0; fout = out_field && g.rowflag[yout] != 0; }
        if(!fin && !fout) {
            unsigned bi[QB_NDW], bo[QB_NDW];
            load_row(plane, yin, bi);
            load_row(plane, out_field ? yout : -QF_PADY, bo);
            qb_windows<HW>([&](const int i) { return qb_tab(qb_byte_x8(bi, i)) - qb_tab(qb_byte_x8(bo, i)); }, [&](const int j, const double S) { V[j] += S; });
            if constexpr (GENERAL) {
                if(in_field != out_field) {
                    const int sign = in_field ? 1 : -1;
#pragma unroll
                    for(int j = 0; j < QB_SEG; j++) Vc[j] += sign * nxw(j);
                }
            }
        }
        else if constexpr (GENERAL) {
            if(fin) flagged_row(yin, 1); else plain_row(yin, 1);
            if(out_field) { if(fout) flagged_row(yout, -1); else plain_row(yout, -1); }
        }
    };
    // run-in step: a row enters the window and nothing leaves (half the table look-ups and no subtraction)
    auto step_in = [&](const int yin) __attribute__((always_inline)) {
        bool fin = false;
        if constexpr (GENERAL) fin = yin >= 0 && yin < Y && g.rowflag[yin] != 0;
        if(!fin) plain_row(yin, 1);
        else if constexpr (GENERAL) flagged_row(yin, 1);
    };
    const double rreps = 1.0 / (double)reps;
    const bool xfull = x0 - HW >= 0 && x0 + QB_SEG - 1 + HW <= X - 1;   // every window of the segment lies inside the field's columns
    // run-in (y < ybeg): rows ybeg - HW .. ybeg + HW - 1 enter and nothing leaves; from ybeg on every step completes a row of output
    for(int y = ybeg - 2 * HW; y < yend; y++) {
#if defined(QB_ABL) && (QB_ABL & 4)      // timing experiment: no window sums (no loads, no table, no prefix sums)
        if(active) {
#pragma unroll
            for(int j = 0; j < QB_SEG; j++) V[j] += (double)(y + j);
        }
#else
        if(active) {
#ifdef QB_NO_RUNIN_SHORTCUT
            step(y + HW, y > ybeg ? y - HW - 1 : -QF_PADY - 1000000);
#else
            if(y > ybeg) step(y + HW, y - HW - 1); else step_in(y + HW);
#endif
        }
#endif
        if(y < ybeg) continue;
        float* const yab = ya + ((y - ybeg) & 1) * T * QB_SW;
        if(active) {
            // mean (:473), E-fold float sum / E (:494-499: `sum += value` E times), clamp (:500-506)
            float o[QB_SEG], acc[QB_SEG];
            const int nyw = min(y + HW, Y - 1) - max(y - HW, 0) + 1;
            if(!GENERAL && xfull) {
                // every window of the segment holds (2 HW + 1) * nyw cells: ONE division per step, then per cell
                // q = V * r, corrected by the exact residual (q' = q + (V - q wc) r).  V / wc is a multiple of 2^-33 / wc away from
                // every float32 rounding boundary or exactly on it, and q' is within an ulp(double) of it and exact when V / wc is
                // representable: (float)q' is the float the double division rounds to.
                const double wcd = (double)((2 * HW + 1) * nyw), r = 1.0 / wcd;
#pragma unroll
                for(int j = 0; j < QB_SEG; j++) {
                    const double q0 = V[j] * r;
                    o[j] = (float)__builtin_fma(__builtin_fma(-q0, wcd, V[j]), r, q0);
                    acc[j] = 0.0f;
                }
            }
            else {
#pragma unroll
                for(int j = 0; j < QB_SEG; j++) {
                    // (no flagged row: the window holds all the field cells it covers)
                    const int wc = GENERAL ? Vc[j] : nxw(j) * nyw;
                    o[j] = (float)(V[j] / (double)max(wc, 1));
                    if(wc <= 0) o[j] = NAN;
                    acc[j] = 0.0f;
                }
            }
#if QB_LAZY
#pragma unroll
            for(int j = 0; j < QB_SEG; j++) yab[p * QB_SW + s * QB_SEG + j] = qb_nv(o[j]) ? o[j] : NAN;   // the means: F where interpolate() looks, qb_interp_lazy
            (void)acc;
        }
#else
            // (a wave whose means are all 0 or 1 -- a threshold below or above everything in sight -- has nothing to add up:
            //  E * 0 and E * 1 are exact, and so is every partial sum)
            bool plain01 = true;
#pragma unroll
            for(int j = 0; j < QB_SEG; j++) plain01 = plain01 && (o[j] == 0.0f || o[j] == 1.0f);
            const bool skip = __all(plain01);
            if(skip) {
#pragma unroll
                for(int j = 0; j < QB_SEG; j++) acc[j] = o[j] * (float)reps;
            }
#if defined(QB_ABL) && (QB_ABL & 1)      // timing experiment: no E-fold sum
            else if(true) {
#pragma unroll
                for(int j = 0; j < QB_SEG; j++) acc[j] = o[j] * (float)reps;
            }
#endif
            else if(reps > 1) {
                // (round 5: the same loop on v_pk_add_f32, two additions per instruction, measured: 970 against 968 us -- a packed instruction issues
                //  like two, tools/ubench/valu_rate.hip; taken out again)
#ifndef QB_UNR
#define QB_UNR 2
#endif
#pragma unroll QB_UNR
                for(int e = 0; e < reps; e++) {
#pragma unroll
                    for(int j = 0; j < QB_SEG; j++) acc[j] += o[j];
                }
            }
#pragma unroll
            for(int j = 0; j < QB_SEG; j++) {
                // acc / E through the double reciprocal: exact for E < 2^8 (acc / E can neither hit nor come within 2^-41 of a float32
                // rounding boundary, the double product is within 2^-52 of it)
                float yv = reps > 1 ? (float)((double)acc[j] * rreps) : o[j];
                yv = yv > 1 ? 1.0f : (yv < 0 ? 0.0f : yv);
                yab[p * QB_SW + s * QB_SEG + j] = qb_nv(o[j]) ? yv : NAN;
            }
        }
#endif
        qb_lds_barrier();
        for(int c = tid; c < QB_SW; c += blockDim.x) {
            const int x = xs + c;
            if(x < X) {
                const long cell = (long)y * X + x;
#if defined(QB_ABL) && (QB_ABL & 2)      // timing experiment: no interpolation
                out[cell] = yab[c] + yab[c + (T - 1) * QB_SW];
#else
#if QB_LAZY
                out[cell] = qb_interp_lazy(yab + c, QB_SW, T, sthr, qfield ? q[cell] : q[0], reps, rreps);
#else
                out[cell] = qb_interp(yab + c, QB_SW, T, sthr, qfield ? q[cell] : q[0]);
#endif
#endif
            }
        }
    }
}

template <int HW>
void launch_hw(const unsigned char* cnt8, const QfGeom& g, int reps, int T, const float* d_thr, const float* d_q, int qfield, float* d_out, hipStream_t st) {
    const int threads = 64 * ((T * (QB_SW / QB_SEG) + 63) / 64);
    const size_t lds = 256 * sizeof(double) + 16 * sizeof(float) + (size_t)2 * T * QB_SW * sizeof(float);
    // rows per workgroup: about QB_SH, chosen so that the workgroups fill the 256 CUs a whole number of times
    const int strips = (g.X + QB_SW - 1) / QB_SW;
    int segs = std::max(1, (g.Y + QB_SH - 1) / QB_SH);
    // (round 5: a field with few rows -- a rank's row tile of the 8-GPU run: 530 rows -- gave 144 workgroups for 256 CUs, each marching 59 + 2 HW + 1
    //  rows; shorter segments, down to 8 rows, until there are about three workgroups per CU (GPP_QB_FILL; tools/qb_fill_sweep.sh: 1 -> 0.544,
    //  2 -> 0.467, 3 -> 0.457, 4 -> 0.490, 6 -> 0.546 ms for quantile_fast on that tile -- every extra segment costs its 2 HW + 1 run-in rows))
    { const char* e = gpp::path_env("GPP_QB_FILL");
      const int fill = e && atoi(e) > 0 ? atoi(e) : 3;
      const int want = std::max(1, (fill * 256) / std::max(1, strips));
      if(segs < want) segs = std::max(segs, std::min(want, std::max(1, g.Y / 8))); }
    if(strips * segs > 256) { const int per = 256 / std::__gcd(256, strips); segs = (segs + per - 1) / per * per; }
    const int SH = std::max(1, (g.Y + segs - 1) / segs);
    const dim3 grid(strips, (g.Y + SH - 1) / SH);
    hipLaunchKernelGGL((k_qf_box<HW, false>), grid, dim3(threads), lds, st, cnt8, g, reps, T, d_thr, d_q, qfield, d_out, SH);
    hipLaunchKernelGGL((k_qf_box<HW, true>), grid, dim3(threads), lds, st, cnt8, g, reps, T, d_thr, d_q, qfield, d_out, SH);
    GPP_HIP(hipGetLastError());
}
}   // namespace

void qf_box_launch(const unsigned char* cnt8, const QfGeom& g, int reps, int hw, int T, const float* d_thr, const float* d_q, int qfield, float* d_out, hipStream_t st) {
    if(!st) st = stream();
    switch(hw) {
#define QB_CASE(n) case n: launch_hw<n>(cnt8, g, reps, T, d_thr, d_q, qfield, d_out, st); break;
        QB_CASE(0) QB_CASE(1) QB_CASE(2) QB_CASE(3) QB_CASE(4) QB_CASE(5) QB_CASE(6) QB_CASE(7) QB_CASE(8)
        QB_CASE(9) QB_CASE(10) QB_CASE(11) QB_CASE(12) QB_CASE(13) QB_CASE(14) QB_CASE(15) QB_CASE(16)
#undef QB_CASE
        default: runtime("Internal error. quantile_fast: halfwidth of the fused box pass");
    }
}
