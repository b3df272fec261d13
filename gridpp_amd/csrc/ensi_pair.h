// k_ensi_pair: ensemble OI with the n x n eigenproblem held in registers (included by ensi.hip).
//
// Same mathematics as k_ensi (B = A A^T = U S U^T, n <= 32 selected observations; oi_ensi.cpp:379-553), different machine
// mapping.  k_ensi kept B and U in LDS and spent most of its time waiting for LDS (every Jacobi step = two barriers, four
// dependent LDS round trips, ~100 LDS instructions).  Here
//   * TWO grid cells with the SAME selection are solved side by side, one per half wave: lane 32 h + i holds row i of B and
//     row i of U of cell h in 128 VGPRs; Y, the Gram matrix Y Y^T and the staging areas are shared by the pair;
//   * the Jacobi sweeps use the odd-even ordering with the swap folded into the rotation (phases alternate between the
//     pairs (2k, 2k+1) and (2k+1, 2k+2); after 32 phases every pair of indices has met exactly once): the partner of a
//     row is always the neighbouring lane (DPP quad_perm / wave shifts, no LDS), the partner of a column always the
//     neighbouring register (static indices);  the only LDS traffic of a step is the broadcast of the 16 (c, s) pairs;
//   * the n x n x n products (warm start U^T B U, U diag U^T, (M_W D) Y) run on the matrix cores through two 8.5 KB
//     staging areas that are reused for Y and for the transposed member-update operands;
//   * any number of valid ensemble members (chunks of 64 lanes); the member update keeps the reference's float
//     accumulation over k (oi_ensi.cpp:505-511) term by term.
#pragma once

// -DGPP_ENSI_PROFILE: cycles per phase of k_ensi_pair (s_memtime, summed over the waves) in counters[40 + phase]
#ifdef GPP_ENSI_PROFILE
#define EPROF(k) { const unsigned long long t_ = __builtin_readcyclecounter(); prof[k] += t_ - tprev; tprev = t_; }
#else
#define EPROF(k)
#endif

#define PP 34             // pitch (doubles) of the staged 32 x 32 matrices: 16-byte aligned rows, 2-way bank conflicts at most
#define YP 65             // pitch (floats) of the Y tile

typedef int v2i __attribute__((ext_vector_type(2)));

#ifndef GPP_ENSI_NSQ
#define GPP_ENSI_NSQ 4    // steps of the square-root iteration in k_ensi_members (NSQ - 1 products)
#endif
#ifndef GPP_ENSI_NNEU
#define GPP_ENSI_NNEU 4   // products of the Neumann series of the inverse in k_ensi_members (terms T1 .. T(NNEU))
#endif
#ifndef GPP_ENSI_JCHUNK
#define GPP_ENSI_JCHUNK 8   // double phases between two tests of the off-diagonal norm in k_ensi_pair (8 = half a sweep; config 5 with the 0.040 c threshold:
                            // 1: 271 ms, 2: 244, 4: 224.5, 6: 221.2, 8: 219.9, 16: 220.9 -- the test is a chain of reductions the single wave of a SIMD waits for)
#endif
// ---- cross-lane moves of doubles on the VALU (DPP) -------------------------------------------------------------------------
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_d(const double old, const double v) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, BANK, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, BANK, false);
    return __hiloint2double(hi, lo);
}
// value of the partner lane (byte address paddr = 4 * partner lane) through the LDS crossbar: keeps the VALU free, and the odd
// phase has no DPP pattern (bank_mask selects groups of four lanes, not a lane parity)
// even phase: the partner is the other lane of the pair (2k, 2k+1) -- a quad permute on the VALU, which has issue slots to spare,
// instead of two more trips through the LDS crossbar, which has not (C5: 515 -> 491 ms; the odd phase done the same way --
// both wave shifts and a select -- costs three VALU instructions per dword and loses: 509 ms)
__device__ __forceinline__ double partner_quad(const double v) { return dpp_d<0xB1, 0xf>(v, v); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ double partner_of(const double v, const int paddr) {
    const int lo = __builtin_amdgcn_ds_bpermute(paddr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(paddr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// sum over the 32 lanes of each half wave, returned to every lane of the half
__device__ __forceinline__ double half_sum_d(double v, const int lane) {
    v += dpp_d<0x111, 0xf>(0.0, v);   // row_shr:1  (old = 0: lanes without a source add 0)
    v += dpp_d<0x112, 0xf>(0.0, v);
    v += dpp_d<0x114, 0xf>(0.0, v);
    v += dpp_d<0x118, 0xf>(0.0, v);
    // lane 15 of every row holds the row sum; rows 1 and 3 add the sum of the row before (row_bcast:15, rows 1 and 3 only)
    {
        int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x142, 0xa, 0xf, false);
        int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x142, 0xa, 0xf, false);
        v += __hiloint2double(hi, lo);
    }
    const double s0 = readlane_d(v, 31), s1 = readlane_d(v, 63);
    return lane < 32 ? s0 : s1;
}

// v(lane) + v(lane ^ 32) in every lane: the upper half of one copy swapped with the lower half of another (v_permlane32_swap, gfx950)
__device__ __forceinline__ double both_halves_sum_d(const double v, const int lane) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto sl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto sh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const unsigned olo = lane < 32 ? sl[1] : sl[0], ohi = lane < 32 ? sh[1] : sh[0];
    return v + __hiloint2double((int)ohi, (int)olo);
}

// a value the optimiser cannot trace back to its load: keeps `select(load, load)` from becoming `load(select(address))`, which
// would turn the register-resident rows into a dynamically indexed stack array
__device__ __forceinline__ double opq(double x) { asm("" : "+v"(x)); return x; }

// 16-way selection x[sel] by four lane-dependent bits (b0 = least significant): a tree of 15 v_cndmask pairs.  The leaves are
// made opaque first, all of them unconditionally (an asm inside a conditional arm would turn the tree into branches).
__device__ __forceinline__ double mux16(const bool b0, const bool b1, const bool b2, const bool b3, double x0, double x1, double x2, double x3,
                                        double x4, double x5, double x6, double x7, double x8, double x9, double x10, double x11, double x12,
                                        double x13, double x14, double x15) {
    x0 = opq(x0); x1 = opq(x1); x2 = opq(x2); x3 = opq(x3); x4 = opq(x4); x5 = opq(x5); x6 = opq(x6); x7 = opq(x7);
    x8 = opq(x8); x9 = opq(x9); x10 = opq(x10); x11 = opq(x11); x12 = opq(x12); x13 = opq(x13); x14 = opq(x14); x15 = opq(x15);
    const double y0 = b0 ? x1 : x0, y1 = b0 ? x3 : x2, y2 = b0 ? x5 : x4, y3 = b0 ? x7 : x6, y4 = b0 ? x9 : x8, y5 = b0 ? x11 : x10,
                 y6 = b0 ? x13 : x12, y7 = b0 ? x15 : x14;
    const double z0 = b1 ? y1 : y0, z1 = b1 ? y3 : y2, z2 = b1 ? y5 : y4, z3 = b1 ? y7 : y6;
    const double w0 = b2 ? z1 : z0, w1 = b2 ? z3 : z2;
    return b3 ? w1 : w0;
}
#define MUX16(...) mux16(m1, m2, m3, m4, __VA_ARGS__)      /* x[(i >> 1) & 15] */
#define MUXD(...) mux16(e0, e1, e2, e3, __VA_ARGS__)       /* x[i & 15] */

__device__ __forceinline__ double d_rcp_n1(const double a) { double r = __builtin_amdgcn_rcp(a); return r * (2.0 - a * r); }
__device__ __forceinline__ double d_rsq_n1(const double a) { double r = __builtin_amdgcn_rsq(a); return r * (1.5 - 0.5 * a * r * r); }

// Jacobi rotation that annihilates apq (dp, dq: the two diagonal entries): t = tan, cs = cos, ci = 1 / cos, tap = t apq.  The angle
// only has to be accurate enough for the quadratic convergence (one Newton step on the reciprocals).
__device__ __forceinline__ void jacobi_rotation(const double apq, const double dp, const double dq, double& t, double& cs, double& ci, double& tap) {
    const double theta = (dq - dp) * 0.5 * d_rcp_n1(apq);
    const double h2 = theta * theta + 1.0;
    const double t_ = (theta >= 0 ? 1.0 : -1.0) * d_rcp_n1(fabs(theta) + h2 * d_rsq_n1(h2));
    const double g2 = t_ * t_ + 1.0;
    const double c_ = d_rsq_nr(g2);
    const bool ok = fabs(theta) < 1e150;      // false for apq == 0 (theta inf / NaN) and for a negligible apq
    t = ok ? t_ : 0.0;
    cs = ok ? c_ : 1.0;
    ci = ok ? g2 * c_ : 1.0;
    tap = ok ? t_ * apq : 0.0;
}

// full 32 x 32 x 32 product on the matrix cores, operands through accessors (no bounds: the staged matrices are zero padded).
// SYM: the product is known to be symmetric (a power of a symmetric matrix, U M U^T, ...): the tile below the diagonal is not computed --
// a quarter of the matrix-core time, and on gfx950 v_mfma_f64_16x16x4 occupies the FP64 pipe of its SIMD for 66 cycles, vector FP64
// instructions queue behind it (tools/ubench/mfma_f64_rate.hip) -- and the stores below write the tile above the diagonal twice.
template <bool SYM = false, class FA, class FB>
__device__ __forceinline__ Acc32 mfma_32_full(const int lane, FA a_at, FB b_at) {
    Acc32 c;
    c.t[0][0] = c.t[0][1] = c.t[1][0] = c.t[1][1] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for(int kk = 0; kk < 32; kk += 4) {
        const int k = kk + kq;
        const double a0 = a_at(r, k), b0 = b_at(k, r), b1 = b_at(k, r + 16);
        c.t[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c.t[0][0], 0, 0, 0);
        c.t[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c.t[0][1], 0, 0, 0);
        const double a1 = a_at(r + 16, k);
        if constexpr(!SYM) c.t[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c.t[1][0], 0, 0, 0);
        c.t[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c.t[1][1], 0, 0, 0);
    }
    return c;
}
template <bool SYM = false>
__device__ __forceinline__ void acc32_store_full(const Acc32& c, const int lane, double* M) {
#pragma unroll
    for(int ti = 0; ti < 2; ++ti)
#pragma unroll
        for(int tj = 0; tj < 2; ++tj) {
            if(SYM && ti == 1 && tj == 0) continue;
#pragma unroll
            for(int r = 0; r < 4; ++r) {
                M[(16 * ti + (lane >> 4) + 4 * r) * PP + 16 * tj + (lane & 15)] = c.t[ti][tj][r];
                if(SYM && ti == 0 && tj == 1) M[(16 + (lane & 15)) * PP + (lane >> 4) + 4 * r] = c.t[0][1][r];   // its mirror image
            }
        }
}

// The same product on the FP32 matrix path (v_mfma_f32_16x16x4_f32: 32 cycles of the matrix pipe instead of 66), for the higher-order
// terms of the perturbation series in k_ensi_members: a term of relative size eps^2 <= 4e-4 carries the 6e-8 of float32 as 2e-11 of the
// result.  Operands are rounded on the way in; the result tiles come back as doubles.  (C/D map of the f32 form: row = 4 (lane >> 4) + reg.)
typedef float v4f __attribute__((ext_vector_type(4)));
struct Acc32f { v4f t[2][2]; };
template <bool SYM = false, class FA, class FB>
__device__ __forceinline__ Acc32f mfma_32_f32(const int lane, FA a_at, FB b_at) {
    Acc32f c;
    c.t[0][0] = c.t[0][1] = c.t[1][0] = c.t[1][1] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for(int kk = 0; kk < 32; kk += 4) {
        const int k = kk + kq;
        const float a0 = (float)a_at(r, k), b0 = (float)b_at(k, r), b1 = (float)b_at(k, r + 16);
        c.t[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, c.t[0][0], 0, 0, 0);
        c.t[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, c.t[0][1], 0, 0, 0);
        const float a1 = (float)a_at(r + 16, k);
        if constexpr(!SYM) c.t[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, c.t[1][0], 0, 0, 0);
        c.t[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, c.t[1][1], 0, 0, 0);
    }
    return c;
}
template <bool SYM = false>
__device__ __forceinline__ void acc32f_store(const Acc32f& c, const int lane, double* M) {
#pragma unroll
    for(int ti = 0; ti < 2; ++ti)
#pragma unroll
        for(int tj = 0; tj < 2; ++tj) {
            if(SYM && ti == 1 && tj == 0) continue;
#pragma unroll
            for(int r = 0; r < 4; ++r) {
                const double v = (double)c.t[ti][tj][r];
                M[(16 * ti + 4 * (lane >> 4) + r) * PP + 16 * tj + (lane & 15)] = v;
                if(SYM && ti == 0 && tj == 1) M[(16 + (lane & 15)) * PP + 4 * (lane >> 4) + r] = v;   // its mirror image
            }
        }
}

// one phase of the odd-even Jacobi ordering on the register-resident rows (b: B, u: U, dg: diagonal of B).
// Round 4: FAST rotations.  A rotation applied as (x, y) <- (c x + s y, c y - s x) is two multiplications and two multiply-adds per pair
// of entries; written (x, y) <- c (x + t y), c (y - t x) the factor c can stay outside, in one scale per row / column: the registers hold
// Bs and Us with  B = D Bs D,  U = Us D,  D = diag(d)  (lane i carries d_i and 1 / d_i), and a rotation of the pair (p, q) is
//      position p <- (entry q) + kL (entry p),   position q <- (entry p) + kP (entry q),    kL = t d_p / d_q,   kP = -t d_q / d_p
// (positions swapped as before), d_p <- c d_q, d_q <- c d_p: ONE multiply-add per entry, 96 FP64 instructions per phase instead of 192 --
// the kernel runs one wave per SIMD and is bound by its FP64 instruction stream.  The diagonal dg stays in true values; the stopping test
// and the hand-over to the park multiply the scales back in (jacobi_unscale below).
template <bool ODD>
__device__ __forceinline__ void jacobi_phase(double (&b)[32], double (&u)[32], double& dg, double& dsc, double& dinv, const int i, const int h,
                                             const bool m1, const bool m2, const bool m3, const bool m4, double* s_cs) {
    // the off-diagonal entry of this lane's pair: leaders (even rows in the even phase, odd rows in the odd phase) hold it in
    // register i + 1
    const double araw = ODD ? MUX16(b[2], b[4], b[6], b[8], b[10], b[12], b[14], b[16], b[18], b[20], b[22], b[24], b[26], b[28], b[30], 0.0)
                            : MUX16(b[1], b[3], b[5], b[7], b[9], b[11], b[13], b[15], b[17], b[19], b[21], b[23], b[25], b[27], b[29], b[31]);
    const bool idle = ODD && (i == 0 || i == 31);
    const bool leader = ODD ? ((i & 1) != 0 && !idle) : ((i & 1) == 0);
    const int paddr = (ODD ? (idle ? (32 * h + i) : ((i & 1) ? 32 * h + i + 1 : 32 * h + i - 1)) : ((32 * h + i) ^ 1)) << 2;
    // both lanes of a pair compute the same rotation from the same numbers in the same order (the partner receives the leader's
    // off-diagonal entry together with its diagonal entry and its scales)
    const double dpart = ODD ? partner_of(dg, paddr) : partner_quad(dg);
    const double apart = ODD ? partner_of(araw, paddr) : partner_quad(araw);
    const double spart = ODD ? partner_of(dsc, paddr) : partner_quad(dsc);
    const double ipart = ODD ? partner_of(dinv, paddr) : partner_quad(dinv);
    const double dL = leader ? dsc : spart, dP = leader ? spart : dsc, iL = leader ? dinv : ipart, iP = leader ? ipart : dinv;
    double t, cs, ci, tap;
    jacobi_rotation((dL * dP) * (leader ? araw : apart), leader ? dg : dpart, leader ? dpart : dg, t, cs, ci, tap);
    if(idle) { t = 0.0; cs = 1.0; ci = 1.0; tap = 0.0; }
    const double kL = t * (dL * iP), kP = -(t * (dP * iL));
    if(leader) { double2 v; v.x = kL; v.y = kP; *reinterpret_cast<double2*>(&s_cs[(h * 16 + (i >> 1)) * 2]) = v; }
    // rotation + swap: position p receives the rotated row / column q and vice versa, so the pairs of the next phase are neighbours again
    //   row p' = c row p - s row q,  row q' = s row p + c row q;  lane p keeps q' (scale c d_q), lane q keeps p' (scale c d_p)
    const double kown = idle ? 0.0 : (leader ? kL : kP);
    dg = idle ? dg : dpart + (leader ? tap : -tap);      // a_pp' = a_pp - t a_pq, a_qq' = a_qq + t a_pq, swapped
    dsc = cs * spart; dinv = ci * ipart;                 // (an idle lane is its own partner, with c = 1)
    // The partner values travel through the LDS crossbar (~100 cycles): all 32 requests of a half of the row are issued before
    // the first one is consumed, and the second half is in flight while the first is rotated.  The scheduling barriers pin
    // that order (left alone, the scheduler issues one request, waits for it, rotates, issues the next).
    __syncthreads();   // (kL, kP) of all pairs visible (one wave: this waits for the LDS write, nothing else)
    double pa[16], pb[16];
    double2 ca[8], cb[8];
#pragma unroll
    for(int j = 0; j < 16; ++j) pa[j] = ODD ? partner_of(b[j], paddr) : partner_quad(b[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for(int j = 0; j < 16; ++j) pb[j] = ODD ? partner_of(b[16 + j], paddr) : partner_quad(b[16 + j]);
#pragma unroll
    for(int j = 0; j < 16; ++j) b[j] = __builtin_fma(kown, b[j], pa[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for(int k = 0; k < 8; ++k) ca[k] = *reinterpret_cast<const double2*>(&s_cs[(h * 16 + k) * 2]);
#pragma unroll
    for(int j = 0; j < 16; ++j) b[16 + j] = __builtin_fma(kown, b[16 + j], pb[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for(int k = 0; k < 8; ++k) cb[k] = *reinterpret_cast<const double2*>(&s_cs[(h * 16 + 8 + k) * 2]);
    // columns (x, y) = (p, q):  position p <- y + kL x,  position q <- x + kP y
#pragma unroll
    for(int k = 0; k < 16; ++k) {
        if(k == 8) __builtin_amdgcn_sched_barrier(0);
        if(ODD && k == 15) continue;
        const double2 k2 = k < 8 ? ca[k & 7] : cb[k & 7];
        const int p = ODD ? 2 * k + 1 : 2 * k, q = p + 1;
        const double x = b[p], y = b[q];
        b[p] = __builtin_fma(k2.x, x, y);
        b[q] = __builtin_fma(k2.y, y, x);
        const double ux = u[p], uy = u[q];
        u[p] = __builtin_fma(k2.x, ux, uy);
        u[q] = __builtin_fma(k2.y, uy, ux);
    }
    __syncthreads();   // before the next phase overwrites s_cs
}


// original index of valid member k (no load when every member is valid: a dependent load costs a trip through a memory system that
// thousands of waves are streaming their park rows through)
__device__ __forceinline__ int ensi_member(const EnsiArgs& a, const int k) { return a.valid_identity ? k : a.validIdx[k]; }

// cell of lane `l` of tile `tile` (the same mapping in both kernels)
__device__ __forceinline__ int ensi_cell_of(const EnsiArgs& a, const int tile, const int l) {
    if(a.tiled2d) {
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int y = ty * (64 >> a.wshift) + (l >> a.wshift), x = (tx << a.wshift) + (l & ((1 << a.wshift) - 1));
        return (y < a.ny && x < a.nx) ? y * a.nx + x : -1;
    }
    const int c = tile * 64 + l;
    return c < a.C ? c : -1;
}

// ---- pass 1: candidate scan of a tile; the selections are parked in HBM in ascending observation index (a canonical order:
//      equal selections give equal lists) together with their length, the "reference sorted" flag and an order-independent
//      signature:  sel[tile][position][lane], meta[tile][lane], hsigs[tile][lane]
template <bool SPATIAL>
__global__ __launch_bounds__(64) void k_ensi_scan(EnsiArgs a) {
    __shared__ unsigned long long s_keys[EN][64];
    const int lane = threadIdx.x;
    const int tile = blockIdx.x;
    const int cell = ensi_cell_of(a, tile, lane);
    float gx = 0, gy = 0, gz = 0, ge = NAN, gl = NAN;
    if(cell >= 0) { gx = a.gx[cell]; gy = a.gy[cell]; gz = a.gz[cell]; ge = a.gelev[cell]; gl = a.glaf[cell]; }
    const bool active = cell >= 0;   // no background validity test here (oi_ensi.cpp:207-213)
    if(__ballot(active) == 0ull) return;
    bool overflow, truncated;
    DevStructure cst = a.s.st;
    if(SPATIAL && cell >= 0) d_structure_at(cst, cst.cell_idx ? cst.cell_idx[cell] : cell);
    int cnt = scan_tile<EN, true>(a.s, cst, active, gx, gy, gz, ge, gl, s_keys, lane, overflow, truncated);
    if(__ballot(overflow) != 0ull) {   // more usable observations than the 32-row tile holds: those cells go to k_ensi_big
        if(a.big_list) { if(overflow) a.big_list[atomicAdd(a.big_count, 1)] = cell; }
        else if(lane == 0) atomicOr(a.err, 1);
        cnt = overflow ? 0 : cnt;
    }
    unsigned* const sel = a.sel + (size_t)tile * EN * 64;
    unsigned long long hsig = 0;
    unsigned idx[EN];
#pragma unroll
    for(int s = 0; s < EN; ++s) idx[s] = (s < cnt) ? ~(unsigned)(s_keys[s][lane] & 0xffffffffull) : 0xffffffffu;
#pragma unroll 1
    for(int s = 0; s < a.s.K; ++s) {
        const unsigned mine = (s < cnt) ? ~(unsigned)(s_keys[s][lane] & 0xffffffffull) : 0xffffffffu;
        int rank = 0;
#pragma unroll
        for(int t = 0; t < EN; ++t) rank += (idx[t] < mine) ? 1 : 0;
        if(s < cnt) {
            sel[rank * 64 + lane] = mine;
            unsigned long long x = (unsigned long long)mine + 0x9e3779b97f4a7c15ull;
            x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
            hsig += x;
        }
    }
    a.meta[(size_t)tile * 64 + lane] = (unsigned)cnt | (truncated ? 0x100u : 0u);
    a.hsigs[(size_t)tile * 64 + lane] = hsig;
}

// ---- pass 2: the solves -------------------------------------------------------------------------------------------------------
template <bool SPATIAL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_ensi_pair(EnsiArgs a) {
    // 17 KB: staging areas A / B of the products; B doubles as Y tile and as the transposed operands of the member update
    __shared__ __attribute__((aligned(16))) double s_ab[2 * 32 * PP];
    __shared__ __attribute__((aligned(16))) double s_ab2[2 * 32 * PP];   // the second pair of staging areas: the warm-start transforms of both halves side by side
    __shared__ __attribute__((aligned(16))) double s_sD[2][32], s_r[2][32], s_z[2][32];
    __shared__ __attribute__((aligned(16))) double s_cs[64];      // (c, s) of the Jacobi pairs [h][16][2]; later t[32]
    __shared__ __attribute__((aligned(16))) double s_dwa[2][2][32];   // [cell][dw | a = sqrt(c + S)][eigenvalue]
    __shared__ int s_i[128];                                      // perm[32] | obs[32] | yhat[32] (floats) | selection[32]
    double* const sA = s_ab;
    double* const sB = s_ab + 32 * PP;
    float* const sBf = reinterpret_cast<float*>(sB);             // Y tile [32][YP] floats (8320 B <= 8704 B)
    double* const s_t = s_cs + 32;
    int* const s_perm = s_i;
    float* const s_ob = reinterpret_cast<float*>(s_i + 32);
    float* const s_yh = reinterpret_cast<float*>(s_i + 64);
    unsigned* const s_sel = reinterpret_cast<unsigned*>(s_i + 96);
    const int lane = threadIdx.x;
    const int tile = a.tile0 + blockIdx.x;
    const int h = lane >> 5, i = lane & 31;
    const bool m1 = (i & 2) != 0, m2 = (i & 4) != 0, m3 = (i & 8) != 0, m4 = (i & 16) != 0;
    const int nV = a.nV;
    if(nV <= 1) return;   // Pinv is the zero matrix: rcond <= 0 -> raw values everywhere (oi_ensi.cpp:386-390)
    const unsigned meta = a.meta[(size_t)tile * 64 + lane];
    const int cnt = ensi_cell_of(a, tile, lane) >= 0 ? (int)(meta & 0xffu) : 0;
    const unsigned long long hsig = a.hsigs[(size_t)tile * 64 + lane];
    const unsigned long long trunc_mask = __ballot((meta & 0x100u) != 0u);
    const unsigned* const sel = a.sel + (size_t)tile * EN * 64;
    double* const gram = a.gram + (size_t)tile * 2 * EN * EN;   // two matrices: one per group in flight

    const double c = (double)((float)(nV - 1));   // diag = 1/delta*(nValidEns-1), float (oi_ensi.cpp:383)
    const double sqc = sqrt(c);
    // The eigenvectors of each half's last cell stay in registers between two pairs (the warm start).  What the ensemble side
    // (k_ensi_members, one wave per cell) needs of a cell -- M' = sD (U Mmid U^T) sD, sD, z, rho -- is parked in HBM.
    double u[32];
#pragma unroll
    for(int j = 0; j < 32; ++j) u[j] = (j == i) ? 1.0 : 0.0;
#ifdef GPP_ENSI_PROFILE
    unsigned long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    unsigned prev_orig = 0xffffffffu; int prev_n = -1;
    unsigned long long todo = __ballot(cnt > 0);
    int ndone = 0, nsweeps = 0;
    while(todo) {
        // ---- next group: the cells whose selection equals that of the lowest cell still to do; and the group after it -----------------
        // Round 4: TWO groups at a time, one per half wave, while there are two (the last odd group is split over the halves as before).
        // A group's first cell starts cold -- sweeps from scratch, ~200 k cycles against ~25 k for a warm cell --, and both halves run every
        // sweep together: with one group split over both halves every group cost one cold step; with two groups side by side two groups do.
        const int l0 = __builtin_ctzll(todo);
        const int nA = __builtin_amdgcn_readlane(cnt, l0);
        const unsigned long long hA = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(hsig >> 32), l0) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)hsig, l0);
        const unsigned long long grpA = __ballot(cnt == nA && hsig == hA) & todo;
        const unsigned long long rest = todo & ~grpA;
#ifdef GPP_ENSI_ONE_GROUP
        bool two = false;
#else
        bool two = rest != 0ull;
#endif
        int nB = nA;
        unsigned long long grpB = 0ull;
        const int cA = __builtin_popcountll(grpA);
        if(two) {   // the partner: of the next groups (up to six) the one closest in size -- a half whose group is exhausted idles through the other's steps
            unsigned long long r = rest;
            int best = 1 << 30;
            for(int it = 0; it < 6 && r != 0ull; ++it) {
                const int l1 = __builtin_ctzll(r);
                const int n1 = __builtin_amdgcn_readlane(cnt, l1);
                const unsigned long long h1 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(hsig >> 32), l1) << 32) |
                                              (unsigned)__builtin_amdgcn_readlane((int)hsig, l1);
                const unsigned long long g1 = __ballot(cnt == n1 && hsig == h1) & r;
                const int d = abs(__builtin_popcountll(g1) - cA);
                if(d < best) { best = d; grpB = g1; nB = n1; }
                r &= ~g1;
            }
            // Side by side costs max(cA, cB) steps and one cold start, one after the other (each split over the halves) (cA + cB) / 2 steps and two: worth it
            // while half the difference in size is below what a cold start costs in warm steps -- ~8 with the sweeps stopped early, ~2.5 with
            // the sweeps run to convergence (every step sweeps then)
            if(best >= (a.jtol2 > 0.0 ? 14 : 3)) { two = false; grpB = 0ull; nB = nA; }
        }
        const int cB = __builtin_popcountll(grpB);
        const int nsteps = two ? max(cA, cB) : (cA + 1) >> 1;
        const unsigned long long below = (1ull << lane) - 1ull;
        const int rkA = __builtin_popcountll(grpA & below), rkB = __builtin_popcountll(grpB & below);   // this lane's rank inside its group
        const bool inA = ((grpA >> lane) & 1ull) != 0ull, inB = ((grpB >> lane) & 1ull) != 0ull;
        const int n = max(nA, nB);                    // bound of the wave-uniform loops
        const int nh = (two && h) ? nB : nA;          // observations of this half's cells
        double* const gh = gram + ((two && h) ? EN * EN : 0);   // this half's Gram matrix
        const unsigned long long hm = h ? 0xffffffff00000000ull : 0x00000000ffffffffull;
        unsigned long long done = 0ull;
        // consecutive cells of a half are neighbours, so the eigenvectors of one cell are an excellent starting basis for the next (warm start below)
#pragma unroll 1
        for(int jp = 0; jp < nsteps; ++jp) {
            int la, lb;
            bool st0 = true, st1 = true;     // the halves' results are stored (a half whose group is exhausted repeats its last cell)
            unsigned orig_i;
            if(two) {
                la = __builtin_ctzll(__ballot(inA && rkA == min(jp, cA - 1)));
                lb = __builtin_ctzll(__ballot(inB && rkB == min(jp, cB - 1)));
                st0 = jp < cA; st1 = jp < cB;
                orig_i = (i < nh) ? sel[i * 64 + (h ? lb : la)] : 0xffffffffu;   // (ascending observation index)
            }
            else {
                la = __builtin_ctzll(__ballot(inA && rkA == jp));
                const unsigned long long mb = __ballot(inA && rkA == jp + nsteps);
                lb = mb ? __builtin_ctzll(mb) : la;
                // both cells must have the very same list (they share the Gram matrix)
                orig_i = (i < nA) ? sel[i * 64 + la] : 0xffffffffu;
                if(lb != la) {
                    const unsigned ob = (i < nA) ? sel[i * 64 + lb] : 0xffffffffu;
                    if(__ballot(orig_i != ob) != 0ull) lb = la;   // equal signature, different lists (never seen): that cell waits for its own group
                }
                st1 = lb != la;              // (else half 1 repeats cell la, its result is not stored)
            }
            const bool store = h ? st1 : st0;
            done |= (st0 ? (1ull << la) : 0ull) | (st1 ? (1ull << lb) : 0ull);
            const bool same = nh == prev_n && (__ballot(orig_i != prev_orig) & hm) == 0ull;   // u still holds the eigenvectors of this half's selection
            const int old_n = prev_n;
            const unsigned old_orig = prev_orig;
            prev_n = nh; prev_orig = orig_i;
            // row i of this half's Gram matrix, asked for now (used behind the per-observation loads below; a new selection reads it again once it is rebuilt)
            double b[32];
#pragma unroll
            for(int j = 0; j < 32; j += 2) { const double2 g2 = *reinterpret_cast<const double2*>(&gh[i * EN + j]); b[j] = g2.x; b[j + 1] = g2.y; }
            const int cell_c = ensi_cell_of(a, tile, h ? lb : la);
            const float cx = a.gx[cell_c], cy = a.gy[cell_c], cz = a.gz[cell_c], ce = a.gelev[cell_c], cl = a.glaf[cell_c];
            ndone += (st0 ? 1 : 0) + (st1 ? 1 : 0);
            // ---- per-observation quantities (lane i < n of each half) ----------------------------------------------------------------
            float4 o0 = make_float4(0, 0, 0, NAN), o1 = make_float4(NAN, 0, 0, 1);
            if(i < nh) { o0 = a.ogeo[orig_i]; o1 = a.oaux[orig_i]; }
            DevStructure lst = a.s.st;
            if(SPATIAL) d_structure_at(lst, lst.cell_idx ? lst.cell_idx[cell_c] : cell_c);
            const float rho = d_corr(lst, cx, cy, cz, ce, cl, o0.x, o0.y, o0.z, o0.w, o1.x, true);   // :227
            const float sig2 = o1.w * o1.w;                                    // float product (:300)
            const double D = (double)rho / (double)sig2;                       // Rinv(i,i)
            const double sD = (i < nh) ? sqrt(D) : 0.0;
            const double dobs = (double)o1.y - (double)o1.z;                   // lObs - lYhat (:437)
            __syncthreads();
            s_sD[h][i] = sD; s_r[h][i] = (i < nh) ? sD * dobs : 0.0;
            __syncthreads();
            EPROF(0)   // pair setup: selection, rho
            // ---- new selection: Gram matrix Y Y^T on the matrix cores (all members, chunks of 64), parked in HBM -----------------------
            const unsigned long long nsm = __ballot(!same);
            for(int w = 0; w < 2; ++w) {   // (two groups: each half's own matrix; one group: the shared one, from half 0's list)
                const bool need = two ? ((nsm >> (32 * w)) & 1ull) != 0ull : (w == 0 && nsm != 0ull);
                if(!need) continue;
                const int nw = w ? nB : nA;
                double* const gw = gram + w * EN * EN;
                __syncthreads();
                if(h == w) s_sel[i] = orig_i;
                __syncthreads();
                Acc32 g;
                g.t[0][0] = g.t[0][1] = g.t[1][0] = g.t[1][1] = (v4d){0.0, 0.0, 0.0, 0.0};
                for(int m0 = 0; m0 < nV; m0 += 64) {
                    __syncthreads();
                    {   // unconditional loads (clamped address, value deselected afterwards): all 32 in flight together
                        float yv[EN];
#pragma unroll
                        for(int r = 0; r < EN; ++r) {
                            const bool on = r < nw && m0 + lane < nV;
                            yv[r] = a.gY[on ? (long)s_sel[r] * nV + m0 + lane : 0];
                            yv[r] = on ? yv[r] : 0.0f;
                        }
#pragma unroll
                        for(int r = 0; r < EN; ++r) sBf[r * YP + lane] = yv[r];
                    }
                    __syncthreads();
                    const int r = lane & 15, kq = lane >> 4;
                    const int kend = min(64, nV - m0);
                    for(int kk = 0; kk < kend; kk += 4) {
                        const double a0 = (double)sBf[r * YP + kk + kq], a1 = (double)sBf[(r + 16) * YP + kk + kq];
                        g.t[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, g.t[0][0], 0, 0, 0);
                        g.t[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a1, g.t[0][1], 0, 0, 0);
                        g.t[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a0, g.t[1][0], 0, 0, 0);
                        g.t[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, g.t[1][1], 0, 0, 0);
                    }
                }
#pragma unroll
                for(int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for(int tj = 0; tj < 2; ++tj)
#pragma unroll
                        for(int r = 0; r < 4; ++r) gw[(16 * ti + (lane >> 4) + 4 * r) * EN + 16 * tj + (lane & 15)] = g.t[ti][tj][r];
                __threadfence();
                __syncthreads();
            }
            // A NEW selection of the same length as the last one (the next group of a tile: typically one or two observations exchanged at the
            // edge of the radius, i.e. the rows with the smallest weights): the eigenvectors of the last cell, with the rows of the observations
            // both selections share moved to their new positions and the rows of the departed observations handed to the newcomers, are an
            // orthogonal matrix again and a far better start than the identity -- 16 % of config 5's cells started cold (two per group), and
            // their sweeps from scratch were all of the Jacobi time (histogram of |E| / c before any sweep: 82 % below the threshold, 16 % above 0.48).
            bool remap = false;   // (per half, like `same`)
#ifndef GPP_ENSI_NO_REMAP
            remap = !same && nh == old_n && nh > 1;
            if(__ballot(remap) != 0ull) {
                int src = -1;            // the row of the old eigenvector matrix this row takes (lane i < n: observation orig_i)
                bool kept = false;       // old row i stays in the new selection
                for(int j = 0; j < n; ++j) {
                    const unsigned p0 = (unsigned)__builtin_amdgcn_readlane((int)old_orig, j), p1 = (unsigned)__builtin_amdgcn_readlane((int)old_orig, 32 + j);
                    const unsigned q0 = (unsigned)__builtin_amdgcn_readlane((int)orig_i, j), q1 = (unsigned)__builtin_amdgcn_readlane((int)orig_i, 32 + j);
                    if(j < nh && (h ? p1 : p0) == orig_i) src = j;
                    if(j < nh && (h ? q1 : q0) == old_orig) kept = true;
                }
                const unsigned long long newm = __ballot(remap && i < nh && src < 0) & hm, oldm = __ballot(remap && i < nh && !kept) & hm;   // newcomers / departed rows (equally many)
                if(remap && i < nh && src < 0) {
                    int r = __builtin_popcountll(newm & below);
                    unsigned long long mm = oldm;
                    while(r-- > 0) mm &= mm - 1ull;
                    src = (mm != 0ull ? __builtin_ctzll(mm) : lane) & 31;
                }
                if(!remap || i >= nh) src = i;
                const int paddr2 = (32 * h + src) << 2;
#pragma unroll
                for(int j = 0; j < 32; ++j) u[j] = partner_of(u[j], paddr2);
            }
#endif
            {
                const bool cold = !same && !remap;
#pragma unroll
                for(int j = 0; j < 32; ++j) u[j] = cold ? ((j == i) ? 1.0 : 0.0) : u[j];
            }
            // ---- B = (sD sD^T) o (Y Y^T): row i of each half ------------------------------------------------------------------------------
            // (plain loads: the Gram matrix, like the parked rows below, was written by lanes of this workgroup before a barrier)
            if(nsm != 0ull) {   // (some Gram matrix was rebuilt: read again -- wave-uniform, both halves)
#pragma unroll
                for(int j = 0; j < 32; j += 2) { const double2 g2 = *reinterpret_cast<const double2*>(&gh[i * EN + j]); b[j] = g2.x; b[j + 1] = g2.y; }
            }
#pragma unroll
            for(int j = 0; j < 32; j += 2) { b[j] *= sD * s_sD[h][j]; b[j + 1] *= sD * s_sD[h][j + 1]; }
            EPROF(1)   // Gram (new selections), B build
            // ---- warm start: B <- U^T B U with the eigenvectors of the previous cell of this half (nearly diagonal already) ------------
            // (both halves' products side by side in two pairs of staging areas: one after the other they were 29 % of the kernel, two thirds of it
            //  barriers and staging around 7 k cycles of matrix-core time)
            const unsigned long long wm = __ballot(same || remap);
            if(wm != 0ull) {
                const bool w0 = (wm & 1ull) != 0ull, w1 = ((wm >> 32) & 1ull) != 0ull;   // (a half that starts cold keeps its B)
                double* const tA = h ? s_ab2 : sA;                 // this half's areas
                double* const tB = h ? s_ab2 + 32 * PP : sB;
                double* const a0p = sA, * const b0p = sB, * const a1p = s_ab2, * const b1p = s_ab2 + 32 * PP;
                __syncthreads();
#pragma unroll
                for(int j = 0; j < 32; j += 2) {
                    double2 v; v.x = b[j]; v.y = b[j + 1]; *reinterpret_cast<double2*>(&tA[i * PP + j]) = v;
                    double2 w; w.x = u[j]; w.y = u[j + 1]; *reinterpret_cast<double2*>(&tB[i * PP + j]) = w;
                }
                __syncthreads();
                Acc32 t0, t1;
                if(w0) t0 = mfma_32_full(lane, [&](int r, int k) { return a0p[r * PP + k]; }, [&](int k, int cc) { return b0p[k * PP + cc]; });
                if(w1) t1 = mfma_32_full(lane, [&](int r, int k) { return a1p[r * PP + k]; }, [&](int k, int cc) { return b1p[k * PP + cc]; });
                __syncthreads();
                if(w0) acc32_store_full(t0, lane, a0p);
                if(w1) acc32_store_full(t1, lane, a1p);
                __syncthreads();
                if(w0) t0 = mfma_32_full<true>(lane, [&](int r, int k) { return b0p[k * PP + r]; }, [&](int k, int cc) { return a0p[k * PP + cc]; });
                if(w1) t1 = mfma_32_full<true>(lane, [&](int r, int k) { return b1p[k * PP + r]; }, [&](int k, int cc) { return a1p[k * PP + cc]; });
                __syncthreads();
                if(w0) acc32_store_full<true>(t0, lane, a0p);   // (symmetric by construction: the tile below the diagonal is the mirror image of the one above)
                if(w1) acc32_store_full<true>(t1, lane, a1p);
                __syncthreads();
                if(same || remap) {
#pragma unroll
                    for(int j = 0; j < 32; j += 2) { const double2 v = *reinterpret_cast<const double2*>(&tA[i * PP + j]); b[j] = v.x; b[j + 1] = v.y; }
                }
                __syncthreads();
            }
            EPROF(2)   // warm-start transform
            // ---- Jacobi sweeps, both cells in lockstep --------------------------------------------------------------------------------------
            const bool e0 = (i & 1) != 0, e1 = (i & 2) != 0, e2 = (i & 4) != 0, e3 = (i & 8) != 0;
            auto diag_of_rows = [&]() {
                const double dhi = MUXD(b[16], b[17], b[18], b[19], b[20], b[21], b[22], b[23], b[24], b[25], b[26], b[27], b[28], b[29], b[30], b[31]);
                const double dlo = MUXD(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9], b[10], b[11], b[12], b[13], b[14], b[15]);
                return m4 ? dhi : dlo;
            };
            double dg = diag_of_rows();
            // scales of the fast rotations (jacobi_phase): the registers hold Bs, Us with B = D Bs D, U = Us D while `scaled`
            double dsc = 1.0, dinv = 1.0;
            bool scaled = false;
            double* const s_dsc = &s_dwa[0][0][0] + 32 * h;   // this half's scales, for the stopping test and the hand-over
            auto jacobi_unscale = [&]() {
                __syncthreads();
                s_dsc[i] = dsc;
                __syncthreads();
#pragma unroll
                for(int j = 0; j < 32; j += 2) {
                    const double2 dj = *reinterpret_cast<const double2*>(&s_dsc[j]);
                    b[j] *= dsc * dj.x; b[j + 1] *= dsc * dj.y;
                    u[j] *= dj.x; u[j + 1] *= dj.y;
                }
                dsc = 1.0; dinv = 1.0; scaled = false;
            };
#pragma unroll 1
            for(int sweep = 0; sweep < 480 / GPP_ENSI_JCHUNK && n > 1 && !GPP_DBG(a, 1); ++sweep) {   // (in chunks of GPP_ENSI_JCHUNK of a sweep's 16 double phases)
                double off = 0.0;   // (not sum(b^2) - dg^2: the off-diagonal part is 20 orders below the diagonal when converged)
                if(scaled) {        // true entries: d_i Bs(i, j) d_j
                    __syncthreads();
                    s_dsc[i] = dsc;
                    __syncthreads();
#pragma unroll
                    for(int j = 0; j < 32; j += 2) {
                        const double2 dj = *reinterpret_cast<const double2*>(&s_dsc[j]);
                        const double v0_ = (j == i) ? 0.0 : b[j] * dj.x, v1_ = (j + 1 == i) ? 0.0 : b[j + 1] * dj.y;
                        off = __builtin_fma(v0_, v0_, off); off = __builtin_fma(v1_, v1_, off);
                    }
                    off *= dsc * dsc;
                }
                else {
#pragma unroll
                    for(int j = 0; j < 32; ++j) { const double v = (j == i) ? 0.0 : b[j]; off = __builtin_fma(v, v, off); }
                }
                off = half_sum_d(off, lane);
                const double tr = half_sum_d(fabs(dg), lane);
                // The sweeps stop at an off-diagonal norm |E| of sqrt(jtol2) * c = 0.040 c (c = nV - 1 bounds every eigenvalue of c I + B from
                // below): what is left of E enters the matrix functions in k_ensi_members as a perturbation series without eigenvalue
                // gaps in any denominator (see there; measured against the LAPACK golden vectors the result stays at the float32
                // rounding floor up to there, tools/ensi_tol.py), tested after every half of a sweep.  Most warm-started cells need no sweep at all that way (C5: 0.52
                // sweeps per cell instead of 1.13 with a threshold of 1e-6 of the trace and only the first-order term)
#ifdef GPP_ENSI_ESTATS
                if(sweep == 0 && i == 0 && store && a.counters) {   // (statistics build: |E|_F / c of every cell BEFORE any sweep, printed in the phases line of GPP_ENSI_STATS=1)
                    const double r = sqrt(off) / c;
                    const double lim[9] = {0.04, 0.08, 0.16, 0.32, 0.64, 1.28, 2.56, 5.12, 10.24};
                    int bk = 0;
                    for(int q = 0; q < 9; ++q) bk += r > lim[q] ? 1 : 0;
                    atomicAdd(&a.counters[80 + (blockIdx.x & 1023) * 32 + bk], 1ull);
                }
#endif
                const bool open = off > a.jtol2 * c * c && off > 1e-24 * tr * tr;   // (and never beyond what double precision resolves)
                if(__ballot(open && store) == 0ull) break;
                nsweeps++;
#pragma unroll 1
                for(int st = 0; st < GPP_ENSI_JCHUNK; ++st) {   // every double phase is a complete similarity transform: the test above may come after any
                    jacobi_phase<false>(b, u, dg, dsc, dinv, i, h, m1, m2, m3, m4, s_cs);
                    jacobi_phase<true>(b, u, dg, dsc, dinv, i, h, m1, m2, m3, m4, s_cs);
                }
                scaled = true;
                // the diagonal from the rows again (the running update drifts in the last bits)
                dg = diag_of_rows() * (dsc * dsc);
                // (every rotation takes a factor cos >= 0.7 into the scales: long before they leave the range of a double they go back into the rows)
                if(__ballot(dsc < 1e-60) != 0ull) jacobi_unscale();
            }
            if(scaled) jacobi_unscale();
            EPROF(3)   // Jacobi
            // ---- park for k_ensi_members: the eigenvector rows, the rows of U^T B U (diagonal: the eigenvalue estimates, off-diagonal: what
            //      the sweeps left, which enters the matrix functions there as a perturbation), sD, sD * (obs - yhat), rho ------------
            if(store) {
                double* const park = a.cpark + ((size_t)(tile - a.tile0) * 64 + (h ? lb : la)) * ENSI_PARK_D;
#pragma unroll
                for(int j = 0; j < 32; j += 2) {
                    // (interleaved: pair j / 2 of every row side by side, U^T B U rows in slots 0..31, U rows in slots 32..63, so that
                    //  k_ensi_members reads 1 KB of contiguous memory per load instruction and every lane still receives ITS row)
                    double2 w; w.x = u[j]; w.y = u[j + 1]; *reinterpret_cast<double2*>(&park[(j >> 1) * 128 + (32 + i) * 2]) = w;
                    double2 v; v.x = b[j]; v.y = b[j + 1]; *reinterpret_cast<double2*>(&park[(j >> 1) * 128 + i * 2]) = v;
                }
                park[2048 + i] = sD;
                park[2080 + i] = (i < nh) ? sD * dobs : 0.0;
                park[2112 + i] = (double)rho;
                park[2144 + i] = __longlong_as_double((long long)(((unsigned long long)__float_as_uint(o1.y) << 32) | (unsigned long long)orig_i));
                park[2176 + i] = __longlong_as_double((long long)(unsigned long long)__float_as_uint(o1.z));
            }
            EPROF(4)   // park
        }
        todo &= ~done;
    }
#ifdef GPP_ENSI_PROFILE
    // (spread over 1024 slots: twelve same-address atomics per wave serialise in the L2 and slow every load of the kernel down)
    if(lane == 0 && a.counters) for(int k = 0; k < 12; ++k) atomicAdd(&a.counters[80 + (blockIdx.x & 1023) * 32 + k], prof[k]);
#endif
    if(lane == 0 && a.counters) { atomicAdd(&a.counters[1], (unsigned long long)ndone); atomicAdd(&a.counters[4 + (blockIdx.x & 31)], (unsigned long long)nsweeps); }
}

// ---- pass 3: the ensemble side, one wave per cell (lane = member, chunks of 64): no dependence between cells, so the chip runs as
//      many of these as its registers hold -- inside k_ensi_pair this part ran at one wave per SIMD with nobody to hide its loads
// ONECHUNK: at most 64 valid members (one chunk): the member update is a matrix-core product as well (below); both forms in one kernel
// do not fit its 256 registers.
template <bool ONECHUNK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ensi_members(EnsiArgs a) {
    __shared__ __attribute__((aligned(16))) double s_ab[2 * 32 * PP];
    __shared__ __attribute__((aligned(16))) double s_small[6 * 32];
    double* const s_sD1 = s_small, * const s_z1 = s_small + 32, * const s_t = s_small + 64, * const s_r1 = s_small + 96, * const s_dw = s_small + 128,
          * const s_rt = s_small + 160;
    double* const s_qt = s_t;                                     // member update: the Q columns of up to four tail members, [pair][row][2] (s_t .. s_rt are free then)
    __shared__ int s_i[160];                                      // perm[32] | obs[32] | yhat[32] | rho[32] (floats) | selection[32]
    double* const sA = s_ab;
    double* const sB = s_ab + 32 * PP;
    float* const sBf = reinterpret_cast<float*>(sB);             // Y tile [32][YP] floats
    int* const s_perm = s_i;
    float* const s_ob = reinterpret_cast<float*>(s_i + 32);
    float* const s_yh = reinterpret_cast<float*>(s_i + 64);
    unsigned* const s_sel = reinterpret_cast<unsigned*>(s_i + 96);
    float* const s_rho = reinterpret_cast<float*>(s_i + 128);
    __shared__ float s_v0[64];                                    // the members' values (first chunk) wait here through the spectral part
    const int lane = threadIdx.x;
    const int h = lane >> 5, i = lane & 31;
    const int nV = a.nV, E = a.E;
    if(nV <= 1) return;
#ifdef GPP_ENSI_PROFILE
    unsigned long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    const int tile = a.tile0 + (int)(blockIdx.x >> 6), lcell = (int)(blockIdx.x & 63);
    const int cell_l = ensi_cell_of(a, tile, lcell);
    if(cell_l < 0) return;
    // Everything this cell needs comes in ONE round of independent loads: its length / order flag, its member values and its park
    // (k_ensi_pair left the selection and the observation records there too: the chain meta -> selection -> observation record
    // was three trips through a saturated memory system, 47 % of this kernel's wave cycles)
    const unsigned meta = a.meta[(size_t)tile * 64 + lcell];
#ifdef GPP_ENSI_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    EPROF(11)   // (profile build: one small load alone, waited for: the latency of the memory system under this kernel's load)
#endif
    const float v0_ld = (lane < nV) ? a.bg[(long)cell_l * E + ensi_member(a, lane)] : 0.0f;
    const double* const park = a.cpark + ((size_t)blockIdx.x) * ENSI_PARK_D;
    const unsigned long long pk0 = __double_as_longlong(park[2144 + i]), pk1 = __double_as_longlong(park[2176 + i]);
    const float rho = (float)park[2112 + i];
    const double p_sD = park[2048 + i], p_r1 = park[2080 + i];
    // lanes 32..63: row i of U; lanes 0..31: row i of U^T B U (e[i]: eigenvalue estimate d_i, the rest: the off-diagonal part E)
    double e[32];
#pragma unroll
    for(int j = 0; j < 32; j += 2) {
        if(GPP_DBG(a, 16)) { e[j] = (j == i) ? 1.0 : 0.0; e[j + 1] = (j + 1 == i) ? 1.0 : 0.0; continue; }   // (timing experiment: no park rows)
        const double2 w = *reinterpret_cast<const double2*>(&park[(j >> 1) * 128 + lane * 2]); e[j] = w.x; e[j + 1] = w.y;
    }
    const int n = (int)(meta & 0xffu);
    if(n == 0) return;   // no observation in range (the output already holds the background) or a cell of k_ensi_big
    const unsigned orig_i = (i < n) ? (unsigned)pk0 : 0xffffffffu;
    float4 o1 = make_float4(NAN, 0, 0, 1);
    if(i < n) { o1.y = __uint_as_float((unsigned)(pk0 >> 32)); o1.z = __uint_as_float((unsigned)pk1); }
    const double c = (double)((float)(nV - 1));   // diag = 1/delta*(nValidEns-1), float (oi_ensi.cpp:383)
    const double sqc = sqrt(c);
#ifdef GPP_ENSI_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    EPROF(9)    // (profile build: the one round of loads, waited for)
#endif
    // spectral functions (lane i < 32: eigenvalue i)
    double ei = 0.0;
#pragma unroll
    for(int j = 0; j < 32; ++j) ei = (j == i) ? e[j] : ei;
    const double S = ei < 0.0 ? 0.0 : ei;
    const double rt = sqrt(c + S);                    // a_i
    const double dwv = -1.0 / (rt * (rt + sqc));      // W_sym = I + A^T g(B) A,  g(S) = -1 / (a (a + sqrt(c))),  a = sqrt(c + S)
    const double inv = 1.0 / (c + S);
    // (what the anti-extrapolation tables need of the observations waits in LDS, not in registers, through the series below)
    s_v0[lane] = v0_ld;
    if(h == 0) { s_sel[i] = orig_i; s_sD1[i] = p_sD; s_r1[i] = p_r1; s_dw[i] = dwv; s_rt[i] = rt; s_ob[i] = o1.y; s_yh[i] = o1.z; s_rho[i] = rho; }
    __syncthreads();
    EPROF(0)   // park loads, spectral scalars
#ifdef GPP_ENSI_ESTATS
    {   // (statistics build: histogram of |E|_F / c in octaves below 0.02, printed by GPP_ENSI_STATS=1 in the members-phases line)
        double off = 0.0;
#pragma unroll
        for(int j = 0; j < 32; ++j) { const double v = (j == i || h == 1) ? 0.0 : e[j]; off = __builtin_fma(v, v, off); }
        off = half_sum_d(off, lane);
        const double r = sqrt((double)__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)off)))) / c;
        int bk = 0;
        for(double lim = 0.01; bk < 11 && r <= lim; lim *= 0.5) bk++;
        if(lane == 0 && a.counters) atomicAdd(&a.counters[80 + (blockIdx.x & 1023) * 32 + 16 + bk], 1ull);
    }
#endif
    // ---- g(D + E) as a series in E, without eigenvalue gaps in any denominator.  With M = c I + D + E:
    //        M^(1/2) = X = diag(a) + R:  simplified Newton steps on X^2 = M with the Sylvester operator of diag(a) kept fixed,
    //                 R(k+1) = R(k) + (M - (diag(a) + R(k))^2) o rinv,   R(0) = 0,   rinv(i, j) = 1 / (a_i + a_j)        (GPP_ENSI_NSQ steps)
    //        g(D + E) = -(M + sqrt(c) M^(1/2))^-1 = -(P + F)^-1,   P = diag(a (a + sqrt(c))) = diag(-1 / dw),   F = E + sqrt(c) R
    //                 = diag(dw) + T0 + T1 + ...,   T0 = H = diag(dw) F diag(dw),   T(k+1) = (T(k) F) diag(dw)          (Neumann, GPP_ENSI_NNEU products)
    //      The first-order part is the Daleckii-Krein term (1 + sqrt(c) / (a_i + a_j)) dw_i dw_j E_ij; the higher orders let the sweeps of
    //      k_ensi_pair stop early -- most warm-started cells need no sweep at all then.  Round 3: two steps / two products, |E| <= 0.010 c (one value
    //      in 10^6 of the soak outside the plain 1e-5 measure: a float32 rounding of a member sum falling the other way under a residual of 1e-8
    //      in W).  Round 4: three / three at 0.020 c -- what the series leaves is (|R1| / a)^4 / 2 <= 5e-9 and (|F| / (2 c))^5 <= 8e-10, nothing of
    //      the soak outside --, then, with the steps at a quarter of their price (float32 products, tile layout: below), FOUR / FOUR at 0.040 c:
    //      (|R1| / a)^5 / 2 <= 2e-9 and (|F| / (2 c))^6 <= 7e-10, the soak's worst deviation equal to that of the converged sweeps (2.5e-6),
    //      config 5 237.8 -> 228 ms (tools/ensi_order_sweep.sh: every order against every threshold).
    //      (rinv comes from v_rcp_f32: its 1e-7 enters every step, and every following step corrects it: left over is 1e-7 of the last step.)
    // square root: R(k+1) = R(k) + (M - diag(a)^2 - diag(a) R(k) - R(k) diag(a) - R(k) R(k)) o rinv, R(0) = 0 -- the bracket without the product is
    // where the cancellation happens and is taken entry by entry in double precision; R(k) R(k) is a second-order term (float32 product).
    // The whole series runs in the register layout of the matrix-core results (lane (kq, r16), register r of tile (ti, tj): row 16 ti + 4 kq + r,
    // column 16 tj + r16; the three tiles on and above the diagonal, the fourth is their mirror image): twelve entries a lane and ALL lanes at
    // work on the entry-wise parts, where the row layout kept 32 entries in each of 32 lanes -- those parts, not the products, were what a step
    // of the series cost (tools/ensi_order_sweep.sh: 5 / 10 ms per step on config 5 against 2 ms of matrix-core time).  The operands of the
    // products are staged as floats (pitch PPF).
    constexpr int PPF = 33;
    float* const sAf = reinterpret_cast<float*>(sA);
    float* const sBf2 = sAf + 32 * PPF;                            // F' beside the T / R operands in area A: area B keeps the rows of U through the series
    const int tr16 = lane & 15, tkq = lane >> 4;
    double et[3][4], ft[3][4];   // E (diagonal: M(i, i) - a_i^2) and R(k) / the sum of the series; tiles (0,0), (0,1), (1,1)
    float rinvt[3][4];
    {   // rows of U^T B U (lanes 0..31) -> area A, rows of U (lanes 32..63) -> area B
        double* const dst = h == 0 ? sA : sB;
#pragma unroll
        for(int j = 0; j < 32; j += 2) { double2 v; v.x = e[j]; v.y = e[j + 1]; *reinterpret_cast<double2*>(&dst[i * PP + j]) = v; }
    }
    __syncthreads();
    // ---- z = U (C + E)^-1 U^T r,  C = diag(c + S):  (C + E)^-1 = C^-1 - C^-1 E C^-1 + C^-1 E C^-1 E C^-1 - ...  (first: E and U are both staged now).
    //      Every sum of 32 terms is split between the two lanes (h, i) of an index i, sixteen terms each, and put together with one
    //      v_permlane32_swap per dword; the iterate alternates between two LDS vectors (one barrier a term).  Round 4: this block took
    //      10 % of the kernel with the sums in lanes 0..31 alone, two barriers a term and the scalar unit in every step.
    {
        int jb = 16 * h;
        asm volatile("" : "+v"(jb));   // (its own value: shared with the 16 h of the member update below, the index arithmetic stayed in registers across everything between)
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for(int rr = 0; rr < 16; rr += 2) {
            p0 = __builtin_fma(sB[(jb + rr) * PP + i], s_r1[jb + rr], p0);
            p1 = __builtin_fma(sB[(jb + rr + 1) * PP + i], s_r1[jb + rr + 1], p1);
        }
        double eh[16];   // E(i, 16 h + jj), diagonal zeroed
#pragma unroll
        for(int jj = 0; jj < 16; jj += 2) {
            const double2 v = *reinterpret_cast<const double2*>(&sA[i * PP + jb + jj]);
            eh[jj] = (jb + jj == i) ? 0.0 : v.x; eh[jj + 1] = (jb + jj + 1 == i) ? 0.0 : v.y;
        }
        double vk = both_halves_sum_d(p0 + p1, lane) * inv;                                          // v0 = C^-1 U^T r
        double tz = vk;
        double* cur = s_t, * nxt = s_r1;   // (s_r1 is read: every lane's loads above are waited for by the barrier below)
        __syncthreads();
        if(h == 0) cur[i] = vk;
        __syncthreads();
        // (the series in E C^-1 has the ratio |E| / (c + S) <= 0.04 -- the stopping threshold of the sweeps --: eight terms leave 7e-12)
#pragma unroll 1
        for(int term = 1; term < 8; ++term) {
            double c0 = 0.0, c1 = 0.0;
#pragma unroll
            for(int jj = 0; jj < 16; jj += 2) {
                const double2 v = *reinterpret_cast<const double2*>(&cur[jb + jj]);
                c0 = __builtin_fma(eh[jj], v.x, c0);
                c1 = __builtin_fma(eh[jj + 1], v.y, c1);
            }
            vk = -inv * both_halves_sum_d(c0 + c1, lane);
            tz += vk;
            if(h == 0) nxt[i] = vk;
            __syncthreads();
            double* const tmp = cur; cur = nxt; nxt = tmp;
        }
        if(h == 0) nxt[i] = tz;
        __syncthreads();
        double z0 = 0.0, z1 = 0.0;
#pragma unroll
        for(int jj = 0; jj < 16; jj += 2) {
            const double2 u2 = *reinterpret_cast<const double2*>(&sB[i * PP + jb + jj]);
            const double2 v = *reinterpret_cast<const double2*>(&nxt[jb + jj]);
            z0 = __builtin_fma(u2.x, v.x, z0);
            z1 = __builtin_fma(u2.y, v.y, z1);
        }
        const double zz = both_halves_sum_d(z0 + z1, lane);
        if(h == 0) s_z1[i] = zz;
    }
    EPROF(2)   // z
#pragma unroll
    for(int t = 0; t < 3; ++t) {
        const int ti = t >> 1, tj = (t + 1) >> 1;   // (0,0), (0,1), (1,1)
        const double acol = s_rt[16 * tj + tr16];
#pragma unroll
        for(int r = 0; r < 4; ++r) {
            const int row = 16 * ti + 4 * tkq + r, col = 16 * tj + tr16;
            const double arow = s_rt[row];
            const double v = sA[row * PP + col];
            const bool isd = ti == tj && row == col;
            et[t][r] = isd ? __builtin_fma(-arow, arow, c + v) : v;   // M(i, i) - a_i^2: 0 up to the rounding of the square root; d_i itself for a negative estimate
            rinvt[t][r] = __builtin_amdgcn_rcpf((float)(arow + acol));
            ft[t][r] = et[t][r] * (double)rinvt[t][r];                 // R(1) = (M - diag(a)^2) o rinv
        }
    }
    __syncthreads();   // (area A is read)
    auto stage_sym = [&](float* const M, const int t, const int r, const float v) {   // entry (t, r) of a symmetric matrix and its mirror image
        const int ti = t >> 1, tj = (t + 1) >> 1;
        const int row = 16 * ti + 4 * tkq + r, col = 16 * tj + tr16;
        M[row * PPF + col] = v;
        if(t == 1) M[col * PPF + row] = v;
    };
#pragma unroll 1
    for(int step = 1; step < GPP_ENSI_NSQ; ++step) {
#pragma unroll
        for(int t = 0; t < 3; ++t)
#pragma unroll
            for(int r = 0; r < 4; ++r) stage_sym(sAf, t, r, (float)ft[t][r]);
        __syncthreads();
        const Acc32f rr = mfma_32_f32<true>(lane, [&](int r, int k) { return sAf[r * PPF + k]; }, [&](int k, int cc) { return sAf[k * PPF + cc]; });   // R(k) R(k)
#pragma unroll
        for(int t = 0; t < 3; ++t) {
            const int ti = t >> 1, tj = (t + 1) >> 1;
            const double acol = s_rt[16 * tj + tr16];
#pragma unroll
            for(int r = 0; r < 4; ++r) {
                const double ssum = s_rt[16 * ti + 4 * tkq + r] + acol;
                ft[t][r] += (__builtin_fma(-ssum, ft[t][r], et[t][r]) - (double)rr.t[ti][tj][r]) * (double)rinvt[t][r];
            }
        }
        __syncthreads();   // (the operands are read)
    }
    // F = E + sqrt(c) R;  F' = F diag(dw) (T(k+1) = T(k) F' needs no scaling pass) -> area B, T0 = H = diag(dw) F' -> area A, running sum: diag(dw) + T0
#pragma unroll
    for(int t = 0; t < 3; ++t) {
        const int ti = t >> 1, tj = (t + 1) >> 1;
        const double dwc = s_dw[16 * tj + tr16];
#pragma unroll
        for(int r = 0; r < 4; ++r) {
            const int row = 16 * ti + 4 * tkq + r, col = 16 * tj + tr16;
            const bool isd = ti == tj && row == col;
            const double dwr = s_dw[row];
            const double fv = (isd ? 0.0 : et[t][r]) + sqc * ft[t][r];
            const double t0 = dwr * fv * dwc;
            sBf2[row * PPF + col] = (float)(fv * dwc);
            if(t == 1) sBf2[col * PPF + row] = (float)(fv * dwr);      // (F is symmetric, F' is not)
            stage_sym(sAf, t, r, (float)t0);
            ft[t][r] = (isd ? dwr : 0.0) + t0;
        }
    }
    __syncthreads();
#pragma unroll 1
    for(int term = 0; term < GPP_ENSI_NNEU; ++term) {   // T(k+1) = T(k) F'
        // (T(k) = (diag(dw) F)^k diag(dw) F diag(dw) is symmetric: F is, and (D F)^k D = D (F D)^k)
        const Acc32f tt = mfma_32_f32<true>(lane, [&](int r, int k) { return sAf[r * PPF + k]; }, [&](int k, int cc) { return sBf2[k * PPF + cc]; });
#pragma unroll
        for(int t = 0; t < 3; ++t)
#pragma unroll
            for(int r = 0; r < 4; ++r) ft[t][r] += (double)tt.t[t >> 1][(t + 1) >> 1][r];
        __syncthreads();   // (T(k) is read)
        if(term + 1 < GPP_ENSI_NNEU) {
#pragma unroll
            for(int t = 0; t < 3; ++t)
#pragma unroll
                for(int r = 0; r < 4; ++r) stage_sym(sAf, t, r, tt.t[t >> 1][(t + 1) >> 1][r]);
        }
        __syncthreads();
    }
    // the middle matrix of W_sym -> area A (doubles, row major)
#pragma unroll
    for(int t = 0; t < 3; ++t) {
        const int ti = t >> 1, tj = (t + 1) >> 1;
#pragma unroll
        for(int r = 0; r < 4; ++r) {
            const int row = 16 * ti + 4 * tkq + r, col = 16 * tj + tr16;
            sA[row * PP + col] = ft[t][r];
            if(t == 1) sA[col * PP + row] = ft[t][r];
        }
    }
    __syncthreads();
    EPROF(1)   // perturbation series (five products)
    // M_W = U Mmid U^T, scaled: M'(i, j) = sD_i M_W(i, j) sD_j   -> area A (stays there for the whole member update)
    {
        const Acc32 tm = mfma_32_full(lane, [&](int r, int k) { return sB[r * PP + k]; }, [&](int k, int cc) { return sA[k * PP + cc]; });
        __syncthreads();
        acc32_store_full(tm, lane, sA);
        __syncthreads();
        const Acc32 mw = mfma_32_full<true>(lane, [&](int r, int k) { return sA[r * PP + k]; }, [&](int k, int cc) { return sB[cc * PP + k]; });
        __syncthreads();
#pragma unroll
        for(int ti = 0; ti < 2; ++ti)
#pragma unroll
            for(int tj = ti; tj < 2; ++tj)
#pragma unroll
                for(int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tj + (lane & 15);
                    const double v = GPP_DBG(a, 8) ? 0.0 : mw.t[ti][tj][r] * (s_sD1[row] * s_sD1[col]);
                    sA[row * PP + col] = v;
                    if(ti != tj) sA[col * PP + row] = v;   // (U Mmid U^T is symmetric: the tile below the diagonal is the mirror image)
                }
    }
    __syncthreads();
    EPROF(3)   // M_W (two products)
        // anti-extrapolation tables (oi_ensi.cpp:520-552): lY[e] is a LINEAR index into the n x nV column-major matrix, so it
        // depends on the ORDER of the selected observations: rho descending when the reference sorted (more usable
        // observations than max_points), candidate (= index) order otherwise
        if(!a.allow_extrap) {
            const bool tr_l = (meta & 0x100u) != 0u;
            unsigned long long* const s_k64 = reinterpret_cast<unsigned long long*>(s_t);   // 32 keys (s_t is free again)
            if(h == 0) s_k64[i] = (i < n) ? (((tr_l ? (unsigned long long)__float_as_uint(s_rho[i]) << 32 : 0ull)) | (unsigned)(~s_sel[i])) : 0ull;
            __syncthreads();
            if(h == 0 && i < n) {
                const unsigned long long mine = s_k64[i];
                int rank = 0;
                for(int j = 0; j < n; ++j) rank += (s_k64[j] > mine) ? 1 : 0;
                s_perm[rank] = (int)s_sel[i];
            }
            __syncthreads();
        }
        // ---- ensemble side: all 64 lanes, lane = member (chunks of 64) -----------------------------------------------------------------
        // ensemble mean: sequential float sum over the valid members in member order (oi_ensi.cpp:447-461)
        const float v0 = s_v0[lane];
        float total = 0.0f;
        for(int m0 = 0; m0 < nV; m0 += 64) {
            const float v = (m0 == 0) ? v0 : ((m0 + lane < nV) ? a.bg[(long)cell_l * E + ensi_member(a, m0 + lane)] : 0.0f);
            const int kend = min(64, nV - m0);
            // (blocks of 16 with the lane numbers as constants: with a loop counter as the lane select every step waits for the scalar unit,
            //  and the chain of dependent additions is what this costs anyway)
#pragma unroll
            for(int b = 0; b < 4; ++b) {
                if(16 * b + 16 <= kend) {
#pragma unroll
                    for(int j = 0; j < 16; ++j) total += readlane_f(v, 16 * b + j);
                }
            }
            for(int k = kend & ~15; k < kend; ++k) total += readlane_f(v, k);
        }
        const float ensMean = total / (float)nV;
        EPROF(4)   // tables, ensemble mean
#pragma unroll 1
        for(int e0_ = 0; e0_ < nV; e0_ += 64) {
            const int e = e0_ + lane;
            // Y tile of this member chunk -> area B (floats)
            __syncthreads();
            {   // unconditional loads (clamped address, value deselected afterwards) so that all of them are in flight together
                float yv[EN];
#pragma unroll
                for(int r = 0; r < EN; ++r) {
                    const bool on = r < n && e < nV;
                    yv[r] = a.gY[(on && !GPP_DBG(a, 32)) ? (long)s_sel[r] * nV + e : 0];   // (bit 32, timing experiment: one address)
                    yv[r] = on ? yv[r] : 0.0f;
                }
#pragma unroll
                for(int r = 0; r < EN; ++r) sBf[r * YP + lane] = yv[r];
            }
            __syncthreads();
            float acc = 0.0f;
            double X = 0.0;
            // Q = M' Y  (32 x 16 NT) on the matrix cores: NT tiles of 16 members
            auto q_product = [&](auto ntc, auto& qa) __attribute__((always_inline)) {
                constexpr int NT = decltype(ntc)::value;
#pragma unroll
                for(int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for(int tj = 0; tj < NT; ++tj) qa[ti][tj] = (v4d){0.0, 0.0, 0.0, 0.0};
                const int r = lane & 15, kq = lane >> 4;
#pragma unroll
                for(int ks = 0; ks < 8; ++ks) {
                    double bop[NT];
#pragma unroll
                    for(int tj = 0; tj < NT; ++tj) bop[tj] = (double)sBf[(4 * ks + kq) * YP + 16 * tj + r];
                    const double am0 = sA[r * PP + 4 * ks + kq], am1 = sA[(r + 16) * PP + 4 * ks + kq];
#pragma unroll
                    for(int tj = 0; tj < NT; ++tj) {
                        qa[0][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(am0, bop[tj], qa[0][tj], 0, 0, 0);
                        qa[1][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(am1, bop[tj], qa[1][tj], 0, 0, 0);
                    }
                }
            };
            if constexpr(ONECHUNK) {
                // Up to 64 members (one chunk): W' = Y^T Q on the matrix cores as well (round 3: config 5 309 -> 296 ms).  The accumulators of the Q product ARE the B
                // operands of this one (lane (kq, r16) holds Q(16 ti + kq + 4 r, 16 te + r16): k-step ks = 4 ti + r), the A operands are
                // the values of the Y tile the Q product read.  One 16-member slab of W' at a time goes through area A (M' is not
                // needed any more) so that lane = member e finds its 16 values W'(k, e) in k order for the float accumulation of
                // oi_ensi.cpp:505-511; X_k and w_k come from lane k.
                // Round 4, tail members: both products cover NT tiles of 16 members.  With 50 valid members four tiles spend 72 of their 192
                // v_mfma_f64 on the 14 padding columns next to members 48 and 49 -- and every one of them holds the FP64 pipe of its SIMD for
                // 66 cycles (tools/ubench/mfma_f64_rate.hip).  Up to four members beyond the last full tile are taken by the vector unit
                // instead: their columns of Q as two 32-term dot products per row, their rows AND columns of W' (it is symmetric:
                // W'(k, e) = sum_r Y(r, k) Q(r, e)) as one dot product per member and tail column, transposed through LDS for the tail lanes.
                const int mraw = nV & 15, nfull = nV >> 4;
                const bool tail = mraw >= 1 && mraw <= 4 && nfull >= 1;
                const int nt = tail ? nfull : (nV + 15) >> 4;
                auto update = [&](auto ntc) __attribute__((always_inline)) {
                    constexpr int NT = decltype(ntc)::value;
                    constexpr int NB = 16 * NT;              // first tail member
                    const int m = tail ? nV - NB : 0;       // tail members (0..4)
                    v4d qa[2][NT];
                    q_product(ntc, qa);
                    EPROF(5)   // Y tile, Q
                    const int r16 = lane & 15, kq = lane >> 4;
                    X = (double)v0 - (double)ensMean;
                    if(m > 0) {   // Q(:, NB + j): row i by lanes (h, i), columns j0 + h
#pragma unroll 1
                        for(int j0 = 0; j0 < m; j0 += 2) {
                            const int col = NB + j0 + h;     // (a column beyond nV holds zeros)
                            double q0 = 0.0, q1 = 0.0;
#pragma unroll 4
                            for(int cc = 0; cc < 32; cc += 2) {
                                const double2 mm = *reinterpret_cast<const double2*>(&sA[i * PP + cc]);
                                q0 = __builtin_fma(mm.x, (double)sBf[cc * YP + col], q0);
                                q1 = __builtin_fma(mm.y, (double)sBf[(cc + 1) * YP + col], q1);
                            }
                            s_qt[(j0 >> 1) * 64 + 2 * i + h] = q0 + q1;
                        }
                    }
                    __syncthreads();   // (the tail columns of Q are visible, and nobody reads M' in area A any more)
                    double wk = 0.0;   // w_k = sum_r sD_r Y(r,k) z_r for member k = lane
                    double wtl[4] = {0.0, 0.0, 0.0, 0.0};   // W'(lane, NB + j)
                    if(m > 0) {
#pragma unroll 8
                        for(int r = 0; r < 32; ++r) {
                            const double yv = (double)sBf[r * YP + lane];
                            const double2 q2 = *reinterpret_cast<const double2*>(&s_qt[2 * r]);
                            wk = __builtin_fma(s_sD1[r] * yv, s_z1[r], wk);
                            wtl[0] = __builtin_fma(yv, q2.x, wtl[0]);
                            wtl[1] = __builtin_fma(yv, q2.y, wtl[1]);
                        }
                        if(m > 2) {
#pragma unroll 8
                            for(int r = 0; r < 32; ++r) {
                                const double yv = (double)sBf[r * YP + lane];
                                const double2 q2 = *reinterpret_cast<const double2*>(&s_qt[64 + 2 * r]);
                                wtl[2] = __builtin_fma(yv, q2.x, wtl[2]);
                                wtl[3] = __builtin_fma(yv, q2.y, wtl[3]);
                            }
                        }
                    }
                    else {
#pragma unroll
                        for(int r = 0; r < 32; ++r) wk = __builtin_fma(s_sD1[r] * (double)sBf[r * YP + lane], s_z1[r], wk);
                    }
                    // W'(k, NB + j) for the tail lanes: behind the slab exchange area [NB][17] in area A (NB <= 48 with a tail: 816 + 256 <= 1088 doubles)
                    double* const s_tail = sA + (NB <= 48 ? NB : 48) * 17;
                    if(m > 0) {
#pragma unroll
                        for(int j = 0; j < 4; ++j) if(j < m) s_tail[j * 64 + lane] = wtl[j];
                    }
                    const double* const src0 = (lane < NB || !tail) ? sA + lane * 17 : s_tail + min(lane - NB, 3) * 64;
                    const int sstep = (lane < NB || !tail) ? 0 : 16;
#pragma unroll 1
                    for(int tk = 0; tk < NT; ++tk) {
                        v4d wt[NT];
#pragma unroll
                        for(int te = 0; te < NT; ++te) wt[te] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for(int ks = 0; ks < 8; ++ks) {
                            const double aop = (double)sBf[(4 * ks + kq) * YP + 16 * tk + r16];   // Y(i = 4 ks + kq, k = 16 tk + r16)
#pragma unroll
                            for(int te = 0; te < NT; ++te) wt[te] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, qa[ks >> 2][te][ks & 3], wt[te], 0, 0, 0);
                        }
                        __syncthreads();   // (the previous slab has been read)
#pragma unroll
                        for(int te = 0; te < NT; ++te)
#pragma unroll
                            for(int r = 0; r < 4; ++r) sA[(16 * te + r16) * 17 + kq + 4 * r] = wt[te][r];   // W'(k = 16 tk + kq + 4 r, e = 16 te + r16)
                        __syncthreads();
                        double wv[16];
                        const double* const src = src0 + sstep * tk;
#pragma unroll
                        for(int kl = 0; kl < 16; ++kl) wv[kl] = src[kl];
#pragma unroll
                        for(int kl = 0; kl < 16; ++kl) {
                            const int k = 16 * tk + kl;
                            if(k < nV) {
                                const double wke = ((k == lane ? 1.0 : 0.0) + wv[kl]) + readlane_d(wk, k);
                                acc = (float)((double)acc + readlane_d(X, k) * wke);
                            }
                        }
                    }
#pragma unroll
                    for(int j = 0; j < 4; ++j) {   // the tail members' own steps: W'(NB + j, e) = W'(e, NB + j)
                        if(j < m) {
                            const int k = NB + j;
                            const double wke = ((k == lane ? 1.0 : 0.0) + wtl[j]) + readlane_d(wk, k);
                            acc = (float)((double)acc + readlane_d(X, k) * wke);
                        }
                    }
                };
                if(nt == 4) update(std::integral_constant<int, 4>{});
                else if(nt == 3) update(std::integral_constant<int, 3>{});
                else if(nt == 2) update(std::integral_constant<int, 2>{});
                else update(std::integral_constant<int, 1>{});
                EPROF(7)   // member update
            }
            else {
            v4d qa[2][4];
            q_product(std::integral_constant<int, 4>{}, qa);
            EPROF(5)   // Y tile, Q
            // columns i and 32 + i of Y, rows [16 h, 16 h + 16), for the tables of the member update: out of the Y tile while it is still
            // there (first member chunk: the tile holds columns 0..63; otherwise they are loaded again below)
            float ya[16], yb[16];
#pragma unroll
            for(int rr = 0; rr < 16; ++rr) { ya[rr] = sBf[(16 * h + rr) * YP + i]; yb[rr] = sBf[(16 * h + rr) * YP + 32 + i]; }
            // transposed through area B, 32 members at a time: Qt[member][row]
            double q[32];
#pragma unroll
            for(int half = 0; half < 2; ++half) {
                __syncthreads();
#pragma unroll
                for(int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for(int tj = 0; tj < 2; ++tj)
#pragma unroll
                        for(int r = 0; r < 4; ++r) sB[(16 * tj + (lane & 15)) * PP + 16 * ti + (lane >> 4) + 4 * r] = qa[ti][2 * half + tj][r];
                __syncthreads();
#pragma unroll
                for(int j = 0; j < 32; j += 2) {   // (selects, not a conditional store: the array must stay in registers)
                    const double2 v = *reinterpret_cast<const double2*>(&sB[i * PP + j]);
                    q[j] = (half == 0 || h == 1) ? v.x : q[j]; q[j + 1] = (half == 0 || h == 1) ? v.y : q[j + 1];
                }
            }
            const float value = (e0_ == 0) ? v0 : ((e < nV) ? a.bg[(long)cell_l * E + ensi_member(a, e)] : 0.0f);
            X = (double)value - (double)ensMean;
            EPROF(6)   // transposition
            // total_e = sum_k X_k W(k,e), W(k,e) = [k == e] + sum_i Y(i,k) q_e(i) + w_k, float accumulation in k order (:505-511)
#pragma unroll 1
            for(int k0 = 0; k0 < nV; k0 += 32) {
                const int kk = k0 + i;    // lanes i and 32 + i share column kk: rows [16 h, 16 h + 16)
                __syncthreads();
                {   // column kk of Y as doubles, X_kk and w_kk = sum_r sD_r Y(r,kk) z_r  -> row i of the table in area B
                    double wk = 0.0;
#pragma unroll
                    for(int rr = 0; rr < 16; rr += 2) {
                        const int r = 16 * h + rr;
                        double2 v;
                        if(e0_ == 0 && k0 < 64) {
                            v.x = (double)(k0 == 0 ? ya[rr] : yb[rr]); v.y = (double)(k0 == 0 ? ya[rr + 1] : yb[rr + 1]);
                        }
                        else {
                            v.x = (r < n && kk < nV) ? (double)a.gY[(long)s_sel[r] * nV + kk] : 0.0;
                            v.y = (r + 1 < n && kk < nV) ? (double)a.gY[(long)s_sel[r + 1] * nV + kk] : 0.0;
                        }
                        wk = __builtin_fma(s_sD1[r] * v.x, s_z1[r], wk);
                        wk = __builtin_fma(s_sD1[r + 1] * v.y, s_z1[r + 1], wk);
                        *reinterpret_cast<double2*>(&sB[i * PP + r]) = v;
                    }
                    const double wo = __shfl_xor(wk, 32);
                    const float vk64 = __shfl(v0, kk & 63);
                    if(h == 0) {
                        const float vk = (kk < 64) ? vk64 : ((kk < nV) ? a.bg[(long)cell_l * E + ensi_member(a, kk)] : 0.0f);
                        double2 xw; xw.x = (double)vk - (double)ensMean; xw.y = wk + wo;
                        *reinterpret_cast<double2*>(&sB[i * PP + 32]) = xw;
                    }
                }
                __syncthreads();
                const int kend = GPP_DBG(a, 2) ? 0 : min(32, nV - k0);
                // one row of the table per step (other waves of the SIMD hide the LDS latency here)
#pragma unroll 1
                for(int k = 0; k < kend; ++k) {
                    double2 row[17];
#pragma unroll
                    for(int r = 0; r < 17; ++r) row[r] = *reinterpret_cast<const double2*>(&sB[k * PP + 2 * r]);
                    double wke = (k0 + k == e) ? 1.0 : 0.0, wk1 = 0.0;   // (two chains: the sum over the rows has no prescribed order)
#pragma unroll
                    for(int r = 0; r < 16; ++r) {
                        wke = __builtin_fma(row[r].x, q[2 * r], wke);
                        wk1 = __builtin_fma(row[r].y, q[2 * r + 1], wk1);
                    }
                    wke = (wke + wk1) + row[16].y;
                    acc = (float)((double)acc + row[16].x * wke);
                }
            }
            EPROF(7)   // member update
            }
            float currIncrement = acc;
            if(!a.allow_extrap && e < nV) {
                const int li = e % n, lk = e / n;
                const double lYe = (double)a.gY[(long)(unsigned)s_perm[li] * nV + lk];
                float maxInc = -INFINITY, minInc = INFINITY;
                for(int r = 0; r < n; ++r) {
                    const float dv = (float)((double)s_ob[r] - (lYe + (double)s_yh[r]));
                    maxInc = fmaxf(maxInc, dv); minInc = fminf(minInc, dv);
                }
                const float memberIncrement = (float)((double)currIncrement - X);
                if(maxInc > 0 && memberIncrement > maxInc) currIncrement = (float)((double)maxInc + X);
                else if(maxInc < 0 && memberIncrement > 0) currIncrement = (float)(0.0 + X);
                else if(minInc < 0 && memberIncrement < minInc) currIncrement = (float)((double)minInc + X);
                else if(minInc > 0 && memberIncrement < 0) currIncrement = (float)(0.0 + X);
            }
            if(e < nV) a.out[(long)cell_l * E + ensi_member(a, e)] = ensMean + currIncrement;   // :553
        }

#ifdef GPP_ENSI_PROFILE
    EPROF(8)   // clamp, store
    if(lane == 0 && a.counters) for(int k = 0; k < 12; ++k) atomicAdd(&a.counters[80 + (blockIdx.x & 1023) * 32 + 16 + k], prof[k]);
#endif
}
