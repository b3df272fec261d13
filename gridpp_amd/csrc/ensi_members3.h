// k_ensi_members3: the ensemble side of optimal_interpolation_ensi at THREE waves per SIMD (included by ensi.hip after ensi_pair.h).
//
// The mathematics of k_ensi_members<true> (ensi_pair.h; oi_ensi.cpp:379-553 -- see there for the series and its error bounds), one wave per
// cell, at most 64 valid members; two things are computed differently: the Neumann series of the inverse in its symmetric form (three products
// instead of four, below) and W' = Y^T M' Y by symmetry (its tiles on and above the diagonal).  k_ensi_members holds two 8.7 KB double-precision staging areas (19.8 KB of LDS per cell, 201 VGPRs): two
// waves per SIMD.  A third needs <= 13.3 KB and <= 168 VGPRs per wave.  Here ONE staging area serves every phase in turn
//      rows of U (z = U (C + E)^-1 U^T r)  ->  rows of U^T B U (the entries of the series)  ->  float operands of the series' products
//      ->  Mmid  ->  U Mmid  ->  M' = sD (U Mmid U^T) sD  ->  the 16-member slabs of W' = Y^T Q
// and what the second area held lives in registers, in the layout the matrix cores read their operands in (lane (kq, r16) = (lane >> 4, lane & 15)):
//      U:  uop[t][ks] = U(r16 + 16 t, 4 ks + kq)  -- the A operand of U Mmid AND the B operand of (U Mmid) U^T -- 16 doubles, loaded from the park a second time;
//      Y:  yop[ks][t] = Y(4 ks + kq, 16 t + r16)  -- the B operand of Q = M' Y AND the A operand of W' = Y^T Q -- 32 floats, loaded from HBM in that layout.
// Sums over the rows of Y that k_ensi_members took per member (lane = member reads its column of the Y tile) are taken per operand lane -- the rows
// 4 ks + kq of four members 16 t + r16 -- and put together over kq with a transposing reduction (member_reduce): the order of these double-precision
// sums differs from k_ensi_members', everything that is rounded to float32 on the way (oi_ensi.cpp:505-511) is accumulated in the same order.
// The operand loads are issued in front of the Neumann products (the square-root steps before them need the registers) and arrive behind them.
// Config 5: 98.8 -> 79 ms for this kernel; 6 values in 10^6 one float32 ulp away from k_ensi_members' (tools/ensi_members_ab.py).
#pragma once

// lanes 0..31: x of the own lane; lanes 32..63: y of lane - 32
__device__ __forceinline__ double lower_y_to_upper(const double x, const double y) {
    const auto sl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto sh = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)sh[0], (int)sl[0]);
}

// lane (kq, r16) holds p[t]: a partial sum for member 16 t + r16 over the rows = kq (mod 4).  Returns to lane l the total of member l (t = l >> 4):
// first the halves of the wave exchange the two members the other half keeps, then neighbouring rows of 16 lanes the one the other keeps.
__device__ __forceinline__ double member_reduce(const double p0, const double p1, const double p2, const double p3, const int lane) {
    const bool up = lane >= 32, odd = (lane & 16) != 0;
    double a[2];
#pragma unroll
    for(int u = 0; u < 2; ++u) {
        const double lo_t = u ? p1 : p0, hi_t = u ? p3 : p2;   // upper lanes send lo_t (members of the lower half), lower lanes send hi_t
        const auto sl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(lo_t), (unsigned)__double2loint(hi_t), false, false);
        const auto sh = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(lo_t), (unsigned)__double2hiint(hi_t), false, false);
        const double recv = __hiloint2double((int)(up ? sh[0] : sh[1]), (int)(up ? sl[0] : sl[1]));
        a[u] = (up ? hi_t : lo_t) + recv;
    }
    const double send = odd ? a[0] : a[1];
    return (odd ? a[1] : a[0]) + __shfl_xor(send, 16);
}

// a Y operand as a double, converted where it is used: the optimiser would otherwise keep all 32 of them converted (64 registers)
__device__ __forceinline__ double yd(float y) { asm volatile("" : "+v"(y)); return (double)y; }

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_ensi_members3(EnsiArgs a) {
    __shared__ __attribute__((aligned(16))) double s_ar[32 * PP];   // THE staging area (8704 bytes)
    __shared__ __attribute__((aligned(16))) double s_small[7 * 32];
    double* const s_sD1 = s_small, * const s_z1 = s_small + 32, * const s_t = s_small + 64, * const s_r1 = s_small + 96, * const s_dw = s_small + 128,
          * const s_rt = s_small + 160, * const s_g = s_small + 192;
    double* const s_qt = s_t;                                     // member update: the Q columns of up to four tail members, [pair][row][2] (s_t .. s_rt are free then)
    __shared__ int s_i[160];                                      // perm[32] | obs[32] | yhat[32] | selection[32] | rho[32] (floats)
    __shared__ float s_v0[64];                                    // the members' values wait here through the spectral part
    __shared__ float s_yt[32 * 4];                                // Y(row, NB + j) of up to four tail members
    int* const s_perm = s_i;
    float* const s_ob = reinterpret_cast<float*>(s_i + 32);
    float* const s_yh = reinterpret_cast<float*>(s_i + 64);
    unsigned* const s_sel = reinterpret_cast<unsigned*>(s_i + 96);
    float* const s_rho = reinterpret_cast<float*>(s_i + 128);
    const int lane = threadIdx.x;
    const int h = lane >> 5, i = lane & 31;
    const int r16 = lane & 15, kq = lane >> 4;
    const int nV = a.nV, E = a.E;
    if(nV <= 1) return;
#ifdef GPP_ENSI_PROFILE
    unsigned long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    const int tile = a.tile0 + (int)(blockIdx.x >> 6), lcell = (int)(blockIdx.x & 63);
    const int cell_l = ensi_cell_of(a, tile, lcell);
    if(cell_l < 0) return;
    // one round of independent loads: length / order flag, member values, the park (rows first, U a second time in operand layout)
    const unsigned meta = a.meta[(size_t)tile * 64 + lcell];
    const float v0_ld = (lane < nV) ? a.bg[(long)cell_l * E + ensi_member(a, lane)] : 0.0f;
    const double* const park = a.cpark + ((size_t)blockIdx.x) * ENSI_PARK_D;
    const unsigned long long pk0 = __double_as_longlong(park[2144 + i]), pk1 = __double_as_longlong(park[2176 + i]);
    const float rho = (float)park[2112 + i];
    const double p_sD = park[2048 + i], p_r1 = park[2080 + i];
    // lanes 32..63: row i of U; lanes 0..31: row i of U^T B U (e[i]: eigenvalue estimate d_i, the rest: the off-diagonal part E)
    double e[32];
#pragma unroll
    for(int j = 0; j < 32; j += 2) {
        const double2 w = *reinterpret_cast<const double2*>(&park[(j >> 1) * 128 + lane * 2]); e[j] = w.x; e[j + 1] = w.y;
    }
    const int n = (int)(meta & 0xffu);
    if(n == 0) return;   // no observation in range (the output already holds the background) or a cell of k_ensi_big
    const unsigned orig_i = (i < n) ? (unsigned)pk0 : 0xffffffffu;
    float4 o1 = make_float4(NAN, 0, 0, 1);
    if(i < n) { o1.y = __uint_as_float((unsigned)(pk0 >> 32)); o1.z = __uint_as_float((unsigned)pk1); }
    const double c = (double)((float)(nV - 1));   // diag = 1/delta*(nValidEns-1), float (oi_ensi.cpp:383)
    const double sqc = sqrt(c);
    // spectral functions (lane i < 32: eigenvalue i)
    double ei = 0.0;
#pragma unroll
    for(int j = 0; j < 32; ++j) ei = (j == i) ? e[j] : ei;
    const double S = ei < 0.0 ? 0.0 : ei;
    const double rt = sqrt(c + S);                    // a_i
    const double dwv = -1.0 / (rt * (rt + sqc));      // W_sym = I + A^T g(B) A,  g(S) = -1 / (a (a + sqrt(c))),  a = sqrt(c + S)
    const double inv = 1.0 / (c + S);
    s_v0[lane] = v0_ld;
    if(h == 0) { s_sel[i] = orig_i; s_sD1[i] = p_sD; s_r1[i] = p_r1; s_dw[i] = dwv; s_g[i] = sqrt(-dwv); s_rt[i] = rt; s_ob[i] = o1.y; s_yh[i] = o1.z; s_rho[i] = rho; }
    // rows of U (lanes 32..63) -> the area
    if(h == 1) {
#pragma unroll
        for(int j = 0; j < 32; j += 2) { double2 v; v.x = e[j]; v.y = e[j + 1]; *reinterpret_cast<double2*>(&s_ar[i * PP + j]) = v; }
    }
    __syncthreads();
    EPROF(0)   // park loads, spectral scalars
    // ---- z = U (C + E)^-1 U^T r,  C = diag(c + S):  (C + E)^-1 = C^-1 - C^-1 E C^-1 + C^-1 E C^-1 E C^-1 - ...   (the area holds the rows of U;
    //      row i of E is in the registers of lane i: lane 32 + i takes its half through v_permlane32_swap)
    constexpr int PPF = 33;
    float* const sAf = reinterpret_cast<float*>(s_ar);
    float* const sBf2 = sAf + 32 * PPF;
    const int tr16 = r16, tkq = kq;
    double et[3][4], ft[3][4];   // E (diagonal: M(i, i) - a_i^2) and R(k) / the sum of the series; tiles (0,0), (0,1), (1,1)
    float rinvt[3][4];
    {
        int jb = 16 * h;
        asm volatile("" : "+v"(jb));
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for(int rr = 0; rr < 16; rr += 2) {
            p0 = __builtin_fma(s_ar[(jb + rr) * PP + i], s_r1[jb + rr], p0);
            p1 = __builtin_fma(s_ar[(jb + rr + 1) * PP + i], s_r1[jb + rr + 1], p1);
        }
        double eh[16];   // E(i, 16 h + jj), diagonal zeroed
#pragma unroll
        for(int jj = 0; jj < 16; ++jj) {
            const double v = lower_y_to_upper(e[jj], e[16 + jj]);
            eh[jj] = (jb + jj == i) ? 0.0 : v;
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
            const double2 u2 = *reinterpret_cast<const double2*>(&s_ar[i * PP + jb + jj]);
            const double2 v = *reinterpret_cast<const double2*>(&nxt[jb + jj]);
            z0 = __builtin_fma(u2.x, v.x, z0);
            z1 = __builtin_fma(u2.y, v.y, z1);
        }
        const double zz = both_halves_sum_d(z0 + z1, lane);
        if(h == 0) s_z1[i] = zz;
    }
    __syncthreads();   // (the rows of U are read)
    EPROF(2)   // z
    // rows of U^T B U (lanes 0..31) -> the area
    if(h == 0) {
#pragma unroll
        for(int j = 0; j < 32; j += 2) { double2 v; v.x = e[j]; v.y = e[j + 1]; *reinterpret_cast<double2*>(&s_ar[i * PP + j]) = v; }
    }
    __syncthreads();
#pragma unroll
    for(int t = 0; t < 3; ++t) {
        const int ti = t >> 1, tj = (t + 1) >> 1;   // (0,0), (0,1), (1,1)
        const double acol = s_rt[16 * tj + tr16];
#pragma unroll
        for(int r = 0; r < 4; ++r) {
            const int row = 16 * ti + 4 * tkq + r, col = 16 * tj + tr16;
            const double arow = s_rt[row];
            const double v = s_ar[row * PP + col];
            const bool isd = ti == tj && row == col;
            et[t][r] = isd ? __builtin_fma(-arow, arow, c + v) : v;   // M(i, i) - a_i^2: 0 up to the rounding of the square root; d_i itself for a negative estimate
            rinvt[t][r] = __builtin_amdgcn_rcpf((float)(arow + acol));
            ft[t][r] = et[t][r] * (double)rinvt[t][r];                 // R(1) = (M - diag(a)^2) o rinv
        }
    }
    __syncthreads();   // (the area is read)
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
    // F = E + sqrt(c) R.  g(D + E) = -(P + F)^-1 = diag(dw) sum_j (F diag(dw))^j = -G (sum_j Xs^j) G with diag(dw) = -G^2, Xs = -G F G SYMMETRIC:
    // the orders 0 .. 5 that k_ensi_members takes as T0 .. T4 (four products) come out of THREE here,
    //      X2 = Xs Xs,   X3 = X2 Xs,   sum = I + Xs + X2 + X3 + (Xs + X2) X3
    // (powers of one symmetric matrix commute: every product is symmetric and only its tiles on and above the diagonal are computed).
    // Orders 0 and 1 entry by entry in double precision as before; running sum: diag(dw) + diag(dw) F diag(dw).
    float xsf[3][4];
#pragma unroll
    for(int t = 0; t < 3; ++t) {
        const int ti = t >> 1, tj = (t + 1) >> 1;
        const double dwc = s_dw[16 * tj + tr16], gc = s_g[16 * tj + tr16];
#pragma unroll
        for(int r = 0; r < 4; ++r) {
            const int row = 16 * ti + 4 * tkq + r, col = 16 * tj + tr16;
            const bool isd = ti == tj && row == col;
            const double dwr = s_dw[row];
            const double fv = (isd ? 0.0 : et[t][r]) + sqc * ft[t][r];
            xsf[t][r] = (float)(-(s_g[row] * fv * gc));
            stage_sym(sAf, t, r, xsf[t][r]);
            ft[t][r] = (isd ? dwr : 0.0) + dwr * fv * dwc;
        }
    }
    __syncthreads();
    // U in operand layout: uop[t][ks] = U(r16 + 16 t, 4 ks + kq)   (the park keeps pair j / 2 of every row side by side: see k_ensi_pair)
    double uop[2][8];
#pragma unroll
    for(int t = 0; t < 2; ++t)
#pragma unroll
        for(int ks = 0; ks < 8; ++ks) {
            const int k = 4 * ks + kq;
            uop[t][ks] = park[(k >> 1) * 128 + (32 + r16 + 16 * t) * 2 + (k & 1)];
        }
    // Y in operand layout: yop[ks][t] = Y(4 ks + kq, 16 t + r16), zero beyond the selection / the valid members; the tail members' columns -> s_yt
    const int mraw = nV & 15, nfull = nV >> 4;
    const bool tail = mraw >= 1 && mraw <= 4 && nfull >= 1;
    const int nt = tail ? nfull : (nV + 15) >> 4;
    float yop[8][4];
    {
        // (unconditional loads from clamped addresses, the value masked afterwards -- bitwise: a select of a loaded value becomes a branch around
        //  the load, one round trip to memory after the other)
        int colc[4];
        unsigned cmask[4];
#pragma unroll
        for(int t = 0; t < 4; ++t) { const int col = 16 * t + r16; cmask[t] = col < nV ? 0xffffffffu : 0u; colc[t] = min(col, nV - 1); }
#pragma unroll
        for(int ks = 0; ks < 8; ++ks) {
            const int row = 4 * ks + kq;
            const unsigned sr = s_sel[row];
            const float* const yrow = a.gY + (long)(row < n ? sr : 0u) * nV;
            const unsigned rmask = row < n ? 0xffffffffu : 0u;
#pragma unroll
            for(int t = 0; t < 4; ++t) {
                if(16 * t < nV) yop[ks][t] = __uint_as_float(__float_as_uint(yrow[colc[t]]) & (rmask & cmask[t]));
                else yop[ks][t] = 0.0f;
            }
        }
    }
    if(tail) {
#pragma unroll
        for(int q = 0; q < 2; ++q) {
            const int row = (lane + 64 * q) >> 2, j = lane & 3;
            const bool on = row < n && j < mraw;
            const unsigned sr = s_sel[row];
            const float v = a.gY[(long)(row < n ? sr : 0u) * nV + min(16 * nfull + j, nV - 1)];
            s_yt[row * 4 + j] = __uint_as_float(__float_as_uint(v) & (on ? 0xffffffffu : 0u));
        }
    }
    {
        const Acc32f x2 = mfma_32_f32<true>(lane, [&](int r, int k) { return sAf[r * PPF + k]; }, [&](int k, int cc) { return sAf[k * PPF + cc]; });    // Xs Xs
#pragma unroll
        for(int t = 0; t < 3; ++t)
#pragma unroll
            for(int r = 0; r < 4; ++r) stage_sym(sBf2, t, r, x2.t[t >> 1][(t + 1) >> 1][r]);
        __syncthreads();
        const Acc32f x3 = mfma_32_f32<true>(lane, [&](int r, int k) { return sBf2[r * PPF + k]; }, [&](int k, int cc) { return sAf[k * PPF + cc]; });   // X2 Xs
        __syncthreads();   // (both operands are read)
#pragma unroll
        for(int t = 0; t < 3; ++t)
#pragma unroll
            for(int r = 0; r < 4; ++r) {
                stage_sym(sAf, t, r, xsf[t][r] + x2.t[t >> 1][(t + 1) >> 1][r]);
                stage_sym(sBf2, t, r, x3.t[t >> 1][(t + 1) >> 1][r]);
                xsf[t][r] = x2.t[t >> 1][(t + 1) >> 1][r] + x3.t[t >> 1][(t + 1) >> 1][r];   // (from here on: X2 + X3 -- a float sum of terms <= 1e-3, like their staging)
            }
        __syncthreads();
        const Acc32f p3 = mfma_32_f32<true>(lane, [&](int r, int k) { return sAf[r * PPF + k]; }, [&](int k, int cc) { return sBf2[k * PPF + cc]; });   // (Xs + X2) X3
#pragma unroll
        for(int t = 0; t < 3; ++t) {
            const int ti = t >> 1, tj = (t + 1) >> 1;
            const double gc = s_g[16 * tj + tr16];
#pragma unroll
            for(int r = 0; r < 4; ++r) {
                const double hs = (double)xsf[t][r] + (double)p3.t[ti][tj][r];
                ft[t][r] -= (s_g[16 * ti + 4 * tkq + r] * gc) * hs;
            }
        }
        __syncthreads();   // (the operands are read)
    }
    // the middle matrix of W_sym -> the area (doubles, row major)
#pragma unroll
    for(int t = 0; t < 3; ++t) {
        const int ti = t >> 1, tj = (t + 1) >> 1;
#pragma unroll
        for(int r = 0; r < 4; ++r) {
            const int row = 16 * ti + 4 * tkq + r, col = 16 * tj + tr16;
            s_ar[row * PP + col] = ft[t][r];
            if(t == 1) s_ar[col * PP + row] = ft[t][r];
        }
    }
    __syncthreads();
    EPROF(1)   // perturbation series (five products)
    // M_W = U Mmid U^T, scaled: M'(i, j) = sD_i M_W(i, j) sD_j   -> the area (stays there through Q = M' Y)
    {
        Acc32 tm;
        tm.t[0][0] = tm.t[0][1] = tm.t[1][0] = tm.t[1][1] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for(int ks = 0; ks < 8; ++ks) {   // U Mmid: A operand out of the registers
            const int k = 4 * ks + kq;
            const double b0 = s_ar[k * PP + r16], b1 = s_ar[k * PP + r16 + 16];
            tm.t[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(uop[0][ks], b0, tm.t[0][0], 0, 0, 0);
            tm.t[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(uop[0][ks], b1, tm.t[0][1], 0, 0, 0);
            tm.t[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(uop[1][ks], b0, tm.t[1][0], 0, 0, 0);
            tm.t[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(uop[1][ks], b1, tm.t[1][1], 0, 0, 0);
        }
        __syncthreads();
        acc32_store_full(tm, lane, s_ar);
        __syncthreads();
        Acc32 mw;
        mw.t[0][0] = mw.t[0][1] = mw.t[1][1] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for(int ks = 0; ks < 8; ++ks) {   // (U Mmid) U^T, symmetric: B operand (k, cc) = U(cc, k) out of the registers
            const int k = 4 * ks + kq;
            const double a0 = s_ar[r16 * PP + k], a1 = s_ar[(r16 + 16) * PP + k];
            mw.t[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, uop[0][ks], mw.t[0][0], 0, 0, 0);
            mw.t[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, uop[1][ks], mw.t[0][1], 0, 0, 0);
            mw.t[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, uop[1][ks], mw.t[1][1], 0, 0, 0);
        }
        __syncthreads();
#pragma unroll
        for(int ti = 0; ti < 2; ++ti)
#pragma unroll
            for(int tj = ti; tj < 2; ++tj)
#pragma unroll
                for(int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + kq + 4 * r, col = 16 * tj + r16;
                    const double v = mw.t[ti][tj][r] * (s_sD1[row] * s_sD1[col]);
                    s_ar[row * PP + col] = v;
                    if(ti != tj) s_ar[col * PP + row] = v;   // (U Mmid U^T is symmetric: the tile below the diagonal is the mirror image)
                }
    }
    __syncthreads();
    EPROF(3)   // M_W (two products)
    // anti-extrapolation tables (oi_ensi.cpp:520-552), as k_ensi_members
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
    // ---- ensemble side: lane = member ------------------------------------------------------------------------------------------
    // ensemble mean: sequential float sum over the valid members in member order (oi_ensi.cpp:447-461)
    const float v0 = s_v0[lane];
    float total = 0.0f;
    {
#pragma unroll
        for(int b = 0; b < 4; ++b) {
            if(16 * b + 16 <= nV) {
#pragma unroll
                for(int j = 0; j < 16; ++j) total += readlane_f(v0, 16 * b + j);
            }
        }
        for(int k = nV & ~15; k < nV; ++k) total += readlane_f(v0, k);
    }
    const float ensMean = total / (float)nV;
    EPROF(4)   // tables, ensemble mean
    const int em = lane;
    float acc = 0.0f;
    const double X = (double)v0 - (double)ensMean;
    auto update = [&](auto ntc) __attribute__((always_inline)) {
        constexpr int NT = decltype(ntc)::value;
        constexpr int NB = 16 * NT;              // first tail member
        const int m = tail ? nV - NB : 0;       // tail members (0..4)
        // Q = M' Y  (32 x 16 NT) on the matrix cores
        v4d qa[2][NT];
#pragma unroll
        for(int ti = 0; ti < 2; ++ti)
#pragma unroll
            for(int tj = 0; tj < NT; ++tj) qa[ti][tj] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for(int ks = 0; ks < 8; ++ks) {
            const double am0 = s_ar[r16 * PP + 4 * ks + kq], am1 = s_ar[(r16 + 16) * PP + 4 * ks + kq];
#pragma unroll
            for(int tj = 0; tj < NT; ++tj) {
                const double bop = yd(yop[ks][tj]);
                qa[0][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(am0, bop, qa[0][tj], 0, 0, 0);
                qa[1][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(am1, bop, qa[1][tj], 0, 0, 0);
            }
        }
        EPROF(5)   // Q
        if(m > 0) {   // Q(:, NB + j): row i by lanes (h, i), columns j0 + h
#pragma unroll 1
            for(int j0 = 0; j0 < m; j0 += 2) {
                const int tc = j0 + h;     // (a column beyond the tail holds zeros)
                double q0 = 0.0, q1 = 0.0;
#pragma unroll 4
                for(int cc = 0; cc < 32; cc += 2) {
                    const double2 mm = *reinterpret_cast<const double2*>(&s_ar[i * PP + cc]);
                    q0 = __builtin_fma(mm.x, (double)s_yt[cc * 4 + tc], q0);
                    q1 = __builtin_fma(mm.y, (double)s_yt[(cc + 1) * 4 + tc], q1);
                }
                s_qt[(j0 >> 1) * 64 + 2 * i + h] = q0 + q1;
            }
        }
        __syncthreads();   // (the tail columns of Q are visible, and nobody reads M' in the area any more)
        // w_k = sum_r sD_r Y(r,k) z_r for member k = lane: per operand lane over its rows, then over kq
        double wk;
        {
            double pw[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for(int ks = 0; ks < 8; ++ks) {
                const double sd = s_sD1[4 * ks + kq], zr = s_z1[4 * ks + kq];
#pragma unroll
                for(int t = 0; t < 4; ++t) pw[t] = __builtin_fma(sd * yd(yop[ks][t]), zr, pw[t]);
            }
            wk = member_reduce(pw[0], pw[1], pw[2], pw[3], lane);
        }
        double wtl[4] = {0.0, 0.0, 0.0, 0.0};   // W'(lane, NB + j) = sum_r Y(r, lane) Q(r, NB + j)
        if(m > 0) {
#pragma unroll
            for(int j = 0; j < 4; ++j) {
                if(j < m) {
                    double pw[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for(int ks = 0; ks < 8; ++ks) {
                        const double qv = s_qt[(j >> 1) * 64 + 2 * (4 * ks + kq) + (j & 1)];
#pragma unroll
                        for(int t = 0; t < 4; ++t) pw[t] = __builtin_fma(yd(yop[ks][t]), qv, pw[t]);
                    }
                    wtl[j] = member_reduce(pw[0], pw[1], pw[2], pw[3], lane);
                }
            }
        }
        // W'(k, NB + j) for the tail lanes: behind the slab exchange area [NB][17] in the area (NB <= 48 with a tail: 816 + 256 <= 1088 doubles)
        double* const s_tail = s_ar + (NB <= 48 ? NB : 48) * 17;
        if(m > 0) {
#pragma unroll
            for(int j = 0; j < 4; ++j) if(j < m) s_tail[j * 64 + lane] = wtl[j];
        }
        const double* const src0 = (lane < NB || !tail) ? s_ar + lane * 17 : s_tail + min(lane - NB, 3) * 64;
        const int sstep = (lane < NB || !tail) ? 0 : 16;
        // W' = Y^T M' Y is symmetric: with NT <= 3 the tiles below the diagonal are not computed -- tile (tk, te), te > tk, stays in its registers until
        // slab te and goes into it transposed (a third of this product's matrix-core time with three tiles a side; with four the six kept tiles
        // do not fit the register budget of three waves)
        constexpr bool SYMW = NT <= 3;
        v4d keep[NT][NT];
#pragma unroll
        for(int tk = 0; tk < NT; ++tk) {
            v4d wt[NT];
#pragma unroll
            for(int te = 0; te < NT; ++te) wt[te] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for(int ks = 0; ks < 8; ++ks) {
                const double aop = yd(yop[ks][tk]);   // Y(i = 4 ks + kq, k = 16 tk + r16)
#pragma unroll
                for(int te = (SYMW ? tk : 0); te < NT; ++te) wt[te] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, qa[ks >> 2][te][ks & 3], wt[te], 0, 0, 0);
            }
            if(tk > 0) __syncthreads();   // (the previous slab has been read; before the first one: the barrier above)
#pragma unroll
            for(int te = 0; te < NT; ++te) {
                if(SYMW && te < tk) {
#pragma unroll
                    for(int r = 0; r < 4; ++r) s_ar[(16 * te + kq + 4 * r) * 17 + r16] = keep[te][tk][r];   // W'(k = 16 tk + r16, e = 16 te + kq + 4 r) = W'(e, k)
                }
                else {
#pragma unroll
                    for(int r = 0; r < 4; ++r) s_ar[(16 * te + r16) * 17 + kq + 4 * r] = wt[te][r];   // W'(k = 16 tk + kq + 4 r, e = 16 te + r16)
                    if(SYMW && te > tk) keep[tk][te] = wt[te];
                }
            }
            __syncthreads();
            const double* const src = src0 + sstep * tk;
#pragma unroll
            for(int k0 = 0; k0 < 16; k0 += 8) {   // (eight values at a time: all sixteen in flight do not fit the register budget of three waves)
                double wv[8];
#pragma unroll
                for(int kl = 0; kl < 8; ++kl) wv[kl] = src[k0 + kl];
#pragma unroll
                for(int kl = 0; kl < 8; ++kl) {
                    const int k = 16 * tk + k0 + kl;
                    if(k < nV) {
                        const double wke = ((k == lane ? 1.0 : 0.0) + wv[kl]) + readlane_d(wk, k);
                        acc = (float)((double)acc + readlane_d(X, k) * wke);
                    }
                }
                asm volatile("" ::: "memory");
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
    float currIncrement = acc;
    if(!a.allow_extrap && em < nV) {
        const int li = em % n, lk = em / n;
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
    if(em < nV) a.out[(long)cell_l * E + ensi_member(a, em)] = ensMean + currIncrement;   // :553
#ifdef GPP_ENSI_PROFILE
    EPROF(8)   // clamp, store
    if(lane == 0 && a.counters) for(int k = 0; k < 12; ++k) atomicAdd(&a.counters[80 + (blockIdx.x & 1023) * 32 + 16 + k], prof[k]);
#endif
}
