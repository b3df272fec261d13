// Shared by oi.hip and ensi.hip: observation index, Barnes device functions, the per-tile candidate scan.
#pragma once
#include "common.h"

#pragma clang fp contract(off)
using gpp::DevBuf;

// -------------------------------------------------------------------------------------------
// observation index (host build, HBM resident)
// -------------------------------------------------------------------------------------------
struct gpp_obs_index {
    int S = 0;
    int axis_a = 0, axis_b = 1;
    float amin = 0, bmin = 0, inv_s = 0;
    int nbx = 1, nby = 1;
    DevBuf<int> d_bin_start;   // [nbx*nby+1]
    DevBuf<int> d_pos;         // orig -> sorted position
    DevBuf<float4> d_sgeo;     // sorted: x,y,z,elev
    DevBuf<float2> d_smeta;    // sorted: laf, orig (int bits)
    DevBuf<float4> d_ogeo;     // original order: x,y,z,elev
    DevBuf<float> d_olaf;      // original order: laf
};

gpp_obs_index* gpp_build_obs_index(gpp_points* pts);   // oi.hip
struct gpp_field {   // spatially varying structure parameters resident in HBM (gpp_field_create)
    gpp_points* grid = nullptr;
    int n = 0, kind = 0;
    float min_rho = 0;
    std::vector<float> h, v, w, R;
    gpp::DevBuf<float> d_h, d_v, d_w, d_R;
    bool uniform = false;   // every point carries the same (valid) h, v and w: the structure is a scalar one (the reference's own
                            // "var len scale" benchmark row, tests/benchmark.py:66,294, passes h * ones)
};
// log2 of the width (in grid columns) of the 64-cell tile of a 2-D grid: the most square tile in metres (8 x 8 cells for
// an isotropic grid, 2 x 32 or 32 x 2 for strongly anisotropic ones) -- the cells of a tile should select the same observations
int gpp_tile_wshift(gpp_points* grid);   // oi.hip
struct DevStructure;
DevStructure gpp_resolve_structure(const gpp_structure* s);   // oi.hip: validation + localization distance
void gpp_bind_field(DevStructure& d, const gpp_structure* s, gpp_points* bgrid, gpp_points* points, gpp::DevBuf<int>& cbuf, gpp::DevBuf<int>& obuf);

// -------------------------------------------------------------------------------------------
// device helpers
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ bool d_valid(float v) { return !isnan(v) && !isinf(v); }

// src/api/structure.cpp:26-34
// exp(x) for x <= 0, accurate to < 1 ulp(double) (the result is only used rounded to float32, where the reference's
// libm exp gives the same value except when exp(x) lies within ~1e-16 relative of a float32 rounding boundary).
// Cody-Waite reduction x = k ln2 + r, |r| <= ln2/2, degree-13 Taylor polynomial in Horner form, exact 2^k scaling.
// one Horner step p*r + c as a single v_fma_f64 (the compiler otherwise emits v_mov + v_fmac for every coefficient)
__device__ __forceinline__ double d_fma_step(double p, double r, double c) {
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "v"(c));
    return o;
}
// Table form (round 3): x = k ln2/128 + r, |r| <= ln2/256, exp(x) = 2^(k >> 7) * T[k & 127] * (1 + r + ... + r^5/120) with
// T[j] = 2^(j/128) correctly rounded, held in LDS (1 KB per workgroup, d_exptab_fill() + a barrier at the top of every kernel that gets here:
// k_oi_union, k_oi, the EnSI scan kernels).  7 double operations behind the reduction instead of 14; < 1 ulp(double) (0.999 measured
// on 3e8 arguments -v^2/2, v float32, against expl; no float32 result different from glibc's among them: tools/ubench/exp_table.c).
static __device__ const double c_exp2_tab[128] = {
    0x1.0000000000000p+0, 0x1.0163da9fb3335p+0, 0x1.02c9a3e778061p+0, 0x1.04315e86e7f85p+0,
    0x1.059b0d3158574p+0, 0x1.0706b29ddf6dep+0, 0x1.0874518759bc8p+0, 0x1.09e3ecac6f383p+0,
    0x1.0b5586cf9890fp+0, 0x1.0cc922b7247f7p+0, 0x1.0e3ec32d3d1a2p+0, 0x1.0fb66affed31bp+0,
    0x1.11301d0125b51p+0, 0x1.12abdc06c31ccp+0, 0x1.1429aaea92de0p+0, 0x1.15a98c8a58e51p+0,
    0x1.172b83c7d517bp+0, 0x1.18af9388c8deap+0, 0x1.1a35beb6fcb75p+0, 0x1.1bbe084045cd4p+0,
    0x1.1d4873168b9aap+0, 0x1.1ed5022fcd91dp+0, 0x1.2063b88628cd6p+0, 0x1.21f49917ddc96p+0,
    0x1.2387a6e756238p+0, 0x1.251ce4fb2a63fp+0, 0x1.26b4565e27cddp+0, 0x1.284dfe1f56381p+0,
    0x1.29e9df51fdee1p+0, 0x1.2b87fd0dad990p+0, 0x1.2d285a6e4030bp+0, 0x1.2ecafa93e2f56p+0,
    0x1.306fe0a31b715p+0, 0x1.32170fc4cd831p+0, 0x1.33c08b26416ffp+0, 0x1.356c55f929ff1p+0,
    0x1.371a7373aa9cbp+0, 0x1.38cae6d05d866p+0, 0x1.3a7db34e59ff7p+0, 0x1.3c32dc313a8e5p+0,
    0x1.3dea64c123422p+0, 0x1.3fa4504ac801cp+0, 0x1.4160a21f72e2ap+0, 0x1.431f5d950a897p+0,
    0x1.44e086061892dp+0, 0x1.46a41ed1d0057p+0, 0x1.486a2b5c13cd0p+0, 0x1.4a32af0d7d3dep+0,
    0x1.4bfdad5362a27p+0, 0x1.4dcb299fddd0dp+0, 0x1.4f9b2769d2ca7p+0, 0x1.516daa2cf6642p+0,
    0x1.5342b569d4f82p+0, 0x1.551a4ca5d920fp+0, 0x1.56f4736b527dap+0, 0x1.58d12d497c7fdp+0,
    0x1.5ab07dd485429p+0, 0x1.5c9268a5946b7p+0, 0x1.5e76f15ad2148p+0, 0x1.605e1b976dc09p+0,
    0x1.6247eb03a5585p+0, 0x1.6434634ccc320p+0, 0x1.6623882552225p+0, 0x1.68155d44ca973p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6c012750bdabfp+0, 0x1.6dfb23c651a2fp+0, 0x1.6ff7df9519484p+0,
    0x1.71f75e8ec5f74p+0, 0x1.73f9a48a58174p+0, 0x1.75feb564267c9p+0, 0x1.780694fde5d3fp+0,
    0x1.7a11473eb0187p+0, 0x1.7c1ed0130c132p+0, 0x1.7e2f336cf4e62p+0, 0x1.80427543e1a12p+0,
    0x1.82589994cce13p+0, 0x1.8471a4623c7adp+0, 0x1.868d99b4492edp+0, 0x1.88ac7d98a6699p+0,
    0x1.8ace5422aa0dbp+0, 0x1.8cf3216b5448cp+0, 0x1.8f1ae99157736p+0, 0x1.9145b0b91ffc6p+0,
    0x1.93737b0cdc5e5p+0, 0x1.95a44cbc8520fp+0, 0x1.97d829fde4e50p+0, 0x1.9a0f170ca07bap+0,
    0x1.9c49182a3f090p+0, 0x1.9e86319e32323p+0, 0x1.a0c667b5de565p+0, 0x1.a309bec4a2d33p+0,
    0x1.a5503b23e255dp+0, 0x1.a799e1330b358p+0, 0x1.a9e6b5579fdbfp+0, 0x1.ac36bbfd3f37ap+0,
    0x1.ae89f995ad3adp+0, 0x1.b0e07298db666p+0, 0x1.b33a2b84f15fbp+0, 0x1.b59728de5593ap+0,
    0x1.b7f76f2fb5e47p+0, 0x1.ba5b030a1064ap+0, 0x1.bcc1e904bc1d2p+0, 0x1.bf2c25bd71e09p+0,
    0x1.c199bdd85529cp+0, 0x1.c40ab5fffd07ap+0, 0x1.c67f12e57d14bp+0, 0x1.c8f6d9406e7b5p+0,
    0x1.cb720dcef9069p+0, 0x1.cdf0b555dc3fap+0, 0x1.d072d4a07897cp+0, 0x1.d2f87080d89f2p+0,
    0x1.d5818dcfba487p+0, 0x1.d80e316c98398p+0, 0x1.da9e603db3285p+0, 0x1.dd321f301b460p+0,
    0x1.dfc97337b9b5fp+0, 0x1.e264614f5a129p+0, 0x1.e502ee78b3ff6p+0, 0x1.e7a51fbc74c83p+0,
    0x1.ea4afa2a490dap+0, 0x1.ecf482d8e67f1p+0, 0x1.efa1bee615a27p+0, 0x1.f252b376bba97p+0,
    0x1.f50765b6e4540p+0, 0x1.f7bfdad9cbe14p+0, 0x1.fa7c1819e90d8p+0, 0x1.fd3c22b8f71f1p+0,
};
__device__ __forceinline__ double* d_exptab() {
    __shared__ double s_exp2_tab[128];
    return s_exp2_tab;
}
template <int NT = 128>
__device__ __forceinline__ void d_exptab_fill() {   // every thread of a workgroup of NT threads; a __syncthreads() before the first use
    if constexpr(NT >= 128) { if(threadIdx.x < 128) d_exptab()[threadIdx.x] = c_exp2_tab[threadIdx.x]; }
    else { for(int i = threadIdx.x; i < 128; i += NT) d_exptab()[i] = c_exp2_tab[i]; }
}
// entry e of a lower triangle stored row by row -> (row i) << 8 | (column p <= i), e = i (i + 1) / 2 + p < 512 (the half-wave
// solves of k_oi / k_oi_pairs walk the triangle 32 entries at a time; the closed form costs a square root and two corrections per entry)
__device__ __forceinline__ unsigned short* d_tritab() {
    __shared__ unsigned short s_tri_tab[512];
    return s_tri_tab;
}
__device__ __forceinline__ void d_tritab_fill() {   // every thread of a workgroup of 256 threads; a __syncthreads() before the first use
    for(int k = 0; k < 2; ++k) {
        const int e = 2 * (int)threadIdx.x + k;
        int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        if(i * (i + 1) / 2 > e) i--;
        if((i + 1) * (i + 2) / 2 <= e) i++;
        d_tritab()[e] = (unsigned short)((i << 8) | (e - i * (i + 1) / 2));
    }
}
__device__ __forceinline__ double d_exp_core(double x) {   // -110 <= x <= 0 (no range check)
    const double kf = rint(x * 184.66496523378730813);                  // 128 / ln2
    double r = __builtin_fma(kf, -6.93147180369123816490e-01 / 128.0, x);
    r = __builtin_fma(kf, -1.90821492927058770002e-10 / 128.0, r);
    const int k = (int)kf;
    const double t = d_exptab()[k & 127];
    double p = 8.33333333333333333333e-03;
    p = d_fma_step(p, r, 4.16666666666666666667e-02);
    p = d_fma_step(p, r, 1.66666666666666666667e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;
    p = __builtin_fma(t, p, t);
    return ldexp(p, k >> 7);
}
__device__ __forceinline__ double d_exp_nonpos(double x) {
    if(x < -110.0) return 0.0;   // (float)exp(x) == 0 below -103.98
    const double kf = rint(x * 1.4426950408889634074);
    double r = __builtin_fma(kf, -6.93147180369123816490e-01, x);
    r = __builtin_fma(kf, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;            // 1/13!
    p = __builtin_fma(p, r, 2.08767569878680989792e-09);
    p = __builtin_fma(p, r, 2.50521083854417187751e-08);
    p = __builtin_fma(p, r, 2.75573192239858906526e-07);
    p = __builtin_fma(p, r, 2.75573192239858906526e-06);
    p = __builtin_fma(p, r, 2.48015873015873015873e-05);
    p = __builtin_fma(p, r, 1.98412698412698412698e-04);
    p = __builtin_fma(p, r, 1.38888888888888888889e-03);
    p = __builtin_fma(p, r, 8.33333333333333333333e-03);
    p = __builtin_fma(p, r, 4.16666666666666666667e-02);
    p = __builtin_fma(p, r, 1.66666666666666666667e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}
// dist / length (float32, correctly rounded) through the double reciprocal rlen = 1.0 / (double)length: the double
// product is within 2^-52 of the exact quotient, and the quotient of two float32 numbers is never closer than 2^-49
// (relative) to a float32 rounding boundary, so rounding the product gives the correctly rounded quotient (a result in
// the subnormal range may differ in the last bit; rho is 1 there either way).  6 instead of 14 instructions.
__device__ __forceinline__ float d_div_by(float dist, double rlen) { return (float)((double)dist * rlen); }
// d_barnes_rho for a valid, non-zero length, without divergent branches.  Same values for every valid dist.  A NaN dist gives exp(-112.5) = 0
// (fminf drops the NaN) -- what the reference returns for !is_valid(dist) (structure.cpp:28-29) -- and not NaN: callers that need to
// know about a missing coordinate test d_valid() themselves (d_barnes_corr_flat does for the elevation / laf factors).  d_exp_core is
// called down to -112.5 here (tests/test_exp_table.py checks [-112.5, -110] too).
__device__ __forceinline__ float d_barnes_rho_flat(float dist, double rlen) {
    // (|v| cut at 15: exp(-112.5) = 1e-49 is 0 in float32 like everything below exp(-103.98) -- one float32 minimum instead of a double
    //  maximum in front of the exp and a compare + select behind it)
    const float v = fminf(fabsf(d_div_by(dist, rlen)), 15.0f);
    const double e = -0.5 * (double)v * (double)v;
    return (float)d_exp_core(e);
}
__device__ __forceinline__ float d_barnes_rho(float dist, float length) {
    if(!d_valid(length) || length == 0) return 1.0f;
    if(!d_valid(dist)) return 0.0f;
    float v = dist / length;
    double e = -0.5 * (double)v * (double)v;
    return (float)d_exp_nonpos(e);
}
// The same two functions through the table form of exp (d_exp_core: the workgroup must have filled the table, d_exptab_fill): the
// generic structure functions of k_oi / k_oi_pairs.  The degree-13 Horner form above and the library exp of d_expf_cr below keep some
// thirty double constants in vector registers for the whole kernel (gfx950 has no 64-bit literals) -- what pushed the 62-row pivoted-LU
// forms of k_oi into the accumulation registers; the table form has five.  Same accuracy class (< 1 ulp of the double, rounded to float32).
__device__ __forceinline__ float d_barnes_rho_tab(float dist, float length) {
    if(!d_valid(length) || length == 0) return 1.0f;
    if(!d_valid(dist)) return 0.0f;
    const float v = dist / length;
    const double e = -0.5 * (double)v * (double)v;
    const float r = (float)d_exp_core(fmax(e, -110.0));
    return e < -110.0 ? 0.0f : r;
}
__device__ __forceinline__ float d_expf_tab(float x) {   // (float)exp((double)x) for finite x
    const double xd = (double)x;
    const float r = (float)d_exp_core(fmin(fmax(xd, -110.0), 90.0));
    return xd < -110.0 ? 0.0f : (xd > 90.0 ? INFINITY : r);
}
// Correctly rounded float32 square root for x = 0 or x >= 2^-96 (NaN gives NaN): v_sqrt_f32 (within one ulp) and the two residual
// tests of the compiler's own expansion of sqrtf, without its rescaling of small arguments (below 2^-96 the residuals underflow) and
// its zero / infinity class test -- 9 instead of 16 instructions; the kernels take ~46 chord lengths per tile.  Every float from 2^-96
// up gives the very bits of sqrtf (tools/ubench/sqrt_cr.hip walks all of them on the GPU); a squared distance below 2^-96 (points
// closer than 4e-15 of their coordinate unit) may come out one ulp off -- rho is 1 there either way, like the quotient of d_div_by.
__device__ __forceinline__ float d_sqrt_cr(const float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float r = (0.0f >= rm) ? sm : s;
    r = (0.0f < rp) ? sp : r;
    return r;
}
// v_sqrt_f32 as it is (within one ulp): for the quantities that only bound the work (tile radii, ring limits, row extents -- all padded)
__device__ __forceinline__ float d_sqrt_raw(const float x) { return __builtin_amdgcn_sqrtf(x); }
// src/api/kdtree.cpp:192-194 (float32, no contraction, correctly rounded sqrt)
__device__ __forceinline__ float d_chord(float x0, float y0, float z0, float x1, float y1, float z1) {
    float dx = x0 - x1, dy = y0 - y1, dz = z0 - z1;
    float s = dx * dx + dy * dy;
    s = s + dz * dz;
    return d_sqrt_cr(s);
}
// src/api/structure.cpp:215-228 (scalar Barnes)
__device__ __forceinline__ float d_barnes_corr(float x1, float y1, float z1, float e1, float l1,
                                               float x2, float y2, float z2, float e2, float l2,
                                               float h, float v, float w, float R) {
    float hdist = d_chord(x1, y1, z1, x2, y2, z2);
    if(hdist > R) return 0.0f;
    float rho = d_barnes_rho(hdist, h);
    if(d_valid(e1) && d_valid(e2)) rho *= d_barnes_rho(e1 - e2, v);
    if(d_valid(l1) && d_valid(l2)) rho *= d_barnes_rho(l1 - l2, w);
    return rho;
}
// ---- generic structure functions (src/api/structure.cpp:26-86,90-138,287-944), scalar forms -------------------
#define SK_BARNES 0
#define SK_CRESSMAN 1
#define SK_SOAR 2
#define SK_TOAR 3
#define SK_POWERLAW 4
#define SK_LINEAR 5
struct DevStructure {
    int kh, kv, kw;       // kernel of the horizontal / vertical / land-area-fraction factor (MultipleStructure mixes them)
    float h, v, w;        // scales
    float R;              // localization distance (of the horizontal structure)
    int cv;               // CrossValidation wrapper active
    float cv_dist;
    // spatially varying form (structure.cpp:168-214): h, v, w, R looked up at the FIRST point of corr(p1, p2) by nearest
    // neighbour in the field's grid; NULL for the scalar forms
    const float *fh, *fv, *fw, *fR;
    const int* cell_idx;  // background point -> field index (NULL: identity)
    const int* obs_idx;   // observation (original order) -> field index
};
// this point's parameters of a spatially varying structure
__device__ __forceinline__ void d_structure_at(DevStructure& s, const int fi) {
    s.h = s.fh[fi]; s.v = s.fv[fi]; s.w = s.fw[fi]; s.R = s.fR[fi];
}
// exp(x) rounded to float32 for any sign of x (soar / toar use exp(float), structure.cpp:53,63)
__device__ __forceinline__ float d_expf_cr(float x) { return (float)exp((double)x); }
template <bool TAB>
__device__ __forceinline__ float d_rho_x(const int kind, const float dist, const float length) {
    if(kind == SK_BARNES) return TAB ? d_barnes_rho_tab(dist, length) : d_barnes_rho(dist, length);
    if(kind == SK_LINEAR) {                                     // structure.cpp:76-86 (length = min_corr)
        if(!d_valid(length) || length < 0) return 1.0f;
        if(!d_valid(dist)) return 0.0f;
        float absdiff = fabsf(dist);
        if(absdiff > 1) absdiff = 1;
        return 1.0f - (1.0f - length) * absdiff;
    }
    if(!d_valid(length) || length == 0) return 1.0f;
    if(!d_valid(dist)) return 0.0f;
    if(kind == SK_CRESSMAN) {                                   // :35-44
        if(dist >= length) return 0.0f;
        return (length * length - dist * dist) / (length * length + dist * dist);
    }
    const float v = dist / length;
    if(kind == SK_SOAR) return (1.0f + v) * (TAB ? d_expf_tab(-v) : d_expf_cr(-v));      // :46-54
    if(kind == SK_TOAR) return (1.0f + v + (v * v) / 3.0f) * (TAB ? d_expf_tab(-v) : d_expf_cr(-v));   // :56-64
    return (float)(1.0 / (1.0 + 0.5 * (double)v * (double)v));  // powerlaw :66-74
}
__device__ __forceinline__ float d_rho(const int kind, const float dist, const float length) { return d_rho_x<false>(kind, dist, length); }
// corr(p1, p2) / corr_background(p1, p2) of a scalar (possibly Multiple / CrossValidation-wrapped) structure.
// PLAIN = true is the compile-time specialisation for an unwrapped Barnes structure (the headline configuration).
template <bool PLAIN>
__device__ __forceinline__ float d_rho_t(const int kind, const float dist, const float length) {
    return PLAIN ? d_barnes_rho(dist, length) : d_rho(kind, dist, length);
}
template <bool PLAIN, bool TAB = false>
__device__ __forceinline__ float d_corr_t(const DevStructure& s, float x1, float y1, float z1, float e1, float l1,
                                          float x2, float y2, float z2, float e2, float l2, const bool background);
template <bool TAB>
__device__ __forceinline__ float d_corr_x(const DevStructure& s, float x1, float y1, float z1, float e1, float l1,
                                          float x2, float y2, float z2, float e2, float l2, const bool background) {
    const float hdist = d_chord(x1, y1, z1, x2, y2, z2);
    if(background && s.cv && hdist <= s.cv_dist) return 0.0f;   // structure.cpp:918-925
    if(s.kh != SK_CRESSMAN && hdist > s.R) return 0.0f;         // :216-217 (Cressman has no cut, :300-312)
    float rho = d_rho_x<TAB>(s.kh, hdist, s.h);
    if(d_valid(e1) && d_valid(e2)) rho *= d_rho_x<TAB>(s.kv, e1 - e2, s.v);
    if(d_valid(l1) && d_valid(l2)) rho *= d_rho_x<TAB>(s.kw, l1 - l2, s.w);
    return rho;
}
__device__ __forceinline__ float d_corr(const DevStructure& s, float x1, float y1, float z1, float e1, float l1,
                                        float x2, float y2, float z2, float e2, float l2, const bool background) {
    return d_corr_x<false>(s, x1, y1, z1, e1, l1, x2, y2, z2, e2, l2, background);
}

// mask of the lanes where p holds: the compare's own scalar mask (HIP's __ballot goes through a 0 / 1 vector and a second compare:
// two vector instructions per ballot, and a tile of k_oi_union asks for more than a hundred)
__device__ __forceinline__ unsigned long long wave_ballot(const bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// Wave-wide min / max on the VALU cross-lane path (DPP row shifts + row broadcasts, then lane 63 holds the result): no LDS round trips.
// Must be called by all 64 lanes; no NaN among the inputs.  The DPP operand sits on the min / max itself and the value is reduced in
// place -- a lane whose shifted source does not exist is switched off by the hardware and keeps its value (bound_ctrl off) -- so a step
// is ONE instruction behind the two wait states a DPP read of a fresh VALU result needs.  Written with update_dpp + fminf the compiler
// made four of each step (identity move, DPP move, canonicalisation, min): 28 instructions per reduction, 13 of them per tile.
#define GPP_DPP_RED(OP)                                                                                   \
    asm("s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                            \
        "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                            \
        "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                            \
        "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                            \
        "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                         \
        "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                         \
        "s_nop 1" : "+v"(v))
__device__ __forceinline__ float wave_min(float v) {
    GPP_DPP_RED("v_min_f32_dpp");
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    GPP_DPP_RED("v_max_f32_dpp");
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#undef GPP_DPP_RED

// maximum of a double over the 64 lanes (DPP moves of its two halves; every lane must hold a value >= -1.0, the identity)
__device__ __forceinline__ double wave_max_d(double v) {
#define GPP_DPP_MAXD(ctrl, rmask)                                                                                        \
    {                                                                                                                   \
        const double id_ = -1.0;                                                                                        \
        const int lo_ = __builtin_amdgcn_update_dpp(__double2loint(id_), __double2loint(v), ctrl, rmask, 0xf, false);    \
        const int hi_ = __builtin_amdgcn_update_dpp(__double2hiint(id_), __double2hiint(v), ctrl, rmask, 0xf, false);    \
        v = fmax(v, __hiloint2double(hi_, lo_));                                                                        \
    }
    GPP_DPP_MAXD(0x111, 0xf) GPP_DPP_MAXD(0x112, 0xf) GPP_DPP_MAXD(0x114, 0xf) GPP_DPP_MAXD(0x118, 0xf)
    GPP_DPP_MAXD(0x142, 0xa) GPP_DPP_MAXD(0x143, 0xc)
#undef GPP_DPP_MAXD
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// inclusive prefix sum over the 64 lanes (same DPP path)
__device__ __forceinline__ int wave_scan_add(int x) {
#define GPP_DPP_ADD(ctrl, rmask) x += __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false);
    GPP_DPP_ADD(0x111, 0xf) GPP_DPP_ADD(0x112, 0xf) GPP_DPP_ADD(0x114, 0xf) GPP_DPP_ADD(0x118, 0xf)
    GPP_DPP_ADD(0x142, 0xa) GPP_DPP_ADD(0x143, 0xc)
#undef GPP_DPP_ADD
    return x;
}

// The n largest of ncand unique 64-bit keys (rho bits << 32 | ~observation index: rho descending, ties -> lower index), for a workgroup of
// 256 threads: an 8-bit radix select instead of sorting every candidate (2 700 candidates for 50 kept ones: 85 of 203 ms of the large-n
// EnSI case went into the full bitonic sort).  A pass counts the keys under the prefix found so far by their next digit; the digit that
// holds the n-th largest key extends the prefix; as soon as every key under the prefix belongs to the selection the remaining bits do not
// matter.  The selection lands in keys[0 .. n) in no particular order (through `out`, n keys of HBM scratch); `hist` is 259 ints of LDS,
// `counter` one more.  All 256 threads call it; keys[] must not be touched by anyone else meanwhile.
__device__ __forceinline__ void block_select_largest(unsigned long long* keys, const int ncand, const int n, unsigned long long* out,
                                                     int* hist, int* counter, const int tid) {
    unsigned long long prefix = 0ull;
    int need = n, ls = 56;
    for(int shift = 56; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        const unsigned long long mask = shift == 56 ? 0ull : (~0ull << (shift + 8));
        for(int i = tid; i < ncand; i += 256) {
            const unsigned long long k = keys[i];
            if((k & mask) == prefix) atomicAdd(&hist[(int)((k >> shift) & 255ull)], 1);
        }
        __syncthreads();
        for(int off = 1; off < 256; off <<= 1) {   // hist[b] <- number of such keys with a digit >= b
            const int v = (tid + off < 256) ? hist[tid + off] : 0;
            __syncthreads();
            hist[tid] += v;
            __syncthreads();
        }
        const int S = hist[tid], Sn = tid < 255 ? hist[tid + 1] : 0;
        __syncthreads();
        if(S >= need && Sn < need) { hist[256] = tid; hist[257] = Sn; hist[258] = S - Sn; }
        __syncthreads();
        const int d = hist[256], above = hist[257], here = hist[258];
        __syncthreads();
        need -= above;
        prefix |= (unsigned long long)d << shift;
        ls = shift;
        if(here == need) break;   // (at the last pass at the latest: one key per value)
    }
    if(tid == 0) *counter = 0;
    __syncthreads();
    for(int i = tid; i < ncand; i += 256) {
        const unsigned long long k = keys[i];
        if((k >> ls) >= (prefix >> ls)) out[atomicAdd(counter, 1)] = k;
    }
    __threadfence_block();
    __syncthreads();
    for(int i = tid; i < n; i += 256) keys[i] = out[i];
    __syncthreads();
}

// Per-call observation pack: validity (oi.cpp:252), variance ratio (oi.cpp:192-195).
static __global__ void k_pack_obs(int S, const float4* __restrict__ sgeo, const int* __restrict__ pos, const float* __restrict__ olaf,
                           const float* __restrict__ obs, const float* __restrict__ obs_var, const float* __restrict__ pbg,
                           const float* __restrict__ bvp, int need_pbg, float4* __restrict__ pgeo, float4* __restrict__ oaux,
                           float4* __restrict__ saux = nullptr,     // saux: the same record at the SORTED position (optional)
                           unsigned long long* __restrict__ zero = nullptr, int nzero = 0) {   // the call's status block to clear (instead of a memset)
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    for(int i = o; i < nzero; i += gridDim.x * blockDim.x) zero[i] = 0ull;
    if(o >= S) return;
    float ob = obs[o], pb = pbg ? pbg[o] : 0.0f;
    float bv = bvp ? bvp[o] : 1.0f;
    float ratio = obs_var[o] / bv;
    oaux[o] = make_float4(olaf[o], ob, pb, ratio);
    int p = pos[o];
    if(saux) saux[p] = make_float4(olaf[o], ob, pb, ratio);
    float4 g = sgeo[p];
    bool ok = d_valid(ob) && (!need_pbg || d_valid(pb));
    if(!ok) g.x = NAN;   // fails the box test of the radius query -> never a candidate
    pgeo[p] = g;
}


// d_barnes_corr as straight-line code: the three factors are independent exp chains, and without the branches of d_barnes_rho
// around them the compiler interleaves them (same values: d_barnes_rho_flat; a disabled factor is skipped by a uniform branch, an
// invalid elevation / laf on either side deselects its factor).  The reciprocals are loop invariants of the calling kernels.
__device__ __forceinline__ float d_barnes_corr_flat(float x1, float y1, float z1, float e1, float l1, float x2, float y2, float z2, float e2,
                                                    float l2, float h, float v, float w, float R) {
    const bool hh = d_valid(h) && h != 0.0f, hv = d_valid(v) && v != 0.0f, hw = d_valid(w) && w != 0.0f;
    const float hdist = d_chord(x1, y1, z1, x2, y2, z2);
    float rho = 1.0f;
    if(hh) rho = d_barnes_rho_flat(hdist, 1.0 / (double)h);
    if(hv) {
        const float f = d_barnes_rho_flat(e1 - e2, 1.0 / (double)v);
        rho = (d_valid(e1) && d_valid(e2)) ? rho * f : rho;
    }
    if(hw) {
        const float f = d_barnes_rho_flat(l1 - l2, 1.0 / (double)w);
        rho = (d_valid(l1) && d_valid(l2)) ? rho * f : rho;
    }
    return hdist > R ? 0.0f : rho;
}
template <bool PLAIN, bool TAB>
__device__ __forceinline__ float d_corr_t(const DevStructure& s, float x1, float y1, float z1, float e1, float l1,
                                          float x2, float y2, float z2, float e2, float l2, const bool background) {
    if(PLAIN) return d_barnes_corr_flat(x1, y1, z1, e1, l1, x2, y2, z2, e2, l2, s.h, s.v, s.w, s.R);
    return d_corr_x<TAB>(s, x1, y1, z1, e1, l1, x2, y2, z2, e2, l2, background);
}

// What the candidate scan needs: the bin-sorted observation block and the structure function
struct ScanArgs {
    const float4* pgeo;      // sorted, per call (x = NaN when the observation is unusable)
    const float2* smeta;     // sorted: laf, orig
    const int* bin_start;
    int axis_a, axis_b, nbx, nby;
    float amin, bmin, inv_s;
    DevStructure st;
    int K;                   // list capacity in use: min(max_points, N) (N if max_points == 0)
    int max_points;
    int q0;                  // half-width (in bins) of the phase-1 square of the candidate scan
    float ring_r0, ring_dr;  // k_oi_union: upper bound of the bulk disc / ring width of the phase-1 visiting order (projected distance from the tile centre)
    unsigned long long* scan_stats;   // optional: [0] candidates iterated, [1] wave-level survivor-branch executions
};

// Per-tile candidate scan (src/api/oi.cpp:229-273 for 64 cells at once).  On return every lane holds `cnt` 64-bit
// keys (rho bits << 32 | ~observation index) in keys[0..cnt)[lane]: the observations the reference would keep for
// that cell (all usable ones if there are at most max_points, otherwise the max_points with the largest rho,
// ties -> lower observation index).  Must be called by the whole wave with at least one active lane.
//
// Order of the walk (it only affects the amount of work, never the result): phase 1 visits the (2q+1)^2 bins around
// the tile first -- they almost always contain the final selection, so the pruning thresholds are tight before
// anything far away is looked at; phase 2 walks the bin rows centre-out with the x-extent and the stop test taken
// from the largest threshold in the wave, skipping the bins phase 1 already did.
template <int N, bool WANT_TRUNC = false, bool PLAIN = false, bool GROUP_MINIMA = false, bool TAB = false>   // TAB: generic kernels through d_exp_core (table filled by the caller)
__device__ __forceinline__ int scan_tile(const ScanArgs& a, const DevStructure& st, const bool active, const float gx, const float gy, const float gz,
                                         const float ge, const float gl, unsigned long long (*keys)[64], const int lane, bool& overflow,
                                         bool& truncated) {
    // truncated: more usable observations than max_points existed, i.e. the reference took its sorted branch
    // (oi.cpp:262-273); only the EnSI anti-extrapolation quirk depends on it (oi_ensi.cpp:523-524)
    int cnt = 0;
    int nins = 0;   // GPP_SCAN_STATS only: wave-level insertion events
    overflow = false;
    truncated = false;
    const float R = st.R;
    const int K = a.K;
    // scalar Barnes structure (PLAIN): which factors are enabled, and the reciprocals of their scales
    const bool p_hh = d_valid(st.h) && st.h != 0.0f, p_hv = d_valid(st.v) && st.v != 0.0f, p_hw = d_valid(st.w) && st.w != 0.0f;
    const double p_rh = p_hh ? 1.0 / (double)st.h : 0.0, p_rv = p_hv ? 1.0 / (double)st.v : 0.0, p_rw = p_hw ? 1.0 / (double)st.w : 0.0;
    const bool bounded = a.max_points > 0 && a.max_points <= N;
    const float h2 = st.h * st.h;
    // h / v and h / w for the combined pruning test (0: factor disabled, or a ratio float32 cannot hold)
    const float p_sv = (p_hh && p_hv && d_valid(st.h / st.v)) ? st.h / st.v : 0.0f, p_sw = (p_hh && p_hw && d_valid(st.h / st.w)) ? st.h / st.w : 0.0f;
    const bool prune = bounded && (PLAIN || st.kh == SK_BARNES);   // rho <= rho_h(d) with the closed-form inverse of the Barnes kernel
    float pa = a.axis_a == 0 ? gx : (a.axis_a == 1 ? gy : gz);
    float pb = a.axis_b == 1 ? gy : (a.axis_b == 2 ? gz : gx);
    const float amin_t = wave_min(active ? pa : INFINITY), amax_t = wave_max(active ? pa : -INFINITY);
    const float bmin_t = wave_min(active ? pb : INFINITY), bmax_t = wave_max(active ? pb : -INFINITY);
    const float sbin = 1.0f / a.inv_s;
    int tby0 = (int)floorf((bmin_t - a.bmin) * a.inv_s), tby1 = (int)floorf((bmax_t - a.bmin) * a.inv_s);
    tby0 = __builtin_amdgcn_readfirstlane(min(max(tby0, 0), a.nby - 1));
    tby1 = __builtin_amdgcn_readfirstlane(min(max(tby1, tby0), a.nby - 1));
    int tbx0 = (int)floorf((amin_t - a.amin) * a.inv_s), tbx1 = (int)floorf((amax_t - a.amin) * a.inv_s);
    tbx0 = __builtin_amdgcn_readfirstlane(min(max(tbx0, 0), a.nbx - 1));
    tbx1 = __builtin_amdgcn_readfirstlane(min(max(tbx1, tbx0), a.nbx - 1));

    // strictly-inside box of the radius query (kdtree.cpp:46,53)
    const float lox = gx - R, hix = gx + R, loy = gy - R, hiy = gy + R, loz = gz - R, hiz = gz + R;
    unsigned long long wkey = 0;   // worst key kept
    int wslot = 0;
    // the slots beyond K never hold a key: an all-ones sentinel there lets the minimum searches run over all N slots without a test per
    // slot (32 wave-uniform masks that the register allocator spilled and reloaded inside the candidate loop)
    for(int s = K; s < N; ++s) keys[s][lane] = ~0ull;
    // minima of eight groups of four slots for the replace-the-worst step (24 registers: asked for by the Cholesky form of k_oi<32>
    // with the scalar Barnes structure only -- the pivoted-LU / spatially varying form lost 5 % to the register pressure)
    constexpr bool GROUPS = GROUP_MINIMA && N == 32;
    unsigned long long gk[8];
    int gs[8];
    bool ginit = false;
    // d2 > thr2 can neither be within R nor beat the worst kept rho (rho <= rho_h(d), monotone in d)
    const float thr2_R = R * R * 1.000001f + 1e-30f;
    float thr2 = active ? thr2_R : -1.0f;
    float thrq = INFINITY;   // the same threshold before the cut at R^2: what the SUM of the three exponents (x h^2) is tested against

    // all observations of bins [xa, xb] of bin row `row`
    auto process_row = [&](const int row, const int xa, const int xb) {
        if(row < 0 || row >= a.nby || xa > xb) return;
        const int js = a.bin_start[row * a.nbx + xa], je = a.bin_start[row * a.nbx + xb + 1];
        for(int base = js; base < je; base += 64) {
            const int mine = base + lane;
            float4 rec = make_float4(NAN, 0, 0, NAN);
            float2 met = make_float2(NAN, 0);
            if(mine < je) { rec = a.pgeo[mine]; met = a.smeta[mine]; }
            const int nc = min(64, je - base);
            if(a.scan_stats && lane == 0) atomicAdd(&a.scan_stats[0], (unsigned long long)nc);
            // Pass A: which candidates of the chunk can matter to THIS cell (one bit per candidate).  With elevation / laf dependent rho
            // on rough terrain nearly every candidate passes for SOME cell of the wave; evaluated wave-wide, all 64 lanes paid three
            // double exps per candidate for the sake of a few.  Pass B lets every lane walk its own list, in the same candidate order
            // (the per-cell sequence of insertions, thresholds and results is unchanged; a stale bit only costs a re-test).
            unsigned long long pm = 0ull;
            for(int c = 0; c < nc; ++c) {
                const float ox = readlane_f(rec.x, c), oy = readlane_f(rec.y, c), oz = readlane_f(rec.z, c);
                const float dx = ox - gx, dy = oy - gy, dz = oz - gz;
                float d2 = dx * dx + dy * dy;
                d2 = d2 + dz * dz;
                bool pass = d2 <= thr2;
                if constexpr(PLAIN && !WANT_TRUNC) {
                    // full list: rho = rho_h rho_v rho_w can only beat the worst kept rho if the SUM of the three exponents stays below
                    // that of the threshold (slack for the separately rounded factors: 1e-4 relative + 1e-4 h^2)
                    if(prune) {
                        const float oe = readlane_f(rec.w, c), ol = readlane_f(met.x, c);
                        const float te = (ge - oe) * p_sv, tl = (gl - ol) * p_sw;
                        float q = d2;
                        q = (p_sv != 0.0f && d_valid(ge) && d_valid(oe)) ? q + te * te : q;
                        q = (p_sw != 0.0f && d_valid(gl) && d_valid(ol)) ? q + tl * tl : q;
                        pass = pass && q <= thrq * 1.0001f + 1e-4f * h2;
                    }
                }
                if(WANT_TRUNC) pass = pass || d2 <= thr2_R;
                pm |= pass ? (1ull << c) : 0ull;
            }
            // Pass B
            while(wave_ballot(pm != 0ull) != 0ull) {
                const bool has = pm != 0ull;
                const int c = has ? __builtin_ctzll(pm) : 0;
                pm &= pm - 1ull;
                // (all six fields here, under the full EXEC mask: ds_bpermute returns 0 for a source lane that is switched off)
                const float ox = __shfl(rec.x, c), oy = __shfl(rec.y, c), oz = __shfl(rec.z, c);
                const float oe = __shfl(rec.w, c), ol = __shfl(met.x, c);
                const unsigned orig = (unsigned)__shfl(__float_as_int(met.y), c);
                const float dx = ox - gx, dy = oy - gy, dz = oz - gz;
                float d2 = dx * dx + dy * dy;
                d2 = d2 + dz * dz;
                if(a.scan_stats && wave_ballot(has && d2 <= thr2) != 0ull && lane == 0) atomicAdd(&a.scan_stats[1], 1ull);
                if(has && d2 <= thr2) {
                    const bool inbox = ox > lox && ox < hix && oy > loy && oy < hiy && oz > loz && oz < hiz;
                    const float dist = d_sqrt_cr(d2);
                    if(inbox && dist <= R) {   // within_radius (kdtree.cpp:255) and the cut inside corr (structure.cpp:216)
                        float rho;
                        if constexpr(PLAIN) {   // straight-line code: the three exp chains interleave (same values as d_barnes_rho)
                            rho = p_hh ? d_barnes_rho_flat(dist, p_rh) : 1.0f;
                            if(p_hv) { const float f = d_barnes_rho_flat(ge - oe, p_rv); rho = (d_valid(ge) && d_valid(oe)) ? rho * f : rho; }
                            if(p_hw) { const float f = d_barnes_rho_flat(gl - ol, p_rw); rho = (d_valid(gl) && d_valid(ol)) ? rho * f : rho; }
                        }
                        else {
                            rho = (st.cv && dist <= st.cv_dist) ? 0.0f : d_rho_x<TAB>(st.kh, dist, st.h);   // corr_background
                            if(d_valid(ge) && d_valid(oe)) rho *= d_rho_x<TAB>(st.kv, ge - oe, st.v);
                            if(d_valid(gl) && d_valid(ol)) rho *= d_rho_x<TAB>(st.kw, gl - ol, st.w);
                        }
                        const bool ins_ = rho > 0.0f && (cnt < K || (((unsigned long long)__float_as_uint(rho) << 32) | 0xffffffffull) > wkey);
                        if(a.scan_stats && wave_ballot(ins_) != 0ull) nins++;
                        if(rho > 0.0f) {   // oi.cpp:253
                            const unsigned long long key = ((unsigned long long)__float_as_uint(rho) << 32) | (unsigned)(~orig);
                            if(cnt < K) {
                                keys[cnt][lane] = key;
                                if(cnt == 0 || key < wkey) { wkey = key; wslot = cnt; }
                                cnt++;
                            }
                            else if(bounded) {
                                truncated = true;
                                if(key > wkey) {   // oi.cpp:262-273, tie-break: lower observation index
                                    keys[wslot][lane] = key;
                                    if constexpr(GROUPS) {
                                        // new worst through the minima of eight groups of four slots (in registers): only the group of the
                                        // replaced slot is read again (4 keys instead of N), then the minimum of the eight
                                        if(!ginit) {
                                            ginit = true;
                                            unsigned long long kv[N];
#pragma unroll
                                            for(int s = 0; s < N; ++s) kv[s] = keys[s][lane];
#pragma unroll
                                            for(int g = 0; g < 8; ++g) {
                                                gk[g] = ~0ull; gs[g] = 0;
#pragma unroll
                                                for(int j = 0; j < 4; ++j) {
                                                    const bool lt = kv[4 * g + j] < gk[g];
                                                    gk[g] = lt ? kv[4 * g + j] : gk[g];
                                                    gs[g] = lt ? 4 * g + j : gs[g];
                                                }
                                            }
                                        }
                                        else {
                                            const int g0 = wslot >> 2;
                                            unsigned long long k4[4];
#pragma unroll
                                            for(int j = 0; j < 4; ++j) k4[j] = keys[4 * g0 + j][lane];
                                            // (trees, not chains: the next candidate's pruning test waits for the new threshold)
                                            const bool l01 = k4[1] < k4[0], l23 = k4[3] < k4[2];
                                            const unsigned long long m01 = l01 ? k4[1] : k4[0], m23 = l23 ? k4[3] : k4[2];
                                            const int s01 = 4 * g0 + (l01 ? 1 : 0), s23 = 4 * g0 + (l23 ? 3 : 2);
                                            const bool lh = m23 < m01;
                                            const unsigned long long mk = lh ? m23 : m01;
                                            const int ms = lh ? s23 : s01;
#pragma unroll
                                            for(int g = 0; g < 8; ++g) { gk[g] = g == g0 ? mk : gk[g]; gs[g] = g == g0 ? ms : gs[g]; }
                                        }
                                        {
                                            unsigned long long t4[4], t2[2];
                                            int u4[4], u2[2];
#pragma unroll
                                            for(int g = 0; g < 4; ++g) { const bool lt = gk[2 * g + 1] < gk[2 * g]; t4[g] = lt ? gk[2 * g + 1] : gk[2 * g]; u4[g] = lt ? gs[2 * g + 1] : gs[2 * g]; }
#pragma unroll
                                            for(int g = 0; g < 2; ++g) { const bool lt = t4[2 * g + 1] < t4[2 * g]; t2[g] = lt ? t4[2 * g + 1] : t4[2 * g]; u2[g] = lt ? u4[2 * g + 1] : u4[2 * g]; }
                                            const bool lt = t2[1] < t2[0];
                                            wkey = lt ? t2[1] : t2[0]; wslot = lt ? u2[1] : u2[0];
                                        }
                                    }
                                    else {
                                    // new worst: static loops, so the LDS reads of a batch issue back to back (one latency, no branches); the
                                    // 62-slot form in two batches of 31 keys (all 62 in flight at once were 124 registers at the kernel's peak)
                                    constexpr int NB = N > 32 ? (N + 1) / 2 : N;
                                    wkey = key;
#pragma unroll
                                    for(int s0 = 0; s0 < N; s0 += NB) {
                                        unsigned long long kv[NB];
#pragma unroll
                                        for(int s = 0; s < NB; ++s) if(s0 + s < N) kv[s] = keys[s0 + s][lane];
#pragma unroll
                                        for(int s = 0; s < NB; ++s) {
                                            if(s0 + s < N) {
                                                const bool lt = kv[s] < wkey;
                                                wkey = lt ? kv[s] : wkey;
                                                wslot = lt ? s0 + s : wslot;
                                            }
                                        }
                                    }
                                    }
                                }
                            }
                            else overflow = true;   // more than N usable observations requested
                            if(prune && cnt == K) {
                                const float wr = __uint_as_float((unsigned)(wkey >> 32));
                                thrq = -2.0f * h2 * logf(wr) * 1.00002f + 2e-5f * h2;
                                thr2 = fminf(thr2_R, thrq);
                            }
                        }
                    }
                }
                else if(WANT_TRUNC && has && !truncated && cnt == K && d2 <= thr2_R) {
                    // pruned by the rho threshold: does it still count as a usable observation?
                    const bool inbox = ox > lox && ox < hix && oy > loy && oy < hiy && oz > loz && oz < hiz;
                    const float dist = d_sqrt_cr(d2);
                    if(inbox && dist <= R) {
                        float rho;
                        if constexpr(PLAIN) {   // straight-line code: the three exp chains interleave (same values as d_barnes_rho)
                            rho = p_hh ? d_barnes_rho_flat(dist, p_rh) : 1.0f;
                            if(p_hv) { const float f = d_barnes_rho_flat(ge - oe, p_rv); rho = (d_valid(ge) && d_valid(oe)) ? rho * f : rho; }
                            if(p_hw) { const float f = d_barnes_rho_flat(gl - ol, p_rw); rho = (d_valid(gl) && d_valid(ol)) ? rho * f : rho; }
                        }
                        else {
                            rho = (st.cv && dist <= st.cv_dist) ? 0.0f : d_rho_x<TAB>(st.kh, dist, st.h);   // corr_background
                            if(d_valid(ge) && d_valid(oe)) rho *= d_rho_x<TAB>(st.kv, ge - oe, st.v);
                            if(d_valid(gl) && d_valid(ol)) rho *= d_rho_x<TAB>(st.kw, gl - ol, st.w);
                        }
                        if(rho > 0.0f) truncated = true;
                    }
                }
            }
        }
    };

    // ---- phase 1: the square of bins around the tile, rows centre-out ------------------------------------------
    const int q = a.q0;
    const int sx0 = max(tbx0 - q, 0), sx1 = min(tbx1 + q, a.nbx - 1);
    const int sy0 = tby0 - q, sy1 = tby1 + q;
    for(int row = tby0; row <= tby1; ++row) process_row(row, sx0, sx1);
    for(int r = 1; r <= q; ++r) { process_row(tby0 - r, sx0, sx1); process_row(tby1 + r, sx0, sx1); }

    // ---- phase 2: every remaining bin that can still matter, rows centre-out ------------------------------------
    for(int r = 0;; ++r) {
        const float t2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wave_max(thr2))));
        if(t2 < 0.0f) break;
        const float gap = (r > 1) ? (float)(r - 1) * sbin * 0.999f : 0.0f;   // min projected distance tile -> row band r
        if(gap * gap > t2) break;
        const int rowA = tby0 - r, rowB = tby1 + r;
        if(rowA < 0 && rowB >= a.nby) break;
        const float wx = d_sqrt_raw(fmaxf(t2 - gap * gap, 0.0f)) * 1.0001f;
        int x0 = (int)floorf((amin_t - wx - a.amin) * a.inv_s) - 1, x1 = (int)floorf((amax_t + wx - a.amin) * a.inv_s) + 1;
        x0 = __builtin_amdgcn_readfirstlane(min(max(x0, 0), a.nbx - 1));
        x1 = __builtin_amdgcn_readfirstlane(min(max(x1, x0), a.nbx - 1));
        const int nrows = (r == 0) ? (tby1 - tby0 + 1) : 2;
        for(int k = 0; k < nrows; ++k) {
            const int row = (r == 0) ? tby0 + k : (k == 0 ? rowA : rowB);
            if(row >= sy0 && row <= sy1) {   // phase 1 did [sx0, sx1] of this row
                process_row(row, x0, min(x1, sx0 - 1));
                process_row(row, max(x0, sx1 + 1), x1);
            }
            else process_row(row, x0, x1);
        }
    }
    if(a.scan_stats && lane == 0) atomicAdd(&a.scan_stats[2 + min(__builtin_amdgcn_readfirstlane(nins), 69)], 1ull);
    return cnt;
}
