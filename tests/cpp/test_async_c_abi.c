/* A plain C caller of the drop-in boundary (include/gridpp_hip.h) with fields resident in HBM: a stream of optimal-interpolation analyses with
 * one call ahead (GPP_MEM_DEVICE | GPP_ASYNC + gpp_wait) must return the very bits of the blocking calls.  Built with gcc against the HIP
 * runtime (hipMalloc / hipMemcpy only) by tests/test_gpu_cpp_host.py.  The reference has no counterpart (its calls return their result,
 * src/api/oi.cpp:26-136); this is what a host that keeps its fields on the GPU binds. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gridpp_hip.h"

#define CHECK(x) do { int rc_ = (x); if(rc_ != GPP_OK) { printf("FAILED %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, gpp_last_error()); return 1; } } while(0)
#define HIP(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while(0)

static unsigned long long s_ = 88172645463325252ull;
static double rnd(void) { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return (double)(s_ >> 11) / 9007199254740992.0; }

int main(void) {
    enum { Y = 160, X = 200, S = 220, K = 6 };
    const int C = Y * X;
    float *lats = malloc(sizeof(float) * C), *lons = malloc(sizeof(float) * C), *bg = malloc(sizeof(float) * C);
    float plat[S], plon[S];
    for(int y = 0; y < Y; y++) for(int x = 0; x < X; x++) { lats[y * X + x] = 60.0f + y / 240.0f; lons[y * X + x] = 10.0f + x / 120.0f; bg[y * X + x] = (float)(2 * rnd() - 1); }
    for(int i = 0; i < S; i++) { plat[i] = 60.0f + (float)(rnd() * Y / 240.0); plon[i] = 10.0f + (float)(rnd() * X / 120.0); }
    gpp_points *grid = NULL, *points = NULL;
    CHECK(gpp_grid_create(lats, lons, NULL, NULL, Y, X, GPP_GEODETIC, &grid));
    CHECK(gpp_points_create(plat, plon, NULL, NULL, S, GPP_GEODETIC, &points));
    gpp_structure st;
    memset(&st, 0, sizeof(st));
    st.kind = GPP_SK_BARNES; st.h = 8000.0f;                       /* BarnesStructure(8000): kind_v = kind_w = 0 (same kernel), v = w = 0 */
    CHECK(gpp_structure_min_rho(st.kind, st.h, NAN, &st.min_rho));   /* hmax = NaN: the default min_rho */
    float *d_bg, *d_obs[K], *d_rat[K], *d_pbg[K], *d_ref[K], *d_out[K];
    HIP(hipMalloc((void**)&d_bg, sizeof(float) * C));
    HIP(hipMemcpy(d_bg, bg, sizeof(float) * C, hipMemcpyHostToDevice));
    for(int k = 0; k < K; k++) {
        float o[S], r[S], p[S];
        for(int i = 0; i < S; i++) { o[i] = (float)(4 * rnd() - 2); r[i] = (float)(0.05 + rnd()); p[i] = (float)(4 * rnd() - 2); }
        if(k == 3) for(int i = 0; i < S; i += 3) o[i] = NAN;      /* other usable observations: the wait may have to run this call again */
        HIP(hipMalloc((void**)&d_obs[k], sizeof(o))); HIP(hipMalloc((void**)&d_rat[k], sizeof(r))); HIP(hipMalloc((void**)&d_pbg[k], sizeof(p)));
        HIP(hipMalloc((void**)&d_ref[k], sizeof(float) * C)); HIP(hipMalloc((void**)&d_out[k], sizeof(float) * C));
        HIP(hipMemcpy(d_obs[k], o, sizeof(o), hipMemcpyHostToDevice)); HIP(hipMemcpy(d_rat[k], r, sizeof(r), hipMemcpyHostToDevice));
        HIP(hipMemcpy(d_pbg[k], p, sizeof(p), hipMemcpyHostToDevice));
        CHECK(gpp_optimal_interpolation_full(grid, d_bg, NULL, points, d_obs[k], d_rat[k], d_pbg[k], NULL, &st, 30, 1, d_ref[k], NULL, GPP_MEM_DEVICE));   /* blocking */
    }
    if(gpp_wait() != GPP_EINVAL) { printf("FAILED: gpp_wait with nothing pending\n"); return 1; }
    for(int rep = 0; rep < 3; rep++) {
        for(int k = 0; k < K; k++) {       /* call k is enqueued, then call k - 1 is completed */
            CHECK(gpp_optimal_interpolation_full(grid, d_bg, NULL, points, d_obs[k], d_rat[k], d_pbg[k], NULL, &st, 30, 1, d_out[k], NULL, GPP_MEM_DEVICE | GPP_ASYNC));
            if(k > 0) CHECK(gpp_wait());
        }
        CHECK(gpp_wait());
        int pending = -1;
        CHECK(gpp_pending(&pending));
        if(pending != 0) { printf("FAILED: %d calls pending after the last wait\n", pending); return 1; }
        CHECK(gpp_synchronize());
        float *a = malloc(sizeof(float) * C), *b = malloc(sizeof(float) * C);
        for(int k = 0; k < K; k++) {
            HIP(hipMemcpy(a, d_ref[k], sizeof(float) * C, hipMemcpyDeviceToHost)); HIP(hipMemcpy(b, d_out[k], sizeof(float) * C, hipMemcpyDeviceToHost));
            if(memcmp(a, b, sizeof(float) * C) != 0) { printf("FAILED: deferred call %d of pass %d differs from the blocking call\n", k, rep); return 1; }
            HIP(hipMemset(d_out[k], 0xFF, sizeof(float) * C));
        }
        free(a); free(b);
    }
    gpp_oi_stats stats;
    CHECK(gpp_oi_last_stats(&stats));
    if(stats.cells != C || stats.cells_updated <= 0) { printf("FAILED: statistics of the last deferred call\n"); return 1; }
    CHECK(gpp_points_destroy(points)); CHECK(gpp_points_destroy(grid));
    printf("C-ABI deferred calls: all checks passed (%d analyses x 3 passes, %d cells)\n", K, C);
    return 0;
}
