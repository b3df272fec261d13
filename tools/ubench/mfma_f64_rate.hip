// FP64 matrix-core probe (gfx950): cycles per v_mfma_f64_16x16x4_f64 from a full-chip launch, with 1 / 2 / 4 / 8 independent accumulators
// per wave and 1 / 2 waves per SIMD; and the same with v_fma_f64 interleaved (do the vector and the matrix FP64 paths share a pipe?).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f64_rate.hip -o /tmp/mfma_f64_rate && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2048
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC, int NFMA>
__global__ __launch_bounds__(256) void k(double* out, double x) {
    v4d acc[8];
#pragma unroll
    for(int i = 0; i < 8; i++) acc[i] = (v4d){x, x, x, x};
    double d[8];
#pragma unroll
    for(int i = 0; i < 8; i++) d[i] = x + i;
    const double a = x + threadIdx.x, b = x - threadIdx.x;
    for(int it = 0; it < N_ITER; it++) {
#pragma unroll
        for(int r = 0; r < 8 / NACC; r++) {
#pragma unroll
            for(int i = 0; i < NACC; i++) {
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                if(NFMA > 0) {
#pragma unroll
                    for(int f = 0; f < NFMA; f++) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[(i * NFMA + f) & 7]) : "v"(a));
                }
            }
        }
    }
    double s = 0;
#pragma unroll
    for(int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + d[i];
    if(s == 12345.678) out[0] = s;
}
template <int NACC, int NFMA>
void run(int waves_per_simd) {
    double* out; hipMalloc(&out, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL((k<NACC, NFMA>), dim3(blocks), dim3(256), 0, 0, out, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, NFMA>), dim3(blocks), dim3(256), 0, 0, out, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double cycles = ms * 1e-3 * clk_khz * 1e3;
    const double mfma_per_simd = (double)N_ITER * 8 * waves_per_simd;
    const double flops = mfma_per_simd * 1024 * (2048.0 + NFMA * 128.0);
    printf("acc %d  fma/mfma %d  waves/SIMD %d: %.3f ms, %.1f cycles per MFMA (+%d FMA) per SIMD at %d MHz, %.1f TFLOP/s\n", NACC, NFMA, waves_per_simd, ms,
           cycles / mfma_per_simd, NFMA, clk_khz / 1000, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<1, 0>(1); run<2, 0>(1); run<4, 0>(1); run<8, 0>(1);
    run<1, 0>(2); run<4, 0>(2); run<8, 0>(2);
    run<4, 4>(1); run<4, 8>(1); run<4, 16>(1);
    run<4, 4>(2); run<4, 8>(2); run<4, 16>(2);
    run<1, 4>(1); run<1, 8>(1); run<1, 16>(1);
    return 0;
}
