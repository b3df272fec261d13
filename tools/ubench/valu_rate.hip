// VALU issue-rate probe (gfx950): cycles per wave-instruction for a few opcodes, from a full-chip launch of dependent-free
// chains.  hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float x) {
    float a[16];
#pragma unroll
    for(int i = 0; i < 16; i++) a[i] = x + i + threadIdx.x;
    double d[8];
#pragma unroll
    for(int i = 0; i < 8; i++) d[i] = x + i;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[8];
#pragma unroll
    for(int i = 0; i < 8; i++) p[i] = (v2f){x + i, x - i};
    unsigned u[16];
#pragma unroll
    for(int i = 0; i < 16; i++) u[i] = threadIdx.x * 7 + i;
    for(int it = 0; it < N_ITER; it++) {
        if(OP == 0) {
#pragma unroll
            for(int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
        }
        else if(OP == 1) {
#pragma unroll
            for(int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
#pragma unroll
            for(int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        }
        else if(OP == 2) {
#pragma unroll
            for(int i = 0; i < 8; i++) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#pragma unroll
            for(int i = 0; i < 8; i++) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        }
        else if(OP == 3) {
#pragma unroll
            for(int i = 0; i < 16; i++) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]));
        }
        else if(OP == 4) {
#pragma unroll
            for(int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(x));
        }
        else if(OP == 5) {
#pragma unroll
            for(int i = 0; i < 16; i++) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
        }
        else if(OP == 6) {
#pragma unroll
            for(int i = 0; i < 16; i++) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(u[i]) : "v"(3u), "v"(u[(i + 1) & 15]));
        }
        else if(OP == 7) {
#pragma unroll
            for(int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#pragma unroll
            for(int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        }
        else if(OP == 8) {
#pragma unroll
            for(int i = 0; i < 16; i++) asm volatile("v_cmp_gt_f32_e32 vcc, %1, %0\n\tv_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 15]) : "vcc");
        }
        else if(OP == 9) {
#pragma unroll
            for(int i = 0; i < 16; i++) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(a[i]));
        }
    }
    float s = 0;
#pragma unroll
    for(int i = 0; i < 16; i++) s += a[i] + (float)u[i];
#pragma unroll
    for(int i = 0; i < 8; i++) s += (float)d[i] + p[i].x + p[i].y;
    if(s == 12345.678f) out[0] = s;
}
template <int OP>
void run(const char* name, int per_iter, int waves_per_simd) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = one per SIMD) x waves_per_simd
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)N_ITER * per_iter * waves_per_simd;
    printf("%-28s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instruction per SIMD (x 2.4 GHz = %.2f cycles)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main() {
    for(int w : {1, 4}) {
        run<0>("v_add_f32", 16, w); run<4>("v_fma_f32", 16, w); run<1>("v_pk_add_f32", 16, w); run<2>("v_add_f64", 16, w); run<7>("v_fma_f64", 16, w);
        run<3>("v_sad_u8", 16, w); run<5>("v_add_u32", 16, w); run<6>("v_lshlrev_b32_sdwa", 16, w); run<8>("v_cmp+v_addc (2 instr)", 32, w); run<9>("v_cvt_pk_u8_f32", 16, w);
    }
    return 0;
}
