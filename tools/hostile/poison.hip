// Test tooling (tools/oi_hostile_soak.py), not part of the product: leaves hostile bit patterns where a kernel that reads
// something it never wrote would find them -- every byte of LDS of every CU, and a block of VGPRs / AGPRs of every SIMD.
// Built by tools/hostile/build.sh into tools/hostile/libpoison.so and loaded with ctypes into the process under test.
#include <hip/hip_runtime.h>
#include <cstdio>

// 160 KB of LDS per workgroup: one workgroup per CU, all 64-byte rows written
__global__ __launch_bounds__(256) void k_poison_lds(unsigned pattern, int nwords, unsigned* sink) {
    extern __shared__ unsigned s_all[];
    for(int i = threadIdx.x; i < nwords; i += 256) s_all[i] = pattern;
    __syncthreads();
    // (read something back so that the stores are not dead)
    if(threadIdx.x == 0 && s_all[(blockIdx.x * 977) % nwords] != pattern) sink[0] = 1u;
    // stay resident for a while: the next workgroup of this launch then lands on another CU
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while(__builtin_amdgcn_s_memtime() - t0 < 20000ull) { }
}

// every lane leaves `pattern` in v0..v255 and a0..a255 (512 registers per lane: one wave per SIMD)
__global__ __launch_bounds__(64, 1) void k_poison_regs(unsigned pattern, unsigned* sink) {
    unsigned acc = 0;
#define W8(b) asm volatile("v_mov_b32 v" #b ", %0" :: "s"(pattern) : "v" #b);
#define A8(b) asm volatile("v_accvgpr_write_b32 a" #b ", %0" :: "s"(pattern) : "a" #b);
#define R16(M, p) M(p##0) M(p##1) M(p##2) M(p##3) M(p##4) M(p##5) M(p##6) M(p##7) M(p##8) M(p##9)
    // v16..v249 and a0..a249 (the low VGPRs hold this kernel's own few values)
    R16(W8, 2) R16(W8, 3) R16(W8, 4) R16(W8, 5) R16(W8, 6) R16(W8, 7) R16(W8, 8) R16(W8, 9)
    R16(W8, 10) R16(W8, 11) R16(W8, 12) R16(W8, 13) R16(W8, 14) R16(W8, 15) R16(W8, 16) R16(W8, 17) R16(W8, 18) R16(W8, 19)
    R16(W8, 20) R16(W8, 21) R16(W8, 22) R16(W8, 23) R16(W8, 24)
    R16(A8, 1) R16(A8, 2) R16(A8, 3) R16(A8, 4) R16(A8, 5) R16(A8, 6) R16(A8, 7) R16(A8, 8) R16(A8, 9)
    R16(A8, 10) R16(A8, 11) R16(A8, 12) R16(A8, 13) R16(A8, 14) R16(A8, 15) R16(A8, 16) R16(A8, 17) R16(A8, 18) R16(A8, 19)
    R16(A8, 20) R16(A8, 21) R16(A8, 22) R16(A8, 23) R16(A8, 24)
    asm volatile("v_accvgpr_write_b32 a0, %0" :: "s"(pattern) : "a0");
    asm volatile("v_accvgpr_write_b32 a1, %0" :: "s"(pattern) : "a1");
    asm volatile("v_accvgpr_write_b32 a2, %0" :: "s"(pattern) : "a2");
    asm volatile("v_accvgpr_write_b32 a3, %0" :: "s"(pattern) : "a3");
    asm volatile("v_accvgpr_write_b32 a4, %0" :: "s"(pattern) : "a4");
    asm volatile("v_accvgpr_write_b32 a5, %0" :: "s"(pattern) : "a5");
    asm volatile("v_accvgpr_write_b32 a6, %0" :: "s"(pattern) : "a6");
    asm volatile("v_accvgpr_write_b32 a7, %0" :: "s"(pattern) : "a7");
    asm volatile("v_accvgpr_write_b32 a8, %0" :: "s"(pattern) : "a8");
    asm volatile("v_accvgpr_write_b32 a9, %0" :: "s"(pattern) : "a9");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while(__builtin_amdgcn_s_memtime() - t0 < 20000ull) acc++;
    if(acc == 0xffffffffu) sink[1] = acc;
}

static unsigned* g_sink = nullptr;
extern "C" int poison_lds(unsigned pattern) {
    if(!g_sink && hipMalloc((void**)&g_sink, 64) != hipSuccess) return 1;
    const int bytes = 160 * 1024;
    if(hipFuncSetAttribute((const void*)k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 2;
    hipLaunchKernelGGL(k_poison_lds, dim3(1024), dim3(256), bytes, 0, pattern, bytes / 4, g_sink);
    if(hipGetLastError() != hipSuccess) return 3;
    return hipDeviceSynchronize() == hipSuccess ? 0 : 4;
}
extern "C" int poison_regs(unsigned pattern) {
    if(!g_sink && hipMalloc((void**)&g_sink, 64) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_poison_regs, dim3(4096), dim3(64), 0, 0, pattern, g_sink);
    if(hipGetLastError() != hipSuccess) return 3;
    return hipDeviceSynchronize() == hipSuccess ? 0 : 4;
}
