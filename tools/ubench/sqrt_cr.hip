// Every float32 from 2^-96 (where the compiler's own expansion starts to rescale its argument) up to FLT_MAX, and 0, inf, NaN: d_sqrt_cr
// against the compiler's correctly rounded sqrtf
// (-fhip-fp32-correctly-rounded-divide-sqrt), on the GPU.  Build + run: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// -fhip-fp32-correctly-rounded-divide-sqrt tools/ubench/sqrt_cr.hip -o /tmp/sqrt_cr && /tmp/sqrt_cr
#include "../../gridpp_amd/csrc/oi_common.h"
#include <cstdio>
__global__ void k_cmp(unsigned long long* bad, unsigned* first) {
    unsigned long long n = 0;
    for(unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < 0x80000000ull; b += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = (unsigned)b;
        if(bits != 0u && bits < 0x0f800000u) { const float a = d_sqrt_cr(__uint_as_float(bits)), r = sqrtf(__uint_as_float(bits)); if(__float_as_uint(a) != __float_as_uint(r)) atomicAdd(bad + 1, 1ull); continue; }   // below 2^-96: outside the contract, counted apart
        const float x = __uint_as_float(bits);
        const float a = d_sqrt_cr(x), r = sqrtf(x);
        const bool same = __float_as_uint(a) == __float_as_uint(r) || (a != a && r != r);
        if(!same) { n++; atomicMin(first, bits); }
    }
    if(n) atomicAdd(bad, n);
}
int main() {
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 16); hipMalloc(&first, 4);
    hipMemset(bad, 0, 16); hipMemset(first, 0xff, 4);
    hipLaunchKernelGGL(k_cmp, dim3(4096), dim3(256), 0, 0, bad, first);
    unsigned long long hb2[2] = {0, 0}; unsigned hf = 0;
    hipMemcpy(hb2, bad, 16, hipMemcpyDeviceToHost); const unsigned long long hb = hb2[0]; hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("d_sqrt_cr vs sqrtf over 0 and every float32 >= 2^-96: %llu differences%s; below 2^-96 (outside the contract): %llu one-ulp differences\n", hb, hb ? "" : " -- identical", hb2[1]);
    if(hb) printf("first differing bit pattern: 0x%08x\n", hf);
    return hb ? 1 : 0;
}
