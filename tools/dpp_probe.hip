#include <hip/hip_runtime.h>
__global__ void k(int* out) {
    int x = threadIdx.x;
    int a = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false);  // wave_shl:1
    int b = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, false);  // wave_shr:1
    int c = __builtin_amdgcn_update_dpp(-1, x, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    int d = __builtin_amdgcn_update_dpp(x, x, 0x130, 0xf, 0xA, false);   // wave_shl, odd banks only
    d = __builtin_amdgcn_update_dpp(d, x, 0x138, 0xf, 0x5, false);       // wave_shr, even banks only
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b; out[128 + threadIdx.x] = c; out[192 + threadIdx.x] = d;
}
int main() {
    int* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); int h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for(int r = 0; r < 4; r++) { for(int i = 0; i < 64; i++) printf("%d ", h[r * 64 + i]); printf("\n"); }
}
