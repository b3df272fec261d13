// k_oi_union with 48 register columns (max_points 33..46; round 6): its own translation unit so that it compiles beside oi.hip and oi_union64.hip.
#include "oi_union.h"

void gpp_launch_union48(const OiArgs& a, const unsigned nblocks, const bool plain, const bool list, hipStream_t stream) {
    constexpr int T = 64 * UnionCfg<48>::WPB;
    const dim3 grid(nblocks), block(T);
    if(plain) {
        if(list) hipLaunchKernelGGL((k_oi_union<true, true, 48>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_oi_union<true, false, 48>), grid, block, 0, stream, a);
    }
    else {
        if(list) hipLaunchKernelGGL((k_oi_union<false, true, 48>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_oi_union<false, false, 48>), grid, block, 0, stream, a);
    }
}
