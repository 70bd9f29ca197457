// Lane exchanges inside a 64-wide wave without LDS memory (gfx950): DPP quad permutes, ds_swizzle in bit mode, ds_bpermute.  Used by the in-wave
// forms that are A/B-ed against the LDS / register forms of the product (zp_set_tuning "ntt_small_wave", "fri_fold_lanes";
// profiles/r5_dpp_ab.txt) -- the north_star's "wavefront-shuffle" clause, measured.
#pragma once
#include <hip/hip_runtime.h>

#include "ctx.hpp"

namespace {

// the value of lane ^ (1 << LH): DPP quad permutes (1, 2), ds_swizzle in bit mode (4, 8, 16: the crossbar, no LDS allocation), ds_bpermute
// across the two halves of the wave (32)
template <int LH>
__device__ __forceinline__ u64 lane_xor(u64 v) {
    int lo = (int)(u32)v, hi = (int)(u32)(v >> 32);
    if constexpr (LH == 0) {
        lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
        hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    } else if constexpr (LH == 1) {
        lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);        // quad_perm [2,3,0,1]
        hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
    } else if constexpr (LH <= 4) {
        constexpr int pat = 0x1F | ((1 << LH) << 10);                   // and 0x1F, or 0, xor 1 << LH
        lo = __builtin_amdgcn_ds_swizzle(lo, pat);
        hi = __builtin_amdgcn_ds_swizzle(hi, pat);
    } else {
        const int src = ((int)(threadIdx.x & 63) ^ 32) << 2;
        lo = __builtin_amdgcn_ds_bpermute(src, lo);
        hi = __builtin_amdgcn_ds_bpermute(src, hi);
    }
    return (u64)(u32)lo | ((u64)(u32)hi << 32);
}

}  // namespace
