// Cross-lane sums on the VALU (DPP) instead of LDS shuffles: the skinny gradients of the reverse
// sweep reduce ~180 values per wave per tile over the 16 / 32 lanes that hold the same feature.
// 16 lanes: 4 x v_add_f32_dpp (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror);
// 32 lanes: + one v_permlane16_swap and an add.  Every lane ends up with the full sum.
#pragma once
#include <hip/hip_runtime.h>

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum16(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float sum32(float v) {
  v = sum16(v);
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int W>
__device__ __forceinline__ float sum_cols(float v) { return W == 32 ? sum32(v) : sum16(v); }
