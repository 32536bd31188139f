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

// Four values per lane (the four features of a register quad) summed over W = 16 / 32 lanes at once:
// the first two butterfly steps also TRANSPOSE, so each lane carries one value from then on and lane
// (l & 3) == e ends up with the full sum of v[e].  14 VALU instead of 4 x 9, and the caller can
// commit all four sums with one masked LDS update (lanes 0..3 of the group) instead of four.
__device__ __forceinline__ float sum_cols4_t(float v0, float v1, float v2, float v3, int lane, bool w32) {
  const bool p1 = lane & 1, p2 = lane & 2;
  // xor 1: keep elements {p1, 2 + p1}, hand the other two to the neighbour
  float k0 = p1 ? v1 : v0, s0 = p1 ? v0 : v1;
  float k1 = p1 ? v3 : v2, s1 = p1 ? v2 : v3;
  float x0 = k0 + dpp_mov<0xB1>(s0);    // quad_perm [1,0,3,2]
  float x1 = k1 + dpp_mov<0xB1>(s1);
  // xor 2: keep element 2 * p2 + p1
  float k = p2 ? x1 : x0, sd = p2 ? x0 : x1;
  float y = k + dpp_mov<0x4E>(sd);      // quad_perm [2,3,0,1]
  // the four quads of the 16-lane row (same l & 3): rotate by 8 and by 4
  y += dpp_mov<0x128>(y);               // row_ror:8
  y += dpp_mov<0x124>(y);               // row_ror:4
  if (w32) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    y = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return y;
}
template <int W>
__device__ __forceinline__ float sum_cols4(float v0, float v1, float v2, float v3, int lane) {
  return sum_cols4_t(v0, v1, v2, v3, lane, W == 32);
}

// sgacc[i] += v as ONE ds_add_f32 (no returned value, no read-modify-write round trip through registers and
// its lgkmcnt wait).  Every accumulator slot is only ever updated by the wave that owns the feature, in program
// order, so the sums stay deterministic.  Call it under an exec mask that leaves only the owning lanes active: the LDS
// float atomic serialises over its ACTIVE lanes (tests/micro/mfma_partner_mix.hip: ~770 cycles per 64-lane
// instruction with four waves issuing them) and holds the LDS pipeline meanwhile.
__device__ __forceinline__ void lds_add(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The same update as a plain read-modify-write, for callers that run it on ALL lanes (non-owners on a per-lane sink, to
// keep the epilogue branch-free): a ds_read / ds_write pair costs the LDS ~4 cycles each whatever the lane count.
// (role-split reverse sweep at 6x256 / 360k points: 3.61 ms with the 64-lane atomic, 3.38 masked, 3.36 this; the
// 8-wave kernels keep the masked atomic - there the exposed LDS round trip of this form costs more: bwd_bf16_wide
// 21.5 -> 27.6 ms at 8x400.)
__device__ __forceinline__ void lds_rmw_add(float* p, float v) { *p += v; }
