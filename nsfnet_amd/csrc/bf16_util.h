// bf16 split helpers and LDS image geometry of the bf16x3 MFMA path (gfx950).
//
// bf16x3: every fp32 operand x is split x = hi + lo (both bf16, RNE) and a product is
// accumulated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with fp32
// accumulate (the dropped a_lo*b_lo term is ~2^-18 relative).  Operand layouts were checked
// on hardware by tests/micro/mfma_bf16_layout.hip.
#pragma once
#include <hip/hip_runtime.h>

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {   // lowers to v_cvt_pk_bf16_f32 (RNE)
  bf16x2_t v;
  v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// four consecutive values -> packed hi (2 dwords) and lo (2 dwords)
__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, u32x2& hi, u32x2& lo) {
  unsigned h0 = pack_bf16(x0, x1), h1 = pack_bf16(x2, x3);
  hi[0] = h0; hi[1] = h1;
  lo[0] = pack_bf16(x0 - bf_lo_f(h0), x1 - bf_hi_f(h0));
  lo[1] = pack_bf16(x2 - bf_lo_f(h1), x3 - bf_hi_f(h1));
}

// ---- 24-bit spill format of the role-split sweeps -------------------------------------------------------------------
// The bf16x3 operand split consumes 16 significant bits of a value (bf16 hi + bf16 lo); the spilled activations and
// z-adjoints are read back only to be split (MFMA operands) or to enter chain-rule products whose other factors are
// bf16x3 GEMM outputs of that accuracy.  So they are spilled ROUNDED TO 24 BITS (sign, exponent, 15 mantissa bits:
// relative error <= 2^-16): the sixteen values of a register quad (4 features x 4 streams) travel as THREE 16-byte
// planes - top halves of streams 0-1, top halves of streams 2-3, third bytes of all four - instead of four fp32
// planes: a quarter fewer vector-memory instructions and bytes with the same 1-KiB-per-wave-instruction coalescing.
// (Measured first as separate 8-byte and 4-byte planes: the same bytes in TWICE the instructions was slower than
// fp32 - the spill is bound by memory instructions through the CU's vector-memory path, not by HBM bytes.)
// Round half up in magnitude on the integer image; v_perm_b32 moves the bytes.
// NaN / infinity through the spill: +-infinity and every NaN whose payload is below 0x7fff80 keep their class (the add
// stays inside the mantissa; the dropped low byte is never the only payload of a NaN that arithmetic produced, because
// the hardware sets the quiet bit 0x400000: its own NaNs are 0x7fc00000 / 0xffc00000).  A NaN with an all-ones payload
// (0x7fffff80 .. 0x7fffffff, either sign) would carry into the exponent and read back as +-0; no instruction of the
// sweeps produces one, and guarding the add costs three VALU per value in the hottest loop (96 per quarter phase),
// so the case is documented and pinned by tests/test_spill_format.py instead.
__device__ __forceinline__ void pack24(const f32x4& x, u32x2& hi, unsigned& lo) {
  const unsigned r0 = __float_as_uint(x[0]) + 0x80u, r1 = __float_as_uint(x[1]) + 0x80u;
  const unsigned r2 = __float_as_uint(x[2]) + 0x80u, r3 = __float_as_uint(x[3]) + 0x80u;
  hi[0] = __builtin_amdgcn_perm(r1, r0, 0x07060302u);      // (selector bytes 0-3: second operand, 4-7: first, 0x0c: zero)
  hi[1] = __builtin_amdgcn_perm(r3, r2, 0x07060302u);
  lo = __builtin_amdgcn_perm(r1, r0, 0x0c0c0501u) | __builtin_amdgcn_perm(r3, r2, 0x05010c0cu);
}
__device__ __forceinline__ f32x4 unpack24(const u32x2& hi, unsigned lo) {
  f32x4 x;
  x[0] = __uint_as_float(__builtin_amdgcn_perm(hi[0], lo, 0x0504000cu));
  x[1] = __uint_as_float(__builtin_amdgcn_perm(hi[0], lo, 0x0706010cu));
  x[2] = __uint_as_float(__builtin_amdgcn_perm(hi[1], lo, 0x0504020cu));
  x[3] = __uint_as_float(__builtin_amdgcn_perm(hi[1], lo, 0x0706030cu));
  return x;
}

// a register quad's four planes <-> the three 16-byte planes of the spill (hi16 of planes 0-1, of planes 2-3, lo8 of all)
__device__ __forceinline__ void pack24_quad(const f32x4& x0, const f32x4& x1, const f32x4& x2, const f32x4& x3, u32x4 (&pk)[3]) {
  u32x2 h; unsigned l;
  pack24(x0, h, l); pk[0][0] = h[0]; pk[0][1] = h[1]; pk[2][0] = l;
  pack24(x1, h, l); pk[0][2] = h[0]; pk[0][3] = h[1]; pk[2][1] = l;
  pack24(x2, h, l); pk[1][0] = h[0]; pk[1][1] = h[1]; pk[2][2] = l;
  pack24(x3, h, l); pk[1][2] = h[0]; pk[1][3] = h[1]; pk[2][3] = l;
}
__device__ __forceinline__ f32x4 unpack24_plane(const u32x4 (&pk)[3], int p) {
  return unpack24(u32x2{pk[p >> 1][2 * (p & 1)], pk[p >> 1][2 * (p & 1) + 1]}, pk[2][p]);
}

// tanh for the bf16 modes: 1 - 2/(exp(2z)+1) on v_exp_f32 / v_rcp_f32 (both 1 ulp: abs. error
// ~2e-7, far below the bf16x3 operand rounding); saturates correctly at +-inf.  The bare v_rcp_f32
// matters: an IEEE 1/x costs 11 VALU instructions, a fifth of the whole chain-rule epilogue.
__device__ __forceinline__ float fast_tanh(float z) {
  float e = __builtin_amdgcn_exp2f(z * 2.8853900817779268f);     // exp(2z) = 2^(2 z log2 e): one multiply
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// Timing-only stand-in (PINN_ABL & 1024, scripts/abl_build.py): the same multiply-accumulate count, operand bytes and
// issue cycles as one 32x32x16 MFMA, as TWO v_mfma_f32_16x16x32_bf16 on half Q of the accumulator (wrong results on
// purpose).  It prices the MFMA SHAPE: the chip holds a higher clock on the 16x16x32 form (MI355X_MICROARCH.md, DVFS
// give-back item 7), at twice the MFMA issue slots.
__device__ __forceinline__ f32x16 mfma_bf16_shape16(int Q, u32x4 a, u32x4 b, f32x16 c) {      // Q: a constant after unrolling
  f32x4 c0 = {c[8 * Q + 0], c[8 * Q + 1], c[8 * Q + 2], c[8 * Q + 3]}, c1 = {c[8 * Q + 4], c[8 * Q + 5], c[8 * Q + 6], c[8 * Q + 7]};
  c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b), __builtin_bit_cast(bf16x8_t, a), c1, 0, 0, 0);      // (operands swapped: not a common subexpression)
#pragma unroll
  for (int i = 0; i < 4; ++i) { c[8 * Q + i] = c0[i]; c[8 * Q + 4 + i] = c1[i]; }
  return c;
}
#ifndef PINN_ABL
#define PINN_ABL_SHAPE16 0
#else
#define PINN_ABL_SHAPE16 ((PINN_ABL) & 1024)
#endif
#define MFMA_Q(q, a, b, c) (PINN_ABL_SHAPE16 ? mfma_bf16_shape16((q) & 1, a, b, c) : mfma_bf16(a, b, c))

// ---- activation image in LDS for the fused fwd/bwd kernels -------------------
// X[hilo(2)][plane(4)][col in plane (PPL)][k (RSE)] bf16, RSE = HP rounded up to a power of two
// (>= 32); 16-byte chunks (8 k) XOR-swizzled so that the ds_read_b128 of one k-chunk by
// different columns is bank-conflict free.  PPL = 32 (128-column tile) or 16 (64-column tile).
template <int HP, int PPL = 32>
struct XImg {
  static constexpr int RSE = HP <= 32 ? 32 : HP <= 64 ? 64 : HP <= 128 ? 128 : HP <= 256 ? 256 : 512;
  static constexpr int NCH = RSE / 8;                    // chunks per row
  static constexpr int R = RSE >= 128 ? 1 : 128 / RSE;   // rows per 256-byte bank row
  static constexpr int MASK = (NCH < 16 ? NCH : 16) - 1;
  static constexpr int PLANE = PPL * RSE;                // elements per (hilo, plane)
  static constexpr int HALF = 4 * PLANE;                 // elements per hilo half
  static constexpr size_t BYTES = (size_t)2 * HALF * 2;
  // byte offset (within one plane) of chunk `ch` of column `col`
  __device__ static __forceinline__ int chunk_off(int col, int ch) {
    return (col * RSE + ((ch ^ ((col / R) & MASK)) << 3)) * 2;
  }
};
