// Shared host/device layout definitions for the PINN HIP pipeline (gfx950 only).
//
// Geometry
//   H   hidden width of the FCNet, HP = H rounded up to a multiple of 32.
//   A "tile" is 128 MFMA columns:  residual mode (4 streams: value, d/dx, d/dy,
//   Laplacian) = 32 points x 4 streams, column = stream*32 + point;
//   value mode (1 stream) = 128 points, column = point.
//   One workgroup = HP/32 waves; wave w owns hidden features [32w, 32w+32).
//
// Saved-activation tile (S) and z-adjoint tile (Zb), per (tile, layer): HP*128
// floats laid out [plane j(4)][feature group o/4 (HP/4)][col%32 (32)][o%4 (4)]
// so that a lane of the 32x32 MFMA accumulator layout (fixed column, four
// consecutive features per register quad) moves it with 16-byte accesses and a
// wave-instruction touches 1 KiB contiguous.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __HIPCC__
#define PINN_HD __host__ __device__ __forceinline__
#else
#define PINN_HD inline
#endif

#define PINN_TILE_COLS 128
#define PINN_MAX_HP 512      // <= 256: 128-column tiles (all precisions); 257..512: 64-column tiles (fp32)
#define PINN_NLOSS 8   // loss partial slots per workgroup
// Dynamic-LDS limit every kernel is configured with (hipFuncAttributeMaxDynamicSharedMemorySize): the device maximum,
// NOT the configuring plan's own byte count - the limit is per kernel instantiation, and two live plans that share an
// instantiation (same width, different depth) must not lower each other's limit.  pinn_plan_create rejects plans
// whose kernels need more.
#define PINN_LDS_MAX 163840

// ---- prepared-parameter buffer (floats) -----------------------------------
//   [w0x HP][w0y HP][b0 HP]
//   per hidden GEMM layer l = 1..L-1:  [Wf HP*HP][WTf HP*HP][b HP]
//   [wout 4*HP][bout 4]
PINN_HD size_t prep_w0x(int HP) { (void)HP; return 0; }
PINN_HD size_t prep_w0y(int HP) { return (size_t)HP; }
PINN_HD size_t prep_b0(int HP) { return (size_t)2 * HP; }
PINN_HD size_t prep_layer_stride(int HP) { return (size_t)2 * HP * HP + HP; }
PINN_HD size_t prep_wf(int HP, int l) { return (size_t)3 * HP + (size_t)(l - 1) * prep_layer_stride(HP); }
PINN_HD size_t prep_wtf(int HP, int l) { return prep_wf(HP, l) + (size_t)HP * HP; }
PINN_HD size_t prep_b(int HP, int l) { return prep_wf(HP, l) + (size_t)2 * HP * HP; }
PINN_HD size_t prep_wout(int HP, int L) { return (size_t)3 * HP + (size_t)(L - 1) * prep_layer_stride(HP); }
PINN_HD size_t prep_bout(int HP, int L) { return prep_wout(HP, L) + (size_t)4 * HP; }
PINN_HD size_t prep_total(int HP, int L) { return prep_bout(HP, L) + 4; }

// ---- per-workgroup small-gradient accumulator (floats) ----------------------
//   [db_l HP] for l = 0..L-1 | [dW0x HP][dW0y HP] | [dWout 4*HP] | [dbout 4]
PINN_HD int sg_db(int HP, int l) { return l * HP; }
PINN_HD int sg_w0x(int HP, int L) { return L * HP; }
PINN_HD int sg_w0y(int HP, int L) { return L * HP + HP; }
PINN_HD int sg_wout(int HP, int L) { return L * HP + 2 * HP; }
PINN_HD int sg_bout(int HP, int L) { return L * HP + 6 * HP; }
PINN_HD int sg_total(int HP, int L) { return L * HP + 6 * HP + 4; }

// row of a 32x32 MFMA accumulator register r (0..15) for lane half h (lane>>5)
PINN_HD int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// number of floats in one (tile, layer) activation block
PINN_HD size_t act_block(int HP) { return (size_t)HP * PINN_TILE_COLS; }

// flat (state_dict order) parameter offsets: layer_0.weight (H,2), layer_0.bias (H),
// layer_l.weight (H,H), layer_l.bias (H) ..., layer_L.weight (n_out,H), layer_L.bias (n_out)
PINN_HD size_t flat_w(int H, int l) { return l == 0 ? 0 : (size_t)3 * H + (size_t)(l - 1) * ((size_t)H * H + H); }
PINN_HD size_t flat_b(int H, int l, int L, int n_out) {
  (void)n_out;
  return flat_w(H, l) + (l == 0 ? (size_t)2 * H : (l == L ? (size_t)n_out * H : (size_t)H * H));
}
PINN_HD size_t flat_total(int H, int L, int n_out) { return flat_b(H, L, L, n_out) + n_out; }
