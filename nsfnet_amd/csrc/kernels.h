// Internal kernel argument blocks and launchers (gfx950).  Not part of the C ABI;
// the exported surface is include/nsfnet_pinn.h.
#pragma once
#include <hip/hip_runtime.h>
#include "layout.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef __HIPCC__
typedef __attribute__((address_space(1))) f32x4 gf32x4;
// A uniform (tile, layer, plane) base pinned to scalar registers: lane addresses then cost ONE VALU (base +
// 32-bit offset) instead of a 64-bit add pair per plane - the epilogues pay full issue time for every VALU.
__device__ __forceinline__ gf32x4* pin_base(const void* q) {
  const unsigned long long b = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return (gf32x4*)(((unsigned long long)hi << 32) | lo);
}
#endif

// Field planes written by the residual forward and read by the backward
// (plane stride = padded point count).
enum { FLD_U = 0, FLD_V, FLD_UX, FLD_UY, FLD_VX, FLD_VY, FLD_EQ1, FLD_EQ2, FLD_EQ3, FLD_EQ4, FLD_P, FLD_COUNT };

struct FwdArgs {
  const float* x; const float* y;
  int n;            // real points
  int ntiles;       // tiles of 32 (residual) / 128 (value) points
  int L;            // hidden layers
  int n_out;        // 3 (u,v,p) or 1 (e)
  const float* prep;
  float* S;         // [tile][L][HP*128] saved (t, z_x, z_y, z_D) or null
  // residual mode (4 streams)
  float* fld;       // [FLD_COUNT][npad]
  const float* e;   // entropy-net output per point or null
  const float* w;   // per-point weights or null
  float* vtm;       // vis_t_minus state (in/out) or null
  float* vis_used;  // artificial viscosity used this step (out) or null
  float inv_re, vis_t0, alpha_evm, scale;
  // value mode (1 stream)
  float* pred[3];        // optional prediction planes
  const float* tgt[3];   // optional targets (NaN target = masked)
  float* oadj;           // [4][npad] output adjoints (written when non-null)
  float coef[3];         // oadj_c = coef[c] * (pred_c - tgt_c)
  float* partials;       // [grid][PINN_NLOSS]
  int stagger;           // start offset unit (x 4096 cycles x (block*5 mod 8)); 0 = off
  int configure;         // 1: do not launch, only raise the kernel's dynamic-LDS limit (pinn_plan_create)
  int s24;               // wide bf16 residual kernels: S / Z-bar in the 24-bit three-plane spill format (bf16_util.h pack24)
  // role-split sweeps (compact spill): block (tile, layer l) of S / Z-bar starts at ((size_t)tile * (L - sl0) + (l - sl0)) * sblk
  // floats - only what the sweeps write is allocated: three 16-byte planes, no layer 0 (sl0 = 1)
  int sl0; size_t sblk;
  // fp32 kernels, residual mode: layer 0's saved activations are not spilled - (tanh(w0x x + w0y y + b0), w0x, w0y, 0)
  // are recomputed where they are read (the reverse sweep's layer-0 epilogue, the layer-1 workgroups of the dW kernel)
  // with the forward's own fmaf chain and tanhf: the same values, a sixth of S at six layers neither written nor read twice
  int s0_skip;
};
// float offset of the spill block of (tile, layer l); `sblk` == 0 selects the classic [tile][L][HP x columns] layout
PINN_HD size_t spill_off(int tile, int l, int L, int sl0, size_t sblk, size_t classic) {
  return sblk ? ((size_t)tile * (L - sl0) + (l > sl0 ? l - sl0 : 0)) * sblk : ((size_t)tile * L + l) * classic;
}

struct BwdArgs {
  const float* x; const float* y;
  int n, ntiles, L, n_out;
  const float* prep;
  const float* S;
  float* Zb;             // [tile][L][HP*128] z-adjoints (layers 1..L-1 used)
  // residual mode
  const float* fld; const float* e; const float* w; const float* vis_used;
  float coef_eq[4];      // 2*alpha_e*c_k/N_total
  float inv_re, scale;
  float* ebar;           // d loss / d e per point (out) or null
  // value mode
  const float* oadj;     // [4][npad]
  float* sg;             // [grid][sg_total]
  int configure;         // see FwdArgs
  int s24;               // see FwdArgs
  int sl0; size_t sblk;  // see FwdArgs
  int s0_skip;           // see FwdArgs
};

struct DwArgs {
  const float* S; const float* Zb;
  int ntiles, L, groups;
  float* slabs;          // [(L-1)][groups][HP*HP]
  int configure;         // see FwdArgs
  // Layer-0 activations recomputed instead of read (the role-split sweeps do not spill them: they are one FMA and one
  // tanh of the point, fwd_bf16_split.hip): the points, the prepared parameters (w0x | w0y | b0 lead them) and n.
  int s0_skip;
  int s24;               // see FwdArgs (dw_bf16_wide)
  const float* x; const float* y; const float* prep; int n;
  int sl0; size_t sblk;  // see FwdArgs
};

struct ReduceSrc { const float* slabs; int groups; const float* sg; int nwg; };
struct ReduceArgs {
  ReduceSrc src[4]; int nsrc;
  int H, HP, L, n_out;
  float* grads; int accumulate;
};

int launch_fwd(int HP, int NS, const FwdArgs& a, int grid, hipStream_t s);
int launch_bwd(int HP, int NS, const BwdArgs& a, int grid, hipStream_t s);
int launch_dw(int HP, int NS, const DwArgs& a, hipStream_t s);
size_t fwd_lds_bytes(int HP);
size_t bwd_lds_bytes(int HP, int L);
size_t dw_lds_bytes(int HP);
int dw_threads(int HP);

int launch_prep(const float* params, float* prep, int H, int HP, int L, int n_out, int prec_fwd, int prec_bwd,
                hipStream_t s);
// wide nets (256 < HP <= 512): 64-column tiles, fp32 MFMA only
int launch_fwd_wide(int HP, int NS, const FwdArgs& a, int grid, hipStream_t s);
int launch_bwd_wide(int HP, int NS, const BwdArgs& a, int grid, hipStream_t s);
int launch_dw_wide(int HP, int NS, const DwArgs& a, hipStream_t s);
size_t fwd_wide_lds_bytes(int HP);
size_t bwd_wide_lds_bytes(int HP, int L);
size_t dw_wide_lds_bytes();
// bf16 MFMA variants (terms = 3: bf16x3 split, terms = 1: plain bf16)
// cols = 128 (32-point tiles) or 64 (16-point tiles, HP 128/256 only)
int launch_fwd_bf16(int HP, int NS, int terms, int cols, const FwdArgs& a, int grid, hipStream_t s);
int launch_bwd_bf16(int HP, int NS, int terms, int cols, const BwdArgs& a, int grid, hipStream_t s);
int launch_dw_bf16(int HP, int NS, int terms, int cols, const DwArgs& a, hipStream_t s);
size_t fwd_bf16_lds_bytes(int HP, int L, int cols);
size_t bwd_bf16_lds_bytes(int HP, int L, int cols);
size_t dw_bf16_lds_bytes(int HP);
// hidden > 256: 64-column tiles, two 32-feature blocks per wave (fwd_bf16_wide.hip / bwd_bf16_wide.hip)
int launch_fwd_bf16_wide(int HP, int NS, int terms, const FwdArgs& a, int grid, hipStream_t s);
int launch_bwd_bf16_wide(int HP, int NS, int terms, const BwdArgs& a, int grid, hipStream_t s);
int launch_dw_bf16_wide(int HP, int NS, int terms, const DwArgs& a, hipStream_t s);
size_t dw_bf16_wide_lds_bytes();
size_t fwd_bf16_wide_lds_bytes(int HP, int L);
size_t bwd_bf16_wide_lds_bytes(int HP, int L);
// one wave per SIMD, two tiles in flight, chain rule in the MFMA shadow (fwd_bf16_pipe.hip): HP = 256, residual mode
int launch_fwd_pipe(int HP, int terms, const FwdArgs& a, int grid, hipStream_t s);
size_t fwd_pipe_lds_bytes(int HP, int L);
// two wave groups in opposite phases (fwd_bf16_split.hip): HP = 256, residual mode
int launch_fwd_split(int HP, int terms, const FwdArgs& a, int grid, hipStream_t s);
size_t fwd_split_lds_bytes(int HP, int L);
int launch_bwd_split(int HP, int terms, const BwdArgs& a, int grid, hipStream_t s);
size_t bwd_split_lds_bytes(int HP, int L);
int launch_bwd_pipe(int HP, int terms, const BwdArgs& a, int grid, hipStream_t s);
size_t bwd_pipe_lds_bytes(int HP, int L);
// wide nets (256 < HP <= 448), residual mode, 24-bit spill: the role-split schedule at 64-column tiles (fwd_bf16_wsplit.hip)
int launch_fwd_wsplit(int HP, int terms, const FwdArgs& a, int grid, hipStream_t s);
size_t fwd_wsplit_lds_bytes(int HP);
int launch_bwd_wsplit(int HP, int terms, const BwdArgs& a, int grid, hipStream_t s);
size_t bwd_wsplit_lds_bytes(int HP, int L);
int launch_reduce(const ReduceArgs& a, hipStream_t s);
int launch_loss_sums(const float* partials, int nparts, float* out, hipStream_t s);
int launch_adam_dev(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                    float eps, long long* step_counter, hipStream_t s);
int launch_adam(float* p, const float* g, float* m, float* v, long n, float step_size, float b1, float b2,
                float eps, float bc2_sqrt, hipStream_t s);
