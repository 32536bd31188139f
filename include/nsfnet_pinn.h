/*
 * nsfnet_pinn.h - C ABI of the MI355X (gfx950) PINN training-step library.
 *
 * The reference (latteine1217/NSFnet) has no FFI: its hot path is Python calling
 * torch autograd.  This header is the boundary a maintainer binds instead of that
 * path; every entry point names the reference code it replaces.  All pointers are
 * DEVICE pointers owned by the caller (e.g. PyTorch's allocator); the library never
 * allocates or frees device memory, launches only on the stream it is given, spawns
 * no threads and keeps no global mutable state besides the last-error string.
 * Return value: 0 on success, negative on error (see pinn_last_error()).
 *
 * Parameters of a network are ONE flat fp32 vector in torch state_dict order of the
 * reference FCNet (NSFnet/net.py:36-46):
 *   layers.layer_0.weight (H,2) | layers.layer_0.bias (H) | layers.layer_l.weight (H,H) |
 *   layers.layer_l.bias (H) ... | layers.layer_L.weight (n_out,H) | layers.layer_L.bias (n_out)
 * so reference checkpoints map onto it by concatenation.
 */
#ifndef NSFNET_PINN_H
#define NSFNET_PINN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pinn_net_s* pinn_net_t;
typedef struct pinn_plan_s* pinn_plan_t;

/* field planes written by pinn_residual_forward (plane stride = pinn_plan_padded_points) */
enum {
  PINN_FLD_U = 0, PINN_FLD_V, PINN_FLD_UX, PINN_FLD_UY, PINN_FLD_VX, PINN_FLD_VY,
  PINN_FLD_EQ1, PINN_FLD_EQ2, PINN_FLD_EQ3, PINN_FLD_EQ4, PINN_FLD_P, PINN_FLD_COUNT
};
#define PINN_NLOSS 8

const char* pinn_last_error(void);
/* 3 = this header (2 + pinn_plan_kernel; 2 = 1 + pinn_adam_step_dev, bf16 modes for every width up to 512). */
int pinn_abi_version(void);

/* ---- network description --------------------------------------------------
 * Replaces FCNet.__init__ (NSFnet/net.py:23-50): 2 inputs, `n_hidden_layers` tanh
 * layers of width `hidden`, `n_out` linear outputs (3 = u,v,p ; 1 = entropy residual e).
 * hidden <= 512. */
int pinn_net_create(int n_out, int n_hidden_layers, int hidden, pinn_net_t* out);
int pinn_net_destroy(pinn_net_t net);
/* Arithmetic of the three MFMA kernel families (forward sweep, reverse sweep, weight-gradient
 * GEMM): 0 = f32-input MFMA (bit-exact fp32 fmaf chains, default), 1 = bf16x3 (fp32 operands
 * split into bf16 hi+lo, three bf16 MFMAs per product, fp32 accumulate; ~2^-17 relative per
 * product), 2 = plain bf16 operands (fast mode, does NOT meet the 1e-4 loss-parity bar).
 * All three modes exist for every supported width (hidden <= 512).  Call before pinn_net_prepare / pinn_plan_create. */
int pinn_net_set_precision(pinn_net_t net, int prec_fwd, int prec_bwd, int prec_dw);
int64_t pinn_net_num_params(pinn_net_t net);
int64_t pinn_net_prep_floats(pinn_net_t net);
/* Re-layout the flat parameters into padded MFMA-fragment order; call after every
 * parameter update.  (No reference counterpart: torch re-reads nn.Linear weights.) */
int pinn_net_prepare(pinn_net_t net, const float* params, float* prep, void* stream);

/* ---- plan: `n_points` points evaluated by `net` ------------------------------
 * streams = 4: residual mode (value, d/dx, d/dy, Laplacian)  - collocation points
 * streams = 1: value mode                                    - boundary / supervised /
 *                                                              evaluation points, entropy net */
int pinn_plan_create(pinn_net_t net, int64_t n_points, int streams, pinn_plan_t* out);
int pinn_plan_destroy(pinn_plan_t plan);
int64_t pinn_plan_padded_points(pinn_plan_t plan);
int64_t pinn_plan_workspace_bytes(pinn_plan_t plan, int with_backward);
/* Name of the kernel family the plan launches for `which` = 0 forward (with saved activations), 1 reverse sweep,
 * 2 weight-gradient GEMM - what appears in a rocprofv3 kernel trace (e.g. "fwd_pipe_kernel", "bwd_bf16_kernel").
 * For profiling harnesses (bench.py keys its roofline / PMC lookup on it); static string, never NULL for a valid plan. */
const char* pinn_plan_kernel(pinn_plan_t plan, int which);

/* ---- residual forward -------------------------------------------------------
 * Replaces neural_net_equations + the PDE half of fwd_computing_loss_2d
 * (NSFnet/pinn_solver.py:132-163,212-222; ev-NSFnet/pinn_solver.py:290-342,384-397).
 *   x,y          [n] collocation coordinates
 *   e            [n] entropy-net output (ev flavour) or NULL (plain NSFnet: eq4 = 0)
 *   w            [n] per-point weights (SDF weights, ev:387-392) or NULL
 *   vis_t_minus  [n] in/out lagged viscosity state alpha_evm*|e_prev| (ev:327-334) or NULL
 *   vis_t_out    [n] out: artificial viscosity min(vis_t0, vis_t_minus) used this call, or NULL
 *   fields       [PINN_FLD_COUNT][padded] out: u,v,u_x,u_y,v_x,v_y,eq1..eq4,p
 *   loss_sums    [PINN_NLOSS] out: sum_i w_i eq_k,i^2 for k=1..4 in slots 0..3
 *   save != 0 keeps the activations in `ws` for pinn_residual_backward. */
int pinn_residual_forward(pinn_plan_t plan, void* ws, const float* prep,
                          const float* x, const float* y, const float* e, const float* w,
                          float* vis_t_minus, float* vis_t_out, float* fields,
                          float Re, float vis_t0, float alpha_evm, float coord_scale,
                          int save, float* loss_sums, void* stream);

/* ---- residual backward ------------------------------------------------------
 * Replaces loss.backward() for the PDE loss (NSFnet/pinn_solver.py:252; ev:469):
 * d/dtheta of sum_k coef_eq[k]/2 * sum_i w_i eq_k,i^2, i.e. the caller passes
 * coef_eq[k] = 2*alpha_e*c_k/N_global (c = 1,1,1,0.1).  Partial gradients stay in `ws`
 * until pinn_grad_reduce.  ebar_out [n] (or NULL) receives d loss/d e.
 * coef_eq4 is a HOST array of 4 floats. */
int pinn_residual_backward(pinn_plan_t plan, void* ws, const float* prep,
                           const float* x, const float* y, const float* e, const float* w,
                           const float* vis_t, const float* fields, const float* coef_eq4,
                           float Re, float coord_scale, float* ebar_out, void* stream);

/* Same, restricted to phases (bit 0: adjoint sweep that spills the z-adjoints, bit 1:
 * weight-gradient GEMM); for per-kernel timing in bench.py.  phases = 3 is the call above. */
int pinn_residual_backward_phases(pinn_plan_t plan, void* ws, const float* prep,
                                  const float* x, const float* y, const float* e, const float* w,
                                  const float* vis_t, const float* fields, const float* coef_eq4,
                                  float Re, float coord_scale, float* ebar_out, int phases, void* stream);

/* ---- value forward / backward -----------------------------------------------
 * Replaces neural_net_u and the boundary / supervised MSE terms
 * (NSFnet/pinn_solver.py:124-130,199-207; ev:280-288,374-379,399-411) and, with
 * save = 0, the inference forward of evaluate/test (NSFnet/pinn_solver.py:308-357).
 *   pred3/tgt3/coef3 are HOST arrays of 3 entries (device pointers / floats):
 *   pred[c]  [n] out planes or NULL ; tgt[c] [n] targets or NULL (NaN target = masked)
 *   coef[c]  output adjoint scale: oadj_c = coef[c]*(pred_c - tgt_c), kept in ws
 *   loss_sums slots 0..2 = sum (pred_c - tgt_c)^2 over valid targets, slot 3 = number
 *   of valid targets of output 2. */
int pinn_value_forward(pinn_plan_t plan, void* ws, const float* prep,
                       const float* x, const float* y,
                       float* const* pred3, const float* const* tgt3, const float* coef3,
                       int save, float* loss_sums, void* stream);
/* out_adj: [n_out][padded] explicit output adjoints, or NULL to use the ones
 * pinn_value_forward left in ws. */
int pinn_value_backward(pinn_plan_t plan, void* ws, const float* prep,
                        const float* x, const float* y, const float* out_adj, void* stream);

/* ---- gradient assembly, optimizer -------------------------------------------
 * Sum the partial gradients of up to 4 (plan, ws) pairs of the SAME net into the flat
 * gradient (state_dict order), fixed summation order. */
int pinn_grad_reduce(pinn_net_t net, int nsrc, const pinn_plan_t* plans, void* const* wss,
                     float* grads, int accumulate, void* stream);
/* torch.optim.Adam step (NSFnet/pinn_solver.py:76-79,253; ev:126-129,472); step >= 1. */
int pinn_adam_step(float* params, const float* grads, float* m, float* v, int64_t n,
                   float lr, float beta1, float beta2, float eps, int64_t step, void* stream);

/* Same update with the step count in DEVICE memory: step_counter points at TWO int64 words, [0] = steps taken
 * so far, [1] = scratch that must be zero on entry (both zero to start / restart the schedule).  The call uses
 * t = step_counter[0] + 1 for the bias corrections and increments step_counter[0] when its last workgroup is
 * done, so a whole training step can be captured in a hipGraph and replayed with no host-side scalar changing
 * between steps. */
int pinn_adam_step_dev(float* params, const float* grads, float* m, float* v, int64_t n,
                       float lr, float beta1, float beta2, float eps, int64_t* step_counter, void* stream);

#ifdef __cplusplus
}
#endif
#endif
