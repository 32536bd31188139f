// Per-point stages shared by every forward kernel (fwd / fwd_wide / fwd_bf16 / fwd_bf16_wide).
// `outv` is the tile's output-layer result in LDS, [3 outputs][COLS] with column = stream * PPL + point in
// residual mode (PPL points x 4 streams) and column = point in value mode (COLS points).
#pragma once
#include "kernels.h"

// Residual mode: NS-residual assembly, lagged artificial viscosity, field planes, loss partial sums.
// Replaces NSFnet/pinn_solver.py:132-163,212-222 and ev-NSFnet/pinn_solver.py:290-342,372-428.
template <int PPL, int COLS>
__device__ __forceinline__ void residual_point_stage(const FwdArgs& a, const float* outv, int tile, int tid, int npad,
                                                     float (&lsum)[4]) {
  if (tid >= PPL) return;
  const int pt = tile * PPL + tid;
  const bool m = pt < a.n;
  const float sc = a.scale, sc2 = a.scale * a.scale;
  float u = outv[tid], ux = outv[PPL + tid] * sc, uy = outv[2 * PPL + tid] * sc, ud = outv[3 * PPL + tid] * sc2;
  float v = outv[COLS + tid], vx = outv[COLS + PPL + tid] * sc, vy = outv[COLS + 2 * PPL + tid] * sc,
        vd = outv[COLS + 3 * PPL + tid] * sc2;
  float p = outv[2 * COLS + tid], pxx = outv[2 * COLS + PPL + tid] * sc, pyy = outv[2 * COLS + 2 * PPL + tid] * sc;
  float vt = 0.f;
  float ev = (a.e && m) ? a.e[pt] : 0.f;
  if (a.vtm && m) {                       // ev-NSFnet/pinn_solver.py:327-334, kept on the device
    vt = fminf(a.vis_t0, a.vtm[pt]);
    a.vtm[pt] = a.alpha_evm * fabsf(ev);
  }
  if (a.vis_used && m) a.vis_used[pt] = vt;
  float nu = a.inv_re + vt;
  float eq1 = (u * ux + v * uy) + pxx - nu * ud;
  float eq2 = (u * vx + v * vy) + pyy - nu * vd;
  float eq3 = ux + vy;
  float eq4 = a.e ? (eq1 * (u - 0.5f) + eq2 * (v - 0.5f)) - ev : 0.f;
  float* f = a.fld + pt;
  f[FLD_U * (size_t)npad] = u; f[FLD_V * (size_t)npad] = v;
  f[FLD_UX * (size_t)npad] = ux; f[FLD_UY * (size_t)npad] = uy;
  f[FLD_VX * (size_t)npad] = vx; f[FLD_VY * (size_t)npad] = vy;
  f[FLD_EQ1 * (size_t)npad] = eq1; f[FLD_EQ2 * (size_t)npad] = eq2;
  f[FLD_EQ3 * (size_t)npad] = eq3; f[FLD_EQ4 * (size_t)npad] = eq4;
  f[FLD_P * (size_t)npad] = p;
  if (m) {
    float ww = a.w ? a.w[pt] : 1.f;
    lsum[0] += ww * eq1 * eq1; lsum[1] += ww * eq2 * eq2;
    lsum[2] += ww * eq3 * eq3; lsum[3] += ww * eq4 * eq4;
  }
}

// Value mode: predictions, squared errors against (possibly masked) targets, output adjoints.
// Replaces the BC / supervised MSE terms (NSFnet/pinn_solver.py:199-207, ev-NSFnet/pinn_solver.py:399-411).
template <int COLS, int NT>
__device__ __forceinline__ void value_point_stage(const FwdArgs& a, const float* outv, int tile, int tid, int npad,
                                                  float (&lsum)[4]) {
  for (int idx = tid; idx < COLS; idx += NT) {
    const int pt = tile * COLS + idx;
    const bool m = pt < a.n;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (c >= a.n_out) break;
      float pv = outv[c * COLS + idx];
      if (a.pred[c] && m) a.pred[c][pt] = pv;
      float adj = 0.f;
      if (a.tgt[c] && m) {
        float t = a.tgt[c][pt];
        if (t == t && fabsf(t) <= 3.0e38f) {   // finite target (NaN pressure = masked, ev:405-410)
          float d = pv - t;
          lsum[c] += d * d;
          lsum[3] += (c == 2) ? 1.f : 0.f;     // count of valid pressure targets
          adj = a.coef[c] * d;
        }
      }
      if (a.oadj) a.oadj[(size_t)c * npad + pt] = adj;
    }
  }
}

// Reverse sweep, start of a tile: adjoints of the three network outputs per MFMA column into LDS `oadjL`
// ([3 outputs][COLS], same column order as `outv` above), this lane's coordinates (px, py: the layer-0 weight
// gradient needs them), the output-bias gradient partials `dbo`, and d loss / d e (ebar).  The residual-mode
// seeds are the hand-derived derivatives of alpha_e * sum_k c_k w eq_k^2 with respect to (u, u_x, u_y, u_D, v, ...,
// p_x, p_y) - what loss.backward() (NSFnet/pinn_solver.py:252, ev-NSFnet/pinn_solver.py:469) propagates into the
// output layer; value mode copies the adjoints the value forward wrote.  px[j] / py[j]: point of column
// 32 j + col (value mode) or of column pp (residual mode, j = 0).
template <int PPL, int COLS, int NS, int NT, int NJ>
__device__ __forceinline__ void output_adjoint_stage(const BwdArgs& a, int tile, int tid, int col, int pp, int npad,
                                                     float* oadjL, float (&dbo)[3], float (&px)[NJ], float (&py)[NJ]) {
  if (NS == 4) {
    const int ptc = tile * PPL + pp;
    px[0] = ptc < a.n ? a.x[ptc] : 0.f;
    py[0] = ptc < a.n ? a.y[ptc] : 0.f;
    if (tid < PPL) {
      const int pt = tile * PPL + tid;
      const bool m = pt < a.n;
      const float* f = a.fld + pt;
      float u = f[FLD_U * (size_t)npad], v = f[FLD_V * (size_t)npad];
      float ux = f[FLD_UX * (size_t)npad], uy = f[FLD_UY * (size_t)npad];
      float vx = f[FLD_VX * (size_t)npad], vy = f[FLD_VY * (size_t)npad];
      float eq1 = f[FLD_EQ1 * (size_t)npad], eq2 = f[FLD_EQ2 * (size_t)npad];
      float eq3 = f[FLD_EQ3 * (size_t)npad], eq4 = f[FLD_EQ4 * (size_t)npad];
      float ww = m ? (a.w ? a.w[pt] : 1.f) : 0.f;
      float g1 = a.coef_eq[0] * ww * eq1, g2 = a.coef_eq[1] * ww * eq2, g3 = a.coef_eq[2] * ww * eq3;
      float g4 = a.e ? a.coef_eq[3] * ww * eq4 : 0.f;
      float r1 = g1 + g4 * (u - 0.5f), r2 = g2 + g4 * (v - 0.5f), r3 = g3;
      float nu = a.inv_re + ((a.vis_used && m) ? a.vis_used[pt] : 0.f);
      const float sc = a.scale, sc2 = a.scale * a.scale;
      float au = r1 * ux + r2 * vx + g4 * eq1;
      float av = r1 * uy + r2 * vy + g4 * eq2;
      oadjL[0 * COLS + 0 * PPL + tid] = au;
      oadjL[0 * COLS + 1 * PPL + tid] = (r1 * u + r3) * sc;
      oadjL[0 * COLS + 2 * PPL + tid] = (r1 * v) * sc;
      oadjL[0 * COLS + 3 * PPL + tid] = -nu * r1 * sc2;
      oadjL[1 * COLS + 0 * PPL + tid] = av;
      oadjL[1 * COLS + 1 * PPL + tid] = (r2 * u) * sc;
      oadjL[1 * COLS + 2 * PPL + tid] = (r2 * v + r3) * sc;
      oadjL[1 * COLS + 3 * PPL + tid] = -nu * r2 * sc2;
      oadjL[2 * COLS + 0 * PPL + tid] = 0.f;
      oadjL[2 * COLS + 1 * PPL + tid] = r1 * sc;
      oadjL[2 * COLS + 2 * PPL + tid] = r2 * sc;
      oadjL[2 * COLS + 3 * PPL + tid] = 0.f;
      if (a.ebar && m) a.ebar[pt] = -g4;
      dbo[0] += au; dbo[1] += av;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int pt = tile * COLS + 32 * j + col;
      px[j] = pt < a.n ? a.x[pt] : 0.f;
      py[j] = pt < a.n ? a.y[pt] : 0.f;
    }
    for (int idx = tid; idx < 3 * COLS; idx += NT) {
      int c3 = idx / COLS, cc = idx % COLS;
      int pt = tile * COLS + cc;
      float v = (c3 < a.n_out && pt < a.n) ? a.oadj[(size_t)c3 * npad + pt] : 0.f;
      oadjL[idx] = v;
      if (c3 == 0) dbo[0] += v; else if (c3 == 1) dbo[1] += v; else dbo[2] += v;
    }
  }
}
