// Fused reverse sweep for WIDE nets (256 < hidden <= 512), fp32-input MFMA.  Same algorithm
// and reference mapping as bwd.hip (loss.backward(), NSFnet/pinn_solver.py:252,
// ev-NSFnet/pinn_solver.py:469); tile geometry and the v_permlane16_swap stream exchange
// as in fwd_wide.hip.
#include "kernels.h"
#include "point_stage.h"
#include "reduce_util.h"

template <int HP, int NS>
__global__ __launch_bounds__(HP * 2) void bwd_wide_kernel(BwdArgs a) {
  constexpr int NT = HP * 2, COLS = 64, PPL = 16, NQ = HP / 8;
  constexpr int PRE = 4, RING = 8;
  extern __shared__ float lds[];
  float* X = lds;                    // [HP][64]
  float* oadjL = X + HP * COLS;      // [4][64]
  float* sgacc = oadjL + 4 * COLS;   // [sg_total]
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
  const int hi = c >> 4, pp = c & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ob = w * 32;
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * (NS == 4 ? PPL : COLS);
  const int SG = sg_total(HP, L);
  for (int i = tid; i < SG; i += NT) sgacc[i] = 0.f;
  float dbo[3] = {0.f, 0.f, 0.f};
  __syncthreads();

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    // ---------------- output adjoints per column (point_stage.h) ----------------
    float px[2], py[2];
    output_adjoint_stage<PPL, COLS, NS, NT, 2>(a, tile, tid, c, pp, npad, oadjL, dbo, px, py);
    __syncthreads();
    f32x16 acc[2];
    // output adjoints this lane needs: per accumulator tile (column 32j + c) and, in residual
    // mode, the four streams of its own point (for the output-layer weight gradient)
    float oc[3][2], oa[3][4];
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) {
#pragma unroll
      for (int j = 0; j < 2; ++j) oc[c3][j] = oadjL[c3 * COLS + 32 * j + c];
#pragma unroll
      for (int s = 0; s < 4; ++s) oa[c3][s] = oadjL[c3 * COLS + 16 * s + pp];
    }
    {
      const float* wo = P + prep_wout(HP, L);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = ob + mfma_row(r, h);
        float w0 = wo[o], w1 = wo[HP + o], w2 = wo[2 * HP + o];
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j][r] = w0 * oc[0][j] + w1 * oc[1][j] + w2 * oc[2][j];
      }
    }
    for (int l = L - 1; l >= 0; --l) {
      f32x4 wq[RING];
      const f32x4* wf = reinterpret_cast<const f32x4*>(P + prep_wtf(HP, l > 0 ? l : 1)) + (size_t)w * NQ * 64 + lane;
      if (l > 0) {
#pragma unroll
        for (int q = 0; q < PRE; ++q) wq[q] = wf[q * 64];
      }
      const float* Sl = a.S + ((size_t)tile * L + l) * ((size_t)HP * COLS);
      float* Zl = a.Zb + ((size_t)tile * L + l) * ((size_t)HP * COLS);
      if (NS == 4) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][q]), __float_as_uint(acc[0][q + 8]), false, false);
          auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[1][q]), __float_as_uint(acc[1][q + 8]), false, false);
          acc[0][q] = __uint_as_float(s01[0]); acc[0][q + 8] = __uint_as_float(s01[1]);
          acc[1][q] = __uint_as_float(s23[0]); acc[1][q + 8] = __uint_as_float(s23[1]);
        }
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          const int g = gq + 2 * hi;
          const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * PPL + pp);
          const f32x4* S4 = reinterpret_cast<const f32x4*>(Sl);
          f32x4 s0, s1, s2, s3;
          if (l == 0 && a.s0_skip) {      // not spilled: the forward's own fmaf chain and tanhf, bit for bit
            const int o0 = ob + 8 * g + 4 * h;
            s1 = *reinterpret_cast<const f32x4*>(P + prep_w0x(HP) + o0); s2 = *reinterpret_cast<const f32x4*>(P + prep_w0y(HP) + o0);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(P + prep_b0(HP) + o0);
#pragma unroll
            for (int e = 0; e < 4; ++e) s0[e] = tanhf(fmaf(s1[e], px[0], fmaf(s2[e], py[0], b4[e])));
            s3 = f32x4{0.f, 0.f, 0.f, 0.f};
          } else {
            s0 = __builtin_nontemporal_load(pin_base(S4 + 0 * (HP / 4) * PPL) + so); s1 = __builtin_nontemporal_load(pin_base(S4 + 1 * (HP / 4) * PPL) + so);
            s2 = __builtin_nontemporal_load(pin_base(S4 + 2 * (HP / 4) * PPL) + so); s3 = __builtin_nontemporal_load(pin_base(S4 + 3 * (HP / 4) * PPL) + so);
          }
          f32x4 z0, z1, z2, z3;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int q = 4 * gq + e;
            const int o = ob + 8 * g + 4 * h + e;
            float t = s0[e], zx = s1[e], zy = s2[e], zd = s3[e];
            float d1 = 1.f - t * t;
            float d2 = -2.f * t * d1;
            float d3 = -2.f * d1 * (1.f - 3.f * t * t);
            float ga = acc[0][q], gx = acc[0][q + 8], gy = acc[1][q], gd = acc[1][q + 8];
            float wo0 = 0.f, wo1 = 0.f, wo2 = 0.f;
            if (l == L - 1) {
              float ax = d1 * zx, ay = d1 * zy, ad = d2 * (zx * zx + zy * zy) + d1 * zd;
              wo0 = oa[0][0] * t + oa[0][1] * ax + oa[0][2] * ay + oa[0][3] * ad;
              wo1 = oa[1][0] * t + oa[1][1] * ax + oa[1][2] * ay + oa[1][3] * ad;
              wo2 = oa[2][0] * t + oa[2][1] * ax + oa[2][2] * ay + oa[2][3] * ad;
            }
            float zbx = d1 * gx + 2.f * d2 * zx * gd;
            float zby = d1 * gy + 2.f * d2 * zy * gd;
            float zbd = d1 * gd;
            float zb = d1 * ga + d2 * (zx * gx + zy * gy) + (d3 * (zx * zx + zy * zy) + d2 * zd) * gd;
            z0[e] = zb; z1[e] = zbx; z2[e] = zby; z3[e] = zbd;
            float dbv = sum16(zb);
            if (pp == 0) sgacc[sg_db(HP, l) + o] += dbv;
            if (l == L - 1) {
              wo0 = sum16(wo0); wo1 = sum16(wo1); wo2 = sum16(wo2);
              if (pp == 0) {
                sgacc[sg_wout(HP, L) + o] += wo0;
                sgacc[sg_wout(HP, L) + HP + o] += wo1;
                sgacc[sg_wout(HP, L) + 2 * HP + o] += wo2;
              }
            }
            if (l == 0) {
              float dwx = sum16(zb * px[0] + zbx), dwy = sum16(zb * py[0] + zby);
              if (pp == 0) { sgacc[sg_w0x(HP, L) + o] += dwx; sgacc[sg_w0y(HP, L) + o] += dwy; }
            } else {
              float* Xo = X + o * COLS + pp;
              Xo[0] = zb; Xo[16] = zbx; Xo[32] = zby; Xo[48] = zbd;
            }
          }
          if (l > 0) {
            const f32x4* Z4 = reinterpret_cast<const f32x4*>(Zl);
            __builtin_nontemporal_store(z0, pin_base(Z4 + 0 * (HP / 4) * PPL) + so); __builtin_nontemporal_store(z1, pin_base(Z4 + 1 * (HP / 4) * PPL) + so);
            __builtin_nontemporal_store(z2, pin_base(Z4 + 2 * (HP / 4) * PPL) + so); __builtin_nontemporal_store(z3, pin_base(Z4 + 3 * (HP / 4) * PPL) + so);
          }
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 zj[2];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const int o = ob + 8 * g + 4 * h + e;
            float dbv = 0.f, dwx = 0.f, dwy = 0.f, wo0 = 0.f, wo1 = 0.f, wo2 = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const f32x4* Sg = reinterpret_cast<const f32x4*>(Sl) + ((size_t)(2 * j + hi) * (HP / 4) + (ob >> 2) + 2 * g + h) * PPL + pp;
              float t = (*Sg)[e];
              if (l == L - 1) { wo0 += oc[0][j] * t; wo1 += oc[1][j] * t; wo2 += oc[2][j] * t; }
              float zq = (1.f - t * t) * acc[j][r];
              zj[j][e] = zq;
              dbv += zq; dwx += zq * px[j]; dwy += zq * py[j];
              if (l > 0) X[o * COLS + 32 * j + c] = zq;
            }
            dbv = sum32(dbv);
            if (c == 0) sgacc[sg_db(HP, l) + o] += dbv;
            if (l == L - 1) {
              wo0 = sum32(wo0); wo1 = sum32(wo1); wo2 = sum32(wo2);
              if (c == 0) {
                sgacc[sg_wout(HP, L) + o] += wo0;
                sgacc[sg_wout(HP, L) + HP + o] += wo1;
                sgacc[sg_wout(HP, L) + 2 * HP + o] += wo2;
              }
            }
            if (l == 0) {
              dwx = sum32(dwx); dwy = sum32(dwy);
              if (c == 0) { sgacc[sg_w0x(HP, L) + o] += dwx; sgacc[sg_w0y(HP, L) + o] += dwy; }
            }
          }
          if (l > 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              f32x4* Zg = reinterpret_cast<f32x4*>(Zl) + ((size_t)(2 * j + hi) * (HP / 4) + (ob >> 2) + 2 * g + h) * PPL + pp;
              *Zg = zj[j];
            }
          }
        }
      }
      if (l == 0) break;
      __syncthreads();
      {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const float* Xr = X + h * COLS + c;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (q + PRE < NQ) wq[(q + PRE) % RING] = wf[(q + PRE) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float* xk = Xr + (2 * (4 * q + e)) * COLS;
            float b0 = xk[0], b1 = xk[32];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q % RING][e], b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q % RING][e], b1, acc[1], 0, 0, 0);
          }
        }
      }
      __syncthreads();
    }
    __syncthreads();
  }
  float* red = X;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) red[k * NT + tid] = dbo[k];
  __syncthreads();
  if (tid < 3) {
    float s = 0.f;
    for (int t = 0; t < NT; ++t) s += red[tid * NT + t];
    sgacc[sg_bout(HP, L) + tid] = s;
  }
  __syncthreads();
  float* out = a.sg + (size_t)blockIdx.x * SG;
  for (int i = tid; i < SG; i += NT) out[i] = sgacc[i];
}

size_t bwd_wide_lds_bytes(int HP, int L) { return ((size_t)HP * 64 + 4 * 64 + sg_total(HP, L)) * sizeof(float); }

template <int HP, int NS>
static int launch_one(const BwdArgs& a, int grid, hipStream_t s) {
  size_t lds = bwd_wide_lds_bytes(HP, a.L);
  if (lds > 163840) return -1001;
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_wide_kernel<HP, NS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((bwd_wide_kernel<HP, NS>), dim3(grid), dim3(HP * 2), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

#define BWD_CASE(hp)                                                        \
  case hp:                                                                  \
    return NS == 4 ? launch_one<hp, 4>(a, grid, s) : launch_one<hp, 1>(a, grid, s);

int launch_bwd_wide(int HP, int NS, const BwdArgs& a, int grid, hipStream_t s) {
  switch (HP) {
    BWD_CASE(128) BWD_CASE(256) BWD_CASE(288) BWD_CASE(320) BWD_CASE(352) BWD_CASE(384)
    BWD_CASE(416) BWD_CASE(448) BWD_CASE(480) BWD_CASE(512)
    default: return -1000;
  }
}
