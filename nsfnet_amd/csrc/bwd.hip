// Fused all-layer reverse pass of the 4-stream (or value-only) forward over one tile.
//
// Replaces (reference, latteine1217/NSFnet) the part of loss.backward()
// (NSFnet/pinn_solver.py:252, ev-NSFnet/pinn_solver.py:469) that walks the
// double-backward graph of the MLP: here the adjoints of the four streams are pulled
// back layer by layer with G_{l-1} = W_l^T Zb_l on v_mfma_f32_32x32x2_f32, the tanh
// adjoint is lane-local on the accumulators, the z-adjoints Zb_l are spilled once for
// the weight-gradient GEMM (dw.hip), and the skinny gradients (all biases, layer 0,
// output layer) are reduced in-kernel.
#include "kernels.h"
#include "point_stage.h"
#include "reduce_util.h"

template <int HP, int NS>
__global__ __launch_bounds__(HP * 2) void bwd_kernel(BwdArgs a) {
  constexpr int NT = HP * 2;
  extern __shared__ float lds[];
  float* X = lds;                 // [HP][128]
  float* oadjL = X + HP * 128;    // [4][128]
  float* sgacc = oadjL + 4 * 128; // [sg_total]
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ob = w * 32;
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * (NS == 4 ? 32 : 128);
  const int SG = sg_total(HP, L);
  for (int i = tid; i < SG; i += NT) sgacc[i] = 0.f;
  float dbo[3] = {0.f, 0.f, 0.f};
  __syncthreads();

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    // ---------------- output adjoints per column (point_stage.h) ----------------
    float px[4], py[4];
    output_adjoint_stage<32, 128, NS, NT, 4>(a, tile, tid, col, col, npad, oadjL, dbo, px, py);
    __syncthreads();
    // ---------------- adjoint of the last hidden layer's activations (3 -> HP, rank-3) ----------------
    f32x16 acc[4];
    float oa[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) oa[c][j] = oadjL[c * 128 + 32 * j + col];
    {
      const float* wo = P + prep_wout(HP, L);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = ob + mfma_row(r, h);
        float w0 = wo[o], w1 = wo[HP + o], w2 = wo[2 * HP + o];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][r] = w0 * oa[0][j] + w1 * oa[1][j] + w2 * oa[2][j];
      }
    }
    for (int l = L - 1; l >= 0; --l) {
      const float* Sl = a.S + ((size_t)tile * L + l) * act_block(HP);
      float* Zl = a.Zb + ((size_t)tile * L + l) * act_block(HP);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * 32 + col);
        const f32x4* S4 = reinterpret_cast<const f32x4*>(Sl);
        f32x4 s0, s1, s2, s3;
        if (NS == 4 && l == 0 && a.s0_skip) {      // not spilled: the forward's own fmaf chain and tanhf, bit for bit
          const int o0 = ob + 8 * g + 4 * h;
          s1 = *reinterpret_cast<const f32x4*>(P + prep_w0x(HP) + o0); s2 = *reinterpret_cast<const f32x4*>(P + prep_w0y(HP) + o0);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(P + prep_b0(HP) + o0);
#pragma unroll
          for (int e = 0; e < 4; ++e) s0[e] = tanhf(fmaf(s1[e], px[0], fmaf(s2[e], py[0], b4[e])));
          s3 = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
          s0 = __builtin_nontemporal_load(pin_base(S4 + 0 * (HP / 4) * 32) + so); s1 = __builtin_nontemporal_load(pin_base(S4 + 1 * (HP / 4) * 32) + so);
          s2 = __builtin_nontemporal_load(pin_base(S4 + 2 * (HP / 4) * 32) + so); s3 = __builtin_nontemporal_load(pin_base(S4 + 3 * (HP / 4) * 32) + so);
        }
        f32x4 z0, z1, z2, z3;
        float dbq[4], wq0[4], wq1[4], wq2[4], dxq[4], dyq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int o = ob + 8 * g + 4 * h + e;
          float zb, dbv, dwx, dwy;
          float wo0 = 0.f, wo1 = 0.f, wo2 = 0.f;
          if (NS == 4) {
            float t = s0[e], zx = s1[e], zy = s2[e], zd = s3[e];
            float d1 = 1.f - t * t;
            float d2 = -2.f * t * d1;
            float d3 = -2.f * d1 * (1.f - 3.f * t * t);
            float ga = acc[0][r], gx = acc[1][r], gy = acc[2][r], gd = acc[3][r];
            if (l == L - 1) {   // dWout[c][o] += sum_s oadj[c][s] * a_s[o]
              float ax = d1 * zx, ay = d1 * zy, ad = d2 * (zx * zx + zy * zy) + d1 * zd;
              wo0 = oa[0][0] * t + oa[0][1] * ax + oa[0][2] * ay + oa[0][3] * ad;
              wo1 = oa[1][0] * t + oa[1][1] * ax + oa[1][2] * ay + oa[1][3] * ad;
              wo2 = oa[2][0] * t + oa[2][1] * ax + oa[2][2] * ay + oa[2][3] * ad;
            }
            float zbx = d1 * gx + 2.f * d2 * zx * gd;
            float zby = d1 * gy + 2.f * d2 * zy * gd;
            float zbd = d1 * gd;
            zb = d1 * ga + d2 * (zx * gx + zy * gy) + (d3 * (zx * zx + zy * zy) + d2 * zd) * gd;
            z0[e] = zb; z1[e] = zbx; z2[e] = zby; z3[e] = zbd;
            dbv = zb;
            dwx = zb * px[0] + zbx;
            dwy = zb * py[0] + zby;
          } else {
            float zq[4];
            dbv = 0.f; dwx = 0.f; dwy = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float t = (j == 0 ? s0[e] : j == 1 ? s1[e] : j == 2 ? s2[e] : s3[e]);
              if (l == L - 1) { wo0 += oa[0][j] * t; wo1 += oa[1][j] * t; wo2 += oa[2][j] * t; }
              zq[j] = (1.f - t * t) * acc[j][r];
              dbv += zq[j]; dwx += zq[j] * px[j]; dwy += zq[j] * py[j];
            }
            z0[e] = zq[0]; z1[e] = zq[1]; z2[e] = zq[2]; z3[e] = zq[3];
          }
          dbq[e] = dbv; wq0[e] = wo0; wq1[e] = wo1; wq2[e] = wo2; dxq[e] = dwx; dyq[e] = dwy;
          if (l != 0) {
            float* Xo = X + o * 128 + col;
            Xo[0] = z0[e]; Xo[32] = z1[e]; Xo[64] = z2[e]; Xo[96] = z3[e];
          }
        }
        // skinny gradients: the four features' sums over the 32 columns of this half-wave at once
        // (reduce_util.h: transposing butterfly); lane col == e commits feature e
        {
          const int o = ob + 8 * g + 4 * h + (col & 3);
          const float dbv = sum_cols4<32>(dbq[0], dbq[1], dbq[2], dbq[3], lane);
          float w0 = 0.f, w1 = 0.f, w2 = 0.f, dx = 0.f, dy = 0.f;
          if (l == L - 1) {
            w0 = sum_cols4<32>(wq0[0], wq0[1], wq0[2], wq0[3], lane);
            w1 = sum_cols4<32>(wq1[0], wq1[1], wq1[2], wq1[3], lane);
            w2 = sum_cols4<32>(wq2[0], wq2[1], wq2[2], wq2[3], lane);
          }
          if (l == 0) {
            dx = sum_cols4<32>(dxq[0], dxq[1], dxq[2], dxq[3], lane);
            dy = sum_cols4<32>(dyq[0], dyq[1], dyq[2], dyq[3], lane);
          }
          if (col < 4) {
            lds_add(&sgacc[sg_db(HP, l) + o], dbv);
            if (l == L - 1) {
              lds_add(&sgacc[sg_wout(HP, L) + o], w0);
              lds_add(&sgacc[sg_wout(HP, L) + HP + o], w1);
              lds_add(&sgacc[sg_wout(HP, L) + 2 * HP + o], w2);
            }
            if (l == 0) { lds_add(&sgacc[sg_w0x(HP, L) + o], dx); lds_add(&sgacc[sg_w0y(HP, L) + o], dy); }
          }
        }
        if (l > 0) {
          const f32x4* Z4 = reinterpret_cast<const f32x4*>(Zl);
          __builtin_nontemporal_store(z0, pin_base(Z4 + 0 * (HP / 4) * 32) + so); __builtin_nontemporal_store(z1, pin_base(Z4 + 1 * (HP / 4) * 32) + so);
          __builtin_nontemporal_store(z2, pin_base(Z4 + 2 * (HP / 4) * 32) + so); __builtin_nontemporal_store(z3, pin_base(Z4 + 3 * (HP / 4) * 32) + so);
        }
      }
      if (l == 0) break;
      __syncthreads();
      // ------------- G_{l-1}[i][col] = sum_o W_l[o][i] Zb_l[o][col] -------------
      {
        const f32x4* wf = reinterpret_cast<const f32x4*>(P + prep_wtf(HP, l)) + (size_t)w * (HP / 8) * 64 + lane;
        f32x4 wq[HP / 8];
#pragma unroll
        for (int q = 0; q < HP / 8; ++q) wq[q] = wf[q * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const float* Xr = X + h * 128 + col;
#pragma unroll
        for (int q = 0; q < HP / 8; ++q) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float* xk = Xr + (2 * (4 * q + e)) * 128;
            float b0 = xk[0], b1 = xk[32], b2 = xk[64], b3 = xk[96];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q][e], b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q][e], b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q][e], b2, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[q][e], b3, acc[3], 0, 0, 0);
          }
        }
      }
      __syncthreads();
    }
    __syncthreads();
  }
  // ---------------- flush ----------------
  // output-bias gradient: per-thread partials -> fixed-order sum
  float* red = X;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 3; ++c) red[c * NT + tid] = dbo[c];
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

size_t bwd_lds_bytes(int HP, int L) { return ((size_t)HP * 128 + 4 * 128 + sg_total(HP, L)) * sizeof(float); }

template <int HP, int NS>
static int launch_one(const BwdArgs& a, int grid, hipStream_t s) {
  size_t lds = bwd_lds_bytes(HP, a.L);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_kernel<HP, NS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((bwd_kernel<HP, NS>), dim3(grid), dim3(HP * 2), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

#define BWD_CASE(hp)                                                        \
  case hp:                                                                  \
    return NS == 4 ? launch_one<hp, 4>(a, grid, s) : launch_one<hp, 1>(a, grid, s);

int launch_bwd(int HP, int NS, const BwdArgs& a, int grid, hipStream_t s) {
  switch (HP) {
    BWD_CASE(32) BWD_CASE(64) BWD_CASE(96) BWD_CASE(128)
    BWD_CASE(160) BWD_CASE(192) BWD_CASE(224) BWD_CASE(256)
    default: return -1000;
  }
}
