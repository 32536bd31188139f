// Fused all-layer forward for WIDE nets (256 < hidden <= 512), fp32-input MFMA.
// Same algorithm and reference mapping as fwd.hip (NSFnet/net.py:52-54,
// NSFnet/pinn_solver.py:132-163,197-226; ev-NSFnet/pinn_solver.py:290-342,372-428).
//
// What changes with the width: the activation tile must still fit LDS, so a tile is 64
// MFMA columns (16 points x 4 streams, column = stream*16 + point; value mode: 64 points)
// and a 32x32 accumulator tile holds TWO streams (lanes 0-15 / 16-31 of each half-wave).
// Before the lane-local tanh chain rule, v_permlane16_swap exchanges register halves so that
// every lane owns all four streams of 8 of the 16 accumulator rows.  With up to 16 waves per
// workgroup a wave has 128 VGPRs, so the weight slice streams through a small register ring.
#include "kernels.h"
#include "point_stage.h"

template <int HP, int NS>
__global__ __launch_bounds__(HP * 2) void fwd_wide_kernel(FwdArgs a) {
  constexpr int NW = HP / 32, NT = HP * 2, COLS = 64, PPL = 16, NQ = HP / 8;
  constexpr int PRE = 4, RING = 8;
  extern __shared__ float lds[];
  float* X = lds;                       // [HP][64]
  float* part = X + HP * COLS;          // [NW][4][64]
  float* outv = part + NW * 4 * COLS;   // [4][64]
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
  const int hi = c >> 4, pp = c & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ob = w * 32;
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * (NS == 4 ? PPL : COLS);
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    f32x16 acc[2];
    {
      float px[2], py[2];
      if (NS == 4) {
        int pt = tile * PPL + pp;
        px[0] = pt < a.n ? a.x[pt] : 0.f;
        py[0] = pt < a.n ? a.y[pt] : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          int pt = tile * COLS + 32 * j + c;
          px[j] = pt < a.n ? a.x[pt] : 0.f;
          py[j] = pt < a.n ? a.y[pt] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int o = ob + mfma_row(r, h);
        float wx = P[prep_w0x(HP) + o], wy = P[prep_w0y(HP) + o], b = P[prep_b0(HP) + o];
        if (NS == 4) {            // tile 0: streams (value | d/dx), tile 1: streams (d/dy | Laplacian)
          acc[0][r] = hi ? wx : fmaf(wx, px[0], fmaf(wy, py[0], b));
          acc[1][r] = hi ? 0.f : wy;
        } else {
          acc[0][r] = fmaf(wx, px[0], fmaf(wy, py[0], b));
          acc[1][r] = fmaf(wx, px[1], fmaf(wy, py[1], b));
        }
      }
    }
    for (int l = 0; l < L; ++l) {
      f32x4 wq[RING];
      const f32x4* wf = reinterpret_cast<const f32x4*>(P + prep_wf(HP, l + 1 < L ? l + 1 : 1)) + (size_t)w * NQ * 64 + lane;
      if (l < L - 1) {
#pragma unroll
        for (int q = 0; q < PRE; ++q) wq[q] = wf[q * 64];
      }
      float* Sl = a.S ? a.S + ((size_t)tile * L + l) * ((size_t)HP * COLS) : nullptr;
      if (NS == 4) {
        // lanes 0-15 keep register rows 0-7, lanes 16-31 rows 8-15; both get all four streams
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
          f32x4 s0, s1, s2, s3;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int q = 4 * gq + e;
            const int o = ob + 8 * g + 4 * h + e;
            float t = tanhf(acc[0][q]);
            float zx = acc[0][q + 8], zy = acc[1][q], zd = acc[1][q + 8];
            float d1 = 1.f - t * t;
            float d2 = -2.f * t * d1;
            float* Xo = X + o * COLS + pp;
            Xo[0] = t; Xo[16] = d1 * zx; Xo[32] = d1 * zy;
            Xo[48] = d2 * (zx * zx + zy * zy) + d1 * zd;
            s0[e] = t; s1[e] = zx; s2[e] = zy; s3[e] = zd;
          }
          if (Sl && !(l == 0 && a.s0_skip)) {      // (layer 0 is recomputed by its readers: FwdArgs::s0_skip)
            const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * PPL + pp);
            const f32x4* S4 = reinterpret_cast<const f32x4*>(Sl);
            __builtin_nontemporal_store(s0, pin_base(S4 + 0 * (HP / 4) * PPL) + so);
            __builtin_nontemporal_store(s1, pin_base(S4 + 1 * (HP / 4) * PPL) + so);
            __builtin_nontemporal_store(s2, pin_base(S4 + 2 * (HP / 4) * PPL) + so);
            __builtin_nontemporal_store(s3, pin_base(S4 + 3 * (HP / 4) * PPL) + so);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int plane = 2 * j + hi;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 s;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int o = ob + 8 * g + 4 * h + e;
              float t = tanhf(acc[j][4 * g + e]);
              X[o * COLS + 32 * j + c] = t;
              s[e] = t;
            }
            if (Sl) {
              f32x4* Sg = reinterpret_cast<f32x4*>(Sl) + ((size_t)plane * (HP / 4) + (ob >> 2) + 2 * g + h) * PPL + pp;
              *Sg = s;
            }
          }
        }
      }
      __syncthreads();
      if (l == L - 1) break;
      {
        const float* bl = P + prep_b(HP, l + 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float b = bl[ob + mfma_row(r, h)];
          if (NS == 4) { acc[0][r] = hi ? 0.f : b; acc[1][r] = 0.f; }
          else { acc[0][r] = b; acc[1][r] = b; }
        }
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
    // ---------------- output layer ----------------
    {
      float po[3] = {0.f, 0.f, 0.f};
      const float* wo = P + prep_wout(HP, L) + ob;
      for (int kk = 0; kk < 32; ++kk) {
        float x0 = X[(ob + kk) * COLS + lane];
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) po[c3] = fmaf(wo[c3 * HP + kk], x0, po[c3]);
      }
#pragma unroll
      for (int c3 = 0; c3 < 3; ++c3) part[(w * 4 + c3) * COLS + lane] = po[c3];
    }
    __syncthreads();
    for (int idx = tid; idx < 3 * COLS; idx += NT) {
      int c3 = idx >> 6, cc = idx & 63;
      float s = (NS == 1 || cc < PPL) ? P[prep_bout(HP, L) + c3] : 0.f;
      for (int ww = 0; ww < NW; ++ww) s += part[(ww * 4 + c3) * COLS + cc];
      outv[c3 * COLS + cc] = s;
    }
    __syncthreads();
    // ---------------- per-point stage (point_stage.h) ----------------
    if (NS == 4) residual_point_stage<PPL, COLS>(a, outv, tile, tid, npad, lsum);
    else value_point_stage<COLS, NT>(a, outv, tile, tid, npad, lsum);
    __syncthreads();
  }
  float* red = lds;
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k * NT + tid] = lsum[k];
  __syncthreads();
  if (tid < 4) {
    float s = 0.f;
    for (int t = 0; t < NT; ++t) s += red[tid * NT + t];
    a.partials[blockIdx.x * PINN_NLOSS + tid] = s;
  } else if (tid < PINN_NLOSS) {
    a.partials[blockIdx.x * PINN_NLOSS + tid] = 0.f;
  }
}

size_t fwd_wide_lds_bytes(int HP) { return ((size_t)HP * 64 + (size_t)(HP / 32) * 4 * 64 + 4 * 64) * sizeof(float); }

template <int HP, int NS>
static int launch_one(const FwdArgs& a, int grid, hipStream_t s) {
  size_t lds = fwd_wide_lds_bytes(HP);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_wide_kernel<HP, NS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((fwd_wide_kernel<HP, NS>), dim3(grid), dim3(HP * 2), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

#define FWD_CASE(hp)                                                        \
  case hp:                                                                  \
    return NS == 4 ? launch_one<hp, 4>(a, grid, s) : launch_one<hp, 1>(a, grid, s);

int launch_fwd_wide(int HP, int NS, const FwdArgs& a, int grid, hipStream_t s) {
  switch (HP) {
    FWD_CASE(128) FWD_CASE(256) FWD_CASE(288) FWD_CASE(320) FWD_CASE(352) FWD_CASE(384)
    FWD_CASE(416) FWD_CASE(448) FWD_CASE(480) FWD_CASE(512)
    default: return -1000;
  }
}
