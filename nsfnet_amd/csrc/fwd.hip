// Fused all-layer forward of the FCNet over one tile of collocation points.
//
// Replaces (reference, latteine1217/NSFnet): FCNet.forward (NSFnet/net.py:52-54) plus
// the nine reverse-mode autograd.grad sweeps of neural_net_equations
// (NSFnet/pinn_solver.py:132-163, ev-NSFnet/pinn_solver.py:290-342) by ONE
// forward-mode sweep that carries four streams per activation (value, d/dx, d/dy,
// Laplacian), then assembles the momentum/continuity/entropy residuals and the
// per-workgroup partial sums of the residual MSE (pinn_solver.py:197-226 / :372-428).
//
// Mapping to CDNA4: one workgroup = HP/32 waves; the tile's activations live in LDS
// as X[k = feature][128 columns]; each wave holds its 32xHP slice of the layer's
// weight matrix in registers as the A operand of v_mfma_f32_32x32x2_f32 and sweeps
// the four 32-column blocks (= the four streams of 32 points), so the tanh chain rule
// that mixes the four streams of one (point, feature) is lane-local on the accumulators.
#include "kernels.h"
#include "point_stage.h"

template <int HP, int NS>
__global__ __launch_bounds__(HP * 2) void fwd_kernel(FwdArgs a) {
  constexpr int NW = HP / 32;
  constexpr int NT = HP * 2;
  extern __shared__ float lds[];
  float* X = lds;                      // [HP][128]
  float* part = X + HP * 128;          // [NW][4][128]
  float* outv = part + NW * 4 * 128;   // [4][128]
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ob = w * 32;
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * (NS == 4 ? 32 : 128);
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    f32x16 acc[4];
    // ---------------- layer 0 (K = 2): VALU ----------------
    {
      float px[4], py[4];
      if (NS == 4) {
        int pt = tile * 32 + col;
        px[0] = pt < a.n ? a.x[pt] : 0.f;
        py[0] = pt < a.n ? a.y[pt] : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int pt = tile * 128 + 32 * j + col;
          px[j] = pt < a.n ? a.x[pt] : 0.f;
          py[j] = pt < a.n ? a.y[pt] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int o = ob + mfma_row(r, h);
        float wx = P[prep_w0x(HP) + o], wy = P[prep_w0y(HP) + o], b = P[prep_b0(HP) + o];
        if (NS == 4) {
          acc[0][r] = fmaf(wx, px[0], fmaf(wy, py[0], b));
          acc[1][r] = wx; acc[2][r] = wy; acc[3][r] = 0.f;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j][r] = fmaf(wx, px[j], fmaf(wy, py[j], b));
        }
      }
    }
    for (int l = 0; l < L; ++l) {
      // ------------- tanh + chain rule on the accumulators; save; restage in LDS -------------
      float* Sl = a.S ? a.S + ((size_t)tile * L + l) * act_block(HP) : nullptr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 s0, s1, s2, s3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int o = ob + 8 * g + 4 * h + e;
          float* Xo = X + o * 128 + col;
          if (NS == 4) {
            float t = tanhf(acc[0][r]);
            float zx = acc[1][r], zy = acc[2][r], zd = acc[3][r];
            float d1 = 1.f - t * t;
            float d2 = -2.f * t * d1;
            Xo[0] = t; Xo[32] = d1 * zx; Xo[64] = d1 * zy;
            Xo[96] = d2 * (zx * zx + zy * zy) + d1 * zd;
            s0[e] = t; s1[e] = zx; s2[e] = zy; s3[e] = zd;
          } else {
            float t0 = tanhf(acc[0][r]), t1 = tanhf(acc[1][r]), t2 = tanhf(acc[2][r]), t3 = tanhf(acc[3][r]);
            Xo[0] = t0; Xo[32] = t1; Xo[64] = t2; Xo[96] = t3;
            s0[e] = t0; s1[e] = t1; s2[e] = t2; s3[e] = t3;
          }
        }
        if (Sl && !(NS == 4 && l == 0 && a.s0_skip)) {      // (layer 0 is recomputed by its readers: FwdArgs::s0_skip)
          const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * 32 + col);
          const f32x4* S4 = reinterpret_cast<const f32x4*>(Sl);
          __builtin_nontemporal_store(s0, pin_base(S4 + 0 * (HP / 4) * 32) + so);   // streamed once: nontemporal
          __builtin_nontemporal_store(s1, pin_base(S4 + 1 * (HP / 4) * 32) + so);
          __builtin_nontemporal_store(s2, pin_base(S4 + 2 * (HP / 4) * 32) + so);
          __builtin_nontemporal_store(s3, pin_base(S4 + 3 * (HP / 4) * 32) + so);
        }
      }
      __syncthreads();
      if (l == L - 1) break;
      // ------------- hidden GEMM l+1:  Z[o][col] = sum_k W[o][k] X[k][col] -------------
      {
        const f32x4* wf = reinterpret_cast<const f32x4*>(P + prep_wf(HP, l + 1)) + (size_t)w * (HP / 8) * 64 + lane;
        f32x4 wq[HP / 8];
#pragma unroll
        for (int q = 0; q < HP / 8; ++q) wq[q] = wf[q * 64];
        const float* bl = P + prep_b(HP, l + 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float b = bl[ob + mfma_row(r, h)];
          acc[0][r] = b;
          if (NS == 4) { acc[1][r] = 0.f; acc[2][r] = 0.f; acc[3][r] = 0.f; }
          else { acc[1][r] = b; acc[2][r] = b; acc[3][r] = b; }
        }
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
    // ---------------- output layer (n_out <= 3 rows): VALU, K split over waves ----------------
    {
      float po[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      const float* wo = P + prep_wout(HP, L) + ob;
      for (int kk = 0; kk < 32; ++kk) {
        float x0 = X[(ob + kk) * 128 + lane], x1 = X[(ob + kk) * 128 + 64 + lane];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float wv = wo[c * HP + kk];
          po[c][0] = fmaf(wv, x0, po[c][0]);
          po[c][1] = fmaf(wv, x1, po[c][1]);
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        part[(w * 4 + c) * 128 + lane] = po[c][0];
        part[(w * 4 + c) * 128 + 64 + lane] = po[c][1];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 3 * 128; idx += NT) {
      int c = idx >> 7, cc = idx & 127;
      float s = (NS == 1 || cc < 32) ? P[prep_bout(HP, L) + c] : 0.f;
      for (int ww = 0; ww < NW; ++ww) s += part[(ww * 4 + c) * 128 + cc];
      outv[c * 128 + cc] = s;
    }
    __syncthreads();
    // ---------------- per-point stage (point_stage.h) ----------------
    if (NS == 4) residual_point_stage<32, 128>(a, outv, tile, tid, npad, lsum);
    else value_point_stage<128, NT>(a, outv, tile, tid, npad, lsum);
    __syncthreads();
  }
  // ---------------- block reduction of the loss partial sums (fixed order) ----------------
  float* red = lds;  // reuse X
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

size_t fwd_lds_bytes(int HP) { return ((size_t)HP * 128 + (size_t)(HP / 32) * 4 * 128 + 4 * 128) * sizeof(float); }

template <int HP, int NS>
static int launch_one(const FwdArgs& a, int grid, hipStream_t s) {
  size_t lds = fwd_lds_bytes(HP);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_kernel<HP, NS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((fwd_kernel<HP, NS>), dim3(grid), dim3(HP * 2), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

#define FWD_CASE(hp)                                                        \
  case hp:                                                                  \
    return NS == 4 ? launch_one<hp, 4>(a, grid, s) : launch_one<hp, 1>(a, grid, s);

int launch_fwd(int HP, int NS, const FwdArgs& a, int grid, hipStream_t s) {
  switch (HP) {
    FWD_CASE(32) FWD_CASE(64) FWD_CASE(96) FWD_CASE(128)
    FWD_CASE(160) FWD_CASE(192) FWD_CASE(224) FWD_CASE(256)
    default: return -1000;
  }
}
