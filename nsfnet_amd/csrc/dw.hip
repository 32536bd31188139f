// Weight gradient of the hidden layers:  dW_l[o][i] = sum over columns (point, stream)
// of Zb_l[o][col] * A_{l-1}[i][col]   -  an HP x HP x (4N) GEMM per layer whose K
// dimension is the (huge) column index, so each workgroup keeps the whole HP x HP
// accumulator in registers (v_mfma_f32_32x32x2_f32 tiles), streams its share of the
// columns through LDS and writes one partial slab; misc.hip sums the slabs in a fixed
// order (bitwise reproducible, no float atomics).
//
// Replaces (reference) the MmBackward0 weight-gradient GEMMs that loss.backward()
// (NSFnet/pinn_solver.py:252, ev-NSFnet/pinn_solver.py:469) issues for every Linear of
// FCNet (NSFnet/net.py:36-46) - 4 streams are batched into one contraction here.
//
// A_{l-1} (the four activation streams of layer l-1) is recomputed on the fly from the
// saved (t, z_x, z_y, z_D) instead of being stored a second time.
#include "kernels.h"

template <int T> struct DwCfg;
template <> struct DwCfg<1> { static constexpr int TM = 1, TN = 1; };
template <> struct DwCfg<2> { static constexpr int TM = 1, TN = 2; };
template <> struct DwCfg<3> { static constexpr int TM = 1, TN = 3; };
template <> struct DwCfg<4> { static constexpr int TM = 2, TN = 2; };
template <> struct DwCfg<5> { static constexpr int TM = 1, TN = 5; };
template <> struct DwCfg<6> { static constexpr int TM = 2, TN = 3; };
template <> struct DwCfg<7> { static constexpr int TM = 1, TN = 7; };
template <> struct DwCfg<8> { static constexpr int TM = 4, TN = 2; };

template <int HP, int NS>
__global__ __launch_bounds__(HP * 2) void dw_kernel(DwArgs a) {
  constexpr int T = HP / 32;
  constexpr int TM = DwCfg<T>::TM, TN = DwCfg<T>::TN;
  constexpr int WN = T / TN;          // waves along N; waves along M = T / TM; WM*WN == T
  constexpr int LDW = HP + 4;
  constexpr int CH = 32;              // columns per chunk (8 points x 4 planes)
  extern __shared__ float lds[];
  float* Zs = lds;                    // [2][CH][LDW]
  float* As = lds + 2 * CH * LDW;     // [2][CH][LDW]
  const int tid = threadIdx.x, lane = tid & 63, i32 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w / WN, wc = w % WN;
  const int l = blockIdx.y + 1;
  const int g = blockIdx.x;
  const int t0 = (int)((long)g * a.ntiles / a.groups), t1 = (int)((long)(g + 1) * a.ntiles / a.groups);
  const int nch = (t1 - t0) * 4;
  const int p = tid & 7, og = tid >> 3;   // this thread's (point-in-chunk, feature quad)

  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  f32x4 zr[4], sr[4];
  const bool rec = NS == 4 && a.s0_skip && l == 1;      // (uniform per workgroup)
  f32x4 wx4 = {0.f, 0.f, 0.f, 0.f}, wy4 = wx4, b4 = wx4;
  if (rec) {
    const f32x4* w0 = reinterpret_cast<const f32x4*>(a.prep + prep_w0x(HP));
    wx4 = w0[og]; wy4 = w0[HP / 4 + og]; b4 = w0[2 * (HP / 4) + og];
  }
  auto gload = [&](int ch) {
    const int tile = t0 + (ch >> 2), c = ch & 3;
    const f32x4* Zg = reinterpret_cast<const f32x4*>(a.Zb + ((size_t)tile * a.L + l) * act_block(HP)) + 8 * c;
    const f32x4* Sg = reinterpret_cast<const f32x4*>(a.S + ((size_t)tile * a.L + (l - 1)) * act_block(HP)) + 8 * c;
    const unsigned lo_ = (unsigned)(og * 32 + p);
    if (rec) {      // layer-1 workgroups, layer 0 not spilled (DwArgs::s0_skip): only the point travels
#pragma unroll
      for (int s = 0; s < 4; ++s) zr[s] = __builtin_nontemporal_load(pin_base(Zg + (size_t)s * (HP / 4) * 32) + lo_);
      const int pt = tile * 32 + 8 * c + p;
      sr[0][0] = pt < a.n ? a.x[pt] : 0.f; sr[0][1] = pt < a.n ? a.y[pt] : 0.f;
      return;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      zr[s] = __builtin_nontemporal_load(pin_base(Zg + (size_t)s * (HP / 4) * 32) + lo_);
      sr[s] = __builtin_nontemporal_load(pin_base(Sg + (size_t)s * (HP / 4) * 32) + lo_);
    }
  };
  auto lstore = [&](int buf) {
    f32x4 a0, a1, a2, a3;
    if (rec) {      // the forward's own fmaf chain and tanhf (fwd.hip layer 0), bit for bit
      const float px = sr[0][0], py = sr[0][1];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = tanhf(fmaf(wx4[e], px, fmaf(wy4[e], py, b4[e]))), zx = wx4[e], zy = wy4[e];
        float d1 = 1.f - t * t, d2 = -2.f * t * d1;
        a0[e] = t; a1[e] = d1 * zx; a2[e] = d1 * zy; a3[e] = d2 * (zx * zx + zy * zy) + d1 * 0.f;
      }
    } else if (NS == 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = sr[0][e], zx = sr[1][e], zy = sr[2][e], zd = sr[3][e];
        float d1 = 1.f - t * t, d2 = -2.f * t * d1;
        a0[e] = t; a1[e] = d1 * zx; a2[e] = d1 * zy; a3[e] = d2 * (zx * zx + zy * zy) + d1 * zd;
      }
    } else {
      a0 = sr[0]; a1 = sr[1]; a2 = sr[2]; a3 = sr[3];
    }
    float* zb = Zs + buf * CH * LDW + p * LDW + og * 4;
    float* ab = As + buf * CH * LDW + p * LDW + og * 4;
    *reinterpret_cast<f32x4*>(zb + 0 * 8 * LDW) = zr[0];
    *reinterpret_cast<f32x4*>(zb + 1 * 8 * LDW) = zr[1];
    *reinterpret_cast<f32x4*>(zb + 2 * 8 * LDW) = zr[2];
    *reinterpret_cast<f32x4*>(zb + 3 * 8 * LDW) = zr[3];
    *reinterpret_cast<f32x4*>(ab + 0 * 8 * LDW) = a0;
    *reinterpret_cast<f32x4*>(ab + 1 * 8 * LDW) = a1;
    *reinterpret_cast<f32x4*>(ab + 2 * 8 * LDW) = a2;
    *reinterpret_cast<f32x4*>(ab + 3 * 8 * LDW) = a3;
  };

  if (nch > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nch) gload(ch + 1);
    const float* zp = Zs + buf * CH * LDW + h * LDW + 32 * (wr * TM) + i32;
    const float* ap = As + buf * CH * LDW + h * LDW + 32 * (wc * TN) + i32;
#pragma unroll
    for (int ks = 0; ks < CH / 2; ++ks) {
      float av[TM], bv[TN];
#pragma unroll
      for (int m = 0; m < TM; ++m) av[m] = zp[2 * ks * LDW + 32 * m];
#pragma unroll
      for (int n = 0; n < TN; ++n) bv[n] = ap[2 * ks * LDW + 32 * n];
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
    }
    if (ch + 1 < nch) lstore(buf ^ 1);
    __syncthreads();
  }
  float* slab = a.slabs + ((size_t)(l - 1) * a.groups + g) * HP * HP;
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int o = 32 * (wr * TM + m) + mfma_row(r, h);
        int i = 32 * (wc * TN + n) + i32;
        slab[(size_t)o * HP + i] = acc[m][n][r];
      }
}

size_t dw_lds_bytes(int HP) { return (size_t)2 * 2 * 32 * (HP + 4) * sizeof(float); }
int dw_threads(int HP) { return HP * 2; }

template <int HP, int NS>
static int launch_one(const DwArgs& a, hipStream_t s) {
  size_t lds = dw_lds_bytes(HP);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_kernel<HP, NS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((dw_kernel<HP, NS>), dim3(a.groups, a.L - 1), dim3(HP * 2), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

#define DW_CASE(hp)                                                         \
  case hp:                                                                  \
    return NS == 4 ? launch_one<hp, 4>(a, s) : launch_one<hp, 1>(a, s);

int launch_dw(int HP, int NS, const DwArgs& a, hipStream_t s) {
  if (a.L <= 1 || a.groups <= 0) return 0;
  switch (HP) {
    DW_CASE(32) DW_CASE(64) DW_CASE(96) DW_CASE(128)
    DW_CASE(160) DW_CASE(192) DW_CASE(224) DW_CASE(256)
    default: return -1000;
  }
}
