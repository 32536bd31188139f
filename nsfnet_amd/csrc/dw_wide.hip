// Weight-gradient GEMM for WIDE nets (256 < hidden <= 512), fp32-input MFMA: the HP x HP
// output is cut into blocks of up to 256 x 256 (blockIdx.z), each handled like dw.hip's
// single block (8 waves x 4x2 accumulator tiles, split-K over tiles, one slab per group).
// Tiles are 64 columns (16 points x 4 planes), see fwd_wide.hip.
#include "kernels.h"

template <int NS>
__global__ __launch_bounds__(512) void dw_wide_kernel(DwArgs a, int HP) {
  constexpr int TM = 4, TN = 2, WN = 4, LDW = 260, CH = 32, PPL = 16, COLS = 64;
  extern __shared__ float lds[];
  float* Zs = lds;
  float* As = lds + 2 * CH * LDW;
  const int tid = threadIdx.x, lane = tid & 63, i32 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w / WN, wc = w % WN;
  const int T = HP / 32, nblk = (T + 7) / 8;
  const int bi = blockIdx.z / nblk, bj = blockIdx.z % nblk;
  const int l = blockIdx.y + 1, g = blockIdx.x;
  const int t0 = (int)((long)g * a.ntiles / a.groups), t1 = (int)((long)(g + 1) * a.ntiles / a.groups);
  const int nch = (t1 - t0) * (PPL / 8);
  const int p = tid & 7, og = tid >> 3;
  const int ogz = bi * 64 + og, oga = bj * 64 + og;
  const bool vz = ogz < HP / 4, va = oga < HP / 4;
  const size_t blk = (size_t)HP * COLS;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  f32x4 zr[4], sr[4];
  const bool rec = NS == 4 && a.s0_skip && l == 1;      // layer-1 workgroups, layer 0 not spilled (uniform per workgroup)
  f32x4 wx4 = {0.f, 0.f, 0.f, 0.f}, wy4 = wx4, b4 = wx4;
  if (rec && va) {
    const f32x4* w0 = reinterpret_cast<const f32x4*>(a.prep + prep_w0x(HP));
    wx4 = w0[oga]; wy4 = w0[HP / 4 + oga]; b4 = w0[2 * (HP / 4) + oga];
  }
  auto gload = [&](int ch) {
    const int tile = t0 + ch / (PPL / 8), c = ch % (PPL / 8);
    const f32x4* Zg = reinterpret_cast<const f32x4*>(a.Zb + ((size_t)tile * a.L + l) * blk) + 8 * c;
    const f32x4* Sg = reinterpret_cast<const f32x4*>(a.S + ((size_t)tile * a.L + (l - 1)) * blk) + 8 * c;
    const unsigned loz = (unsigned)(ogz * PPL + p), loa = (unsigned)(oga * PPL + p);
    if (rec) {
#pragma unroll
      for (int s = 0; s < 4; ++s) zr[s] = vz ? __builtin_nontemporal_load(pin_base(Zg + (size_t)s * (HP / 4) * PPL) + loz) : f32x4{0.f, 0.f, 0.f, 0.f};
      const int pt = tile * PPL + 8 * c + p;
      sr[0][0] = pt < a.n ? a.x[pt] : 0.f; sr[0][1] = pt < a.n ? a.y[pt] : 0.f;
      return;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      zr[s] = vz ? __builtin_nontemporal_load(pin_base(Zg + (size_t)s * (HP / 4) * PPL) + loz) : f32x4{0.f, 0.f, 0.f, 0.f};
      sr[s] = va ? __builtin_nontemporal_load(pin_base(Sg + (size_t)s * (HP / 4) * PPL) + loa) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](int buf) {
    f32x4 a0, a1, a2, a3;
    if (rec) {      // the forward's own fmaf chain and tanhf (fwd_wide.hip layer 0), bit for bit; padded features: all zero
      const float px = sr[0][0], py = sr[0][1];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = va ? tanhf(fmaf(wx4[e], px, fmaf(wy4[e], py, b4[e]))) : 0.f, zx = wx4[e], zy = wy4[e];
        float d1 = 1.f - t * t, d2 = -2.f * t * d1;
        a0[e] = t; a1[e] = d1 * zx; a2[e] = d1 * zy; a3[e] = d2 * (zx * zx + zy * zy);
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
    for (int n = 0; n < TN; ++n) {
      const int tr = bi * 8 + wr * TM + m, tc = bj * 8 + wc * TN + n;
      if (tr >= T || tc >= T) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int o = 32 * tr + mfma_row(r, h);
        int i = 32 * tc + i32;
        slab[(size_t)o * HP + i] = acc[m][n][r];
      }
    }
}

size_t dw_wide_lds_bytes() { return (size_t)2 * 2 * 32 * 260 * sizeof(float); }

int launch_dw_wide(int HP, int NS, const DwArgs& a, hipStream_t s) {
  if (a.L <= 1 || a.groups <= 0) return 0;
  const int T = HP / 32, nblk = (T + 7) / 8;
  size_t lds = dw_wide_lds_bytes();
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_wide_kernel<4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_wide_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  dim3 grid(a.groups, a.L - 1, nblk * nblk);
  if (NS == 4) hipLaunchKernelGGL((dw_wide_kernel<4>), grid, dim3(512), lds, s, a, HP);
  else hipLaunchKernelGGL((dw_wide_kernel<1>), grid, dim3(512), lds, s, a, HP);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
