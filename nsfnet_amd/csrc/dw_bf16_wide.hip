// bf16x3 / bf16 weight-gradient GEMM for WIDE nets (256 < hidden <= 512): dw_bf16.hip's kernel (transposed
// ds_read_b64_tr_b16 fragment reads from plain [col][feature] staging images, split-K over tiles, one slab per
// group; reference: the MmBackward0 weight-gradient GEMMs of loss.backward(), NSFnet/pinn_solver.py:252,
// ev-NSFnet/pinn_solver.py:469) applied to output blocks of up to 256 x 256 (blockIdx.z) with the row stride
// HP at run time, exactly as dw_wide.hip blocks the fp32 kernel.  Tiles are 64 columns (16 points x 4 planes).
#include "kernels.h"
#include "bf16_util.h"

namespace {
struct DI {                                      // staging image of one 256-feature block (dw_bf16.hip DwImg<256>)
  static constexpr int RSB = 2 * 256 + 64;       // row stride == 64 (mod 256) bytes
  static constexpr int CH = 32;                  // columns (rows of the image) per chunk
  static constexpr int ARR = CH * RSB;           // bytes per array per buffer; arrays: Z hi, Z lo, A hi, A lo
  static constexpr size_t BYTES = (size_t)2 * 4 * ARR;
};
typedef __attribute__((address_space(3))) unsigned char lds_u8;   // LDS-space pointer end to end (no generic casts)
__device__ __forceinline__ u32x2 tr_read(const lds_u8* p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  return __builtin_bit_cast(u32x2, v);
}
}  // namespace

// P24: S / Z-bar in the 24-bit three-plane spill format (DwArgs::s24; residual mode only) - a template parameter, the
// steady-state loop must stay one basic block
template <int NS, int TERMS, bool P24>
__global__ __launch_bounds__(512) void dw_bf16_wide_kernel(DwArgs a, int HP) {
  constexpr int TM = 4, TN = 2, WN = 4, PPL = 16, COLS = 64, CPT = PPL / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, i32 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w / WN, wc = w % WN;
  const int T = HP / 32, nblk = (T + 7) / 8;
  const int bi = blockIdx.z / nblk, bj = blockIdx.z % nblk;
  const int l = blockIdx.y + 1, g = blockIdx.x;
  const int t0 = (int)((long)g * a.ntiles / a.groups), t1 = (int)((long)(g + 1) * a.ntiles / a.groups);
  const int nch = (t1 - t0) * CPT;
  const int p = tid & 7, og = tid >> 3;
  const int ogz = bi * 64 + og, oga = bj * 64 + og;
  const bool vz = ogz < HP / 4, va = oga < HP / 4;
  const size_t blk = (size_t)HP * COLS;
  // accumulator tiles of this wave that lie inside the matrix (uniform per wave)
  bool live[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) live[m][n] = (bi * 8 + wr * TM + m < T) && (bj * 8 + wc * TN + n < T);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  f32x4 zr[4], sr[4];
  u32x4 zp[3], sp[3];
  auto gload = [&](int ch) {
    const int tile = t0 + ch / CPT, c = ch % CPT;
    const f32x4* Zg = reinterpret_cast<const f32x4*>(a.Zb + ((size_t)tile * a.L + l) * blk) + 8 * c;
    const f32x4* Sg = reinterpret_cast<const f32x4*>(a.S + ((size_t)tile * a.L + (l - 1)) * blk) + 8 * c;
    const unsigned loz = (unsigned)(ogz * PPL + p), loa = (unsigned)(oga * PPL + p);
    if (P24) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        zp[k] = vz ? __builtin_bit_cast(u32x4, __builtin_nontemporal_load(pin_base(Zg + (size_t)k * (HP / 4) * PPL) + loz)) : u32x4{0u, 0u, 0u, 0u};
        sp[k] = va ? __builtin_bit_cast(u32x4, __builtin_nontemporal_load(pin_base(Sg + (size_t)k * (HP / 4) * PPL) + loa)) : u32x4{0u, 0u, 0u, 0u};
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      zr[s] = vz ? __builtin_nontemporal_load(pin_base(Zg + (size_t)s * (HP / 4) * PPL) + loz) : f32x4{0.f, 0.f, 0.f, 0.f};
      sr[s] = va ? __builtin_nontemporal_load(pin_base(Sg + (size_t)s * (HP / 4) * PPL) + loa) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](int buf) {
    if (P24) {
#pragma unroll
      for (int s = 0; s < 4; ++s) { zr[s] = unpack24_plane(zp, s); sr[s] = unpack24_plane(sp, s); }
    }
    f32x4 av[4];
    if (NS == 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = sr[0][e], zx = sr[1][e], zy = sr[2][e], zd = sr[3][e];
        float d1 = 1.f - t * t, d2 = -2.f * t * d1;
        av[0][e] = t; av[1][e] = d1 * zx; av[2][e] = d1 * zy; av[3][e] = d2 * (zx * zx + zy * zy) + d1 * zd;
      }
    } else {
      av[0] = sr[0]; av[1] = sr[1]; av[2] = sr[2]; av[3] = sr[3];
    }
    unsigned char* base = ldsb + (size_t)buf * 4 * DI::ARR + p * DI::RSB + (og ^ (p & 6)) * 8;     // XOR swizzle: see dw_bf16.hip
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x2 hi, lo;
      unsigned char* row = base + s * 8 * DI::RSB;
      split4(zr[s][0], zr[s][1], zr[s][2], zr[s][3], hi, lo);
      *reinterpret_cast<u32x2*>(row + 0 * DI::ARR) = hi;
      if (TERMS == 3) *reinterpret_cast<u32x2*>(row + 1 * DI::ARR) = lo;
      split4(av[s][0], av[s][1], av[s][2], av[s][3], hi, lo);
      *reinterpret_cast<u32x2*>(row + 2 * DI::ARR) = hi;
      if (TERMS == 3) *reinterpret_cast<u32x2*>(row + 3 * DI::ARR) = lo;
    }
  };

  // transposed-read lane geometry: 16-lane group gq = lane>>4 -> feature half fb, k half (== h)
  const int li = lane & 15, fb = (lane >> 4) & 1, q = li >> 2, pp = li & 3;
  const int lane_off = (8 * h + q) * DI::RSB + ((4 * fb + pp) ^ (q & 2)) * 8;
  const int lane_off4 = (8 * h + q + 4) * DI::RSB + ((4 * fb + pp) ^ ((q & 2) | 4)) * 8 - lane_off;
  auto mfma_chunk = [&](int buf) {
    const lds_u8* B0 = (const lds_u8*)ldsb + buf * 4 * DI::ARR + lane_off;
#pragma unroll
    for (int ks = 0; ks < DI::CH / 16; ++ks) {
      u32x4 zh[TM], zl[TM], ah[TN], al[TN];
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const lds_u8* pz = B0 + ks * 16 * DI::RSB + 64 * (wr * TM + m);
        u32x2 x0 = tr_read(pz), x1 = tr_read(pz + lane_off4);
        zh[m][0] = x0[0]; zh[m][1] = x0[1]; zh[m][2] = x1[0]; zh[m][3] = x1[1];
        if (TERMS == 3) {
          u32x2 y0 = tr_read(pz + DI::ARR), y1 = tr_read(pz + DI::ARR + lane_off4);
          zl[m][0] = y0[0]; zl[m][1] = y0[1]; zl[m][2] = y1[0]; zl[m][3] = y1[1];
        }
      }
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const lds_u8* pa = B0 + 2 * DI::ARR + ks * 16 * DI::RSB + 64 * (wc * TN + n);
        u32x2 x0 = tr_read(pa), x1 = tr_read(pa + lane_off4);
        ah[n][0] = x0[0]; ah[n][1] = x0[1]; ah[n][2] = x1[0]; ah[n][3] = x1[1];
        if (TERMS == 3) {
          u32x2 y0 = tr_read(pa + DI::ARR), y1 = tr_read(pa + DI::ARR + lane_off4);
          al[n][0] = y0[0]; al[n][1] = y0[1]; al[n][2] = y1[0]; al[n][3] = y1[1];
        }
      }
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
          if (live[m][n]) {
            if (TERMS == 3) {
              acc[m][n] = mfma_bf16(zh[m], al[n], acc[m][n]);
              acc[m][n] = mfma_bf16(zl[m], ah[n], acc[m][n]);
            }
            acc[m][n] = mfma_bf16(zh[m], ah[n], acc[m][n]);
          }
    }
  };
  // software pipeline as in dw_bf16.hip: the conversion of chunk ch+1 (requested one iteration earlier) and the
  // request for chunk ch+2 sit in the same basic block as the MFMAs of chunk ch
  if (nch > 0) {
    gload(0);
    lstore(0);
    if (nch > 1) gload(1);
  }
  __syncthreads();
  for (int ch = 0; ch + 1 < nch; ++ch) {
    const int buf = ch & 1;
    mfma_chunk(buf);
    lstore(buf ^ 1);
    gload(ch + 2 < nch ? ch + 2 : nch - 1);
    constexpr int NMF = (DI::CH / 16) * TM * TN * (TERMS == 3 ? 3 : 1);
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);        // conversion VALU in its shadow
    }
    __syncthreads();
  }
  if (nch > 0) mfma_chunk((nch - 1) & 1);
  __syncthreads();
  float* slab = a.slabs + ((size_t)(l - 1) * a.groups + g) * HP * HP;
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      if (!live[m][n]) continue;
      const int tr = bi * 8 + wr * TM + m, tc = bj * 8 + wc * TN + n;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int o = 32 * tr + mfma_row(r, h);
        int i = 32 * tc + i32;
        slab[(size_t)o * HP + i] = acc[m][n][r];
      }
    }
}

size_t dw_bf16_wide_lds_bytes() { return DI::BYTES; }

template <int NS, int TERMS, bool P24 = false>
static int launch_one(int HP, const DwArgs& a, hipStream_t s) {
  if (!P24 && NS == 4 && a.s24) return launch_one<NS, TERMS, NS == 4>(HP, a, s);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_bf16_wide_kernel<NS, TERMS, P24>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  const int nblk = (HP / 32 + 7) / 8;
  hipLaunchKernelGGL((dw_bf16_wide_kernel<NS, TERMS, P24>), dim3(a.groups, a.L - 1, nblk * nblk), dim3(512), DI::BYTES, s, a, HP);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int launch_dw_bf16_wide(int HP, int NS, int terms, const DwArgs& a, hipStream_t s) {
  if (a.L <= 1 || a.groups <= 0) return 0;
  if (HP <= 256 || HP > 512 || HP % 32) return -1000;
  if (terms == 3) return NS == 4 ? launch_one<4, 3>(HP, a, s) : launch_one<1, 3>(HP, a, s);
  return NS == 4 ? launch_one<4, 1>(HP, a, s) : launch_one<1, 1>(HP, a, s);
}
