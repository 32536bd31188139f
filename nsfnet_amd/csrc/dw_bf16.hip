// bf16x3 (and plain bf16) variant of the weight-gradient GEMM (see dw.hip for the algorithm
// and the reference lines it replaces).  dW_l[o][i] = sum_col Zb_l[o][col] A_{l-1}[i][col]:
// the contraction index is the column (point, stream), which is the ROW index of the natural
// staging image [col][feature]; ds_read_b64_tr_b16 delivers the K-along-rows fragments the
// 32x32x16 bf16 MFMA wants (lane map verified by tests/micro/mfma_bf16_layout.hip), so the
// images are written with plain 8-byte stores and no second (transposed) copy exists.
#include "kernels.h"
#include "bf16_util.h"

template <int T> struct DwCfgB;
template <> struct DwCfgB<1> { static constexpr int TM = 1, TN = 1; };
template <> struct DwCfgB<2> { static constexpr int TM = 1, TN = 2; };
template <> struct DwCfgB<3> { static constexpr int TM = 1, TN = 3; };
template <> struct DwCfgB<4> { static constexpr int TM = 2, TN = 2; };
template <> struct DwCfgB<5> { static constexpr int TM = 1, TN = 5; };
template <> struct DwCfgB<6> { static constexpr int TM = 2, TN = 3; };
template <> struct DwCfgB<7> { static constexpr int TM = 1, TN = 7; };
template <> struct DwCfgB<8> { static constexpr int TM = 4, TN = 2; };

template <int HP>
struct DwImg {
  static constexpr int PADB = ((64 - 2 * HP) % 256 + 256) % 256;   // row stride == 64 (mod 256) bytes
  static constexpr int RSB = 2 * HP + PADB;                         // row stride in bytes
  static constexpr int CH = 32;                                     // columns (rows of the image) per chunk
  static constexpr int ARR = CH * RSB;                              // bytes per array per buffer
  // arrays: 0 = Z hi, 1 = Z lo, 2 = A hi, 3 = A lo
  static constexpr size_t BYTES = (size_t)2 * 4 * ARR;
};

// ds_read_b64_tr_b16 through the compiler builtin (hipcc tracks its lgkmcnt itself)
// (the address stays an LDS-space pointer end to end: a generic pointer would cost an address-space cast,
// i.e. three VALU instructions, per read)
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ u32x2 tr_read(const lds_u8* p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  return __builtin_bit_cast(u32x2, v);
}

// PPL = points per plane of a tile: 32 (128-column tiles) or 16 (64-column tiles)
// REC: the activations are layer 0's and were not spilled (DwArgs::s0_skip) - recomputed from the point.  A template
// parameter, not a run-time flag: the steady-state loop must stay one basic block (see below).
template <int HP, int NS, int TERMS, int PPL, bool REC, bool P24>
__device__ __forceinline__ void dw_bf16_body(const DwArgs& a) {
  constexpr int CPT = PPL / 8;                 // 32-column chunks per tile
  constexpr size_t ABLK = (size_t)HP * 4 * PPL; // floats per (tile, layer) activation block
  using DI = DwImg<HP>;
  constexpr int T = HP / 32;
  constexpr int TM = DwCfgB<T>::TM, TN = DwCfgB<T>::TN;
  constexpr int WN = T / TN;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, i32 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w / WN, wc = w % WN;
  const int l = blockIdx.y + 1;
  const int g = blockIdx.x;
  const int t0 = (int)((long)g * a.ntiles / a.groups), t1 = (int)((long)(g + 1) * a.ntiles / a.groups);
  const int nch = (t1 - t0) * CPT;
  const int p = tid & 7, og = tid >> 3;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // Software pipeline: while the MFMAs of chunk ch run out of LDS buffer ch&1, the SAME wave converts chunk
  // ch+1 (requested one whole iteration earlier, so its data has landed) to bf16 hi/lo and stores it into the
  // other buffer - that VALU / ds_write work can sit in the MFMA shadow (4 VALU per MFMA are free on gfx950,
  // and a partner wave's VALU does NOT overlap this wave's MFMAs: tests/micro/mfma_valu_*.hip) - and then
  // requests chunk ch+2 into the registers it has just freed.
  // Layer-0 activations that the role-split forward did not spill (DwArgs::s0_skip): recomputed from the point with the
  // forward's own two FMAs and tanh; the point travels in sr[0][0..1] instead of the four saved quads.
  constexpr bool rec = REC;
  f32x4 wx4 = {0.f, 0.f, 0.f, 0.f}, wy4 = wx4, b4 = wx4;
  if (rec) {
    const f32x4* w0 = reinterpret_cast<const f32x4*>(a.prep + prep_w0x(HP));
    wx4 = w0[og]; wy4 = w0[HP / 4 + og]; b4 = w0[2 * (HP / 4) + og];
  }
  f32x4 zrA[4], srA[4];
  // P24: S and Z-bar arrive in the 24-bit spill format of the role-split sweeps (bf16_util.h pack24): THREE 16-byte
  // planes per quad (hi16 of streams 0-1, hi16 of streams 2-3, lo8 of all four) instead of four, same plane geometry,
  // same coalescing; unpacked at the head of the conversion.
  // Two raw sets (A, B): with the 24-bit format TWO chunks are in flight per workgroup (PINN_DWDEPTH 2) - one chunk in
  // flight (48 KB per CU, 12 MB chip-wide) does not cover the HBM latency at the rate the MFMAs consume it.
  struct Raw24 { u32x4 z[3], s[3]; float px, py; };
  Raw24 rawA, rawB;
  auto gload24 = [&](int ch, Raw24& R) {
    const int tile = t0 + ch / CPT, c = ch % CPT;
    const unsigned lo_ = (unsigned)(og * PPL + p);
#ifndef PINN_ABL
#define PINN_ABL 0      // timing-only (scripts/abl_build.py): 512 = the last hidden layer's Z-bar is not streamed (one resident block re-read)
#endif
    const f32x4* Zg = reinterpret_cast<const f32x4*>(a.Zb + spill_off(((PINN_ABL & 512) && l == a.L - 1) ? 0 : tile, l, a.L, a.sl0, a.sblk, ABLK)) + 8 * c;
#pragma unroll
    for (int k = 0; k < 3; ++k) R.z[k] = __builtin_bit_cast(u32x4, __builtin_nontemporal_load(pin_base(Zg + (size_t)k * (HP / 4) * PPL) + lo_));
    if (rec) {
      const int pt = tile * PPL + 8 * c + p;
      R.px = pt < a.n ? a.x[pt] : 0.f; R.py = pt < a.n ? a.y[pt] : 0.f;
    } else {
      const f32x4* Sg = reinterpret_cast<const f32x4*>(a.S + spill_off(tile, l - 1, a.L, a.sl0, a.sblk, ABLK)) + 8 * c;
#pragma unroll
      for (int k = 0; k < 3; ++k) R.s[k] = __builtin_bit_cast(u32x4, __builtin_nontemporal_load(pin_base(Sg + (size_t)k * (HP / 4) * PPL) + lo_));
    }
  };
  auto gload = [&](int ch, f32x4 (&zr)[4], f32x4 (&sr)[4]) {
    if (P24) { gload24(ch, rawA); return; }
    const int tile = t0 + ch / CPT, c = ch % CPT;
    if (rec) {
      const f32x4* Zg = reinterpret_cast<const f32x4*>(a.Zb + ((size_t)tile * a.L + l) * ABLK) + 8 * c;
      const unsigned lo_ = (unsigned)(og * PPL + p);
#pragma unroll
      for (int s = 0; s < 4; ++s) zr[s] = __builtin_nontemporal_load(pin_base(Zg + (size_t)s * (HP / 4) * PPL) + lo_);
      const int pt = tile * PPL + 8 * c + p;
      sr[0][0] = pt < a.n ? a.x[pt] : 0.f; sr[0][1] = pt < a.n ? a.y[pt] : 0.f;
      return;
    }
    // (tile, layer, plane, chunk) bases are uniform: pinned to scalar registers, one 32-bit lane offset
    const f32x4* Zg = reinterpret_cast<const f32x4*>(a.Zb + ((size_t)tile * a.L + l) * ABLK) + 8 * c;
    const f32x4* Sg = reinterpret_cast<const f32x4*>(a.S + ((size_t)tile * a.L + (l - 1)) * ABLK) + 8 * c;
    const unsigned lo_ = (unsigned)(og * PPL + p);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      zr[s] = __builtin_nontemporal_load(pin_base(Zg + (size_t)s * (HP / 4) * PPL) + lo_);
      sr[s] = __builtin_nontemporal_load(pin_base(Sg + (size_t)s * (HP / 4) * PPL) + lo_);
    }
  };
  auto lstore_r = [&](int buf, f32x4 (&zr)[4], f32x4 (&sr)[4], const Raw24& R) {
    if (P24) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        zr[s] = unpack24(u32x2{R.z[s >> 1][2 * (s & 1)], R.z[s >> 1][2 * (s & 1) + 1]}, R.z[2][s]);
        if (!rec) sr[s] = unpack24(u32x2{R.s[s >> 1][2 * (s & 1)], R.s[s >> 1][2 * (s & 1) + 1]}, R.s[2][s]);
      }
    }
    f32x4 av[4];
    if (rec) {
      const float px = P24 ? R.px : sr[0][0], py = P24 ? R.py : sr[0][1];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = fast_tanh(fmaf(wx4[e], px, fmaf(wy4[e], py, b4[e]))), zx = wx4[e], zy = wy4[e];
        const float d1 = 1.f - t * t, d2 = -2.f * t * d1;
        av[0][e] = t; av[1][e] = d1 * zx; av[2][e] = d1 * zy; av[3][e] = d2 * (zx * zx + zy * zy);
      }
    } else if (NS == 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = sr[0][e], zx = sr[1][e], zy = sr[2][e], zd = sr[3][e];
        float d1 = 1.f - t * t, d2 = -2.f * t * d1;
        av[0][e] = t; av[1][e] = d1 * zx; av[2][e] = d1 * zy; av[3][e] = d2 * (zx * zx + zy * zy) + d1 * zd;
      }
    } else {
      av[0] = sr[0]; av[1] = sr[1]; av[2] = sr[2]; av[3] = sr[3];
    }
    // 8-byte column XOR-swizzled by the row: with the row stride == 16 dwords (mod 32) the 16 lanes of a
    // ds_write_b64 group (8 rows x 2 columns) would otherwise fall on 8 banks (4-way conflict: SQ_LDS_BANK_CONFLICT
    // 3.5e8 cycles per launch at 6x256 / 360k points, more than the LDS-active cycles of the reads)
    unsigned char* base = ldsb + (size_t)buf * 4 * DI::ARR + p * DI::RSB + (og ^ (p & 6)) * 8;
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
  auto lstore = [&](int buf, f32x4 (&zr)[4], f32x4 (&sr)[4]) { lstore_r(buf, zr, sr, rawA); };
  // transposed-read lane geometry: 16-lane group gq = lane>>4 -> feature half fb, k half (== h)
  const int li = lane & 15, fb = (lane >> 4) & 1, q = li >> 2, pp = li & 3;
  // (the two 4-row blocks of a fragment sit in rows q and q + 4 of their 8-row group: un-swizzle each with its row)
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
        for (int n = 0; n < TN; ++n) {
          if (TERMS == 3) {
            acc[m][n] = MFMA_Q(ks, zh[m], al[n], acc[m][n]);
            acc[m][n] = MFMA_Q(ks + 1, zl[m], ah[n], acc[m][n]);
          }
          acc[m][n] = MFMA_Q(ks, zh[m], ah[n], acc[m][n]);
        }
    }
  };

#ifndef PINN_DWDEPTH
#define PINN_DWDEPTH 1      // 2 = two chunks in flight (built and measured in round 3: no gain, profiles/r03_ablations.txt B)
#endif
  constexpr int NMF = (DI::CH / 16) * TM * TN * (TERMS == 3 ? 3 : 1);
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // one MFMA
      // conversion VALU in its shadow: 4 per MFMA (192 per chunk) cover the fp32 spill's conversion; the 24-bit spill's
      // unpack adds 32 byte moves, and 5 per MFMA measured 2.60 against 2.71 ms (6-8: 2.60-2.66, 3: 2.70)
      __builtin_amdgcn_sched_group_barrier(0x002, P24 ? 5 : 4, 0);
    }
  };
  if (P24 && PINN_DWDEPTH == 2) {
    // chunk ch computes out of buffer ch & 1; raw set A carries the odd chunks, B the even ones (from chunk 2 on)
    if (nch > 0) {
      gload24(0, rawA);
      lstore_r(0, zrA, srA, rawA);
      gload24(nch > 1 ? 1 : 0, rawA);
      gload24(nch > 2 ? 2 : nch - 1, rawB);
    }
    __syncthreads();
    int ch = 0;
    for (; ch + 2 < nch; ch += 2) {
      mfma_chunk(0);
      lstore_r(1, zrA, srA, rawA);                           // chunk ch + 1
      gload24(ch + 3 < nch ? ch + 3 : nch - 1, rawA);
      interleave();
      __syncthreads();
      mfma_chunk(1);
      lstore_r(0, zrA, srA, rawB);                           // chunk ch + 2
      gload24(ch + 4 < nch ? ch + 4 : nch - 1, rawB);
      interleave();
      __syncthreads();
    }
    if (ch < nch) mfma_chunk(0);                             // (ch is even: its chunk sits in buffer 0)
    if (ch + 1 < nch) {
      lstore_r(1, zrA, srA, rawA);
      __syncthreads();
      mfma_chunk(1);
    }
    __syncthreads();
  } else {
  if (nch > 0) {
    gload(0, zrA, srA);
    lstore(0, zrA, srA);
    if (nch > 1) gload(1, zrA, srA);
  }
  __syncthreads();
  // steady state: MFMAs of chunk ch and the conversion of chunk ch+1 in ONE basic block (no branches: the
  // request for chunk ch+2 is clamped to the last chunk), with the interleave prescribed to the scheduler:
  // per MFMA one transposed LDS read for a later MFMA and a few VALU / one ds_write of the conversion.
  for (int ch = 0; ch + 1 < nch; ++ch) {
    const int buf = ch & 1;
    mfma_chunk(buf);
    lstore(buf ^ 1, zrA, srA);
    gload(ch + 2 < nch ? ch + 2 : nch - 1, zrA, srA);
    interleave();
    __syncthreads();
  }
  if (nch > 0) mfma_chunk((nch - 1) & 1);
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

// P24 = the plan's sweeps are the role-split pair (DwArgs::s0_skip): 24-bit spill format, layer 0 not spilled
template <int HP, int NS, int TERMS, int PPL, bool P24>
__global__ __launch_bounds__(HP * 2) void dw_bf16_kernel(DwArgs a) {
  if (P24 && blockIdx.y == 0) dw_bf16_body<HP, NS, TERMS, PPL, P24, P24>(a);
  else dw_bf16_body<HP, NS, TERMS, PPL, false, P24>(a);
}

template <int HP>
static size_t lds_bytes_t() { return DwImg<HP>::BYTES; }

size_t dw_bf16_lds_bytes(int HP) {
  switch (HP) {
    case 32: return lds_bytes_t<32>(); case 64: return lds_bytes_t<64>(); case 96: return lds_bytes_t<96>();
    case 128: return lds_bytes_t<128>(); case 160: return lds_bytes_t<160>(); case 192: return lds_bytes_t<192>();
    case 224: return lds_bytes_t<224>(); default: return lds_bytes_t<256>();
  }
}

template <int HP, int NS, int TERMS, int PPL, bool P24 = false>
static int launch_one(const DwArgs& a, hipStream_t s) {
  if (!P24 && HP == 256 && NS == 4 && PPL == 32 && a.s0_skip) return launch_one<HP, NS, TERMS, PPL, HP == 256 && NS == 4 && PPL == 32>(a, s);
  size_t lds = lds_bytes_t<HP>();
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_bf16_kernel<HP, NS, TERMS, PPL, P24>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((dw_bf16_kernel<HP, NS, TERMS, PPL, P24>), dim3(a.groups, a.L - 1), dim3(HP * 2), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

template <int HP, int PPL>
static int launch_hp(int NS, int terms, const DwArgs& a, hipStream_t s) {
  if (terms == 3) return NS == 4 ? launch_one<HP, 4, 3, PPL>(a, s) : launch_one<HP, 1, 3, PPL>(a, s);
  return NS == 4 ? launch_one<HP, 4, 1, PPL>(a, s) : launch_one<HP, 1, 1, PPL>(a, s);
}

int launch_dw_bf16(int HP, int NS, int terms, int cols, const DwArgs& a, hipStream_t s) {
  if (a.L <= 1 || a.groups <= 0) return 0;
  if (cols == 64) {
    switch (HP) {
      case 128: return launch_hp<128, 16>(NS, terms, a, s);
      case 256: return launch_hp<256, 16>(NS, terms, a, s);
      default: return -1000;
    }
  }
  switch (HP) {
    case 32: return launch_hp<32, 32>(NS, terms, a, s);
    case 64: return launch_hp<64, 32>(NS, terms, a, s);
    case 96: return launch_hp<96, 32>(NS, terms, a, s);
    case 128: return launch_hp<128, 32>(NS, terms, a, s);
    case 160: return launch_hp<160, 32>(NS, terms, a, s);
    case 192: return launch_hp<192, 32>(NS, terms, a, s);
    case 224: return launch_hp<224, 32>(NS, terms, a, s);
    case 256: return launch_hp<256, 32>(NS, terms, a, s);
    default: return -1000;
  }
}
