// Role-split bf16x3 forward sweep for WIDE nets (256 < hidden <= 448, e.g. BASELINE config 5's 8x400), residual mode:
// the schedule of fwd_bf16_split.hip (two wave groups per workgroup in opposite phases, SIMD partners overlap one
// group's MFMAs with the other's VALU / LDS / memory instructions) at the tile geometry of fwd_bf16_wide.hip (64 columns =
// 16 points x 4 streams, S in the 24-bit three-plane format, layer 0 spilled: dw_bf16_wide.hip and bwd_bf16_wsplit.hip
// read exactly what fwd_bf16_wide.hip would have written).  Reference lines replaced: NSFnet/net.py:52-54,
// NSFnet/pinn_solver.py:132-163,197-226, ev-NSFnet/pinn_solver.py:290-342,372-428 (see fwd.hip).
//
// Geometry.  NB = HP / 32 feature blocks.  A group is four waves; wave w owns the blocks 4 q + w (q = 0 .. MQ - 1, those
// that exist), so the accumulators are acc[MQ][2 column blocks] - 128 registers at MQ = 4, as in the hidden-256 kernel.
// The ONE shared image (bf16 hi/lo, [stream][point][k], 512-element rows) is split along K into MQ regions of four
// blocks (128 features): a phase is MQ quarters with a workgroup barrier after each; the M group reads region q in
// quarter q while the E group computes its block q (one per wave), parks the result in 32 registers and writes it into
// region q in quarter q + 1, when the M group has finished with that region.
// The LAST region needs no parking: the image rows have 512 - HP spare elements behind the features, enough for a
// second copy of that (short) region, and each group keeps its own copy - a group's E phase writes its last block
// straight into its copy while the other group's M phase reads the other one.  (The hidden-256 kernel parks it through
// the first quarter of the following M phase; here those 32 registers are the second half of the weight ring: four
// feature blocks per wave need 64 registers of weight fragments for two k-steps.)  Being private, that copy can be
// written at ANY time of the E phase: the last block's two quads are computed in quarters 0 and 1 (three quads per
// quarter there), and the last quarter - in which the M group has only the short last region to multiply - carries
// nothing but the dump of block MQ - 2.  (Measured neutral at 8x400: the wide sweeps wait on their weight stream.)
//
// A 64-column accumulator block holds two streams (lanes 0-15 / 16-31): v_permlane16_swap brings the four streams
// of a point into one lane (fwd_bf16_wide.hip), so lane (pp, hi, h) of a wave ends up with the eight features
// 32 b + 8 (gq + 2 hi) + 4 h + e (gq = 0, 1; e = 0..3) of point pp: two register quads per quarter, as in the hidden-256
// kernel.
#include "kernels.h"
#include "point_stage.h"
#include "bf16_util.h"
#ifndef PINN_ABL
#define PINN_ABL 0      // timing-only ablation switches (scripts/abl_build.py)
#endif

template <int HP>
struct WSplitGeo {
  using XI = XImg<HP, 16>;
  static constexpr int NB = HP / 32, MQ = (NB + 3) / 4, KS = HP / 16;
  static constexpr int LASTK = 128 * (MQ - 1);          // first feature of the last region
  static constexpr int LASTN = HP - LASTK;              // its width (32 .. 128)
  static constexpr bool FITS = XI::RSE - HP >= LASTN;   // room for the second copy of the last region
  static constexpr size_t X_BYTES = XI::BYTES;
  static constexpr size_t PART_F = (size_t)2 * 4 * 12 * 16;   // [group][wave][3 outputs x 4 streams][16 points]
  static constexpr size_t OUTV_F = (size_t)2 * 3 * 64;        // [group][3][64]
  static size_t fwd_bytes() { return X_BYTES + (PART_F + OUTV_F + 6 * HP) * sizeof(float); }
};

template <int HP, int TERMS>
__global__ __launch_bounds__(512, 1) void fwd_wsplit_kernel(FwdArgs a) {
  using G = WSplitGeo<HP>;
  using XI = typename G::XI;
  static_assert(HP > 256 && HP <= 512 && G::FITS, "hidden widths whose last K region fits twice in the image rows");
  constexpr int GT = 256, NB = G::NB, MQ = G::MQ, KS = G::KS, PPL = 16, COLS = 64;
  constexpr int RING = 2;
  constexpr size_t PLQ = (size_t)(HP / 4) * PPL;                   // f32x4 per S plane
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* const X = ldsb;
  float* const part = reinterpret_cast<float*>(ldsb + G::X_BYTES);
  float* const outv = part + G::PART_F;
  float* const woutL = outv + G::OUTV_F;                  // [3][HP]
  float* const w0L = woutL + 3 * HP;                      // [w0x | w0y | b0][HP]
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w = wave & 3;
  const int gtid = tid - grp * GT;
  const int mc = (NB - w + 3) / 4;                        // feature blocks of this wave: 4 q + w, q < mc
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * PPL;
  float* const partG = part + (size_t)grp * 4 * 12 * 16;
  float* const outvG = outv + (size_t)grp * 3 * 64;
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < 3 * HP; i += 2 * GT) { woutL[i] = P[prep_wout(HP, L) + i]; w0L[i] = P[prep_w0x(HP) + i]; }
  __syncthreads();

#define WS_LANE()                                      \
  int lane = lane0;                                    \
  asm volatile("" : "+v"(lane));                       \
  const int col = lane & 31, h = lane >> 5;            \
  const int hi = col >> 4, pp = col & 15;              \
  (void)h; (void)hi; (void)pp

  f32x16 acc[MQ][2];                      // [feature block of this wave][column block: streams 2 j, 2 j + 1]
  u32x2 st[2][4][2];                      // parked epilogue output of one block: [quad][stream][hi | lo]

  // image chunk (8 k) of feature o for group g: the last region's second copy sits behind the features
  auto img_chunk = [&](int o, int g) { return (o >> 3) + ((o >= G::LASTK && g) ? G::LASTN / 8 : 0); };
  // quad (block b, g8 = gq + 2 hi), stream p -> image (this group's copy of the last region)
  auto dump_kp = [&](int b, int k, int p, int pp, int hi, int h) {
    const int off = XI::chunk_off(pp, img_chunk(32 * b + 8 * (k + 2 * hi), grp)) + 8 * h;
    *reinterpret_cast<u32x2*>(X + p * XI::PLANE * 2 + off) = st[k][p][0];
    if (TERMS == 3) *reinterpret_cast<u32x2*>(X + XI::HALF * 2 + p * XI::PLANE * 2 + off) = st[k][p][1];
  };
  auto dump_k = [&](int b, int k, int pp, int hi, int h) {
#pragma unroll
    for (int p = 0; p < 4; ++p) dump_kp(b, k, p, pp, hi, h);
  };

  // weight-fragment ring [block][k-step % RING]; lives across phases (the first k-step of M_{l+1} is requested during
  // the last quad of E_l)
  u32x4 wh[MQ][RING], wl[MQ][RING];
  typedef __attribute__((address_space(1))) u32x4 gu32x4;
  auto wload_l = [&](int l, int s, int lane) {
    const gu32x4* const wf = reinterpret_cast<const gu32x4*>(pin_base(reinterpret_cast<const u32x4*>(P + prep_wf(HP, l))));
#pragma unroll
    for (int m = 0; m < MQ; ++m) {
      if (m == MQ - 1 && m >= mc) continue;            // (only the last block can be missing: mc >= MQ - 1)
      wh[m][s % RING] = (wf + (size_t)(4 * m + w) * KS * 64 + s * 64)[lane];
      if (TERMS == 3 && !((PINN_ABL & 128) && (m & 1)))      // (PINN_ABL 128, timing only: lo fragments of every other block -> 3/4 of the weight bytes)
        wl[m][s % RING] = (wf + (size_t)(HP * HP / 8) + (size_t)(4 * m + w) * KS * 64 + s * 64)[lane];
    }
  };
#define WL_W(m, i) (((PINN_ABL & 128) && ((m) & 1)) ? wl[(m) - 1][i] : wl[m][i])

  // ---------------- M phase: acc <- W_l x image, region q in quarter q ----------------
  auto mphase = [&](int l) {
    WS_LANE();
    u32x4 bh[2], bo[2];
    auto bload = [&](int u) {                          // u = 2 s + j
      const int s = u >> 1, j = u & 1;
      const int off = XI::chunk_off(pp, img_chunk(16 * s, grp) + h) + (2 * j + hi) * XI::PLANE * 2;
      bh[u & 1] = *reinterpret_cast<const u32x4*>(X + off);
      if (TERMS == 3) bo[u & 1] = *reinterpret_cast<const u32x4*>(X + XI::HALF * 2 + off);
    };
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
      const int s0 = 8 * q, s1 = (8 * q + 8 < KS) ? 8 * q + 8 : KS;
      bload(2 * s0);
#pragma unroll
      for (int u = 2 * s0; u < 2 * s1; ++u) {
        const int s = u >> 1, j = u & 1;
        if (j == 0 && s + 1 < KS) wload_l(l, s + 1, lane);
        if (u + 1 < 2 * s1) bload(u + 1);
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
          if (m == MQ - 1 && m >= mc) continue;
          if (s == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[m][j] = TERMS == 3 ? mfma_bf16(wh[m][0], bo[u & 1], zero) : mfma_bf16(wh[m][0], bh[u & 1], zero);
            if (TERMS == 3) {
              acc[m][j] = mfma_bf16(WL_W(m, 0), bh[u & 1], acc[m][j]);
              acc[m][j] = mfma_bf16(wh[m][0], bh[u & 1], acc[m][j]);
            }
          } else {
            if (TERMS == 3) {
              acc[m][j] = mfma_bf16(wh[m][s % RING], bo[u & 1], acc[m][j]);
              acc[m][j] = mfma_bf16(WL_W(m, s % RING), bh[u & 1], acc[m][j]);
            }
            acc[m][j] = mfma_bf16(wh[m][s % RING], bh[u & 1], acc[m][j]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);      // requests stay where they are written (one k-step / one step ahead)
      }
      __syncthreads();
    }
  };

  // ---------------- E phase: chain rule of layer lE of this group's tile ----------------
  // EK: 0 = layer 0 (pre-activations from (x, y) on the VALU), 1 = hidden layer 1..L-2, 2 = last hidden layer (output
  // layer folded in, nothing written to the image).  The point stage of the group's PREVIOUS tile rides in quarters 0 / 1.
  auto ephase = [&](auto EKIND, int lE, int tileE, int pstage_tile) {
    constexpr int EK = decltype(EKIND)::value;
    constexpr bool last = EK == 2, first = EK == 0;
    WS_LANE();
    float* const Sl = a.S + ((size_t)tileE * L + lE) * ((size_t)HP * COLS);
    const float* const bE = P + (first ? prep_b0(HP) : prep_b(HP, lE));      // (global: 160 KiB of LDS do not hold L x HP biases too)
    float po[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s) po[c][s] = 0.f;
    float px = 0.f, py = 0.f;
    if (first) {
      const int pt = tileE * PPL + pp;
      px = pt < a.n ? a.x[pt] : 0.f; py = pt < a.n ? a.y[pt] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
      // ---- the previous tile's point stage (output-layer bias + cross-wave sum, then residuals / loss) ----
      if (pstage_tile >= 0 && q == 0) {
        for (int idx = gtid; idx < 3 * COLS; idx += GT) {
          const int c3 = idx / COLS, cc = idx % COLS;
          float s = cc < PPL ? P[prep_bout(HP, L) + c3] : 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) s += partG[(ww * 12 + c3 * 4 + cc / PPL) * 16 + (cc % PPL)];
          outvG[c3 * COLS + cc] = s;
        }
      }
      if (pstage_tile >= 0 && pstage_tile < a.ntiles && q == 1)
        residual_point_stage<PPL, COLS>(a, outvG, pstage_tile, gtid, npad, lsum);
      // Blocks 0 .. MQ - 2 (every wave owns them) ride in their own quarter.  The LAST block's two quads ride in quarters
      // 0 and 1 instead of a quarter of their own: its region has a per-group copy that nobody else touches during this
      // phase, so it can be written at any time - and the last quarter, where the M group has only the short last
      // region to multiply (2 k-steps at hidden 416), is left with the dump of block MQ - 2 alone.
      const bool mainb = q < MQ - 1;
      const bool prev = q > 0 && !last;                  // block q - 1 is parked and its region is free now
      const bool extra = q < 2 && mc == MQ;              // (uniform: this wave owns a block in the last region)
      auto swaps = [&](int bq) {                          // the four streams of a point into one lane (see the header)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[bq][0][r]), __float_as_uint(acc[bq][0][r + 8]), false, false);
          auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[bq][1][r]), __float_as_uint(acc[bq][1][r + 8]), false, false);
          acc[bq][0][r] = __uint_as_float(s01[0]); acc[bq][0][r + 8] = __uint_as_float(s01[1]);
          acc[bq][1][r] = __uint_as_float(s23[0]); acc[bq][1][r + 8] = __uint_as_float(s23[1]);
        }
      };
      // chain rule of quad k of block bq: a-streams av, saved values sv
      auto compute = [&](int bq, int k, f32x4 (&av)[4], f32x4 (&sv)[4]) {
        const int o = 32 * (4 * bq + w) + 8 * (k + 2 * hi) + 4 * h;
        f32x4 b4, wx4, wy4;
        if (first) {
          wx4 = *reinterpret_cast<const f32x4*>(w0L + o); wy4 = *reinterpret_cast<const f32x4*>(w0L + HP + o);
          b4 = *reinterpret_cast<const f32x4*>(w0L + 2 * HP + o);
        } else {
          b4 = *reinterpret_cast<const f32x4*>(bE + o);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * k + e;
          float z, zx, zy, zd;
          if (first) {
            z = fmaf(wx4[e], px, fmaf(wy4[e], py, b4[e])); zx = wx4[e]; zy = wy4[e]; zd = 0.f;
          } else {
            z = acc[bq][0][r] + b4[e]; zx = acc[bq][0][r + 8]; zy = acc[bq][1][r]; zd = acc[bq][1][r + 8];
          }
          const float t = fast_tanh(z);
          const float d1 = 1.f - t * t;
          const float d2 = -2.f * t * d1;
          av[0][e] = t; av[1][e] = d1 * zx; av[2][e] = d1 * zy; av[3][e] = d2 * (zx * zx + zy * zy) + d1 * zd;
          sv[0][e] = t; sv[1][e] = zx; sv[2][e] = zy; sv[3][e] = zd;
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // hi/lo split of the a-streams (parked in st[k], or straight into the image: `direct`), output layer (last), S spill
      auto finish = [&](int bq, int k, f32x4 (&av)[4], f32x4 (&sv)[4], bool direct) {
        const int o = 32 * (4 * bq + w) + 8 * (k + 2 * hi) + 4 * h;
        const unsigned so = (unsigned)(o >> 2) * PPL + pp;
        u32x4 pk[3];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (!last) {
            if (direct) {
              u32x2 th, tl;
              split4(av[p][0], av[p][1], av[p][2], av[p][3], th, tl);
              const int off = XI::chunk_off(pp, img_chunk(o - 4 * h, grp)) + 8 * h;
              *reinterpret_cast<u32x2*>(X + p * XI::PLANE * 2 + off) = th;
              if (TERMS == 3) *reinterpret_cast<u32x2*>(X + XI::HALF * 2 + p * XI::PLANE * 2 + off) = tl;
            } else {
              split4(av[p][0], av[p][1], av[p][2], av[p][3], st[k][p][0], st[k][p][1]);
            }
          } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const f32x4 wo = *reinterpret_cast<const f32x4*>(woutL + c * HP + o);
#pragma unroll
              for (int e = 0; e < 4; ++e) po[c][p] = fmaf(wo[e], av[p][e], po[c][p]);
            }
          }
          u32x2 hi24; unsigned lo24;      // 24-bit spill (bf16_util.h pack24): three 16-byte planes
          pack24(sv[p], hi24, lo24);
          pk[p >> 1][2 * (p & 1)] = hi24[0]; pk[p >> 1][2 * (p & 1) + 1] = hi24[1]; pk[2][p] = lo24;
          if (p & 1) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, pk[p >> 1]), pin_base(reinterpret_cast<const f32x4*>(Sl) + (p >> 1) * PLQ) + so);
          if (p == 3) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, pk[2]), pin_base(reinterpret_cast<const f32x4*>(Sl) + 2 * PLQ) + so);
          if (last) asm volatile("" : "+v"(po[0][p]), "+v"(po[1][p]), "+v"(po[2][p]));
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (mainb && !first) swaps(q);
      if (q == 0 && mc == MQ && !first) swaps(MQ - 1);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (!last && q == MQ - 1 && k == 1) wload_l(lE + 1, 0, lane);      // first weight k-step of M_{lE+1}
        f32x4 av[4], sv[4];
        if (mainb) compute(q, k, av, sv);
        // block q - 1, parked in the previous quarter: quad k leaves its registers just before they are refilled
        if (prev) dump_k(4 * (q - 1) + w, k, pp, hi, h);
        if (mainb) finish(q, k, av, sv, false);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (extra) {
        f32x4 av[4], sv[4];
        compute(MQ - 1, q, av, sv);
        finish(MQ - 1, q, av, sv, true);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (last && q == MQ - 1) {
        // the lanes (pp, hi, h) of a point hold different features: add the four of them (all publish the same value)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            float v = po[c][s];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            partG[(w * 12 + c * 4 + s) * 16 + pp] = v;
          }
      }
      __syncthreads();
    }
  };
  auto idle = [&]() {
#pragma unroll
    for (int q = 0; q < MQ; ++q) __syncthreads();
  };

  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  // Program of a group: per tile E0 M1 E1 ... M_{L-1} E_{L-1}; group 1 runs it one phase behind group 0
  // (fwd_bf16_split.hip).  Tile of pair i: 2 i + grp.
  const int npairs = (a.ntiles + 1) / 2;
  if (grp == 1) idle();
  int prev_tile = -1;
  for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
    const int tile = 2 * pair + grp;
    ephase(K0{}, 0, tile, prev_tile);
    for (int l = 1; l < L - 1; ++l) {
      mphase(l);
      ephase(K1{}, l, tile, -1);
    }
    mphase(L - 1);
    ephase(K2{}, L - 1, tile, -1);
    prev_tile = tile;
  }
  for (int q = 0; q < MQ; ++q) {                      // drain: point stage of the last tile
    if (q == 0 && prev_tile >= 0)
      for (int i2 = gtid; i2 < 3 * COLS; i2 += GT) {
        const int c3 = i2 / COLS, cc = i2 % COLS;
        float s = cc < PPL ? P[prep_bout(HP, L) + c3] : 0.f;
        for (int ww = 0; ww < 4; ++ww) s += partG[(ww * 12 + c3 * 4 + cc / PPL) * 16 + (cc % PPL)];
        outvG[c3 * COLS + cc] = s;
      }
    if (q == 1 && prev_tile >= 0 && prev_tile < a.ntiles) residual_point_stage<PPL, COLS>(a, outvG, prev_tile, gtid, npad, lsum);
    __syncthreads();
  }
  if (grp == 0) idle();
  float* red = reinterpret_cast<float*>(ldsb);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k * 2 * GT + tid] = lsum[k];
  __syncthreads();
  if (tid < 4) {
    float s = 0.f;
    for (int t = 0; t < 2 * GT; ++t) s += red[tid * 2 * GT + t];
    a.partials[blockIdx.x * PINN_NLOSS + tid] = s;
  } else if (tid < PINN_NLOSS) {
    a.partials[blockIdx.x * PINN_NLOSS + tid] = 0.f;
  }
}

template <int HP>
static constexpr bool wsplit_ok() { return WSplitGeo<HP>::FITS; }

size_t fwd_wsplit_lds_bytes(int HP) {
  switch (HP) {
    case 288: return WSplitGeo<288>::fwd_bytes(); case 320: return WSplitGeo<320>::fwd_bytes();
    case 352: return WSplitGeo<352>::fwd_bytes(); case 384: return WSplitGeo<384>::fwd_bytes();
    case 416: return WSplitGeo<416>::fwd_bytes(); case 448: return WSplitGeo<448>::fwd_bytes();
    default: return (size_t)1 << 30;      // 480, 512: the last K region does not fit twice - the 8-wave kernel stays
  }
}

template <int HP, int TERMS>
static int launch_one(const FwdArgs& a, int grid, hipStream_t s) {
  const size_t lds = WSplitGeo<HP>::fwd_bytes();
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_wsplit_kernel<HP, TERMS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((fwd_wsplit_kernel<HP, TERMS>), dim3(grid), dim3(512), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
template <int HP>
static int launch_hp(int terms, const FwdArgs& a, int grid, hipStream_t s) {
  return terms == 3 ? launch_one<HP, 3>(a, grid, s) : launch_one<HP, 1>(a, grid, s);
}

// residual mode, saved activations in the 24-bit format, L >= 2 hidden layers (the caller checks)
int launch_fwd_wsplit(int HP, int terms, const FwdArgs& a, int grid, hipStream_t s) {
  switch (HP) {
    case 288: return launch_hp<288>(terms, a, grid, s); case 320: return launch_hp<320>(terms, a, grid, s);
    case 352: return launch_hp<352>(terms, a, grid, s); case 384: return launch_hp<384>(terms, a, grid, s);
    case 416: return launch_hp<416>(terms, a, grid, s); case 448: return launch_hp<448>(terms, a, grid, s);
    default: return -1000;
  }
}
