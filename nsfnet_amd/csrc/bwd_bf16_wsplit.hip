// Role-split bf16x3 reverse sweep for WIDE nets (256 < hidden <= 448), residual mode: the schedule of bwd_bf16_split.hip
// at the tile geometry of bwd_bf16_wide.hip (64 columns = 16 points x 4 streams, S and Z-bar in the 24-bit three-plane
// format, classic [tile][L] blocks) - see fwd_bf16_wsplit.hip for the geometry (blocks 4 q + w per wave, MQ K regions,
// the last region's per-group copy) and bwd.hip for the algorithm and the reference lines it replaces
// (loss.backward(), NSFnet/pinn_solver.py:252, ev-NSFnet/pinn_solver.py:469).  Results layout (Z-bar, per-workgroup
// skinny-gradient accumulators, ebar) is that of bwd_bf16_wide.hip: dw_bf16_wide.hip / reduce do not care which reverse
// sweep ran.  Layer 0's saved activations are recomputed from the point (they are one FMA pair and one tanh), not read.
//
//     group 0:  E_{L-1}(A)  G_{L-1}(A)  E_{L-2}(A)  ...  G_1(A)  E_0(A) | E_{L-1}(A') ...
//     group 1:              E_{L-1}(B)  G_{L-1}(B)  ...          G_1(B)   E_0(B) | ...
#include "kernels.h"
#include "point_stage.h"
#include "bf16_util.h"
#ifndef PINN_ABL
#define PINN_ABL 0      // timing-only ablation switches (scripts/abl_build.py)
#endif
#include "reduce_util.h"

template <int HP>
struct WSplitBwdGeo {
  using XI = XImg<HP, 16>;
  static constexpr int NB = HP / 32, MQ = (NB + 3) / 4, KS = HP / 16;
  static constexpr int LASTK = 128 * (MQ - 1), LASTN = HP - LASTK;
  static constexpr bool FITS = XI::RSE - HP >= LASTN;
  static constexpr size_t X_BYTES = XI::BYTES;
  static constexpr size_t OADJ_F = (size_t)2 * 4 * 64;                  // [group][4][64] (3 outputs used)
  static constexpr size_t DUMMY_F = 64 * 8;                             // sink of the lanes that own no accumulator slot
  static size_t bytes(int L) { return X_BYTES + (OADJ_F + DUMMY_F + (size_t)sg_total(HP, L)) * sizeof(float); }
};

template <int HP, int TERMS>
__global__ __launch_bounds__(512, 1) void bwd_wsplit_kernel(BwdArgs a) {
  using G = WSplitBwdGeo<HP>;
  using XI = typename G::XI;
  static_assert(HP > 256 && HP <= 512 && G::FITS, "hidden widths whose last K region fits twice in the image rows");
  constexpr int GT = 256, NB = G::NB, MQ = G::MQ, KS = G::KS, PPL = 16, COLS = 64;
  constexpr int RING = 2, SQ = 2, NQD = 2 * MQ;           // NQD: register quads of a wave per phase, qq = 2 q + k
  constexpr size_t PLQ = (size_t)(HP / 4) * PPL;          // f32x4 per plane of S / Z-bar
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* const X = ldsb;
  float* const oadjL = reinterpret_cast<float*>(ldsb + G::X_BYTES);         // [2][4][64]
  float* const dummy = oadjL + G::OADJ_F;
  float* const sgacc = dummy + G::DUMMY_F;                                   // [sg_total]
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w = wave & 3;
  const int gtid = tid - grp * GT;
  const int mc = (NB - w + 3) / 4;
  const float* __restrict__ P = a.prep;
  const float* const woutG = P + prep_wout(HP, a.L);      // [3][HP]  (global: the LDS holds the image and the accumulators)
  const float* const w0G = P + prep_w0x(HP);              // [w0x | w0y | b0][HP]
  const int L = a.L;
  const int npad = a.ntiles * PPL;
  const int SG = sg_total(HP, L);
  float* const oadjG = oadjL + (size_t)grp * 4 * 64;
  for (int i = tid; i < SG; i += 2 * GT) sgacc[i] = 0.f;
  for (int i = tid; i < (int)G::DUMMY_F; i += 2 * GT) dummy[i] = 0.f;
  float dbo[3] = {0.f, 0.f, 0.f};

#define WSB_LANE()                                     \
  int lane = lane0;                                    \
  asm volatile("" : "+v"(lane));                       \
  const int col = lane & 31, h = lane >> 5;            \
  const int hi = col >> 4, pp = col & 15;              \
  (void)h; (void)hi; (void)pp

  f32x16 acc[MQ][2];
  u32x2 st[2][4][2];

  // saved-activation quads in flight (24-bit format): requested SQ quads ahead, the first SQ of a phase already during
  // the last k-step of the G phase before it
  u32x4 sq[SQ + 1][3];
  // Items of a phase in processing order (fwd_bf16_wsplit.hip: the last block's two quads ride in quarters 0 and 1):
  //   quarter 0: (block 0, quad 0) (0, 1) (MQ-1, 0) | quarter 1: (1, 0) (1, 1) (MQ-1, 1) | quarter q >= 2: (q, 0) (q, 1) | last: none
  auto item_bq = [](int i) { return i < 6 ? ((i % 3) == 2 ? MQ - 1 : i / 3) : 2 + (i - 6) / 2; };
  auto item_k = [](int i) { return i < 6 ? ((i % 3) == 2 ? i / 3 : i % 3) : (i - 6) % 2; };
  auto quad_o = [&](int bq, int k, int hi, int h) { return 32 * (4 * bq + w) + 8 * (k + 2 * hi) + 4 * h; };
  auto sload = [&](const float* Sl, int i, int pp, int hi, int h) {
    if (item_bq(i) >= mc) return;                       // (uniform: this wave owns no block in the last region)
    const unsigned so = (unsigned)(quad_o(item_bq(i), item_k(i), hi, h) >> 2) * PPL + pp;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      sq[i % (SQ + 1)][k] = __builtin_bit_cast(u32x4, __builtin_nontemporal_load(pin_base(reinterpret_cast<const f32x4*>(Sl) + k * PLQ) + so));
  };
  auto unpack_plane = [&](const u32x4 (&pk)[3], int p) {
    return unpack24(u32x2{pk[p >> 1][2 * (p & 1)], pk[p >> 1][2 * (p & 1) + 1]}, pk[2][p]);
  };
  auto s_layer = [&](int tile, int l) {      // the dummy partner of an odd tile count reads tile 0's (finite) S
    return a.S + ((size_t)(tile < a.ntiles ? tile : 0) * L + l) * ((size_t)HP * COLS);
  };
  u32x4 wh[MQ][RING], wl[MQ][RING];
  typedef __attribute__((address_space(1))) u32x4 gu32x4;
  auto wload = [&](int l, int s, int lane) {
    const gu32x4* const wf = reinterpret_cast<const gu32x4*>(pin_base(reinterpret_cast<const u32x4*>(P + prep_wtf(HP, l))));
#pragma unroll
    for (int m = 0; m < MQ; ++m) {
      if (m == MQ - 1 && m >= mc) continue;
      wh[m][s % RING] = (wf + (size_t)(4 * m + w) * KS * 64 + s * 64)[lane];
      if (TERMS == 3 && !((PINN_ABL & 128) && (m & 1)))      // (PINN_ABL 128, timing only: lo fragments of every other block -> 3/4 of the weight bytes)
        wl[m][s % RING] = (wf + (size_t)(HP * HP / 8) + (size_t)(4 * m + w) * KS * 64 + s * 64)[lane];
    }
  };
#define WL_W(m, i) (((PINN_ABL & 128) && ((m) & 1)) ? wl[(m) - 1][i] : wl[m][i])
  auto img_chunk = [&](int o, int g) { return (o >> 3) + ((o >= G::LASTK && g) ? G::LASTN / 8 : 0); };
  auto dump_k = [&](int b, int k, int pp, int hi, int h) {
    const int off = XI::chunk_off(pp, img_chunk(32 * b + 8 * (k + 2 * hi), grp)) + 8 * h;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *reinterpret_cast<u32x2*>(X + p * XI::PLANE * 2 + off) = st[k][p][0];
      if (TERMS == 3) *reinterpret_cast<u32x2*>(X + XI::HALF * 2 + p * XI::PLANE * 2 + off) = st[k][p][1];
    }
  };

  // ---------------- G phase: acc <- W_l^T x z-bar image, region q in quarter q ----------------
  auto gphase = [&](int l, int tile, auto PRE_S) {
    constexpr bool pre_s = decltype(PRE_S)::value;      // (layer 0 is recomputed, not read: nothing to request before E_0)
    WSB_LANE();
    u32x4 bh[2], bo[2];
    const float* const Snext = s_layer(tile, l - 1);
    auto bload = [&](int u) {
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
        if (j == 0 && s + 1 < KS) wload(l, s + 1, lane);
        // the next E phase's first saved-activation quads: younger than every weight request of this phase
        if (pre_s && s == KS - 1) sload(Snext, j, pp, hi, h);
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
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  };

  // ---- output adjoints of a tile (point_stage.h) into the group's LDS block; zero for the dummy partner tile ----
  auto seeds = [&](int tile, float& px, float& py) {
    const int col = lane0 & 31;
    if (tile < a.ntiles) {
      float pxa[1], pya[1];
      output_adjoint_stage<PPL, COLS, 4, GT, 1>(a, tile, gtid, col, col & 15, npad, oadjG, dbo, pxa, pya);
      px = pxa[0]; py = pya[0];
    } else {
      for (int i = gtid; i < 3 * COLS; i += GT) oadjG[i] = 0.f;
      px = py = 0.f;
    }
  };

  // ---------------- E phase: tanh adjoint of layer lE of this group's tile ----------------
  // EK: 0 = last hidden layer L-1 (a-stream adjoints from the output adjoints on the VALU, dW_out), 1 = layer L-2..1,
  //     2 = layer 0 (dW_0; no image, nothing parked, no spill; the NEXT tile's output adjoints ride in the last quarter).
  auto ephase = [&](auto EKIND, int lE, int tileE, float pxE, float pyE, int next_tile, float& pxN, float& pyN) {
    constexpr int EK = decltype(EKIND)::value;
    constexpr bool first = EK == 0, last = EK == 2;
    WSB_LANE();
    const float* const Sl = s_layer(tileE, lE);
    float* const Zl = a.Zb + ((size_t)tileE * L + lE) * ((size_t)HP * COLS);
    float oc[3][4];
    if (first) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s) oc[c][s] = oadjG[c * COLS + s * PPL + pp];
    }
    auto commit = [&](int base, int o4, float v) {        // lanes pp < 4 of each 16-lane row own feature o4 + pp (reduce_util.h)
      float* p = pp < 4 ? &sgacc[base + o4 + pp] : &dummy[wave * 64 + lane];
      lds_rmw_add(p, v);
    };
    if (!last && first) {      // (every other E phase follows a G phase, which has requested them)
#pragma unroll
      for (int qq = 0; qq < SQ; ++qq) sload(Sl, qq, pp, hi, h);
    }
    auto swaps = [&](int bq) {      // the four streams of a point into one lane (fwd_bf16_wsplit.hip)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[bq][0][r]), __float_as_uint(acc[bq][0][r + 8]), false, false);
        auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[bq][1][r]), __float_as_uint(acc[bq][1][r + 8]), false, false);
        acc[bq][0][r] = __uint_as_float(s01[0]); acc[bq][0][r + 8] = __uint_as_float(s01[1]);
        acc[bq][1][r] = __uint_as_float(s23[0]); acc[bq][1][r + 8] = __uint_as_float(s23[1]);
      }
    };
    // item i = quad k of block bq: z-bar of its four features x four streams, skinny-gradient column sums
    auto compute = [&](int i, f32x4 (&zq)[4]) {
      const int bq = item_bq(i), k = item_k(i), o = quad_o(bq, k, hi, h);
      f32x4 sc[4];
      if (last) {
        // layer 0: same two FMAs and tanh as the forward, bit for bit
        const f32x4 wx4 = *reinterpret_cast<const f32x4*>(w0G + o), wy4 = *reinterpret_cast<const f32x4*>(w0G + HP + o);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(w0G + 2 * HP + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) sc[0][e] = fast_tanh(fmaf(wx4[e], pxE, fmaf(wy4[e], pyE, b4[e])));
        sc[1] = wx4; sc[2] = wy4; sc[3] = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) sc[p] = unpack_plane(sq[i % (SQ + 1)], p);
      }
      f32x4 wov[3], dwv[2], wo4[3];
      if (first) {
#pragma unroll
        for (int c = 0; c < 3; ++c) wo4[c] = *reinterpret_cast<const f32x4*>(woutG + c * HP + o);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * k + e;
        float ga, gx, gy, gd;
        if (first) {      // adjoint of the last hidden layer's a-streams: rank-3 update from the output adjoints
          ga = wo4[0][e] * oc[0][0] + wo4[1][e] * oc[1][0] + wo4[2][e] * oc[2][0];
          gx = wo4[0][e] * oc[0][1] + wo4[1][e] * oc[1][1] + wo4[2][e] * oc[2][1];
          gy = wo4[0][e] * oc[0][2] + wo4[1][e] * oc[1][2] + wo4[2][e] * oc[2][2];
          gd = wo4[0][e] * oc[0][3] + wo4[1][e] * oc[1][3] + wo4[2][e] * oc[2][3];
        } else {
          ga = acc[bq][0][r]; gx = acc[bq][0][r + 8]; gy = acc[bq][1][r]; gd = acc[bq][1][r + 8];
        }
        const float t = sc[0][e], zx = sc[1][e], zy = sc[2][e], zd = sc[3][e];
        const float d1 = 1.f - t * t;
        const float d2 = -2.f * t * d1;
        const float d3 = -2.f * d1 * (1.f - 3.f * t * t);
        const float zz = zx * zx + zy * zy;
        zq[1][e] = d1 * gx + 2.f * d2 * zx * gd;
        zq[2][e] = d1 * gy + 2.f * d2 * zy * gd;
        zq[3][e] = d1 * gd;
        zq[0][e] = d1 * ga + d2 * (zx * gx + zy * gy) + (d3 * zz + d2 * zd) * gd;
        if (first) {      // dWout[c][o] += sum_s oadj[c][s] * a_s[o]
          const float ax = d1 * zx, ay = d1 * zy, ad = d2 * zz + d1 * zd;
#pragma unroll
          for (int c = 0; c < 3; ++c) wov[c][e] = oc[c][0] * t + oc[c][1] * ax + oc[c][2] * ay + oc[c][3] * ad;
        }
        if (last) { dwv[0][e] = zq[0][e] * pxE + zq[1][e]; dwv[1][e] = zq[0][e] * pyE + zq[2][e]; }
        __builtin_amdgcn_sched_barrier(0);
      }
      // column sums over the 16 points of a lane row, four features at once (reduce_util.h)
      commit(sg_db(HP, lE), o, sum_cols4<16>(zq[0][0], zq[0][1], zq[0][2], zq[0][3], lane));
      if (first) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
          commit(sg_wout(HP, L) + c * HP, o, sum_cols4<16>(wov[c][0], wov[c][1], wov[c][2], wov[c][3], lane));
      }
      if (last) {
        commit(sg_w0x(HP, L), o, sum_cols4<16>(dwv[0][0], dwv[0][1], dwv[0][2], dwv[0][3], lane));
        commit(sg_w0y(HP, L), o, sum_cols4<16>(dwv[1][0], dwv[1][1], dwv[1][2], dwv[1][3], lane));
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // hi/lo split of z-bar (parked in st[k], or straight into this group's copy of the last region) and its 24-bit spill
    auto finish = [&](int i, f32x4 (&zq)[4], bool direct) {
      const int bq = item_bq(i), k = item_k(i), o = quad_o(bq, k, hi, h);
      const unsigned so = (unsigned)(o >> 2) * PPL + pp;
      u32x4 pk[3];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (direct) {
          u32x2 th, tl;
          split4(zq[p][0], zq[p][1], zq[p][2], zq[p][3], th, tl);
          const int off = XI::chunk_off(pp, img_chunk(o - 4 * h, grp)) + 8 * h;
          *reinterpret_cast<u32x2*>(X + p * XI::PLANE * 2 + off) = th;
          if (TERMS == 3) *reinterpret_cast<u32x2*>(X + XI::HALF * 2 + p * XI::PLANE * 2 + off) = tl;
        } else {
          split4(zq[p][0], zq[p][1], zq[p][2], zq[p][3], st[k][p][0], st[k][p][1]);
        }
        u32x2 hi24; unsigned lo24;      // 24-bit spill (bf16_util.h pack24): three 16-byte planes
        pack24(zq[p], hi24, lo24);
        pk[p >> 1][2 * (p & 1)] = hi24[0]; pk[p >> 1][2 * (p & 1) + 1] = hi24[1]; pk[2][p] = lo24;
        if (p & 1) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, pk[p >> 1]), pin_base(reinterpret_cast<const f32x4*>(Zl) + (p >> 1) * PLQ) + so);
        if (p == 3) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, pk[2]), pin_base(reinterpret_cast<const f32x4*>(Zl) + 2 * PLQ) + so);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
      const bool mainb = q < MQ - 1;                     // blocks 0 .. MQ - 2 ride in their own quarter (every wave owns them)
      const bool prev = q > 0 && !last;                  // block q - 1 is parked and its region is free now
      const bool extra = q < 2 && mc == MQ;              // the last block's quads ride in quarters 0 and 1 (uniform)
      const int i0 = q < 2 ? 3 * q : 6 + 2 * (q - 2);    // first item of this quarter
      if (mainb && !first) swaps(q);
      if (q == 0 && mc == MQ && !first) swaps(MQ - 1);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = i0 + k;
        if (mainb && !last && i + SQ < NQD) sload(Sl, i + SQ, pp, hi, h);
        // first weight k-step of the G phase that follows (its first MFMA would otherwise wait out an L2 round trip)
        if (!last && q == MQ - 1 && k == 1) wload(lE, 0, lane);
        f32x4 zq[4];
        if (mainb) compute(i, zq);
        // block q - 1, parked in the previous quarter: its region is free now; quad k leaves its registers before the refill
        if (prev) dump_k(4 * (q - 1) + w, k, pp, hi, h);
        if (mainb && !last) finish(i, zq, false);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (q < 2) {
        const int i = i0 + 2;
        if (!last && i + SQ < NQD) sload(Sl, i + SQ, pp, hi, h);
        if (extra) {
          f32x4 zq[4];
          compute(i, zq);
          if (!last) finish(i, zq, true);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (last && q == MQ - 1 && next_tile >= 0) seeds(next_tile, pxN, pyN);      // the group's next tile: its output adjoints
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
  const int npairs = (a.ntiles + 1) / 2;
  float px = 0.f, py = 0.f, pxN = 0.f, pyN = 0.f;
  if ((int)blockIdx.x < npairs) seeds(2 * (int)blockIdx.x + grp, px, py);
  __syncthreads();
  if (grp == 1) idle();
  for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
    const int tile = 2 * pair + grp;
    const int next_tile = pair + (int)gridDim.x < npairs ? 2 * (pair + (int)gridDim.x) + grp : -1;
    ephase(K0{}, L - 1, tile, px, py, -1, pxN, pyN);
    for (int l = L - 1; l >= 2; --l) {
      gphase(l, tile, std::true_type{});
      ephase(K1{}, l - 1, tile, px, py, -1, pxN, pyN);
    }
    gphase(1, tile, std::false_type{});
    ephase(K2{}, 0, tile, px, py, next_tile, pxN, pyN);
    px = pxN; py = pyN;
  }
  if (grp == 0) idle();
  // ---------------- flush ----------------
  float* red = reinterpret_cast<float*>(ldsb);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 3; ++c) red[c * 2 * GT + tid] = dbo[c];
  __syncthreads();
  if (tid < 3) {
    float s = 0.f;
    for (int t = 0; t < 2 * GT; ++t) s += red[tid * 2 * GT + t];
    sgacc[sg_bout(HP, L) + tid] = s;
  }
  __syncthreads();
  float* out = a.sg + (size_t)blockIdx.x * SG;
  for (int i = tid; i < SG; i += 2 * GT) out[i] = sgacc[i];
}

size_t bwd_wsplit_lds_bytes(int HP, int L) {
  switch (HP) {
    case 288: return WSplitBwdGeo<288>::bytes(L); case 320: return WSplitBwdGeo<320>::bytes(L);
    case 352: return WSplitBwdGeo<352>::bytes(L); case 384: return WSplitBwdGeo<384>::bytes(L);
    case 416: return WSplitBwdGeo<416>::bytes(L); case 448: return WSplitBwdGeo<448>::bytes(L);
    default: return (size_t)1 << 30;
  }
}

template <int HP, int TERMS>
static int launch_one(const BwdArgs& a, int grid, hipStream_t s) {
  const size_t lds = WSplitBwdGeo<HP>::bytes(a.L);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_wsplit_kernel<HP, TERMS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((bwd_wsplit_kernel<HP, TERMS>), dim3(grid), dim3(512), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
template <int HP>
static int launch_hp(int terms, const BwdArgs& a, int grid, hipStream_t s) {
  return terms == 3 ? launch_one<HP, 3>(a, grid, s) : launch_one<HP, 1>(a, grid, s);
}

// residual mode, 24-bit spill, L >= 2 hidden layers (the caller checks)
int launch_bwd_wsplit(int HP, int terms, const BwdArgs& a, int grid, hipStream_t s) {
  switch (HP) {
    case 288: return launch_hp<288>(terms, a, grid, s); case 320: return launch_hp<320>(terms, a, grid, s);
    case 352: return launch_hp<352>(terms, a, grid, s); case 384: return launch_hp<384>(terms, a, grid, s);
    case 416: return launch_hp<416>(terms, a, grid, s); case 448: return launch_hp<448>(terms, a, grid, s);
    default: return -1000;
  }
}
