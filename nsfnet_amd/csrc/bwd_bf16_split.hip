// Role-split bf16x3 reverse sweep for residual mode (4 streams): the schedule of fwd_bf16_split.hip (two wave groups
// per workgroup in opposite phases; SIMD partners overlap one group's MFMAs with the other's VALU / LDS / memory
// instructions) applied to bwd_bf16.hip - see there and bwd.hip for the algorithm and the reference lines it replaces
// (loss.backward(), NSFnet/pinn_solver.py:252, ev-NSFnet/pinn_solver.py:469).  Results layout (Z-bar spill,
// per-workgroup skinny-gradient accumulators, ebar) is unchanged: the dW / reduce kernels do not care which reverse
// sweep ran.
//
//     group 0:  E_{L-1}(A)  G_{L-1}(A)  E_{L-2}(A)  ...  G_1(A)  E_0(A) | E_{L-1}(A') ...
//     group 1:              E_{L-1}(B)  G_{L-1}(B)  ...          G_1(B)   E_0(B) | ...
//
// E_l = tanh adjoint of layer l: reads the saved (t, z_x, z_y, z_D) quads (requested SQ quads ahead: they stream from
// HBM; layer 0's are recomputed from the point instead - the role-split forward does not spill them), turns the a-stream adjoints (accumulators of G_{l+1}; for l = L-1 the rank-3 update W_out^T o-bar) into z-bar,
// column-sums the skinny gradients into the LDS accumulator, splits z-bar into bf16 hi/lo, spills it.
// G_l = W_l^T z-bar_l (MFMA only).  One shared z-bar image, four K regions, 32 parked registers: fwd_bf16_split.hip.
#include "kernels.h"
#include "point_stage.h"
#include "bf16_util.h"
#include "reduce_util.h"

__device__ __forceinline__ float acc_read_sb(float acc_elem) {      // just-in-time AGPR -> VGPR (see fwd_bf16_split.hip)
#if !defined(PINN_ACCV) || PINN_ACCV      // default: the accumulators live in arch VGPRs (MFMA in VGPR form), the epilogue reads them in place
  return acc_elem;
#else      // PINN_ACCV=0: accumulators pinned to AGPRs, one v_accvgpr_read per element (round 2; same speed, profiles/r03_ablations.txt C)
  float v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc_elem));
  return v;
#endif
}

template <int HP>
struct SplitBwdLds {
  using XI = XImg<HP, 32>;
  static constexpr size_t X_BYTES = XI::BYTES;                          // THE z-bar image (shared by the two groups)
  static constexpr size_t OADJ_F = (size_t)2 * 4 * 128;                 // [group][4][128] (3 outputs used)
  static constexpr size_t DUMMY_F = 64 * 8;                             // sink of the lanes that own no accumulator slot
  static size_t bytes(int L) { return X_BYTES + (OADJ_F + DUMMY_F + 6 * HP + (size_t)sg_total(HP, L)) * sizeof(float); }
};

template <int HP, int TERMS>
__global__ __launch_bounds__(2 * HP, 1) void bwd_split_kernel(BwdArgs a) {
  static_assert(HP == 256, "four waves x 64 features per group");
  using G = SplitBwdLds<HP>;
  using XI = typename G::XI;
  constexpr int GT = HP, KS = HP / 16, PPL = 32, COLS = 128;
#ifndef PINN_SRING
#define PINN_SRING 2
#endif
#ifndef PINN_ESB
#define PINN_ESB 1      // see fwd_bf16_split.hip
#endif
#ifndef PINN_PRIO
#define PINN_PRIO 0     // see fwd_bf16_split.hip
#endif
#define E_SB() do { if (PINN_ESB) __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef PINN_ABL
#define PINN_ABL 0      // timing-only ablation switches (scripts/abl_build.py): 1 = no Z-bar spill, 4 = S quads loaded once per phase,
                        // 8 = G phase without its MFMAs (operands still fetched), 16 = E phase reduced to its barriers,
                        // 2 = weights loaded once per phase, 128 / 256 = lo weight fragments for one feature block only / none, 512 = no Z-bar spill of the last hidden layer
#endif
#ifndef PINN_BDS
#define PINN_BDS 1      // B fragments (image reads) requested this many column-block steps ahead
#endif
  constexpr int RING = PINN_SRING, WPRE = RING - 1, BD = PINN_BDS;
  constexpr size_t PLQ = (size_t)(HP / 4) * PPL;          // f32x4 per plane of S / Z-bar
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* const X = ldsb;
  float* const oadjL = reinterpret_cast<float*>(ldsb + G::X_BYTES);         // [2][4][128]
  float* const dummy = oadjL + G::OADJ_F;
  float* const woutL = dummy + G::DUMMY_F;                                   // [3][HP]
  float* const w0L = woutL + 3 * HP;                                         // [w0x | w0y | b0][HP]
  float* const sgacc = w0L + 3 * HP;                                         // [sg_total]
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w = wave & 3;
  const int gtid = tid - grp * GT;
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * PPL;
  const int SG = sg_total(HP, L);
  float* const oadjG = oadjL + (size_t)grp * 4 * 128;
  for (int i = tid; i < SG; i += 2 * GT) sgacc[i] = 0.f;
  for (int i = tid; i < 3 * HP; i += 2 * GT) { woutL[i] = P[prep_wout(HP, L) + i]; w0L[i] = P[prep_w0x(HP) + i]; }
  for (int i = tid; i < (int)G::DUMMY_F; i += 2 * GT) dummy[i] = 0.f;
  float dbo[3] = {0.f, 0.f, 0.f};

  auto qbase = [&](int fb, int g) { return 64 * (2 * fb + (g >> 1)) + 16 * w + 8 * (g & 1); };
#define PHASE_LANE_B()                                 \
  int lane = lane0;                                    \
  asm volatile("" : "+v"(lane));                       \
  const int col = lane & 31, h = lane >> 5;            \
  (void)col; (void)h

  f32x16 acc[2][4];
  u32x2 st[2][4][2];
  bool have_parked = false;
#ifdef PINN_STAMP
  // diagnostic build only: s_memtime stamps of workgroup 0, wave 0 of each group, third pair, into the ebar buffer
  long long* const stamp = reinterpret_cast<long long*>(a.ebar) + grp * 1024;
  bool stamp_on = false;
  int nstamp = 0;
#define STAMP() do { if (stamp_on && nstamp < 1024) { if (lane0 == 0) stamp[nstamp] = __builtin_amdgcn_s_memtime(); ++nstamp; } } while (0)
#else
#define STAMP() do {} while (0)
#endif

  // saved-activation quads in flight: requested SQ quads ahead of their use, the first SQ of a phase already during the
  // LAST quarter of the G phase before it (whose registers are idle): no phase starts by waiting out the HBM latency
#ifndef PINN_SQ
#define PINN_SQ 2
#endif
  constexpr int SQ = PINN_SQ;
  u32x4 sq[SQ + 1][3];       // 24-bit spill format (bf16_util.h pack24): hi16 of streams 0-1, of streams 2-3, lo8 of all four
  auto quad_o = [&](int qq, int h) { return qbase(qq >> 2, qq & 3) + 4 * h; };      // qq = 4 fb + g in processing order
  auto sload = [&](const float* Sl, int qq, int col, int h) {
    const int o = quad_o(qq, h);
    const unsigned so = (unsigned)(((o - 4 * h) >> 2) + h) * PPL + col;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      sq[qq % (SQ + 1)][k] = __builtin_bit_cast(u32x4, __builtin_nontemporal_load(pin_base(reinterpret_cast<const f32x4*>(Sl) + k * PLQ) + so));
  };
  auto unpack_plane = [&](const u32x4 (&pk)[3], int p) {
    return unpack24(u32x2{pk[p >> 1][2 * (p & 1)], pk[p >> 1][2 * (p & 1) + 1]}, pk[2][p]);
  };
  auto s_layer = [&](int tile, int l) {      // the dummy partner of an odd tile count reads tile 0's (finite) S
    return a.S + spill_off(tile < a.ntiles ? tile : 0, l, L, a.sl0, a.sblk, (size_t)HP * COLS);
  };
#ifndef PINN_XPRE
#define PINN_XPRE 3     // cross-phase prefetch: 1 = S quads of E_{l-1} during G_l, 2 = first weight k-steps of G_l during E_l
#endif
  // weight-fragment ring of the G phases [feature block][k-step % RING]; lives across phases (PINN_XPRE & 2)
  u32x4 wh[2][RING], wl[2][RING];
  typedef __attribute__((address_space(1))) u32x4 gu32x4;
  auto wload = [&](int l, int s, int wlane) {
    const gu32x4* const wf = reinterpret_cast<const gu32x4*>(pin_base(reinterpret_cast<const u32x4*>(P + prep_wtf(HP, l))));
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
      wh[fb][s % RING] = (wf + (size_t)fb * 4 * KS * 64 + s * 64)[wlane];
      if (TERMS == 3 && !(PINN_ABL & 256) && !((PINN_ABL & 128) && fb == 1))      // (timing only: 128 = one lo fragment for both feature blocks, 256 = none)
        wl[fb][s % RING] = (wf + (size_t)(HP * HP / 8) + (size_t)fb * 4 * KS * 64 + s * 64)[wlane];
    }
  };
#define WLB_(fb, i) ((PINN_ABL & 256) ? wh[fb][i] : (PINN_ABL & 128) ? wl[0][i] : wl[fb][i])
  auto w_lane = [&](int col, int h) { return ((2 * (col >> 4) + (w >> 1)) * KS) * 64 + 16 * (w & 1) + (col & 15) + 32 * h; };

  auto dump = [&](int fb, int g0, int col, int h) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int off = XI::chunk_off(col, qbase(fb, g0 + k) >> 3) + 8 * h;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        *reinterpret_cast<u32x2*>(X + p * XI::PLANE * 2 + off) = st[k][p][0];
        if (TERMS == 3) *reinterpret_cast<u32x2*>(X + XI::HALF * 2 + p * XI::PLANE * 2 + off) = st[k][p][1];
      }
    }
  };

  // ---------------- G phase: acc <- W_l^T x z-bar image, region q in quarter q ----------------
  auto gphase = [&](int l, int tile, auto PRE_S) {
    constexpr bool pre_s = decltype(PRE_S)::value;      // (layer 0 is recomputed, not read: nothing to request before E_0)
    PHASE_LANE_B();
    if (PINN_PRIO) __builtin_amdgcn_s_setprio(PINN_PRIO == 2 ? 2 : 0);
    const int wlane = w_lane(col, h);
    u32x4 bh[BD + 1], bo[BD + 1];
    const float* const Snext = s_layer(tile, l - 1);
    auto bload = [&](int u) {
      const int s = u >> 2, j = u & 3;
      const int off = XI::chunk_off(col, 2 * s + h);
      bh[u % (BD + 1)] = *reinterpret_cast<const u32x4*>(X + j * XI::PLANE * 2 + off);
      if (TERMS == 3) bo[u % (BD + 1)] = *reinterpret_cast<const u32x4*>(X + XI::HALF * 2 + j * XI::PLANE * 2 + off);
    };
    if ((PINN_ABL & 2) || !(PINN_XPRE & 2)) {
#pragma unroll
      for (int s = 0; s < ((PINN_ABL & 2) ? RING : WPRE); ++s) wload(l, s, wlane);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      STAMP();
      if (q == 0 && have_parked) dump(1, 2, col, h);
#pragma unroll
      for (int d = 0; d < BD; ++d) bload(16 * q + d);
#pragma unroll
      for (int u = 16 * q; u < 16 * q + 16; ++u) {
        const int s = u >> 2, j = u & 3;
        if (j == 0 && s + WPRE < KS && !(PINN_ABL & 2)) wload(l, s + WPRE, wlane);
        // the next E phase's first saved-activation quads: younger than every weight request of this phase
        if ((PINN_XPRE & 1) && pre_s && u >= 4 * (KS - WPRE) && u < 4 * (KS - WPRE) + SQ) sload(Snext, u - 4 * (KS - WPRE), col, h);
        if ((u & 15) + BD <= 15) bload(u + BD);
        if (PINN_ABL & 8) {
          asm volatile("" :: "v"(bh[u % (BD + 1)]), "v"(bo[u % (BD + 1)]), "v"(wh[0][s % RING]), "v"(wh[1][s % RING]),
                       "v"(WLB_(0, s % RING)), "v"(WLB_(1, s % RING)));
          continue;
        }
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
          if (s == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[fb][j] = TERMS == 3 ? MFMA_Q(0, wh[fb][0], bo[u % (BD + 1)], zero) : MFMA_Q(0, wh[fb][0], bh[u % (BD + 1)], zero);
            if (TERMS == 3) {
              acc[fb][j] = MFMA_Q(1, WLB_(fb, 0), bh[u % (BD + 1)], acc[fb][j]);
              acc[fb][j] = MFMA_Q(0, wh[fb][0], bh[u % (BD + 1)], acc[fb][j]);
              if (PINN_ABL_SHAPE16) acc[fb][j] = MFMA_Q(1, wh[fb][0], bo[u % (BD + 1)], acc[fb][j]);      // (timing only: initialise the other half too)
            }
          } else {
            if (TERMS == 3) {
              acc[fb][j] = MFMA_Q(s, wh[fb][s % RING], bo[u % (BD + 1)], acc[fb][j]);
              acc[fb][j] = MFMA_Q(s + 1, WLB_(fb, s % RING), bh[u % (BD + 1)], acc[fb][j]);
            }
            acc[fb][j] = MFMA_Q(s, wh[fb][s % RING], bh[u % (BD + 1)], acc[fb][j]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      STAMP();
      __syncthreads();
    }
    have_parked = false;
  };

  // ---- output adjoints of a tile (point_stage.h) into the group's LDS block; zero for the dummy partner tile ----
  auto seeds = [&](int tile, float& px, float& py) {
    const int col = lane0 & 31;
    if (tile < a.ntiles) {
      float pxa[1], pya[1];
      output_adjoint_stage<PPL, COLS, 4, GT, 1>(a, tile, gtid, col, col, npad, oadjG, dbo, pxa, pya);
      px = pxa[0]; py = pya[0];
    } else {
      for (int i = gtid; i < 3 * COLS; i += GT) oadjG[i] = 0.f;
      px = py = 0.f;
    }
  };

  auto idle_ = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) __syncthreads();
  };
  // ---------------- E phase: tanh adjoint of layer lE of this group's tile ----------------
  // EK: 0 = last hidden layer L-1 (a-stream adjoints from the output adjoints on the VALU, dW_out), 1 = layer L-2..1,
  //     2 = layer 0 (dW_0; no image, nothing parked, no spill; the NEXT tile's output adjoints ride in quarter 3).
  auto ephase = [&](auto EKIND, int lE, int tileE, float pxE, float pyE, int next_tile, float& pxN, float& pyN) {
    constexpr int EK = decltype(EKIND)::value;
    constexpr bool first = EK == 0, last = EK == 2;
    PHASE_LANE_B();
    if (PINN_PRIO) __builtin_amdgcn_s_setprio(PINN_PRIO == 1 ? 2 : 0);
    if (PINN_ABL & 16) {
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[fb][j]));
      idle_();
      return;
    }
    const float* const Sl = s_layer(tileE, lE);
    float* const Zl = a.Zb + spill_off(tileE, lE, L, a.sl0, a.sblk, (size_t)HP * COLS);
    float oc[3][4];
    if (first) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s) oc[c][s] = oadjG[c * COLS + s * PPL + col];
    }
    auto commit = [&](int base, int o4, float v) {        // lanes col < 4 of each half own feature o4 + col (reduce_util.h)
      float* p = col < 4 ? &sgacc[base + o4 + (col & 3)] : &dummy[wave * 64 + lane];
      lds_rmw_add(p, v);      // (unconditional, the other lanes hit a sink: a plain read-modify-write costs the same for 8 lanes as for 64)
    };
    if (!last && (first || !(PINN_XPRE & 1))) {      // (every other E phase follows a G phase, which has requested them)
#pragma unroll
      for (int qq = 0; qq < SQ; ++qq) sload(Sl, qq, col, h);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      STAMP();
      if (q > 0 && !last) dump((q - 1) >> 1, 2 * ((q - 1) & 1), col, h);
      const int fb = q >> 1;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int g = 2 * (q & 1) + k, qq = 2 * q + k, o = quad_o(qq, h);
        if (!last && qq + SQ < 8 && !(PINN_ABL & 4)) sload(Sl, qq + SQ, col, h);
        // first weight k-steps of the G phase that follows (its first MFMA would otherwise wait out an L2 round trip)
        if ((PINN_XPRE & 2) && !(PINN_ABL & 2) && !last && qq == 7) {
#pragma unroll
          for (int s = 0; s < WPRE; ++s) wload(lE, s, w_lane(col, h));
        }
        f32x4 sc[4];
        if (last) {
          // layer 0 is not spilled (fwd_bf16_split.hip): same two FMAs and tanh as the forward, bit for bit
          const f32x4 wx4 = *reinterpret_cast<const f32x4*>(w0L + o), wy4 = *reinterpret_cast<const f32x4*>(w0L + HP + o);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(w0L + 2 * HP + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) sc[0][e] = fast_tanh(fmaf(wx4[e], pxE, fmaf(wy4[e], pyE, b4[e])));
          sc[1] = wx4; sc[2] = wy4; sc[3] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
          for (int p = 0; p < 4; ++p) sc[p] = unpack_plane(sq[qq % (SQ + 1)], p);
        }
        f32x4 zq[4], wov[3], dwv[2], wo4[3];
        if (first) {
#pragma unroll
          for (int c = 0; c < 3; ++c) wo4[c] = *reinterpret_cast<const f32x4*>(woutL + c * HP + o);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float ga, gx, gy, gd;
          if (first) {      // adjoint of the last hidden layer's a-streams: rank-3 update from the output adjoints
            ga = wo4[0][e] * oc[0][0] + wo4[1][e] * oc[1][0] + wo4[2][e] * oc[2][0];
            gx = wo4[0][e] * oc[0][1] + wo4[1][e] * oc[1][1] + wo4[2][e] * oc[2][1];
            gy = wo4[0][e] * oc[0][2] + wo4[1][e] * oc[1][2] + wo4[2][e] * oc[2][2];
            gd = wo4[0][e] * oc[0][3] + wo4[1][e] * oc[1][3] + wo4[2][e] * oc[2][3];
          } else {
            ga = acc_read_sb(acc[fb][0][r]); gx = acc_read_sb(acc[fb][1][r]); gy = acc_read_sb(acc[fb][2][r]);
            gd = acc_read_sb(acc[fb][3][r]);
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
          E_SB();
        }
        // column sums of the four features at once (reduce_util.h); lane col == e of each half commits feature e
        const int o4 = o;     // (= qbase + 4h: the lane half's four features)
        commit(sg_db(HP, lE), o4, sum_cols4<32>(zq[0][0], zq[0][1], zq[0][2], zq[0][3], lane));
        if (first) {
#pragma unroll
          for (int c = 0; c < 3; ++c)
            commit(sg_wout(HP, L) + c * HP, o4, sum_cols4<32>(wov[c][0], wov[c][1], wov[c][2], wov[c][3], lane));
        }
        if (last) {
          commit(sg_w0x(HP, L), o4, sum_cols4<32>(dwv[0][0], dwv[0][1], dwv[0][2], dwv[0][3], lane));
          commit(sg_w0y(HP, L), o4, sum_cols4<32>(dwv[1][0], dwv[1][1], dwv[1][2], dwv[1][3], lane));
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP();
        if (!last) {
          const unsigned so = (unsigned)(((o - 4 * h) >> 2) + h) * PPL + col;
          u32x4 pk[3];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            split4(zq[p][0], zq[p][1], zq[p][2], zq[p][3], st[k][p][0], st[k][p][1]);
            if (!(PINN_ABL & 1) && !((PINN_ABL & 512) && first)) {      // 24-bit spill (bf16_util.h pack24): three 16-byte planes instead of four
              u32x2 hi24; unsigned lo24;
              pack24(zq[p], hi24, lo24);
              pk[p >> 1][2 * (p & 1)] = hi24[0]; pk[p >> 1][2 * (p & 1) + 1] = hi24[1]; pk[2][p] = lo24;
              if (p & 1) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, pk[p >> 1]), pin_base(reinterpret_cast<const f32x4*>(Zl) + (p >> 1) * PLQ) + so);
              if (p == 3) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, pk[2]), pin_base(reinterpret_cast<const f32x4*>(Zl) + 2 * PLQ) + so);
            }
            E_SB();
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (last && q == 3 && next_tile >= 0) seeds(next_tile, pxN, pyN);      // the group's next tile: its output adjoints
      STAMP();
      __syncthreads();
    }
    have_parked = !last;
  };
  auto idle = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) __syncthreads();
  };

  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  // Straight-line program per group (fwd_bf16_split.hip): per tile E_{L-1} G_{L-1} E_{L-2} ... G_1 E_0, group 1 one
  // phase behind group 0.  Tile of pair i: 2 i + grp.
  const int npairs = (a.ntiles + 1) / 2;
  float px = 0.f, py = 0.f, pxN = 0.f, pyN = 0.f;
  if ((int)blockIdx.x < npairs) seeds(2 * (int)blockIdx.x + grp, px, py);
  __syncthreads();
  if (grp == 1) idle();
  for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
    const int tile = 2 * pair + grp;
#ifdef PINN_STAMP
    stamp_on = blockIdx.x == 0 && w == 0 && pair == (int)blockIdx.x + 2 * (int)gridDim.x;
#endif
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

size_t bwd_split_lds_bytes(int HP, int L) { (void)HP; return SplitBwdLds<256>::bytes(L); }

template <int HP, int TERMS>
static int launch_one(const BwdArgs& a, int grid, hipStream_t s) {
  const size_t lds = SplitBwdLds<HP>::bytes(a.L);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_split_kernel<HP, TERMS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((bwd_split_kernel<HP, TERMS>), dim3(grid), dim3(2 * HP), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// residual mode, L >= 2 hidden layers, HP = 256 (the caller checks)
int launch_bwd_split(int HP, int terms, const BwdArgs& a, int grid, hipStream_t s) {
  if (HP != 256) return -1000;
  return terms == 3 ? launch_one<256, 3>(a, grid, s) : launch_one<256, 1>(a, grid, s);
}
