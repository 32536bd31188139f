// Role-split bf16x3 forward sweep for residual mode (4 streams): two wave groups per workgroup in opposite phases.
//
// Same algorithm and results layout (S, field planes, loss partials) as fwd_bf16.hip - see there and fwd.hip for the
// reference lines replaced (NSFnet/net.py:52-54, NSFnet/pinn_solver.py:132-163,197-226,
// ev-NSFnet/pinn_solver.py:290-342,372-428).
//
// What the schedule rests on (tests/micro/mfma_valu_partner_pad.hip, MI355X round 2): on one SIMD, a wave issuing
// v_mfma_f32_32x32x16_bf16 back to back keeps its 32 cycles per MFMA while its PARTNER wave's VALU stream runs at
// 4.7 instructions per MFMA slot (78 % of its solo rate), and the partner's LDS / vector-memory instructions issue on
// ports the MFMA wave does not use.  One wave alone cannot do that for itself (every instruction of a wave issues in
// order: fwd_bf16_pipe.hip, one wave per SIMD, is issue-bound at ~45 cycles per MFMA).  What the schedule does NOT
// escape: the CU's one in-order vector-memory path - the M group's weight-fragment loads queue behind
// the E group's S stores, which drain at the HBM rate, so the M quarters run 1.2-1.4x their solo time.
//
// So: 512 threads = two groups of four waves; waves w and w + 4 are SIMD partners.  Group 0 owns tile A, group 1 tile
// B (32 points x 4 streams each); within a group wave w owns 64 features.  The groups run the SAME program one phase
// apart:          group 0:  E0(A)  M1(A)  E1(A)  M2(A) ...  M_{L-1}(A)  E_{L-1}(A) | E0(A') ...
//                 group 1:         E0(B)  M1(B)  E1(B) ...              M_{L-1}(B)  E_{L-1}(B) | ...
// M_l = hidden GEMM l (MFMA only: W fragments from L2 through a register ring, B fragments from the LDS image),
// E_l = tanh chain rule of layer l, bf16 hi/lo split, S spill (VALU / LDS / VMEM only).  In every phase but one per
// tile pair a SIMD has one wave in M and its partner in E.  A wave keeps ONE tile's accumulators (128 registers): no
// per-wave tile duplication, no hand interleave of two instruction streams - the hardware overlaps the partners.
//
// LDS: one tile's hi/lo image is 128 KB at HP = 256, two do not fit.  The groups SHARE one image, split along K into
// four 64-feature regions R0..R3; a phase is four quarters with a workgroup barrier after each.  The M group reads
// region q in quarter q.  A wave owns 16 features of every region (rows 0-15 / 16-31 of its two 32-row MFMA blocks map
// to regions 2fb / 2fb+1), so the E group computes its region-q quads in quarter q, parks them in 32 registers, and
// writes them into the image in quarter q + 1, when the M group has finished with that region (region 3: in quarter 0
// of the wave's own following M phase).  The output layer is folded into the last epilogue.
#include "kernels.h"
#include "point_stage.h"
#include "bf16_util.h"

// Just-in-time AGPR -> VGPR read of one accumulator element (keeps the register allocator from copying the whole
// 128-register accumulator set into VGPRs at the top of the epilogue)
__device__ __forceinline__ float acc_read_s(float acc_elem) {
#if !defined(PINN_ACCV) || PINN_ACCV      // default: the accumulators live in arch VGPRs (MFMA in VGPR form), the epilogue reads them in place
  return acc_elem;
#else      // PINN_ACCV=0: accumulators pinned to AGPRs, one v_accvgpr_read per element (round 2; same speed, profiles/r03_ablations.txt C)
  float v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc_elem));
  return v;
#endif
}

template <int HP>
struct SplitLds {
  using XI = XImg<HP, 32>;
  static constexpr size_t X_BYTES = XI::BYTES;                         // THE tile image (shared by the two groups)
  static constexpr size_t PART_F = (size_t)2 * 4 * 12 * 32;            // [group][wave][3 outputs x 4 streams][32 points]
  static constexpr size_t OUTV_F = (size_t)2 * 3 * 128;                // [group][3][128]
  static size_t bytes(int L) { return X_BYTES + (PART_F + OUTV_F + (size_t)L * HP + 6 * HP) * sizeof(float); }
};

template <int HP, int TERMS>
__global__ __launch_bounds__(2 * HP, 1) void fwd_split_kernel(FwdArgs a) {
  static_assert(HP == 256, "four waves x 64 features per group");
  using G = SplitLds<HP>;
  using XI = typename G::XI;
  constexpr int GT = HP, KS = HP / 16, PPL = 32, COLS = 128;      // GT: threads per group
#ifndef PINN_ABL
#define PINN_ABL 0      // timing-only ablation switches (scripts/abl_build.py): 1 = no S spill, 2 = weights loaded once per phase,
                        // 8 = M phase without its MFMAs (operands still fetched), 16 = E phase reduced to its barriers,
                        // 32 = no image writes, 64 = no tanh
#endif
#ifndef PINN_SRING
#define PINN_SRING 2
#endif
#ifndef PINN_ESB
#define PINN_ESB 1      // 1: the epilogue's elements / planes are scheduled one at a time (sched_barrier between them); 0: the compiler may interleave them
#endif
#ifndef PINN_PRIO
#define PINN_PRIO 0     // wave priority by phase: 1 = raised in the E phases, 2 = raised in the M phases (s_setprio; SIMD partners arbitrate by priority, then age)
#endif
#define E_SB() do { if (PINN_ESB) __builtin_amdgcn_sched_barrier(0); } while (0)
  constexpr int RING = PINN_SRING, WPRE = RING - 1;                // weight k-steps in the register ring / requested ahead
  constexpr size_t PLQ = (size_t)(HP / 4) * PPL;                   // f32x4 per S plane
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* const X = ldsb;
  float* const part = reinterpret_cast<float*>(ldsb + G::X_BYTES);
  float* const outv = part + G::PART_F;
  float* const biasL = outv + G::OUTV_F;                  // [L][HP], row 0 = zeros
  float* const woutL = biasL + (size_t)a.L * HP;          // [3][HP]
  float* const w0L = woutL + 3 * HP;                      // [w0x | w0y | b0][HP]
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w = wave & 3;
  const int gtid = tid - grp * GT;                        // thread index inside the group
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * PPL;
  float* const partG = part + (size_t)grp * 4 * 12 * 32;
  float* const outvG = outv + (size_t)grp * 3 * 128;
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < L * HP; i += 2 * GT) biasL[i] = i < HP ? 0.f : P[prep_b(HP, i / HP) + (i % HP)];
  for (int i = tid; i < 3 * HP; i += 2 * GT) { woutL[i] = P[prep_wout(HP, L) + i]; w0L[i] = P[prep_w0x(HP) + i]; }
  __syncthreads();

  // feature geometry: row i of the wave's 32-row block fb is feature 64 (2 fb + (i >> 4)) + 16 w + (i & 15), i.e. the
  // register quad (fb, g) (rows 8g + 4h + e) holds features qbase(fb, g) + 4h + e of region 2 fb + (g >> 1)
  auto qbase = [&](int fb, int g) { return 64 * (2 * fb + (g >> 1)) + 16 * w + 8 * (g & 1); };
  // Lane geometry is re-derived inside every phase from an opaque copy of the lane id: address arithmetic then lives
  // in the phase that uses it instead of being hoisted in front of the phase loop (it was: 139 spilled registers).
#define PHASE_LANE()                                   \
  int lane = lane0;                                    \
  asm volatile("" : "+v"(lane));                       \
  const int col = lane & 31, h = lane >> 5;            \
  (void)col; (void)h

  f32x16 acc[2][4];                       // this wave's accumulators: [feature block][stream]
  u32x2 st[2][4][2];                      // parked epilogue output of one region: [quad][stream][hi | lo]

#ifdef PINN_STAMP
  // diagnostic build only: s_memtime stamps of workgroup 0, wave 0 of each group, third pair, into the buffer passed as `e`
  long long* const stamp = reinterpret_cast<long long*>(const_cast<float*>(a.e)) + grp * 1024;
  bool stamp_on = false;
  int nstamp = 0;
#define STAMP() do { if (stamp_on && nstamp < 1024) { if (lane0 == 0) stamp[nstamp] = __builtin_amdgcn_s_memtime(); ++nstamp; } } while (0)
#else
#define STAMP() do {} while (0)
#endif
  // parked quad (fb, g0 + k), stream p -> image.  The image writes are SPREAD over the phase that issues them (one
  // stream per MFMA step / one quad per epilogue quad) instead of bursting right behind a barrier: a burst of 64 writes
  // per CU sits in the LDS queue in front of the partner group's first B-fragment reads of the quarter (PINN_DUMP=0:
  // the burst; forward 2.05 -> 1.63 ms without the writes when nothing else limits it)
#ifndef PINN_DUMP
#define PINN_DUMP 1
#endif
  auto dump_kp = [&](int fb, int g0, int k, int p, int col, int h) {
    if (PINN_ABL & 32) { asm volatile("" :: "v"(st[k][p][0]), "v"(st[k][p][1])); return; }      // (timing only: no image writes)
    const int off = XI::chunk_off(col, qbase(fb, g0 + k) >> 3) + 8 * h;
    *reinterpret_cast<u32x2*>(X + p * XI::PLANE * 2 + off) = st[k][p][0];
    if (TERMS == 3) *reinterpret_cast<u32x2*>(X + XI::HALF * 2 + p * XI::PLANE * 2 + off) = st[k][p][1];
  };
  auto dump_k = [&](int fb, int g0, int k, int col, int h) {
#pragma unroll
    for (int p = 0; p < 4; ++p) dump_kp(fb, g0, k, p, col, h);
  };
  auto dump = [&](int fb, int g0, int col, int h) { dump_k(fb, g0, 0, col, h); dump_k(fb, g0, 1, col, h); };

  // weight-fragment ring of the M phases [feature block][k-step % RING].  It lives across phases: the first WPRE k-steps
  // of M_{l+1} are requested during the last quad of E_l (PINN_XPRE & 2), so no M phase opens with an L2 round trip.
#ifndef PINN_XPRE
#define PINN_XPRE 2
#endif
  u32x4 wh[2][RING], wl[2][RING];
  typedef __attribute__((address_space(1))) u32x4 gu32x4;
  // this wave's rows in the prepared weight image (32-row blocks b, lane slot r + 32 h): per-lane offset in u32x4
  // units, plus fb * 4 * KS * 64 + s * 64 (uniform)
  auto w_lane = [&](int col, int h) { return ((2 * (col >> 4) + (w >> 1)) * KS) * 64 + 16 * (w & 1) + (col & 15) + 32 * h; };
  auto wload_l = [&](int l, int s, int wlane) {
    const gu32x4* const wf = reinterpret_cast<const gu32x4*>(pin_base(reinterpret_cast<const u32x4*>(P + prep_wf(HP, l))));
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
      wh[fb][s % RING] = (wf + (size_t)fb * 4 * KS * 64 + s * 64)[wlane];
      if (TERMS == 3 && !(PINN_ABL & 256) && !((PINN_ABL & 128) && fb == 1))      // (timing only: 128 = one lo fragment for both feature blocks, 256 = none)
        wl[fb][s % RING] = (wf + (size_t)(HP * HP / 8) + (size_t)fb * 4 * KS * 64 + s * 64)[wlane];
    }
  };
#define WL_(fb, i) ((PINN_ABL & 256) ? wh[fb][i] : (PINN_ABL & 128) ? wl[0][i] : wl[fb][i])

  // ---------------- M phase: acc <- W_l x image, region q in quarter q ----------------
  auto mphase = [&](int l) {
    PHASE_LANE();
    if (PINN_PRIO) __builtin_amdgcn_s_setprio(PINN_PRIO == 2 ? 2 : 0);
    const int wlane = w_lane(col, h);
    u32x4 bh[2], bo[2];
    auto wload = [&](int s) { wload_l(l, s, wlane); };
    auto bload = [&](int u) {
      const int s = u >> 2, j = u & 3;
      const int off = XI::chunk_off(col, 2 * s + h);
      bh[u & 1] = *reinterpret_cast<const u32x4*>(X + j * XI::PLANE * 2 + off);
      if (TERMS == 3) bo[u & 1] = *reinterpret_cast<const u32x4*>(X + XI::HALF * 2 + j * XI::PLANE * 2 + off);
    };
    if ((PINN_ABL & 2) || !(PINN_XPRE & 2)) {
#pragma unroll
      for (int s = 0; s < ((PINN_ABL & 2) ? RING : WPRE); ++s) wload(s);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      STAMP();
      // region 3 of this tile's previous layer is free since the partner group finished with it a phase ago, and is
      // read here in quarter 3 only: written during quarter 0, one stream of a quad every other step
      if (q == 0 && !PINN_DUMP) dump(1, 2, col, h);      // (every M phase follows an E phase that parked)
      bload(16 * q);
#pragma unroll
      for (int u = 16 * q; u < 16 * q + 16; ++u) {
        const int s = u >> 2, j = u & 3;
        if (j == 0 && s + WPRE < KS && !(PINN_ABL & 2)) wload(s + WPRE);
        if ((u & 15) != 15) bload(u + 1);
        if (PINN_DUMP && q == 0 && (u & 1)) dump_kp(1, 2, u >> 3, (u >> 1) & 3, col, h);
        if (PINN_ABL & 8) {
          asm volatile("" :: "v"(bh[u & 1]), "v"(bo[u & 1]), "v"(wh[0][s % RING]), "v"(wh[1][s % RING]),
                       "v"(WL_(0, s % RING)), "v"(WL_(1, s % RING)));
          continue;
        }
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
          if (s == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[fb][j] = TERMS == 3 ? MFMA_Q(0, wh[fb][0], bo[u & 1], zero) : MFMA_Q(0, wh[fb][0], bh[u & 1], zero);
            if (TERMS == 3) {
              acc[fb][j] = MFMA_Q(1, WL_(fb, 0), bh[u & 1], acc[fb][j]);
              acc[fb][j] = MFMA_Q(0, wh[fb][0], bh[u & 1], acc[fb][j]);
              if (PINN_ABL_SHAPE16) acc[fb][j] = MFMA_Q(1, wh[fb][0], bo[u & 1], acc[fb][j]);      // (timing only: initialise the other half too)
            }
          } else {
            if (TERMS == 3) {
              acc[fb][j] = MFMA_Q(s, wh[fb][s % RING], bo[u & 1], acc[fb][j]);
              acc[fb][j] = MFMA_Q(s + 1, WL_(fb, s % RING), bh[u & 1], acc[fb][j]);
            }
            acc[fb][j] = MFMA_Q(s, wh[fb][s % RING], bh[u & 1], acc[fb][j]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);      // requests stay where they are written (one k-step / one step ahead)
      }
      STAMP();
      __syncthreads();
    }
  };

  // ---------------- E phase: chain rule of layer lE of this group's tile ----------------
  // EK: 0 = layer 0 (pre-activations from (x, y) on the VALU), 1 = hidden layer 1..L-2, 2 = last hidden layer (output
  // layer folded in, nothing parked).  `pstage`: the point stage of the group's PREVIOUS tile rides in quarters 0 / 1.
  auto ephase = [&](auto EKIND, int lE, int tileE, int pstage_tile) {
    constexpr int EK = decltype(EKIND)::value;
    constexpr bool last = EK == 2, first = EK == 0;
    PHASE_LANE();
    if (PINN_PRIO) __builtin_amdgcn_s_setprio(PINN_PRIO == 1 ? 2 : 0);
    if (PINN_ABL & 16) {
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[fb][j]));
#pragma unroll
      for (int q = 0; q < 4; ++q) __syncthreads();
      return;
    }
    float* const Sl = a.S + spill_off(tileE, lE, L, a.sl0, a.sblk, (size_t)HP * COLS);
    const float* const bE = biasL + (size_t)lE * HP;
    float po[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s) po[c][s] = 0.f;
    float px = 0.f, py = 0.f;
    if (first) {
      const int pt = tileE * PPL + col;
      px = pt < a.n ? a.x[pt] : 0.f; py = pt < a.n ? a.y[pt] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      STAMP();
      // ---- the previous tile's point stage (output-layer bias + cross-wave sum, then residuals / loss) ----
      if (pstage_tile >= 0 && q == 0) {
        for (int idx = gtid; idx < 3 * COLS; idx += GT) {
          const int c3 = idx / COLS, cc = idx % COLS;
          float s = cc < PPL ? P[prep_bout(HP, L) + c3] : 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) s += partG[(ww * 12 + c3 * 4 + cc / PPL) * 32 + (cc % PPL)];
          outvG[c3 * COLS + cc] = s;
        }
      }
      if (pstage_tile >= 0 && pstage_tile < a.ntiles && q == 1)
        residual_point_stage<PPL, COLS>(a, outvG, pstage_tile, gtid, npad, lsum);
      // ---- region q - 1, parked in the previous quarter, is free now ----
      if (q > 0 && !last && !PINN_DUMP) dump((q - 1) >> 1, 2 * ((q - 1) & 1), col, h);
      // ---- the two register quads of region q ----
      const int fb = q >> 1;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int g = 2 * (q & 1) + k, o = qbase(fb, g) + 4 * h;
        if ((PINN_XPRE & 2) && !(PINN_ABL & 2) && !last && q == 3 && k == 1) {      // first weight k-steps of M_{lE+1}
#pragma unroll
          for (int s = 0; s < WPRE; ++s) wload_l(lE + 1, s, w_lane(col, h));
        }
        f32x4 av[4], sv[4];
        f32x4 b4, wx4, wy4;
        if (first) {
          wx4 = *reinterpret_cast<const f32x4*>(w0L + o); wy4 = *reinterpret_cast<const f32x4*>(w0L + HP + o);
          b4 = *reinterpret_cast<const f32x4*>(w0L + 2 * HP + o);
        } else {
          b4 = *reinterpret_cast<const f32x4*>(bE + o);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float z, zx, zy, zd;
          if (first) {
            z = fmaf(wx4[e], px, fmaf(wy4[e], py, b4[e])); zx = wx4[e]; zy = wy4[e]; zd = 0.f;
          } else {
            z = acc_read_s(acc[fb][0][r]) + b4[e]; zx = acc_read_s(acc[fb][1][r]); zy = acc_read_s(acc[fb][2][r]);
            zd = acc_read_s(acc[fb][3][r]);
          }
          const float t = (PINN_ABL & 64) ? z : fast_tanh(z);
          const float d1 = 1.f - t * t;
          const float d2 = -2.f * t * d1;
          av[0][e] = t; av[1][e] = d1 * zx; av[2][e] = d1 * zy; av[3][e] = d2 * (zx * zx + zy * zy) + d1 * zd;
          sv[0][e] = t; sv[1][e] = zx; sv[2][e] = zy; sv[3][e] = zd;
          E_SB();
        }
        const unsigned so = (unsigned)(((o - 4 * h) >> 2) + h) * PPL + col;
        u32x4 pk[3];      // the quad's 24-bit spill: hi16 of streams 0-1, hi16 of streams 2-3, lo8 of all four
        STAMP();
        // region q - 1, parked in the previous quarter, is free now: quad k leaves its registers just before they are refilled
        if (PINN_DUMP && q > 0 && !last) dump_k((q - 1) >> 1, 2 * ((q - 1) & 1), k, col, h);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (!last) {
            split4(av[p][0], av[p][1], av[p][2], av[p][3], st[k][p][0], st[k][p][1]);
          } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const f32x4 wo = *reinterpret_cast<const f32x4*>(woutL + c * HP + o);
#pragma unroll
              for (int e = 0; e < 4; ++e) po[c][p] = fmaf(wo[e], av[p][e], po[c][p]);
            }
          }
          // (layer 0 is not spilled: t = tanh(w0x x + w0y y + b0), z_x = w0x, z_y = w0y, z_D = 0 cost the reverse sweep
          // and the dW kernel one FMA pair and one tanh to recompute - a sixth of the spill at 6 layers)
          if (!first && !(PINN_ABL & 1)) {      // 24-bit spill (bf16_util.h pack24): three 16-byte planes instead of four
            u32x2 hi24; unsigned lo24;
            pack24(sv[p], hi24, lo24);
            pk[p >> 1][2 * (p & 1)] = hi24[0]; pk[p >> 1][2 * (p & 1) + 1] = hi24[1]; pk[2][p] = lo24;
            if (p & 1) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, pk[p >> 1]), pin_base(reinterpret_cast<const f32x4*>(Sl) + (p >> 1) * PLQ) + so);
            if (p == 3) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, pk[2]), pin_base(reinterpret_cast<const f32x4*>(Sl) + 2 * PLQ) + so);
          }
          if (last) asm volatile("" : "+v"(po[0][p]), "+v"(po[1][p]), "+v"(po[2][p]));   // (no sinking behind the loop)
          E_SB();
        }
        __builtin_amdgcn_sched_barrier(0);        // 128 arch VGPRs: do not interleave the two quads' live ranges
      }
      if (last && q == 3) {
        // the lane pair (l, l + 32) holds the same column: add the halves (both publish the same value)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            partG[(w * 12 + c * 4 + s) * 32 + col] = po[c][s] + __shfl_xor(po[c][s], 32, 64);
      }
      STAMP();
      __syncthreads();
    }
  };
  auto idle = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) __syncthreads();
  };

  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  // Program of a group, one op per phase: per tile E0 M1 E1 ... M_{L-1} E_{L-1} (2L - 1 phases, four barriers each);
  // group 1 runs it one phase behind group 0 (an idle phase in front, group 0 idles one phase at the end), then one
  // drain phase for the last tile's point stage.  Straight-line per group: no per-phase dispatch (with one, the
  // register allocator spilled the whole accumulator set around the phase loop).  Tile of pair i: 2 i + grp.
  const int npairs = (a.ntiles + 1) / 2;
  if (grp == 1) idle();
  int prev_tile = -1;
  for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
    const int tile = 2 * pair + grp;
#ifdef PINN_STAMP
    stamp_on = blockIdx.x == 0 && w == 0 && pair == (int)blockIdx.x + 2 * (int)gridDim.x;
#endif
    ephase(K0{}, 0, tile, prev_tile);
    for (int l = 1; l < L - 1; ++l) {
      mphase(l);
      ephase(K1{}, l, tile, -1);
    }
    mphase(L - 1);
    ephase(K2{}, L - 1, tile, -1);
    prev_tile = tile;
  }
  for (int q = 0; q < 4; ++q) {                       // drain: point stage of the last tile
    if (q == 0 && prev_tile >= 0)
      for (int i2 = gtid; i2 < 3 * COLS; i2 += GT) {
        const int c3 = i2 / COLS, cc = i2 % COLS;
        float s = cc < PPL ? P[prep_bout(HP, L) + c3] : 0.f;
        for (int ww = 0; ww < 4; ++ww) s += partG[(ww * 12 + c3 * 4 + cc / PPL) * 32 + (cc % PPL)];
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

size_t fwd_split_lds_bytes(int HP, int L) { (void)HP; return SplitLds<256>::bytes(L); }

template <int HP, int TERMS>
static int launch_one(const FwdArgs& a, int grid, hipStream_t s) {
  const size_t lds = SplitLds<HP>::bytes(a.L);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_split_kernel<HP, TERMS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((fwd_split_kernel<HP, TERMS>), dim3(grid), dim3(2 * HP), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// residual mode, saved activations, L >= 2 hidden layers, HP = 256 (the caller checks)
int launch_fwd_split(int HP, int terms, const FwdArgs& a, int grid, hipStream_t s) {
  if (HP != 256) return -1000;
  return terms == 3 ? launch_one<256, 3>(a, grid, s) : launch_one<256, 1>(a, grid, s);
}
