// Software-pipelined bf16x3 forward sweep for residual mode (4 streams), one wave per SIMD.
//
// Same algorithm, same results layout (S, field planes, loss partials) as fwd_bf16.hip - see there and fwd.hip for
// the reference lines replaced (NSFnet/net.py:52-54, NSFnet/pinn_solver.py:132-163,197-226,
// ev-NSFnet/pinn_solver.py:290-342,372-428).  What changes is the schedule.  Measured on gfx950 (DESIGN.md section 4,
// tests/micro/mfma_valu_*.hip, MI355X_MICROARCH.md "Two waves per SIMD"): a SIMD never overlaps one wave's MFMAs with
// its partner wave's VALU, but inside ONE wave ~24 cycles of vector issue per 32-cycle MFMA are free.  fwd_bf16.hip
// (8 waves, phases epilogue -> barrier -> GEMM -> barrier in lockstep) therefore pays MFMA time PLUS chain-rule
// time.  Here a workgroup is HP/64 = 4 waves (one per SIMD, two 32-row MFMA blocks each, up to 512 registers) and
// keeps TWO tiles (A, B: 32 points x 4 streams each) in flight in opposite phases:
//
//     slot:   E0(A) | M1(A)+E0(B) | M1(B)+E1(A) | M2(A)+E1(B) | ... | M_{L-1}(B)+E_{L-1}(A) | E_{L-1}(B) | points
//
// M_l(T) = hidden GEMM l of tile T on v_mfma_f32_32x32x16_bf16 (3 MFMAs per product), E_l(T) = tanh chain rule of
// layer l, hi/lo split, S spill.  Inside a slot both are in ONE basic block of the SAME wave, six MFMAs and one
// epilogue slice per step, so the chain rule issues in the MFMA shadow; every weight fragment streamed from L2 still
// feeds 12 MFMAs (full 32-point tiles).
//
// One tile's bf16 hi/lo image is 128 KB at HP = 256, so two images do not fit the 160 KB of LDS.  The two tiles SHARE
// one image, split along K: R1 = features 0..HP/2-1, R2 = the rest.  Wave w owns the 32-row blocks w (in R1) and
// NW + w (in R2).  A slot is two sub-slots with a barrier between them: in sub-slot 1 the GEMM reads R1 and the
// epilogue computes its R1 block, in sub-slot 2 the GEMM reads R2 and the epilogue computes its R2 block.  What an
// epilogue produces is parked in 64 registers and written into the image half the GEMM has just finished with, half
// a slot later (R1 data during sub-slot 2, R2 data during sub-slot 1 of the next slot), one register quad per step.
// The output layer is folded into the last epilogue (partial dot products straight from the registers), so the last
// layer needs no image.
#include "kernels.h"
#include "point_stage.h"
#include "bf16_util.h"

#include <type_traits>

// timing-only ablation switches (scripts/abl_build.py builds variant libraries; never set in the product build):
// 1 = no S spill, 2 = weight fragments loaded once per slot, 4 = no epilogue arithmetic, 8 = no mid-slot barrier,
// 16 = no parked-quad writes into the image, 32 = no B-fragment LDS reads after the first
#ifndef PINN_ABL
#define PINN_ABL 0
#endif

// Just-in-time AGPR -> VGPR read of one accumulator element.  The two tiles' accumulators fill the 256 AccVGPRs; the
// epilogue's VALU cannot read those directly.  Left to itself the register allocator copies a whole 128-register
// accumulator set into VGPRs at the top of the slot (and spills around it); the asm pins each read to its use.
__device__ __forceinline__ float acc_read(float acc_elem) {
  float v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc_elem));
  return v;
}

template <int HP>
struct PipeLds {
  using XI = XImg<HP, 32>;
  static constexpr int NW = HP / 64;
  static constexpr size_t X_BYTES = XI::BYTES;                         // THE tile image (shared by the two tiles)
  static constexpr size_t PART_F = (size_t)2 * NW * 12 * 32;           // [tile][wave][3 outputs x 4 streams][32 points]
  static constexpr size_t OUTV_F = (size_t)2 * 3 * 128;                // [tile][3][128]
  static size_t bytes(int L) { return X_BYTES + (PART_F + OUTV_F + (size_t)L * HP + 6 * HP) * sizeof(float); }
};

template <int HP, int TERMS>
__global__ __launch_bounds__(HP, 1) void fwd_pipe_kernel(FwdArgs a) {
  using G = PipeLds<HP>;
  using XI = typename G::XI;
  constexpr int NW = HP / 64, NT = HP, KS = HP / 16, PPL = 32, COLS = 128;
#ifndef PINN_PRE
#define PINN_PRE 1
#endif
  constexpr int PRE = PINN_PRE, RING = PRE + 1;      // weight k-steps in flight ahead of their MFMAs (behind the S stores in vmcnt order)
#ifndef PINN_BD
#define PINN_BD 1
#endif
  constexpr int BD = PINN_BD;           // B-fragment (LDS) requests in flight ahead of their MFMAs, in 6-MFMA steps
  constexpr size_t PLQ = (size_t)(HP / 4) * PPL;          // f32x4 per S plane
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* const X = ldsb;
  float* const part = reinterpret_cast<float*>(ldsb + G::X_BYTES);
  float* const outv = part + G::PART_F;
  float* const biasL = outv + G::OUTV_F;                  // [L][HP], row 0 = zeros (layer 0's bias is in its pre-activation)
  float* const woutL = biasL + (size_t)a.L * HP;          // [3][HP]
  float* const w0L = woutL + 3 * HP;                      // [w0x | w0y | b0][HP]: layer 0 (K = 2) runs on the VALU
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * PPL;
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < L * HP; i += NT) biasL[i] = i < HP ? 0.f : P[prep_b(HP, i / HP) + (i % HP)];
  for (int i = tid; i < 3 * HP; i += NT) { woutL[i] = P[prep_wout(HP, L) + i]; w0L[i] = P[prep_w0x(HP) + i]; }
  __syncthreads();

  using T_ = std::true_type;
  using F_ = std::false_type;
#ifdef PINN_STAMP
  // diagnostic build only: s_memtime stamps of workgroup 0 (first pair of tiles) into the buffer passed as `e`
  long long* const stamp = reinterpret_cast<long long*>(const_cast<float*>(a.e)) + w * 64;
  int nstamp = 0;
#define STAMP() do { if (blockIdx.x == 0 && nstamp < 64) { stamp[nstamp] = __builtin_amdgcn_s_memtime(); } ++nstamp; } while (0)
#else
#define STAMP() do {} while (0)
#endif

  // parked epilogue output: [register quad g of the block][stream p][hi | lo], 4 bf16 each.  Lives across slots.
  u32x2 st[4][4][2];
  // weight-fragment ring [feature block][k-step % RING]: lives across slots, because the first PRE k-steps of the
  // NEXT slot's GEMM are requested during the last k-steps of this one (a load issued right after the barrier would
  // expose the L2 latency and, vmcnt being in order, the drain of every S store issued before it)
  u32x4 wh[2][RING], wl[2][RING];

  // ---- one slot: GEMM `lM` of tile M (accM <- W_lM x image) and the epilogue of layer `lE` of tile E ----
  // EK: 0 = layer 0 (pre-activations from (x, y) on the VALU, nothing read from accE), 1 = hidden layer 1..L-2,
  //     2 = last hidden layer (output layer folded in, nothing parked).  DUMP1: the parked registers hold the R2 block
  //     of the previous slot's epilogue (to be written during sub-slot 1).
  // lNext: layer of the NEXT slot's GEMM (0: the next slot has none)
  auto slot = [&](auto DO_M, auto EKIND, auto DUMP1_, f32x16 (&accM)[2][4], int lM, int lNext, f32x16 (&accE)[2][4],
                  int lE, int tileE, float* partE) {
    constexpr bool doM = decltype(DO_M)::value, dump1 = decltype(DUMP1_)::value;
    constexpr int EK = decltype(EKIND)::value;
    constexpr bool last = EK == 2, first = EK == 0;
    // lane geometry re-derived per slot from an opaque copy: address arithmetic then lives inside the slot that uses
    // it instead of being hoisted over all slot bodies (it was: ~80 long-lived VGPRs, spilled around the layer loop)
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    const int col = lane_ & 31, h = lane_ >> 5;
    // ------------- GEMM state -------------
    u32x4 bh[BD + 1], bo[BD + 1];      // B fragments: one column block in use, BD in flight
    // the wave's fragment slices are UNIFORM bases (scalar registers) + lane * 16 bytes: every load is the
    // saddr + voffset form, no 64-bit vector address per fragment.  Row block of (wave, fb) = fb * NW + w.
    typedef __attribute__((address_space(1))) u32x4 gu32x4;
    const gu32x4* const wf = reinterpret_cast<const gu32x4*>(
        pin_base(reinterpret_cast<const u32x4*>(P + prep_wf(HP, doM ? lM : 1)) + (size_t)w * KS * 64));
    const gu32x4* const wfn = reinterpret_cast<const gu32x4*>(
        pin_base(reinterpret_cast<const u32x4*>(P + prep_wf(HP, lNext > 0 ? lNext : 1)) + (size_t)w * KS * 64));
    static_assert(KS % RING == 0, "the ring index of k-step s of the next slot must be s % RING");
    auto wload = [&](const gu32x4* base, int s) {
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        wh[fb][s % RING] = (base + (size_t)fb * NW * KS * 64 + s * 64)[lane_];
        if (TERMS == 3) wl[fb][s % RING] = (base + (size_t)(HP * HP / 8) + (size_t)fb * NW * KS * 64 + s * 64)[lane_];
      }
    };
    // B fragments of column block j (= stream j) at k-step s: one conflict-free ds_read_b128 per hi / lo image
    auto bload = [&](int u) {
      const int s = u >> 2, j = u & 3;
      const int off = XI::chunk_off(col, 2 * s + h);
      bh[u % (BD + 1)] = *reinterpret_cast<const u32x4*>(X + j * XI::PLANE * 2 + off);
      if (TERMS == 3) bo[u % (BD + 1)] = *reinterpret_cast<const u32x4*>(X + XI::HALF * 2 + j * XI::PLANE * 2 + off);
    };
    // step u = (k-step s, column block j): 2 feature blocks x 3 MFMAs on the fragments requested BD steps earlier
    // (the requests for the first BD steps of sub-slot 2 are made after the mid-slot barrier, not here)
    auto jstep = [&](int u) {
      const int s = u >> 2, j = u & 3;
      if (j == 0 && !(PINN_ABL & 2)) {
        if (s + PRE < KS) wload(wf, s + PRE);
        else if (lNext > 0) wload(wfn, s + PRE - KS);      // (uniform branch, once per k-step of the slot's tail)
      }
      if (u + BD < 4 * KS && !(u < 2 * KS && u + BD >= 2 * KS) && !(PINN_ABL & 32)) bload(u + BD);
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        if (s == 0) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          accM[fb][j] = TERMS == 3 ? mfma_bf16(wh[fb][0], bo[u % (BD + 1)], zero) : mfma_bf16(wh[fb][0], bh[u % (BD + 1)], zero);
          if (TERMS == 3) {
            accM[fb][j] = mfma_bf16(wl[fb][0], bh[u % (BD + 1)], accM[fb][j]);
            accM[fb][j] = mfma_bf16(wh[fb][0], bh[u % (BD + 1)], accM[fb][j]);
          }
        } else {
          if (TERMS == 3) {
            accM[fb][j] = mfma_bf16(wh[fb][s % RING], bo[u % (BD + 1)], accM[fb][j]);
            accM[fb][j] = mfma_bf16(wl[fb][s % RING], bh[u % (BD + 1)], accM[fb][j]);
          }
          accM[fb][j] = mfma_bf16(wh[fb][s % RING], bh[u % (BD + 1)], accM[fb][j]);
        }
      }
    };
    // ------------- epilogue state -------------
    float* const Sl = a.S + ((size_t)tileE * L + lE) * ((size_t)HP * COLS);
    const float* const bE = biasL + (size_t)lE * HP;
    float po[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s) po[c][s] = 0.f;
    f32x4 av[4], sv[4];                   // a-streams / saved streams (t, z_x, z_y, z_D) of the register quad in flight
    f32x4 b4, wx4, wy4, b4n, wx4n, wy4n;      // (n: requested one quad ahead, so no step waits on its own LDS read)
    float px = 0.f, py = 0.f;
    if (first) {
      const int pt = tileE * PPL + col;
      px = pt < a.n ? a.x[pt] : 0.f; py = pt < a.n ? a.y[pt] : 0.f;
    }
    // The epilogue of register quad q = (fb, g) (features ob + 8g + 4h + e, this lane's column) in EIGHT slices, one
    // per 6-MFMA step of the GEMM: slices 0-3 = tanh chain rule of element e, slices 4-7 = plane (stream) p: write
    // the parked quad (g, p) of the OTHER block into the image half the GEMM is not reading, split the new values
    // into bf16 hi/lo and park them (or fold them into the output layer), spill the saved plane.
    // per-quad parameters from LDS: the bias (layer 0: w0x, w0y, b0) of features ob + 8g + 4h + 0..3
    auto qparams = [&](int q) {
      const int o = 32 * ((q >> 2) * NW + w) + 8 * (q & 3) + 4 * h;
      if (first) {
        wx4n = *reinterpret_cast<const f32x4*>(w0L + o); wy4n = *reinterpret_cast<const f32x4*>(w0L + HP + o);
        b4n = *reinterpret_cast<const f32x4*>(w0L + 2 * HP + o);
      } else {
        b4n = *reinterpret_cast<const f32x4*>(bE + o);
      }
    };
    auto eslice = [&](int q, int i) {
      const int fb = q >> 2, g = q & 3, ob = 32 * (fb * NW + w);
      if (PINN_ABL & 4) return;
      if (i < 4) {
        const int e = i, r = 4 * g + e;
        float z, zx, zy, zd;
        if (e == 0) { b4 = b4n; if (first) { wx4 = wx4n; wy4 = wy4n; } }
        if (first) {
          z = fmaf(wx4[e], px, fmaf(wy4[e], py, b4[e])); zx = wx4[e]; zy = wy4[e]; zd = 0.f;
        } else {
          z = acc_read(accE[fb][0][r]) + b4[e]; zx = acc_read(accE[fb][1][r]); zy = acc_read(accE[fb][2][r]);
          zd = acc_read(accE[fb][3][r]);
        }
        const float t = fast_tanh(z);
        const float d1 = 1.f - t * t;
        const float d2 = -2.f * t * d1;
        av[0][e] = t; av[1][e] = d1 * zx; av[2][e] = d1 * zy; av[3][e] = d2 * (zx * zx + zy * zy) + d1 * zd;
        sv[0][e] = t; sv[1][e] = zx; sv[2][e] = zy; sv[3][e] = zd;
      } else {
        const int p = i - 4;
        if (p == 0 && q + 1 < 8) qparams(q + 1);
        if (fb == 0 ? dump1 : !last) {
          // parked quad (g, p) of the other block: R2 data of the previous slot while R1 is read, R1 data of this
          // slot while R2 is read
          const int obo = 32 * ((1 - fb) * NW + w);
          const int off = XI::chunk_off(col, (obo >> 3) + g) + 8 * h;
          if (!(PINN_ABL & 16)) {
            *reinterpret_cast<u32x2*>(X + p * XI::PLANE * 2 + off) = st[g][p][0];
            if (TERMS == 3) *reinterpret_cast<u32x2*>(X + XI::HALF * 2 + p * XI::PLANE * 2 + off) = st[g][p][1];
          }
        }
        if (!last) {
          split4(av[p][0], av[p][1], av[p][2], av[p][3], st[g][p][0], st[g][p][1]);
        } else {
          // output layer (3 x HP, VALU): stream p of this lane's column, this quad's four features
          const int o = ob + 8 * g + 4 * h;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const f32x4 wo = *reinterpret_cast<const f32x4*>(woutL + c * HP + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) po[c][p] = fmaf(wo[e], av[p][e], po[c][p]);
          }
          // keep the partial dot products HERE: left alone the optimiser sinks all 384 of them behind the step loop
          // and spills the a-streams they need (46 serialised scratch reloads per tile)
          asm volatile("" : "+v"(po[0][p]), "+v"(po[1][p]), "+v"(po[2][p]));
        }
        const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * PPL + col);
        if (!(PINN_ABL & 1)) __builtin_nontemporal_store(sv[p], pin_base(reinterpret_cast<const f32x4*>(Sl) + p * PLQ) + so);
      }
    };

    qparams(0);
    if (doM) {
      if (PINN_ABL & 2) {
#pragma unroll
        for (int s = 0; s < RING; ++s) wload(wf, s);
      }
#pragma unroll
      for (int u = 0; u < BD; ++u) bload(u);
      if (PINN_ABL & 32) bload(BD);
    }
    // 4 KS steps of 6 MFMAs, each with one epilogue slice in its shadow; nothing crosses a step boundary, so the
    // requests (weights PRE k-steps ahead, B fragments one step ahead) stay where they are written
    constexpr int NSTEP = 4 * KS;
    static_assert(NSTEP == 64, "HP must be 256 (8 steps per register quad, 4 quads per sub-slot)");
#pragma unroll
    for (int u = 0; u < NSTEP; ++u) {
      if (u == 0) STAMP();
      if (u == NSTEP / 2) {
        STAMP();
        if (!(PINN_ABL & 8)) __syncthreads();                  // R1 has been read by every wave, R2 is complete
        STAMP();
        if (doM) {
#pragma unroll
          for (int k = 0; k < BD; ++k) bload(u + k);
        }
      }
      if (doM) jstep(u);
      eslice(u / 8, u % 8);
      if (doM) {
#pragma unroll
        for (int i = 0; i < (TERMS == 3 ? 6 : 2); ++i) {
#ifndef PINN_VPM
#define PINN_VPM 3
#endif
          if (PINN_VPM > 0) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, TERMS == 3 ? PINN_VPM : 3 * PINN_VPM, 0);   // epilogue VALU in its shadow
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP();
    if (!doM && lNext > 0) {      // an epilogue-only slot in front of a GEMM slot: its first weight fragments
#pragma unroll
      for (int s = 0; s < PRE; ++s) wload(wfn, s);
    }
    if (last) {
      // the lane pair (l, l + 32) holds the same column: add the halves (both publish the same value)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          partE[(w * 12 + c * 4 + s) * 32 + col] = po[c][s] + __shfl_xor(po[c][s], 32, 64);
    }
  };

  // ---- output-layer bias + cross-wave sum, then the per-point residual / loss stage (point_stage.h) ----
  auto points = [&](int tile, const float* partT, float* outvT) {
    for (int idx = tid; idx < 3 * COLS; idx += NT) {
      const int c3 = idx / COLS, cc = idx % COLS;
      float s = cc < PPL ? P[prep_bout(HP, L) + c3] : 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) s += partT[(ww * 12 + c3 * 4 + cc / PPL) * 32 + (cc % PPL)];
      outvT[c3 * COLS + cc] = s;
    }
  };

  const int npairs = (a.ntiles + 1) / 2;
  float* const partA = part, *const partB = part + (size_t)NW * 12 * 32;
  float* const outvA = outv, *const outvB = outv + 3 * 128;
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
    const int tA = 2 * pair, tB = 2 * pair + 1;      // tB == ntiles: a dummy tile (masked points, scratch S block)
    f32x16 accA[2][4], accB[2][4];
    slot(F_{}, K0{}, F_{}, accB, 1, 1, accA, 0, tA, partA);                     //            E_0(A)
    __syncthreads();
    slot(T_{}, K0{}, T_{}, accA, 1, 1, accB, 0, tB, partB);                     // M_1(A)   + E_0(B)
    __syncthreads();
    for (int l = 1; l < L - 1; ++l) {
      slot(T_{}, K1{}, T_{}, accB, l, l + 1, accA, l, tA, partA);               // M_l(B)   + E_l(A)
      __syncthreads();
      slot(T_{}, K1{}, T_{}, accA, l + 1, l + 1, accB, l, tB, partB);           // M_l+1(A) + E_l(B)
      __syncthreads();
    }
    slot(T_{}, K2{}, T_{}, accB, L - 1, 0, accA, L - 1, tA, partA);             // M_L-1(B) + E_L-1(A), output layer
    __syncthreads();
    slot(F_{}, K2{}, F_{}, accA, 1, 0, accB, L - 1, tB, partB);                 //            E_L-1(B)
    __syncthreads();
    points(tA, partA, outvA);
    points(tB, partB, outvB);
    __syncthreads();
    residual_point_stage<PPL, COLS>(a, outvA, tA, tid, npad, lsum);
    if (tB < a.ntiles) residual_point_stage<PPL, COLS>(a, outvB, tB, tid, npad, lsum);
    __syncthreads();
  }
  float* red = reinterpret_cast<float*>(ldsb);
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

size_t fwd_pipe_lds_bytes(int HP, int L) { (void)HP; return PipeLds<256>::bytes(L); }

template <int HP, int TERMS>
static int launch_one(const FwdArgs& a, int grid, hipStream_t s) {
  const size_t lds = PipeLds<HP>::bytes(a.L);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_pipe_kernel<HP, TERMS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((fwd_pipe_kernel<HP, TERMS>), dim3(grid), dim3(HP), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// residual mode, saved activations, L >= 2 hidden layers, HP = 256 (the caller checks)
int launch_fwd_pipe(int HP, int terms, const FwdArgs& a, int grid, hipStream_t s) {
  if (HP != 256) return -1000;
  return terms == 3 ? launch_one<256, 3>(a, grid, s) : launch_one<256, 1>(a, grid, s);
}
