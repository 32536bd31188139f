// Software-pipelined bf16x3 reverse sweep for residual mode (4 streams), one wave per SIMD: the schedule of
// fwd_bf16_pipe.hip applied to bwd_bf16.hip (see there and bwd.hip for the algorithm and the reference lines it
// replaces: loss.backward(), NSFnet/pinn_solver.py:252, ev-NSFnet/pinn_solver.py:469).  Results layout (Z-bar spill,
// per-workgroup skinny-gradient accumulators, ebar) is unchanged, so the dW / reduce kernels do not care which reverse
// sweep ran.
//
//     slot:  E_{L-1}(A) | G_{L-1}(A)+E_{L-1}(B) | G_{L-1}(B)+E_{L-2}(A) | G_{L-2}(A)+E_{L-2}(B) | ... | G_1(B)+E_0(A) | E_0(B)
//
// E_l(T) = tanh adjoint of layer l of tile T: reads the saved (t, z_x, z_y, z_D) quad by quad, turns the a-stream
// adjoints g (accumulators of G_{l+1}(T); for l = L-1 the rank-3 update W_out^T o-bar, on the VALU) into the z-stream
// adjoints z-bar, column-sums the skinny gradients (biases, layer 0, output layer) into the LDS accumulator, restages
// z-bar as bf16 hi/lo into T's LDS image and spills it.  G_l(T) = W_l^T z-bar_l on v_mfma_f32_32x32x16_bf16.
// Inside a slot both run in ONE wave, six MFMAs and one epilogue slice per step, the slice in the MFMA shadow.  The two
// tiles share ONE z-bar image split along K, with the freshly computed blocks parked in registers for half a slot
// (fwd_bf16_pipe.hip explains the rotation); W^T fragments for the next slot are requested during this slot's tail.
#include "kernels.h"
#include "point_stage.h"
#include "bf16_util.h"
#include "reduce_util.h"

#include <type_traits>

// Just-in-time AGPR -> VGPR read of one accumulator element (see fwd_bf16_pipe.hip).
__device__ __forceinline__ float acc_read_b(float acc_elem) {
  float v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc_elem));
  return v;
}

template <int HP>
struct PipeBwdLds {
  using XI = XImg<HP, 32>;
  static constexpr int NW = HP / 64;
  static constexpr size_t X_BYTES = XI::BYTES;                          // THE z-bar image (shared by the two tiles)
  static constexpr size_t OADJ_F = (size_t)2 * 4 * 128;                 // [tile][4][128] (3 outputs used)
  static constexpr size_t DUMMY_F = 64 * NW;                            // sink of the lanes that own no accumulator slot
  static size_t bytes(int L) { return X_BYTES + (OADJ_F + DUMMY_F + 3 * HP + (size_t)sg_total(HP, L)) * sizeof(float); }
};

template <int HP, int TERMS>
__global__ __launch_bounds__(HP, 1) void bwd_pipe_kernel(BwdArgs a) {
  using G = PipeBwdLds<HP>;
  using XI = typename G::XI;
  constexpr int NW = HP / 64, NT = HP, KS = HP / 16, PPL = 32, COLS = 128;
#ifndef PINN_PRE
#define PINN_PRE 1
#endif
  constexpr int PRE = PINN_PRE, RING = PRE + 1, BD = 1;
  constexpr size_t PLQ = (size_t)(HP / 4) * PPL;          // f32x4 per plane of S / Z-bar
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* const X = ldsb;
  float* const oadjL = reinterpret_cast<float*>(ldsb + G::X_BYTES);         // [2][4][128]
  float* const dummy = oadjL + G::OADJ_F;
  float* const woutL = dummy + G::DUMMY_F;                                   // [3][HP]
  float* const sgacc = woutL + 3 * HP;                                       // [sg_total]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * PPL;
  const int SG = sg_total(HP, L);
  for (int i = tid; i < SG; i += NT) sgacc[i] = 0.f;
  for (int i = tid; i < 3 * HP; i += NT) woutL[i] = P[prep_wout(HP, L) + i];
  for (int i = tid; i < (int)G::DUMMY_F; i += NT) dummy[i] = 0.f;
  float dbo[3] = {0.f, 0.f, 0.f};
  __syncthreads();

  using T_ = std::true_type;
  using F_ = std::false_type;

  u32x2 st[4][4][2];                      // parked z-bar hi/lo of one 32-row block (see fwd_bf16_pipe.hip)
  u32x4 wh[2][RING], wl[2][RING];         // W^T fragment ring, lives across slots (cross-slot prefetch)

  // EK: 0 = last hidden layer L-1 (a-stream adjoints from the output adjoints on the VALU, dW_out), 1 = layer L-2..1,
  //     2 = layer 0 (dW_0; no image, nothing parked, no spill).  DUMP1 / lNext as in fwd_bf16_pipe.hip.
  auto slot = [&](auto DO_M, auto EKIND, auto DUMP1_, f32x16 (&accM)[2][4], int lM, int lNext, f32x16 (&accE)[2][4],
                  int lE, int tileE, const float* oadjE, float pxE, float pyE) {
    constexpr bool doM = decltype(DO_M)::value, dump1 = decltype(DUMP1_)::value;
    constexpr int EK = decltype(EKIND)::value;
    constexpr bool first = EK == 0, last = EK == 2;
    int lane_ = lane;                         // (opaque copy: keeps the address arithmetic inside the slot)
    asm volatile("" : "+v"(lane_));
    const int col = lane_ & 31, h = lane_ >> 5;
    // ------------- GEMM state (W_l^T fragments as the A operand) -------------
    u32x4 bh[BD + 1], bo[BD + 1];
    typedef __attribute__((address_space(1))) u32x4 gu32x4;
    const gu32x4* const wf = reinterpret_cast<const gu32x4*>(
        pin_base(reinterpret_cast<const u32x4*>(P + prep_wtf(HP, doM ? lM : 1)) + (size_t)w * KS * 64));
    const gu32x4* const wfn = reinterpret_cast<const gu32x4*>(
        pin_base(reinterpret_cast<const u32x4*>(P + prep_wtf(HP, lNext > 0 ? lNext : 1)) + (size_t)w * KS * 64));
    static_assert(KS % RING == 0, "the ring index of k-step s of the next slot must be s % RING");
    auto wload = [&](const gu32x4* base, int s) {
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        wh[fb][s % RING] = (base + (size_t)fb * NW * KS * 64 + s * 64)[lane_];
        if (TERMS == 3) wl[fb][s % RING] = (base + (size_t)(HP * HP / 8) + (size_t)fb * NW * KS * 64 + s * 64)[lane_];
      }
    };
    auto bload = [&](int u) {
      const int s = u >> 2, j = u & 3;
      const int off = XI::chunk_off(col, 2 * s + h);
      bh[u % (BD + 1)] = *reinterpret_cast<const u32x4*>(X + j * XI::PLANE * 2 + off);
      if (TERMS == 3) bo[u % (BD + 1)] = *reinterpret_cast<const u32x4*>(X + XI::HALF * 2 + j * XI::PLANE * 2 + off);
    };
    auto jstep = [&](int u) {
      const int s = u >> 2, j = u & 3;
      if (j == 0) {
        if (s + PRE < KS) wload(wf, s + PRE);
        else if (lNext > 0) wload(wfn, s + PRE - KS);
      }
      if (u + BD < 4 * KS && !(u < 2 * KS && u + BD >= 2 * KS)) bload(u + BD);
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
    const int tileS = tileE < a.ntiles ? tileE : 0;      // the dummy partner of an odd tile count reads tile 0's (finite) S
    const float* const Sl = a.S + ((size_t)tileS * L + lE) * ((size_t)HP * COLS);
    float* const Zl = a.Zb + ((size_t)tileE * L + lE) * ((size_t)HP * COLS);
    float oc[3][4];                                       // output adjoints of this lane's point: [output][stream]
    if (first) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s) oc[c][s] = oadjE[c * COLS + s * PPL + col];
    }
#ifndef PINN_SD
#define PINN_SD 1
#endif
    constexpr int SD = PINN_SD;    // S quads requested ahead of their chain rule (HBM latency: several quads' worth)
    f32x4 sq[SD + 1][4];           // saved (t, z_x, z_y, z_D): ring over quads
    f32x4 zq[4];                   // z-bar of the quad: [stream][element]
    f32x4 wov[3], dwv[2];          // per-element column terms of dW_out (first) / dW_0 (last)
    f32x4 wo4[3];
    auto sload = [&](int q, f32x4 (&dst)[4]) {
      const int fb = q >> 2, g = q & 3, ob = 32 * (fb * NW + w);
      const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * PPL + col);
#pragma unroll
      for (int p = 0; p < 4; ++p)
        dst[p] = __builtin_nontemporal_load(pin_base(reinterpret_cast<const f32x4*>(Sl) + p * PLQ) + so);
    };
    // slot of the skinny-gradient accumulator owned by this lane after a transposing column sum (lanes col < 4 of
    // each half own feature ob + 8g + 4h + col); every other lane adds its (discarded) value to a private sink, so
    // the update is unconditional (a plain read-modify-write, reduce_util.h): no branch splits the MFMA block
    auto commit = [&](int base, int q, float v) {
      const int fb = q >> 2, g = q & 3, o = 32 * (fb * NW + w) + 8 * g + 4 * h + (col & 3);
      float* p = col < 4 ? &sgacc[base + o] : &dummy[w * 64 + lane_];
      lds_rmw_add(p, v);
    };
    // The adjoint of register quad q = (fb, g) in EIGHT slices (one per 6-MFMA step): 0-3 = chain rule of element e,
    // 4-7 = stream p: write the parked quad of the other block into the free image half, split the new z-bar into
    // bf16 hi/lo and park it, spill it; the column sums of the skinny gradients are spread over slices 4-7.
    auto eslice = [&](int q, int i) {
      const int fb = q >> 2, g = q & 3, ob = 32 * (fb * NW + w);
      if (i < 4) {
        const int e = i, r = 4 * g + e;
        if (e == 0) {
          if (q + SD < 8) sload(q + SD, sq[(q + SD) % (SD + 1)]);
          if (first) {
#pragma unroll
            for (int c = 0; c < 3; ++c) wo4[c] = *reinterpret_cast<const f32x4*>(woutL + c * HP + ob + 8 * g + 4 * h);
          }
        }
        float ga, gx, gy, gd;
        if (first) {      // adjoint of the last hidden layer's a-streams: rank-3 update from the output adjoints
          ga = wo4[0][e] * oc[0][0] + wo4[1][e] * oc[1][0] + wo4[2][e] * oc[2][0];
          gx = wo4[0][e] * oc[0][1] + wo4[1][e] * oc[1][1] + wo4[2][e] * oc[2][1];
          gy = wo4[0][e] * oc[0][2] + wo4[1][e] * oc[1][2] + wo4[2][e] * oc[2][2];
          gd = wo4[0][e] * oc[0][3] + wo4[1][e] * oc[1][3] + wo4[2][e] * oc[2][3];
        } else {
          ga = acc_read_b(accE[fb][0][r]); gx = acc_read_b(accE[fb][1][r]); gy = acc_read_b(accE[fb][2][r]);
          gd = acc_read_b(accE[fb][3][r]);
        }
        const f32x4 (&sc)[4] = sq[q % (SD + 1)];
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
      } else {
        const int p = i - 4;
        if (fb == 0 ? dump1 : !last) {
          const int obo = 32 * ((1 - fb) * NW + w);
          const int off = XI::chunk_off(col, (obo >> 3) + g) + 8 * h;
          *reinterpret_cast<u32x2*>(X + p * XI::PLANE * 2 + off) = st[g][p][0];
          if (TERMS == 3) *reinterpret_cast<u32x2*>(X + XI::HALF * 2 + p * XI::PLANE * 2 + off) = st[g][p][1];
        }
        if (!last) {
          split4(zq[p][0], zq[p][1], zq[p][2], zq[p][3], st[g][p][0], st[g][p][1]);
          const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * PPL + col);
          __builtin_nontemporal_store(zq[p], pin_base(reinterpret_cast<const f32x4*>(Zl) + p * PLQ) + so);
        }
        // column sums of the four features at once (reduce_util.h); lane col == e of each half commits feature e.
        // One sum per slice: db in slice 4; dW_out rows 0..2 (first kind) / dW_0 x, y (last kind) in slices 5-7.
        if (p == 0) commit(sg_db(HP, lE), q, sum_cols4<32>(zq[0][0], zq[0][1], zq[0][2], zq[0][3], lane_));
        if (first && p >= 1) {
          const int c = p - 1;
          commit(sg_wout(HP, L) + c * HP, q, sum_cols4<32>(wov[c][0], wov[c][1], wov[c][2], wov[c][3], lane_));
        }
        if (last && (p == 1 || p == 2)) {
          const int c = p - 1;
          commit(c == 0 ? sg_w0x(HP, L) : sg_w0y(HP, L), q, sum_cols4<32>(dwv[c][0], dwv[c][1], dwv[c][2], dwv[c][3], lane_));
        }
      }
    };

    if (doM) {
#pragma unroll
      for (int u = 0; u < BD; ++u) bload(u);
    }
#pragma unroll
    for (int q = 0; q < SD; ++q) sload(q, sq[q]);
    constexpr int NSTEP = 4 * KS;
    static_assert(NSTEP == 64, "HP must be 256");
#pragma unroll
    for (int u = 0; u < NSTEP; ++u) {
      if (u == NSTEP / 2) {
        __syncthreads();                  // R1 has been read by every wave, R2 is complete
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
#define PINN_VPM 4
#endif
          if (PINN_VPM > 0) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, TERMS == 3 ? PINN_VPM : 3 * PINN_VPM, 0);   // adjoint VALU in its shadow
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!doM && lNext > 0) {      // an epilogue-only slot in front of a GEMM slot: its first weight fragments
#pragma unroll
      for (int s = 0; s < PRE; ++s) wload(wfn, s);
    }
  };

  // ---- output adjoints of a tile (point_stage.h) into its LDS block; zero for the dummy partner tile ----
  auto seeds = [&](int tile, float* oadjT, float& px, float& py) {
    const int col = lane & 31;
    if (tile < a.ntiles) {
      float pxa[1], pya[1];
      output_adjoint_stage<PPL, COLS, 4, NT, 1>(a, tile, tid, col, col, npad, oadjT, dbo, pxa, pya);
      px = pxa[0]; py = pya[0];
    } else {
      for (int i = tid; i < 3 * COLS; i += NT) oadjT[i] = 0.f;
      px = py = 0.f;
    }
  };

  const int npairs = (a.ntiles + 1) / 2;
  float* const oadjA = oadjL, *const oadjB = oadjL + 4 * 128;
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
    const int tA = 2 * pair, tB = 2 * pair + 1;      // tB == ntiles: dummy tile (zero adjoints, scratch Z-bar block)
    f32x16 accA[2][4], accB[2][4];
    float pxA, pyA, pxB, pyB;
    seeds(tA, oadjA, pxA, pyA);
    seeds(tB, oadjB, pxB, pyB);
    __syncthreads();
    slot(F_{}, K0{}, F_{}, accB, 1, L - 1, accA, L - 1, tA, oadjA, pxA, pyA);             //              E_L-1(A)
    __syncthreads();
    slot(T_{}, K0{}, T_{}, accA, L - 1, L - 1, accB, L - 1, tB, oadjB, pxB, pyB);         // G_L-1(A)   + E_L-1(B)
    __syncthreads();
    for (int l = L - 1; l >= 2; --l) {
      slot(T_{}, K1{}, T_{}, accB, l, l - 1, accA, l - 1, tA, oadjA, pxA, pyA);           // G_l(B)     + E_l-1(A)
      __syncthreads();
      slot(T_{}, K1{}, T_{}, accA, l - 1, l - 1, accB, l - 1, tB, oadjB, pxB, pyB);       // G_l-1(A)   + E_l-1(B)
      __syncthreads();
    }
    slot(T_{}, K2{}, T_{}, accB, 1, 0, accA, 0, tA, oadjA, pxA, pyA);                     // G_1(B)     + E_0(A)
    __syncthreads();
    slot(F_{}, K2{}, F_{}, accA, 1, 0, accB, 0, tB, oadjB, pxB, pyB);                     //              E_0(B)
    __syncthreads();
  }
  // ---------------- flush ----------------
  float* red = reinterpret_cast<float*>(ldsb);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 3; ++c) red[c * NT + tid] = dbo[c];
  __syncthreads();
  if (tid < 3) {
    float s = 0.f;
    for (int t = 0; t < NT; ++t) s += red[tid * NT + t];
    sgacc[sg_bout(HP, L) + tid] = s;
  }
  __syncthreads();
  float* out = a.sg + (size_t)blockIdx.x * SG;
  for (int i = tid; i < SG; i += NT) out[i] = sgacc[i];
}

size_t bwd_pipe_lds_bytes(int HP, int L) { return PipeBwdLds<256>::bytes(L); }

template <int HP, int TERMS>
static int launch_one(const BwdArgs& a, int grid, hipStream_t s) {
  const size_t lds = PipeBwdLds<HP>::bytes(a.L);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_pipe_kernel<HP, TERMS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((bwd_pipe_kernel<HP, TERMS>), dim3(grid), dim3(HP), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// residual mode, L >= 2 hidden layers, HP = 256 (the caller checks)
int launch_bwd_pipe(int HP, int terms, const BwdArgs& a, int grid, hipStream_t s) {
  if (HP != 256) return -1000;
  return terms == 3 ? launch_one<256, 3>(a, grid, s) : launch_one<256, 1>(a, grid, s);
}
