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
// Inside a slot both run in ONE wave, six MFMAs and one epilogue slice per step, the slice in the MFMA shadow.
#include "kernels.h"
#include "point_stage.h"
#include "bf16_util.h"
#include "reduce_util.h"

#include <type_traits>

template <int HP>
struct PipeBwdLds {
  using XI = XImg<HP, 32>;
  static constexpr int NW = HP / 64;
  static constexpr size_t X_BYTES = XI::BYTES;
  static constexpr size_t OADJ_F = (size_t)2 * 4 * 128;                 // [tile][4][128] (3 outputs used)
  static constexpr size_t DUMMY_F = 64 * NW;                            // sink of the lanes that own no accumulator slot
  static size_t bytes(int L) { return 2 * X_BYTES + (OADJ_F + DUMMY_F + 3 * HP + (size_t)sg_total(HP, L)) * sizeof(float); }
};

template <int HP, int TERMS>
__global__ __launch_bounds__(HP, 1) void bwd_pipe_kernel(BwdArgs a) {
  using G = PipeBwdLds<HP>;
  using XI = typename G::XI;
  constexpr int NT = HP, KS = HP / 16, PPL = 32, COLS = 128;
  constexpr int PRE = 3, RING = 4;
  constexpr size_t PLQ = (size_t)(HP / 4) * PPL;          // f32x4 per plane of S / Z-bar
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* const XA = ldsb;
  unsigned char* const XB = ldsb + G::X_BYTES;
  float* const oadjL = reinterpret_cast<float*>(ldsb + 2 * G::X_BYTES);     // [2][4][128]
  float* const dummy = oadjL + G::OADJ_F;
  float* const woutL = dummy + G::DUMMY_F;                                   // [3][HP]
  float* const sgacc = woutL + 3 * HP;                                       // [sg_total]
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ob0 = w * 64;
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

  // EK: 0 = last hidden layer L-1 (a-stream adjoints from the output adjoints, dW_out), 1 = layer L-2..1,
  //     2 = layer 0 (dW_0, no image, no spill)
  auto slot = [&](auto DO_M, auto EKIND, f32x16 (&accM)[2][4], const unsigned char* __restrict__ XM, int lM,
                  f32x16 (&accE)[2][4], unsigned char* __restrict__ XE, int lE, int tileE, const float* oadjE,
                  float pxE, float pyE) {
    constexpr bool doM = decltype(DO_M)::value;
    constexpr int EK = decltype(EKIND)::value;
    constexpr bool first = EK == 0, last = EK == 2;
    int lane_ = lane;                         // (opaque copy: keeps the address arithmetic inside the slot, see fwd_bf16_pipe.hip)
    asm volatile("" : "+v"(lane_));
    const int col = lane_ & 31, h = lane_ >> 5, lane = lane_;
    // ------------- GEMM state (W_l^T fragments as the A operand) -------------
    u32x4 wh[2][RING], wl[2][RING], bh[2], bo[2];
    typedef __attribute__((address_space(1))) u32x4 gu32x4;
    const gu32x4* const wf = reinterpret_cast<const gu32x4*>(
        pin_base(reinterpret_cast<const u32x4*>(P + prep_wtf(HP, doM ? lM : 1)) + (size_t)(2 * w) * KS * 64));
    auto wload = [&](int s) {
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        wh[fb][s % RING] = (wf + (size_t)fb * KS * 64 + s * 64)[lane];
        if (TERMS == 3) wl[fb][s % RING] = (wf + (size_t)(HP * HP / 8) + (size_t)fb * KS * 64 + s * 64)[lane];
      }
    };
    auto bload = [&](int u) {
      const int s = u >> 2, j = u & 3;
      const int off = XI::chunk_off(col, 2 * s + h);
      bh[u & 1] = *reinterpret_cast<const u32x4*>(XM + j * XI::PLANE * 2 + off);
      if (TERMS == 3) bo[u & 1] = *reinterpret_cast<const u32x4*>(XM + XI::HALF * 2 + j * XI::PLANE * 2 + off);
    };
    auto jstep = [&](int u) {
      const int s = u >> 2, j = u & 3;
      if (j == 0 && s + PRE < KS) wload(s + PRE);
      if (u + 1 < 4 * KS) bload(u + 1);
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        if (s == 0) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          accM[fb][j] = TERMS == 3 ? mfma_bf16(wh[fb][0], bo[u & 1], zero) : mfma_bf16(wh[fb][0], bh[u & 1], zero);
          if (TERMS == 3) {
            accM[fb][j] = mfma_bf16(wl[fb][0], bh[u & 1], accM[fb][j]);
            accM[fb][j] = mfma_bf16(wh[fb][0], bh[u & 1], accM[fb][j]);
          }
        } else {
          if (TERMS == 3) {
            accM[fb][j] = mfma_bf16(wh[fb][s % RING], bo[u & 1], accM[fb][j]);
            accM[fb][j] = mfma_bf16(wl[fb][s % RING], bh[u & 1], accM[fb][j]);
          }
          accM[fb][j] = mfma_bf16(wh[fb][s % RING], bh[u & 1], accM[fb][j]);
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
    f32x4 sc[4], sn[4];            // saved (t, z_x, z_y, z_D) of the quad in flight / of the next quad (requested a quad ahead)
    f32x4 zq[4];                   // z-bar of the quad: [stream][element]
    f32x4 wov[3], dwv[2];          // per-element column terms of dW_out (first) / dW_0 (last)
    f32x4 wo4[3];
    auto sload = [&](int q, f32x4 (&dst)[4]) {
      const int fb = q >> 2, g = q & 3, ob = ob0 + 32 * fb;
      const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * PPL + col);
#pragma unroll
      for (int p = 0; p < 4; ++p)
        dst[p] = __builtin_nontemporal_load(pin_base(reinterpret_cast<const f32x4*>(Sl) + p * PLQ) + so);
    };
    // slot of the skinny-gradient accumulator owned by this lane after a transposing column sum (lanes col < 4 of
    // each half own feature ob + 8g + 4h + col); every other lane adds its (discarded) value to a private sink, so
    // the update is one unconditional ds_add_f32: no branch splits the MFMA block
    auto commit = [&](int base, int q, float v) {
      const int fb = q >> 2, g = q & 3, o = ob0 + 32 * fb + 8 * g + 4 * h + (col & 3);
      float* p = col < 4 ? &sgacc[base + o] : &dummy[w * 64 + lane];
      lds_add(p, v);
    };
    // The adjoint of register quad q = (fb, g) in EIGHT slices (one per 6-MFMA step): 0-3 = chain rule of element e,
    // 4-7 = stream p: bf16 hi/lo restage + spill, with the column sums of the skinny gradients spread over them.
    auto eslice = [&](int q, int i) {
      const int fb = q >> 2, g = q & 3, ob = ob0 + 32 * fb;
      if (i < 4) {
        const int e = i, r = 4 * g + e;
        if (e == 0) {
#pragma unroll
          for (int p = 0; p < 4; ++p) sc[p] = sn[p];
          if (q + 1 < 8) sload(q + 1, sn);
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
          ga = accE[fb][0][r]; gx = accE[fb][1][r]; gy = accE[fb][2][r]; gd = accE[fb][3][r];
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
      } else {
        const int p = i - 4;
        if (!last) {
          const int off = XI::chunk_off(col, (ob >> 3) + g) + 8 * h;
          u32x2 vh, vl;
          split4(zq[p][0], zq[p][1], zq[p][2], zq[p][3], vh, vl);
          *reinterpret_cast<u32x2*>(XE + p * XI::PLANE * 2 + off) = vh;
          if (TERMS == 3) *reinterpret_cast<u32x2*>(XE + XI::HALF * 2 + p * XI::PLANE * 2 + off) = vl;
          const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * PPL + col);
          __builtin_nontemporal_store(zq[p], pin_base(reinterpret_cast<const f32x4*>(Zl) + p * PLQ) + so);
        }
        // column sums of the four features at once (reduce_util.h); lane col == e of each half commits feature e.
        // One sum per slice: db in slice 4; dW_out (first kind) / dW_0 (last kind) in slices 5-7.
        if (p == 0) commit(sg_db(HP, lE), q, sum_cols4<32>(zq[0][0], zq[0][1], zq[0][2], zq[0][3], lane));
        if (first && p >= 1)
          commit(sg_wout(HP, L) + (p - 1) * HP, q, sum_cols4<32>(wov[(p + 2) % 3][0], wov[(p + 2) % 3][1], wov[(p + 2) % 3][2], wov[(p + 2) % 3][3], lane));
        if (last && (p == 1 || p == 2))
          commit(p == 1 ? sg_w0x(HP, L) : sg_w0y(HP, L), q, sum_cols4<32>(dwv[(p + 1) % 2][0], dwv[(p + 1) % 2][1], dwv[(p + 1) % 2][2], dwv[(p + 1) % 2][3], lane));
      }
    };

    if (doM) {
#pragma unroll
      for (int s = 0; s < PRE; ++s) wload(s);
      bload(0);
    }
    sload(0, sn);
    constexpr int NSTEP = 4 * KS, SPQ = NSTEP / 8;
    static_assert(SPQ == 8, "HP must be 256");
#pragma unroll
    for (int u = 0; u < NSTEP; ++u) {
      if (doM) jstep(u);
      eslice(u / SPQ, u % SPQ);
      if (doM) {
#pragma unroll
        for (int i = 0; i < (TERMS == 3 ? 6 : 2); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, TERMS == 3 ? 4 : 12, 0);   // adjoint VALU in its shadow
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- output adjoints of a tile (point_stage.h) into its LDS block; zero for the dummy partner tile ----
  auto seeds = [&](int tile, float* oadjT, float& px, float& py) {
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
    slot(F_{}, K0{}, accB, XB, 1, accA, XA, L - 1, tA, oadjA, pxA, pyA);                  //              E_L-1(A)
    __syncthreads();
    slot(T_{}, K0{}, accA, XA, L - 1, accB, XB, L - 1, tB, oadjB, pxB, pyB);              // G_L-1(A)   + E_L-1(B)
    __syncthreads();
    for (int l = L - 1; l >= 2; --l) {
      slot(T_{}, K1{}, accB, XB, l, accA, XA, l - 1, tA, oadjA, pxA, pyA);                // G_l(B)     + E_l-1(A)
      __syncthreads();
      slot(T_{}, K1{}, accA, XA, l - 1, accB, XB, l - 1, tB, oadjB, pxB, pyB);            // G_l-1(A)   + E_l-1(B)
      __syncthreads();
    }
    slot(T_{}, K2{}, accB, XB, 1, accA, XA, 0, tA, oadjA, pxA, pyA);                      // G_1(B)     + E_0(A)
    __syncthreads();
    slot(F_{}, K2{}, accA, XA, 1, accB, XB, 0, tB, oadjB, pxB, pyB);                      //              E_0(B)
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
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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
