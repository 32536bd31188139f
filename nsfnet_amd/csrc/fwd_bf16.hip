// bf16x3 (and plain bf16) variant of the fused all-layer forward (see fwd.hip for the
// algorithm and the reference lines it replaces: NSFnet/net.py:52-54,
// NSFnet/pinn_solver.py:132-163,197-226, ev-NSFnet/pinn_solver.py:290-342,372-428).
//
// Differences from the fp32-MFMA kernel: the hidden GEMMs run on
// v_mfma_f32_32x32x16_bf16 (16x the f32-input MFMA rate) with every fp32 operand split
// into bf16 hi + lo and three products per term (TERMS = 3; TERMS = 1 = plain bf16 fast
// mode); the tile's activations live in LDS as X[hi|lo][stream][col][k] bf16 with
// XOR-swizzled 16-byte chunks so each B fragment is one conflict-free ds_read_b128; the
// wave's weight slice (A operand, hi and lo fragments) stays in VGPRs for the layer.
// Saved activations (S) keep the fp32 layout of layout.h, so fwd / bwd / dW kernels of
// different precisions interoperate.
#include "kernels.h"
#include "point_stage.h"
#include "bf16_util.h"

// COLS = 128: tile = 32 points x 4 streams, one workgroup per CU at HP = 256.
// COLS = 64 : tile = 16 points x 4 streams (two streams per 32-column accumulator tile, exchanged
//             with v_permlane16_swap before the lane-local epilogue); half the LDS and <= 128 VGPRs,
//             so two workgroups share a CU and cover each other's epilogue / spill / MFMA phases.
template <int HP, int NS, int TERMS, int COLS>
__global__ __launch_bounds__(HP * 2, (COLS == 64 && HP == 256) ? 4 : 2) void fwd_bf16_kernel(FwdArgs a) {
  constexpr int PPL = COLS / 4, NTL = COLS / 32;
  using XI = XImg<HP, PPL>;
  constexpr int NW = HP / 32, NT = HP * 2, KS = HP / 16;
#ifndef PINN_PREK
#define PINN_PREK 4
#endif
  constexpr int PREK = PINN_PREK;   // weight k-steps requested ahead of the epilogue's store burst
  constexpr int PRE = COLS == 64 ? (KS < 2 ? KS : 2) : (KS < PREK ? KS : PREK);
  constexpr int RING = (PRE + 2 < KS) ? PRE + 2 : KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* Xb = ldsb;                                   // [2][4][32][RSE] bf16
  float* part = reinterpret_cast<float*>(ldsb + XI::BYTES);   // [NW][4][COLS]
  float* outv = part + NW * 4 * COLS;                         // [4][COLS]
  float* biasL = outv + 4 * COLS;                             // [L][HP] hidden-layer biases (l >= 1)
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
  const int hi = COLS == 64 ? (col >> 4) : 0;                 // which of the tile's two planes (64-col tiles)
  const int pp = COLS == 64 ? (col & 15) : col;               // column inside its plane
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ob = w * 32;
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * (NS == 4 ? PPL : COLS);
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < (a.L - 1) * HP; i += NT) biasL[HP + i] = a.prep[prep_b(HP, 1 + i / HP) + (i % HP)];
  // De-phase the workgroups: every workgroup runs the same store-burst / MFMA cadence, and in
  // lockstep the whole chip hits HBM in the same windows.  A one-off start offset spreads them.
  if (a.stagger > 0) {
    const int slots = (blockIdx.x * 5) & 7;
    for (int i = 0; i < slots * a.stagger; ++i) __builtin_amdgcn_s_sleep(64);
  }
  __syncthreads();

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    f32x16 acc[NTL];
    {
      float px[NTL], py[NTL];
      if (NS == 4) {
        int pt = tile * PPL + pp;
        px[0] = pt < a.n ? a.x[pt] : 0.f;
        py[0] = pt < a.n ? a.y[pt] : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          int pt = tile * COLS + 32 * j + col;
          px[j] = pt < a.n ? a.x[pt] : 0.f;
          py[j] = pt < a.n ? a.y[pt] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int o = ob + mfma_row(r, h);
        float wx = P[prep_w0x(HP) + o], wy = P[prep_w0y(HP) + o], b = P[prep_b0(HP) + o];
        if (NS == 4) {
          float z = fmaf(wx, px[0], fmaf(wy, py[0], b));
          if (COLS == 128) { acc[0][r] = z; acc[1][r] = wx; acc[2 % NTL][r] = wy; acc[3 % NTL][r] = 0.f; }
          else { acc[0][r] = hi ? wx : z; acc[1][r] = hi ? 0.f : wy; }
        } else {
#pragma unroll
          for (int j = 0; j < NTL; ++j) acc[j][r] = fmaf(wx, px[j], fmaf(wy, py[j], b));
        }
      }
    }
    for (int l = 0; l < L; ++l) {
      // Next layer's weight fragments and bias are requested BEFORE this layer's activation
      // stores: vmcnt retires in order, so a weight load issued behind the 16 S stores would
      // make the MFMA loop wait for the whole HBM store burst.
      // (only the first PRE k-steps; the rest stream through a small register ring inside the MFMA
      // loop so that the B fragments can be double-buffered: see gemm_ring below)
      u32x4 wh[RING], wl[RING];
      const u32x4* wf = reinterpret_cast<const u32x4*>(P + prep_wf(HP, l + 1 < L ? l + 1 : 1)) + (size_t)w * KS * 64 + lane;
      if (l < L - 1) {
#pragma unroll
        for (int s = 0; s < PRE; ++s) {
          wh[s] = wf[s * 64];
          if (TERMS == 3) wl[s] = wf[(size_t)(HP * HP / 8) + s * 64];
        }
      }
      asm volatile("" ::: "memory");
      float* Sl = a.S ? a.S + ((size_t)tile * L + l) * ((size_t)HP * COLS) : nullptr;
      // one register quad (4 consecutive features `o4*4..`, one column) of the four planes:
      // restage as bf16 hi/lo (8 bytes at [pp][chunk] + 8h per plane) and spill the saved values
      auto emit = [&](int g, const f32x4& a0, const f32x4& a1, const f32x4& a2, const f32x4& a3,
                      const f32x4& s0, const f32x4& s1, const f32x4& s2, const f32x4& s3) {
        const int off = XI::chunk_off(pp, (ob >> 3) + g) + 8 * h;
        u32x2 vh, vl;
        split4(a0[0], a0[1], a0[2], a0[3], vh, vl);
        *reinterpret_cast<u32x2*>(Xb + 0 * XI::PLANE * 2 + off) = vh;
        if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + 0 * XI::PLANE * 2 + off) = vl;
        split4(a1[0], a1[1], a1[2], a1[3], vh, vl);
        *reinterpret_cast<u32x2*>(Xb + 1 * XI::PLANE * 2 + off) = vh;
        if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + 1 * XI::PLANE * 2 + off) = vl;
        split4(a2[0], a2[1], a2[2], a2[3], vh, vl);
        *reinterpret_cast<u32x2*>(Xb + 2 * XI::PLANE * 2 + off) = vh;
        if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + 2 * XI::PLANE * 2 + off) = vl;
        split4(a3[0], a3[1], a3[2], a3[3], vh, vl);
        *reinterpret_cast<u32x2*>(Xb + 3 * XI::PLANE * 2 + off) = vh;
        if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + 3 * XI::PLANE * 2 + off) = vl;
        if (Sl) {
          // uniform plane bases (scalar registers) + ONE 32-bit lane offset: the stores then take the
          // saddr + voffset form instead of a 64-bit VALU address per plane
          const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * PPL + pp);
          constexpr size_t PLQ = (size_t)(HP / 4) * PPL;          // f32x4 per plane
          const f32x4* S4 = reinterpret_cast<const f32x4*>(Sl);
          // streamed once: nontemporal keeps the spill out of L2's way (measured -3 % on the kernel)
          __builtin_nontemporal_store(s0, pin_base(S4 + 0 * PLQ) + so);
          __builtin_nontemporal_store(s1, pin_base(S4 + 1 * PLQ) + so);
          __builtin_nontemporal_store(s2, pin_base(S4 + 2 * PLQ) + so);
          __builtin_nontemporal_store(s3, pin_base(S4 + 3 * PLQ) + so);
        }
      };
      auto chain = [&](float z, float zx, float zy, float zd, int e, f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3,
                       f32x4& s0, f32x4& s1, f32x4& s2, f32x4& s3) {
        float t = fast_tanh(z);
        float d1 = 1.f - t * t;
        float d2 = -2.f * t * d1;
        a0[e] = t; a1[e] = d1 * zx; a2[e] = d1 * zy; a3[e] = d2 * (zx * zx + zy * zy) + d1 * zd;
        s0[e] = t; s1[e] = zx; s2[e] = zy; s3[e] = zd;
      };
      if (NS == 4 && COLS == 128) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 a0, a1, a2, a3, s0, s1, s2, s3;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            chain(acc[0][r], acc[1][r], acc[2 % NTL][r], acc[3 % NTL][r], e, a0, a1, a2, a3, s0, s1, s2, s3);
          }
          emit(g, a0, a1, a2, a3, s0, s1, s2, s3);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (NS == 4) {
        // 64-column tile: lanes 0-15 keep accumulator rows 0-7, lanes 16-31 rows 8-15; after the
        // swaps (acc[0][q], acc[0][q+8], acc[1][q], acc[1][q+8]) are the four streams of one row
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][q]), __float_as_uint(acc[0][q + 8]), false, false);
          auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[1 % NTL][q]), __float_as_uint(acc[1 % NTL][q + 8]), false, false);
          acc[0][q] = __uint_as_float(s01[0]); acc[0][q + 8] = __uint_as_float(s01[1]);
          acc[1 % NTL][q] = __uint_as_float(s23[0]); acc[1 % NTL][q + 8] = __uint_as_float(s23[1]);
        }
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          f32x4 a0, a1, a2, a3, s0, s1, s2, s3;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int q = 4 * gq + e;
            chain(acc[0][q], acc[0][q + 8], acc[1 % NTL][q], acc[1 % NTL][q + 8], e, a0, a1, a2, a3, s0, s1, s2, s3);
          }
          emit(gq + 2 * hi, a0, a1, a2, a3, s0, s1, s2, s3);
        }
      } else {
        // value mode: every 32-column accumulator tile is one (COLS = 128) or two (COLS = 64) planes
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          const int plane = COLS == 128 ? j : 2 * j + hi;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 t4;
#pragma unroll
            for (int e = 0; e < 4; ++e) t4[e] = fast_tanh(acc[j][4 * g + e]);
            const int off = XI::chunk_off(pp, (ob >> 3) + g) + 8 * h;
            u32x2 vh, vl;
            split4(t4[0], t4[1], t4[2], t4[3], vh, vl);
            *reinterpret_cast<u32x2*>(Xb + plane * XI::PLANE * 2 + off) = vh;
            if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + plane * XI::PLANE * 2 + off) = vl;
            if (Sl) {
              f32x4* Sg = reinterpret_cast<f32x4*>(Sl) + ((size_t)plane * (HP / 4) + (ob >> 2) + 2 * g + h) * PPL + pp;
              *Sg = t4;
            }
          }
        }
      }
      __syncthreads();
      if (l == L - 1) break;
      // ------------- hidden GEMM l+1 on bf16 MFMA -------------
      {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float b = biasL[(l + 1) * HP + ob + mfma_row(r, h)];
          if (NS == 4) {
            if (COLS == 128) { acc[0][r] = b; acc[1][r] = 0.f; acc[2 % NTL][r] = 0.f; acc[3 % NTL][r] = 0.f; }
            else { acc[0][r] = hi ? 0.f : b; acc[1][r] = 0.f; }
          } else {
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[j][r] = b;
          }
        }
        // B fragment of accumulator tile j: plane j (128 cols) or plane 2j+hi (64 cols), column pp
        const unsigned char* Xl = Xb + (COLS == 128 ? 0 : hi * XI::PLANE * 2);
        constexpr int TSTR = (COLS == 128 ? 1 : 2) * XI::PLANE * 2;
        u32x4 bh[NTL], bo[NTL];
        {
          const int off0 = XI::chunk_off(pp, h);
#pragma unroll
          for (int j = 0; j < NTL; ++j) {
            bh[j] = *reinterpret_cast<const u32x4*>(Xl + j * TSTR + off0);
            if (TERMS == 3) bo[j] = *reinterpret_cast<const u32x4*>(Xl + XI::HALF * 2 + j * TSTR + off0);
          }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          if (s + PRE < KS) {     // stream the weight fragments PRE k-steps ahead
            wh[(s + PRE) % RING] = wf[(s + PRE) * 64];
            if (TERMS == 3) wl[(s + PRE) % RING] = wf[(size_t)(HP * HP / 8) + (s + PRE) * 64];
          }
          u32x4 nh[NTL], no[NTL];
          if (s + 1 < KS) {       // next k-step's B fragments in flight during this step's MFMAs
            const int off = XI::chunk_off(pp, 2 * (s + 1) + h);
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
              nh[j] = *reinterpret_cast<const u32x4*>(Xl + j * TSTR + off);
              if (TERMS == 3) no[j] = *reinterpret_cast<const u32x4*>(Xl + XI::HALF * 2 + j * TSTR + off);
            }
          }
#pragma unroll
          for (int j = 0; j < NTL; ++j) {
            if (TERMS == 3) {
              acc[j] = mfma_bf16(wh[s % RING], bo[j], acc[j]);
              acc[j] = mfma_bf16(wl[s % RING], bh[j], acc[j]);
            }
            acc[j] = mfma_bf16(wh[s % RING], bh[j], acc[j]);
          }
          if (s + 1 < KS) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) { bh[j] = nh[j]; if (TERMS == 3) bo[j] = no[j]; }
          }
        }
      }
      __syncthreads();
    }
    // ---------------- output layer: VALU, K split over waves ----------------
    {
      constexpr int NH = COLS / 64;           // columns per lane
      float po[3][NH];
#pragma unroll
      for (int c3 = 0; c3 < 3; ++c3)
#pragma unroll
        for (int half = 0; half < NH; ++half) po[c3][half] = 0.f;
      const float* wo = P + prep_wout(HP, L) + ob;
#pragma unroll
      for (int half = 0; half < NH; ++half) {
        const int c2 = lane + 64 * half, plane = c2 / PPL, cp = c2 % PPL;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          const int off = XI::chunk_off(cp, 4 * w + ch);
          u32x4 vh = *reinterpret_cast<const u32x4*>(Xb + plane * XI::PLANE * 2 + off);
          u32x4 vl = {0u, 0u, 0u, 0u};
          if (TERMS == 3) vl = *reinterpret_cast<const u32x4*>(Xb + XI::HALF * 2 + plane * XI::PLANE * 2 + off);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float x0 = bf_lo_f(vh[q]) + bf_lo_f(vl[q]);
            float x1 = bf_hi_f(vh[q]) + bf_hi_f(vl[q]);
            const int kk = 8 * ch + 2 * q;
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) {
              po[c3][half] = fmaf(wo[c3 * HP + kk], x0, po[c3][half]);
              po[c3][half] = fmaf(wo[c3 * HP + kk + 1], x1, po[c3][half]);
            }
          }
        }
      }
#pragma unroll
      for (int c3 = 0; c3 < 3; ++c3)
#pragma unroll
        for (int half = 0; half < NH; ++half) part[(w * 4 + c3) * COLS + 64 * half + lane] = po[c3][half];
    }
    __syncthreads();
    for (int idx = tid; idx < 3 * COLS; idx += NT) {
      int c3 = idx / COLS, cc = idx % COLS;
      float s = (NS == 1 || cc < PPL) ? P[prep_bout(HP, L) + c3] : 0.f;
      for (int ww = 0; ww < NW; ++ww) s += part[(ww * 4 + c3) * COLS + cc];
      outv[c3 * COLS + cc] = s;
    }
    __syncthreads();
    // ---------------- per-point stage (point_stage.h) ----------------
    if (NS == 4) residual_point_stage<PPL, COLS>(a, outv, tile, tid, npad, lsum);
    else value_point_stage<COLS, NT>(a, outv, tile, tid, npad, lsum);
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

template <int HP, int COLS>
static size_t lds_bytes_t(int L) {
  return XImg<HP, COLS / 4>::BYTES + ((size_t)(HP / 32) * 4 * COLS + 4 * COLS + (size_t)L * HP) * sizeof(float);
}

size_t fwd_bf16_lds_bytes(int HP, int L, int cols) {
#define LB(hp) case hp: return cols == 64 ? lds_bytes_t<hp, 64>(L) : lds_bytes_t<hp, 128>(L);
  switch (HP) { LB(32) LB(64) LB(96) LB(128) LB(160) LB(192) LB(224) default: return cols == 64 ? lds_bytes_t<256, 64>(L) : lds_bytes_t<256, 128>(L); }
#undef LB
}

template <int HP, int NS, int TERMS, int COLS>
static int launch_one(const FwdArgs& a, int grid, hipStream_t s) {
  size_t lds = lds_bytes_t<HP, COLS>(a.L);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_bf16_kernel<HP, NS, TERMS, COLS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((fwd_bf16_kernel<HP, NS, TERMS, COLS>), dim3(grid), dim3(HP * 2), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

template <int HP, int COLS>
static int launch_hp(int NS, int terms, const FwdArgs& a, int grid, hipStream_t s) {
  if (terms == 3) return NS == 4 ? launch_one<HP, 4, 3, COLS>(a, grid, s) : launch_one<HP, 1, 3, COLS>(a, grid, s);
  return NS == 4 ? launch_one<HP, 4, 1, COLS>(a, grid, s) : launch_one<HP, 1, 1, COLS>(a, grid, s);
}

int launch_fwd_bf16(int HP, int NS, int terms, int cols, const FwdArgs& a, int grid, hipStream_t s) {
  if (cols == 64) {
    switch (HP) {
      case 128: return launch_hp<128, 64>(NS, terms, a, grid, s);
      case 256: return launch_hp<256, 64>(NS, terms, a, grid, s);
      default: return -1000;
    }
  }
  switch (HP) {
    case 32: return launch_hp<32, 128>(NS, terms, a, grid, s);
    case 64: return launch_hp<64, 128>(NS, terms, a, grid, s);
    case 96: return launch_hp<96, 128>(NS, terms, a, grid, s);
    case 128: return launch_hp<128, 128>(NS, terms, a, grid, s);
    case 160: return launch_hp<160, 128>(NS, terms, a, grid, s);
    case 192: return launch_hp<192, 128>(NS, terms, a, grid, s);
    case 224: return launch_hp<224, 128>(NS, terms, a, grid, s);
    case 256: return launch_hp<256, 128>(NS, terms, a, grid, s);
    default: return -1000;
  }
}
