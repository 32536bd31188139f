// bf16x3 / bf16 fused all-layer forward for WIDE nets (256 < hidden <= 512, e.g. 8x400).
// Same algorithm and data layouts as fwd_bf16.hip at 64-column tiles (16 points x 4 streams, S in the
// [plane][feature/4][16 cols][4] fp32 layout shared with fwd_wide / bwd_wide / dw_wide), replacing the
// reference lines listed there (NSFnet/net.py:52-54, NSFnet/pinn_solver.py:132-163,197-226,
// ev-NSFnet/pinn_solver.py:290-342,372-428).
//
// What differs: one wave owns TWO 32-feature blocks (64 features), so a workgroup is HP/64 <= 8 waves
// and every wave keeps the 256-register budget of two waves per SIMD.  The narrow kernel's shape (one
// block per wave) would need 13-16 waves at 128 registers each and spills 33-91 of them.  Per k-step a
// wave streams 2 x (hi, lo) weight fragments from L2 (register ring, 2 k-steps ahead) and reads 2 x
// (hi, lo) B fragments from the LDS image for 12 MFMAs - the same ratio as the 128-column kernel.
// If HP/32 is odd the last wave owns one block.
#include "kernels.h"
#include "point_stage.h"
#include "bf16_util.h"
#include <type_traits>

template <int HP, int NS, int TERMS>
__global__ __launch_bounds__(((HP / 32 + 1) / 2) * 64) void fwd_bf16_wide_kernel(FwdArgs a) {
  constexpr int COLS = 64, PPL = 16, NTL = 2, MT = 2;
  using XI = XImg<HP, PPL>;
  constexpr int NB = HP / 32, NWV = (NB + 1) / 2, NT = NWV * 64, KS = HP / 16;
  constexpr int PRE = 2, RING = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* Xb = ldsb;                                   // [2][4][16][RSE] bf16
  float* part = reinterpret_cast<float*>(ldsb + XI::BYTES);   // [NWV][4][COLS]
  float* outv = part + NWV * 4 * COLS;                        // [4][COLS]
  float* biasL = outv + 4 * COLS;                             // [L][HP] hidden-layer biases (l >= 1)
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
  const int hi = col >> 4;                                    // which of an accumulator tile's two planes
  const int pp = col & 15;                                    // column inside its plane
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mcount = (2 * w + 1 < NB) ? 2 : 1;                // 32-feature blocks of this wave
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * (NS == 4 ? PPL : COLS);
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < (a.L - 1) * HP; i += NT) biasL[HP + i] = a.prep[prep_b(HP, 1 + i / HP) + (i % HP)];
  __syncthreads();

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    f32x16 acc[MT][NTL];
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
      for (int m = 0; m < MT; ++m) {
        const int ob = (2 * w + (m < mcount ? m : 0)) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int o = ob + mfma_row(r, h);
          float wx = P[prep_w0x(HP) + o], wy = P[prep_w0y(HP) + o], b = P[prep_b0(HP) + o];
          if (NS == 4) {
            float z = fmaf(wx, px[0], fmaf(wy, py[0], b));
            acc[m][0][r] = hi ? wx : z; acc[m][1][r] = hi ? 0.f : wy;
          } else {
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[m][j][r] = fmaf(wx, px[j], fmaf(wy, py[j], b));
          }
        }
      }
    }
    for (int l = 0; l < L; ++l) {
      // next layer's first weight fragments are requested BEFORE this layer's activation stores
      // (vmcnt retires in order); the rest stream through the register ring inside the MFMA loop
      u32x4 wh[MT][RING], wl[MT][RING];
      const u32x4* wf = reinterpret_cast<const u32x4*>(P + prep_wf(HP, l + 1 < L ? l + 1 : 1)) + (size_t)(2 * w) * KS * 64 + lane;
      if (l < L - 1) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
          if (m < mcount) {
#pragma unroll
            for (int s = 0; s < PRE; ++s) {
              wh[m][s] = wf[(size_t)m * KS * 64 + s * 64];
              if (TERMS == 3) wl[m][s] = wf[(size_t)(HP * HP / 8) + (size_t)m * KS * 64 + s * 64];
            }
          }
      }
      asm volatile("" ::: "memory");
      float* Sl = a.S ? a.S + ((size_t)tile * L + l) * ((size_t)HP * COLS) : nullptr;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (m >= mcount) continue;
        const int ob = (2 * w + m) * 32;
        if (NS == 4) {
          // lanes 0-15 keep accumulator rows 0-7, lanes 16-31 rows 8-15; after the swaps
          // (acc[0][q], acc[0][q+8], acc[1][q], acc[1][q+8]) are the four streams of one row
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[m][0][q]), __float_as_uint(acc[m][0][q + 8]), false, false);
            auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[m][1][q]), __float_as_uint(acc[m][1][q + 8]), false, false);
            acc[m][0][q] = __uint_as_float(s01[0]); acc[m][0][q + 8] = __uint_as_float(s01[1]);
            acc[m][1][q] = __uint_as_float(s23[0]); acc[m][1][q + 8] = __uint_as_float(s23[1]);
          }
#pragma unroll
          for (int gq = 0; gq < 2; ++gq) {
            f32x4 a0, a1, a2, a3, s0, s1, s2, s3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int q = 4 * gq + e;
              float z = acc[m][0][q], zx = acc[m][0][q + 8], zy = acc[m][1][q], zd = acc[m][1][q + 8];
              float t = fast_tanh(z);
              float d1 = 1.f - t * t;
              float d2 = -2.f * t * d1;
              a0[e] = t; a1[e] = d1 * zx; a2[e] = d1 * zy; a3[e] = d2 * (zx * zx + zy * zy) + d1 * zd;
              s0[e] = t; s1[e] = zx; s2[e] = zy; s3[e] = zd;
            }
            // restage as bf16 hi/lo (8 bytes at [pp][chunk] + 8h per plane) and spill the saved values
            const int g = gq + 2 * hi;
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
              const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * PPL + pp);
              const f32x4* S4 = reinterpret_cast<const f32x4*>(Sl);
              if (a.s24) {      // 24-bit three-plane spill (bf16_util.h pack24): three instructions instead of four
                u32x4 pk[3];
                pack24_quad(s0, s1, s2, s3, pk);
#pragma unroll
                for (int k = 0; k < 3; ++k) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, pk[k]), pin_base(S4 + k * (HP / 4) * PPL) + so);
              } else {
                __builtin_nontemporal_store(s0, pin_base(S4 + 0 * (HP / 4) * PPL) + so);
                __builtin_nontemporal_store(s1, pin_base(S4 + 1 * (HP / 4) * PPL) + so);
                __builtin_nontemporal_store(s2, pin_base(S4 + 2 * (HP / 4) * PPL) + so);
                __builtin_nontemporal_store(s3, pin_base(S4 + 3 * (HP / 4) * PPL) + so);
              }
            }
          }
        } else {
          // value mode: every 32-column accumulator tile is two planes (2j + hi)
#pragma unroll
          for (int j = 0; j < NTL; ++j) {
            const int plane = 2 * j + hi;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 t4;
#pragma unroll
              for (int e = 0; e < 4; ++e) t4[e] = fast_tanh(acc[m][j][4 * g + e]);
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
      }
      __syncthreads();
      if (l == L - 1) break;
      // ------------- hidden GEMM l+1 on bf16 MFMA -------------
      {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int ob = (2 * w + (m < mcount ? m : 0)) * 32;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float b = biasL[(l + 1) * HP + ob + mfma_row(r, h)];
            if (NS == 4) { acc[m][0][r] = hi ? 0.f : b; acc[m][1][r] = 0.f; }
            else { acc[m][0][r] = b; acc[m][1][r] = b; }
          }
        }
        // B fragment of accumulator tile j: plane 2j+hi, column pp.  The XOR swizzle only touches the low
        // four chunk bits, so k-steps s and s+8 are exactly 256 bytes apart: eight base offsets serve the
        // whole K range, and the k loop runs in groups of eight (runtime outer loop, static ring slots).
        const unsigned char* Xl = Xb + hi * XI::PLANE * 2;
        constexpr int TSTR = 2 * XI::PLANE * 2;
        int base8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) base8[i] = XI::chunk_off(pp, 2 * i + h);
        u32x4 bh[NTL], bo[NTL];
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          bh[j] = *reinterpret_cast<const u32x4*>(Xl + j * TSTR + base8[0]);
          if (TERMS == 3) bo[j] = *reinterpret_cast<const u32x4*>(Xl + XI::HALF * 2 + j * TSTR + base8[0]);
        }
        auto steps = [&](auto cnt, int s0) {
          constexpr int CNT = decltype(cnt)::value;
          const unsigned char* Xg = Xl + s0 * 32;
#pragma unroll
          for (int i = 0; i < CNT; ++i) {
            const int sk = s0 + i;
            if (sk + PRE < KS) {     // stream the weight fragments PRE k-steps ahead
#pragma unroll
              for (int m = 0; m < MT; ++m)
                if (m < mcount) {
                  wh[m][(i + PRE) % RING] = wf[(size_t)m * KS * 64 + (sk + PRE) * 64];
                  if (TERMS == 3) wl[m][(i + PRE) % RING] = wf[(size_t)(HP * HP / 8) + (size_t)m * KS * 64 + (sk + PRE) * 64];
                }
            }
            u32x4 nh[NTL], no[NTL];
            if (sk + 1 < KS) {       // next k-step's B fragments in flight during this step's MFMAs
              const int off = base8[(i + 1) & 7] + ((i + 1) >> 3) * 256;
#pragma unroll
              for (int j = 0; j < NTL; ++j) {
                nh[j] = *reinterpret_cast<const u32x4*>(Xg + j * TSTR + off);
                if (TERMS == 3) no[j] = *reinterpret_cast<const u32x4*>(Xg + XI::HALF * 2 + j * TSTR + off);
              }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
              if (m < mcount) {
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                  if (TERMS == 3) {
                    acc[m][j] = mfma_bf16(wh[m][i % RING], bo[j], acc[m][j]);
                    acc[m][j] = mfma_bf16(wl[m][i % RING], bh[j], acc[m][j]);
                  }
                  acc[m][j] = mfma_bf16(wh[m][i % RING], bh[j], acc[m][j]);
                }
              }
            if (sk + 1 < KS) {
#pragma unroll
              for (int j = 0; j < NTL; ++j) { bh[j] = nh[j]; if (TERMS == 3) bo[j] = no[j]; }
            }
          }
        };
        int s0 = 0;
        for (; s0 + 8 <= KS; s0 += 8) steps(std::integral_constant<int, 8>{}, s0);
        steps(std::integral_constant<int, KS % 8>{}, s0);
      }
      __syncthreads();
    }
    // ---------------- output layer: VALU, K split over waves (this wave's 32-feature blocks) ----------------
    {
      float po[3] = {0.f, 0.f, 0.f};
      const int plane = lane / PPL, cp = lane % PPL;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (m >= mcount) continue;
        const int ob = (2 * w + m) * 32;
        const float* wo = P + prep_wout(HP, L) + ob;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          const int off = XI::chunk_off(cp, (ob >> 3) + ch);
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
              po[c3] = fmaf(wo[c3 * HP + kk], x0, po[c3]);
              po[c3] = fmaf(wo[c3 * HP + kk + 1], x1, po[c3]);
            }
          }
        }
      }
#pragma unroll
      for (int c3 = 0; c3 < 3; ++c3) part[(w * 4 + c3) * COLS + lane] = po[c3];
    }
    __syncthreads();
    for (int idx = tid; idx < 3 * COLS; idx += NT) {
      int c3 = idx / COLS, cc = idx % COLS;
      float s = (NS == 1 || cc < PPL) ? P[prep_bout(HP, L) + c3] : 0.f;
      for (int ww = 0; ww < NWV; ++ww) s += part[(ww * 4 + c3) * COLS + cc];
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

template <int HP>
static size_t lds_bytes_t(int L) {
  return XImg<HP, 16>::BYTES + ((size_t)((HP / 32 + 1) / 2) * 4 * 64 + 4 * 64 + (size_t)L * HP) * sizeof(float);
}

size_t fwd_bf16_wide_lds_bytes(int HP, int L) {
  // the image row stride is 512 elements for every HP in (256, 512]
  return XImg<512, 16>::BYTES + ((size_t)((HP / 32 + 1) / 2) * 4 * 64 + 4 * 64 + (size_t)L * HP) * sizeof(float);
}
int fwd_bf16_wide_threads(int HP) { return ((HP / 32 + 1) / 2) * 64; }

template <int HP, int NS, int TERMS>
static int launch_one(const FwdArgs& a, int grid, hipStream_t s) {
  size_t lds = lds_bytes_t<HP>(a.L);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_bf16_wide_kernel<HP, NS, TERMS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((fwd_bf16_wide_kernel<HP, NS, TERMS>), dim3(grid), dim3(((HP / 32 + 1) / 2) * 64), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

template <int HP>
static int launch_hp(int NS, int terms, const FwdArgs& a, int grid, hipStream_t s) {
  if (terms == 3) return NS == 4 ? launch_one<HP, 4, 3>(a, grid, s) : launch_one<HP, 1, 3>(a, grid, s);
  return NS == 4 ? launch_one<HP, 4, 1>(a, grid, s) : launch_one<HP, 1, 1>(a, grid, s);
}

int launch_fwd_bf16_wide(int HP, int NS, int terms, const FwdArgs& a, int grid, hipStream_t s) {
  switch (HP) {
    case 288: return launch_hp<288>(NS, terms, a, grid, s);
    case 320: return launch_hp<320>(NS, terms, a, grid, s);
    case 352: return launch_hp<352>(NS, terms, a, grid, s);
    case 384: return launch_hp<384>(NS, terms, a, grid, s);
    case 416: return launch_hp<416>(NS, terms, a, grid, s);
    case 448: return launch_hp<448>(NS, terms, a, grid, s);
    case 480: return launch_hp<480>(NS, terms, a, grid, s);
    case 512: return launch_hp<512>(NS, terms, a, grid, s);
    default: return -1000;
  }
}
