// bf16x3 / bf16 fused reverse sweep for WIDE nets (256 < hidden <= 512): the 64-column-tile algorithm of
// bwd_bf16.hip (reference: loss.backward(), NSFnet/pinn_solver.py:252, ev-NSFnet/pinn_solver.py:469) with
// TWO 32-feature blocks per wave, so the workgroup is HP/64 <= 8 waves at the 256-register budget (see
// fwd_bf16_wide.hip).  S / Z-bar use the [plane][feature/4][16 cols][4] fp32 layout of the other 64-column
// kernels, so forward, reverse sweep and dW kernels of different precisions interoperate.
#include "kernels.h"
#include "point_stage.h"
#include "bf16_util.h"
#include "reduce_util.h"
#include <type_traits>

template <int HP, int NS, int TERMS>
__global__ __launch_bounds__(((HP / 32 + 1) / 2) * 64) void bwd_bf16_wide_kernel(BwdArgs a) {
  constexpr int COLS = 64, PPL = 16, NTL = 2, MT = 2;
  using XI = XImg<HP, PPL>;
  constexpr int NB = HP / 32, NWV = (NB + 1) / 2, NT = NWV * 64, KS = HP / 16;
  constexpr int PRE = 2, RING = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* Xb = ldsb;                                       // [2][4][PPL][RSE] bf16
  float* oadjL = reinterpret_cast<float*>(ldsb + XI::BYTES);      // [4][COLS]
  float* sgacc = oadjL + 4 * COLS;                                // [sg_total]
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
  const int hi = col >> 4;
  const int pp = col & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mcount = (2 * w + 1 < NB) ? 2 : 1;                    // 32-feature blocks of this wave
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * (NS == 4 ? PPL : COLS);
  const int SG = sg_total(HP, L);
  for (int i = tid; i < SG; i += NT) sgacc[i] = 0.f;
  float dbo[3] = {0.f, 0.f, 0.f};
  __syncthreads();

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    // ---------------- output adjoints per column (point_stage.h) ----------------
    float px[NTL], py[NTL];
    output_adjoint_stage<PPL, COLS, NS, NT, NTL>(a, tile, tid, col, pp, npad, oadjL, dbo, px, py);
    __syncthreads();
    // ---------------- adjoint of the last hidden layer's activations (rank-3 update) ----------------
    f32x16 acc[MT][NTL];
    float oc[3][NTL];   // output adjoints of this lane's column in accumulator tile j
    float oa[3][4];     // residual mode: the four streams of this lane's point (output-layer dW)
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) {
#pragma unroll
      for (int j = 0; j < NTL; ++j) oc[c3][j] = oadjL[c3 * COLS + 32 * j + col];
#pragma unroll
      for (int s = 0; s < 4; ++s) oa[c3][s] = oadjL[c3 * COLS + s * PPL + pp];
    }
    {
      const float* wo = P + prep_wout(HP, L);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int ob = (2 * w + (m < mcount ? m : 0)) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = ob + mfma_row(r, h);
          float w0 = wo[o], w1 = wo[HP + o], w2 = wo[2 * HP + o];
#pragma unroll
          for (int j = 0; j < NTL; ++j) acc[m][j][r] = w0 * oc[0][j] + w1 * oc[1][j] + w2 * oc[2][j];
        }
      }
    }
    for (int l = L - 1; l >= 0; --l) {
      // W_l^T fragments: the first PRE k-steps are requested before this layer's Zb stores
      // (vmcnt retires in order), the rest stream through the register ring in the MFMA loop
      u32x4 wh[MT][RING], wl[MT][RING];
      const u32x4* wf = reinterpret_cast<const u32x4*>(P + prep_wtf(HP, l > 0 ? l : 1)) + (size_t)(2 * w) * KS * 64 + lane;
      if (l > 0) {
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
      const float* Sl = a.S + ((size_t)tile * L + l) * ((size_t)HP * COLS);
      float* Zl = a.Zb + ((size_t)tile * L + l) * ((size_t)HP * COLS);
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
            const int g = gq + 2 * hi;
            const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * PPL + pp);
            const f32x4* S4 = reinterpret_cast<const f32x4*>(Sl);
            f32x4 s0, s1, s2, s3;
            if (a.s24) {      // 24-bit three-plane spill (bf16_util.h pack24)
              u32x4 pk[3];
#pragma unroll
              for (int k = 0; k < 3; ++k) pk[k] = __builtin_bit_cast(u32x4, __builtin_nontemporal_load(pin_base(S4 + k * (HP / 4) * PPL) + so));
              s0 = unpack24_plane(pk, 0); s1 = unpack24_plane(pk, 1); s2 = unpack24_plane(pk, 2); s3 = unpack24_plane(pk, 3);
            } else {
              s0 = __builtin_nontemporal_load(pin_base(S4 + 0 * (HP / 4) * PPL) + so); s1 = __builtin_nontemporal_load(pin_base(S4 + 1 * (HP / 4) * PPL) + so);
              s2 = __builtin_nontemporal_load(pin_base(S4 + 2 * (HP / 4) * PPL) + so); s3 = __builtin_nontemporal_load(pin_base(S4 + 3 * (HP / 4) * PPL) + so);
            }
            f32x4 z0, z1, z2, z3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int q = 4 * gq + e;
              float ga = acc[m][0][q], gx = acc[m][0][q + 8], gy = acc[m][1][q], gd = acc[m][1][q + 8];
              float t = s0[e], zx = s1[e], zy = s2[e], zd = s3[e];
              float d1 = 1.f - t * t;
              float d2 = -2.f * t * d1;
              float d3 = -2.f * d1 * (1.f - 3.f * t * t);
              z1[e] = d1 * gx + 2.f * d2 * zx * gd;
              z2[e] = d1 * gy + 2.f * d2 * zy * gd;
              z3[e] = d1 * gd;
              z0[e] = d1 * ga + d2 * (zx * gx + zy * gy) + (d3 * (zx * zx + zy * zy) + d2 * zd) * gd;
            }
            f32x4 wo0v, wo1v, wo2v, dwxv, dwyv;      // per-element column terms of the skinny gradients
            if (l == L - 1) {   // dWout[c][o] += sum_s oadj[c][s] * a_s[o]
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float t = s0[e], zx = s1[e], zy = s2[e], zd = s3[e];
                float d1 = 1.f - t * t, d2 = -2.f * t * d1;
                float ax = d1 * zx, ay = d1 * zy, ad = d2 * (zx * zx + zy * zy) + d1 * zd;
                wo0v[e] = oa[0][0] * t + oa[0][1] * ax + oa[0][2] * ay + oa[0][3] * ad;
                wo1v[e] = oa[1][0] * t + oa[1][1] * ax + oa[1][2] * ay + oa[1][3] * ad;
                wo2v[e] = oa[2][0] * t + oa[2][1] * ax + oa[2][2] * ay + oa[2][3] * ad;
              }
            }
            if (l == 0) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { dwxv[e] = z0[e] * px[0] + z1[e]; dwyv[e] = z0[e] * py[0] + z2[e]; }
            }
            {   // column sums of the four features at once; lane pp == e of each 16-lane group commits feature e
              const int o = ob + 8 * g + 4 * h + (pp & 3);
              const float dbv = sum_cols4<PPL>(z0[0], z0[1], z0[2], z0[3], lane);
              float w0 = 0.f, w1 = 0.f, w2 = 0.f, dx = 0.f, dy = 0.f;
              if (l == L - 1) {
                w0 = sum_cols4<PPL>(wo0v[0], wo0v[1], wo0v[2], wo0v[3], lane);
                w1 = sum_cols4<PPL>(wo1v[0], wo1v[1], wo1v[2], wo1v[3], lane);
                w2 = sum_cols4<PPL>(wo2v[0], wo2v[1], wo2v[2], wo2v[3], lane);
              }
              if (l == 0) {
                dx = sum_cols4<PPL>(dwxv[0], dwxv[1], dwxv[2], dwxv[3], lane);
                dy = sum_cols4<PPL>(dwyv[0], dwyv[1], dwyv[2], dwyv[3], lane);
              }
              if (pp < 4) {
                lds_add(&sgacc[sg_db(HP, l) + o], dbv);
                if (l == L - 1) {
                  lds_add(&sgacc[sg_wout(HP, L) + o], w0);
                  lds_add(&sgacc[sg_wout(HP, L) + HP + o], w1);
                  lds_add(&sgacc[sg_wout(HP, L) + 2 * HP + o], w2);
                }
                if (l == 0) { lds_add(&sgacc[sg_w0x(HP, L) + o], dx); lds_add(&sgacc[sg_w0y(HP, L) + o], dy); }
              }
            }
            if (l > 0) {
              const int off = XI::chunk_off(pp, (ob >> 3) + g) + 8 * h;
              u32x2 vh, vl;
              split4(z0[0], z0[1], z0[2], z0[3], vh, vl);
              *reinterpret_cast<u32x2*>(Xb + 0 * XI::PLANE * 2 + off) = vh;
              if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + 0 * XI::PLANE * 2 + off) = vl;
              split4(z1[0], z1[1], z1[2], z1[3], vh, vl);
              *reinterpret_cast<u32x2*>(Xb + 1 * XI::PLANE * 2 + off) = vh;
              if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + 1 * XI::PLANE * 2 + off) = vl;
              split4(z2[0], z2[1], z2[2], z2[3], vh, vl);
              *reinterpret_cast<u32x2*>(Xb + 2 * XI::PLANE * 2 + off) = vh;
              if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + 2 * XI::PLANE * 2 + off) = vl;
              split4(z3[0], z3[1], z3[2], z3[3], vh, vl);
              *reinterpret_cast<u32x2*>(Xb + 3 * XI::PLANE * 2 + off) = vh;
              if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + 3 * XI::PLANE * 2 + off) = vl;
              const f32x4* Z4 = reinterpret_cast<const f32x4*>(Zl);
              if (a.s24) {
                u32x4 pk[3];
                pack24_quad(z0, z1, z2, z3, pk);
#pragma unroll
                for (int k = 0; k < 3; ++k) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, pk[k]), pin_base(Z4 + k * (HP / 4) * PPL) + so);
              } else {
                __builtin_nontemporal_store(z0, pin_base(Z4 + 0 * (HP / 4) * PPL) + so); __builtin_nontemporal_store(z1, pin_base(Z4 + 1 * (HP / 4) * PPL) + so);
                __builtin_nontemporal_store(z2, pin_base(Z4 + 2 * (HP / 4) * PPL) + so); __builtin_nontemporal_store(z3, pin_base(Z4 + 3 * (HP / 4) * PPL) + so);
              }
            }
          }
        } else {
          // value mode: z-bar = (1 - t^2) * g per column; every accumulator tile is two planes (2j + hi)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 zj[NTL];
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
              const int plane = 2 * j + hi;
              const f32x4 t4 = *(reinterpret_cast<const f32x4*>(Sl) + ((size_t)plane * (HP / 4) + (ob >> 2) + 2 * g + h) * PPL + pp);
#pragma unroll
              for (int e = 0; e < 4; ++e) zj[j][e] = (1.f - t4[e] * t4[e]) * acc[m][j][4 * g + e];
              if (l == L - 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const int o = ob + 8 * g + 4 * h + e;
                  float w0 = sum_cols<32>(oc[0][j] * t4[e]), w1 = sum_cols<32>(oc[1][j] * t4[e]), w2 = sum_cols<32>(oc[2][j] * t4[e]);
                  if (col == 0) {
                    lds_add(&sgacc[sg_wout(HP, L) + o], w0);
                    lds_add(&sgacc[sg_wout(HP, L) + HP + o], w1);
                    lds_add(&sgacc[sg_wout(HP, L) + 2 * HP + o], w2);
                  }
                }
              }
              if (l > 0) {
                const int off = XI::chunk_off(pp, (ob >> 3) + g) + 8 * h;
                u32x2 vh, vl;
                split4(zj[j][0], zj[j][1], zj[j][2], zj[j][3], vh, vl);
                *reinterpret_cast<u32x2*>(Xb + plane * XI::PLANE * 2 + off) = vh;
                if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + plane * XI::PLANE * 2 + off) = vl;
                *(reinterpret_cast<f32x4*>(Zl) + ((size_t)plane * (HP / 4) + (ob >> 2) + 2 * g + h) * PPL + pp) = zj[j];
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int o = ob + 8 * g + 4 * h + e;
              float dbv = 0.f, dwx = 0.f, dwy = 0.f;
#pragma unroll
              for (int j = 0; j < NTL; ++j) { dbv += zj[j][e]; dwx += zj[j][e] * px[j]; dwy += zj[j][e] * py[j]; }
              dbv = sum_cols<32>(dbv);
              if (col == 0) lds_add(&sgacc[sg_db(HP, l) + o], dbv);
              if (l == 0) {
                dwx = sum_cols<32>(dwx); dwy = sum_cols<32>(dwy);
                if (col == 0) { lds_add(&sgacc[sg_w0x(HP, L) + o], dwx); lds_add(&sgacc[sg_w0y(HP, L) + o], dwy); }
              }
            }
          }
        }
      }
      if (l == 0) break;
      __syncthreads();
      // ------------- G_{l-1}[i][col] = sum_o W_l[o][i] Zb_l[o][col]  (bf16 MFMA) -------------
      {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
        // eight base offsets serve the whole K range (see fwd_bf16_wide.hip); k loop in groups of eight
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
            if (sk + PRE < KS) {
#pragma unroll
              for (int m = 0; m < MT; ++m)
                if (m < mcount) {
                  wh[m][(i + PRE) % RING] = wf[(size_t)m * KS * 64 + (sk + PRE) * 64];
                  if (TERMS == 3) wl[m][(i + PRE) % RING] = wf[(size_t)(HP * HP / 8) + (size_t)m * KS * 64 + (sk + PRE) * 64];
                }
            }
            u32x4 nh[NTL], no[NTL];
            if (sk + 1 < KS) {
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

size_t bwd_bf16_wide_lds_bytes(int HP, int L) {
  return XImg<512, 16>::BYTES + ((size_t)4 * 64 + sg_total(HP, L)) * sizeof(float);
}

template <int HP, int NS, int TERMS>
static int launch_one(const BwdArgs& a, int grid, hipStream_t s) {
  size_t lds = bwd_bf16_wide_lds_bytes(HP, a.L);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_bf16_wide_kernel<HP, NS, TERMS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((bwd_bf16_wide_kernel<HP, NS, TERMS>), dim3(grid), dim3(((HP / 32 + 1) / 2) * 64), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

template <int HP>
static int launch_hp(int NS, int terms, const BwdArgs& a, int grid, hipStream_t s) {
  if (terms == 3) return NS == 4 ? launch_one<HP, 4, 3>(a, grid, s) : launch_one<HP, 1, 3>(a, grid, s);
  return NS == 4 ? launch_one<HP, 4, 1>(a, grid, s) : launch_one<HP, 1, 1>(a, grid, s);
}

int launch_bwd_bf16_wide(int HP, int NS, int terms, const BwdArgs& a, int grid, hipStream_t s) {
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
