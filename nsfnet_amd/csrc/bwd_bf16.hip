// bf16x3 (and plain bf16) variant of the fused reverse sweep (see bwd.hip for the algorithm
// and the reference lines it replaces: loss.backward(), NSFnet/pinn_solver.py:252,
// ev-NSFnet/pinn_solver.py:469).  G_{l-1} = W_l^T Zb_l runs on v_mfma_f32_32x32x16_bf16 with
// hi/lo split operands (bf16_util.h); the z-adjoint tile lives in LDS as
// X[hi|lo][plane][col][k] bf16 (swizzled); Zb spilled to HBM stays fp32 (layout.h).
// COLS = 128 / 64: tile geometry as in fwd_bf16.hip.
#include "kernels.h"
#include "point_stage.h"
#include "bf16_util.h"
#include "reduce_util.h"

template <int HP, int NS, int TERMS, int COLS>
__global__ __launch_bounds__(HP * 2, 2) void bwd_bf16_kernel(BwdArgs a) {
  constexpr int PPL = COLS / 4, NTL = COLS / 32;
  using XI = XImg<HP, PPL>;
  constexpr int NT = HP * 2, KS = HP / 16;
  constexpr int PRE = COLS == 64 ? (KS < 2 ? KS : 2) : (KS < 4 ? KS : 4);
  constexpr int RING = (PRE + 2 < KS) ? PRE + 2 : KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* Xb = ldsb;                                       // [2][4][PPL][RSE] bf16
  float* oadjL = reinterpret_cast<float*>(ldsb + XI::BYTES);      // [4][COLS]
  float* sgacc = oadjL + 4 * COLS;                                // [sg_total]
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
  const int hi = COLS == 64 ? (col >> 4) : 0;
  const int pp = COLS == 64 ? (col & 15) : col;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ob = w * 32;
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
    f32x16 acc[NTL];
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
      for (int r = 0; r < 16; ++r) {
        const int o = ob + mfma_row(r, h);
        float w0 = wo[o], w1 = wo[HP + o], w2 = wo[2 * HP + o];
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[j][r] = w0 * oc[0][j] + w1 * oc[1][j] + w2 * oc[2][j];
      }
    }
    for (int l = L - 1; l >= 0; --l) {
      // W_l^T fragments: the first PRE k-steps are requested before this layer's Zb stores
      // (vmcnt retires in order), the rest stream through the register ring in the MFMA loop
      u32x4 wh[RING], wl[RING];
      const u32x4* wf = reinterpret_cast<const u32x4*>(P + prep_wtf(HP, l > 0 ? l : 1)) + (size_t)w * KS * 64 + lane;
      if (l > 0) {
#pragma unroll
        for (int s = 0; s < PRE; ++s) {
          wh[s] = wf[s * 64];
          if (TERMS == 3) wl[s] = wf[(size_t)(HP * HP / 8) + s * 64];
        }
      }
      asm volatile("" ::: "memory");
      const float* Sl = a.S + ((size_t)tile * L + l) * ((size_t)HP * COLS);
      float* Zl = a.Zb + ((size_t)tile * L + l) * ((size_t)HP * COLS);

      // residual mode: tanh adjoint of one register quad (features ob+8g+4h+e, column pp), all streams
      auto adj_quad = [&](int g, const f32x4& ga, const f32x4& gx, const f32x4& gy, const f32x4& gd) {
        // plane bases pinned to scalar registers + one 32-bit lane offset: one VALU per address instead of a
        // 64-bit add pair (every VALU instruction of the epilogue costs its full issue time)
        const unsigned so = (unsigned)(((ob >> 2) + 2 * g + h) * PPL + pp);
        constexpr size_t PLQ = (size_t)(HP / 4) * PPL;          // f32x4 per plane
        auto plane = [&](const float* base, int k) { return pin_base(reinterpret_cast<const f32x4*>(base) + k * PLQ); };
        f32x4 s0 = __builtin_nontemporal_load(plane(Sl, 0) + so), s1 = __builtin_nontemporal_load(plane(Sl, 1) + so);
        f32x4 s2 = __builtin_nontemporal_load(plane(Sl, 2) + so), s3 = __builtin_nontemporal_load(plane(Sl, 3) + so);
        f32x4 z0, z1, z2, z3;
        // branch-free chain for the four features (the layer-specific skinny-gradient terms follow below,
        // once per quad: a branch per element splits this into blocks the scheduler cannot pack)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = s0[e], zx = s1[e], zy = s2[e], zd = s3[e];
          float d1 = 1.f - t * t;
          float d2 = -2.f * t * d1;
          float d3 = -2.f * d1 * (1.f - 3.f * t * t);
          float zbx = d1 * gx[e] + 2.f * d2 * zx * gd[e];
          float zby = d1 * gy[e] + 2.f * d2 * zy * gd[e];
          float zbd = d1 * gd[e];
          float zb = d1 * ga[e] + d2 * (zx * gx[e] + zy * gy[e]) + (d3 * (zx * zx + zy * zy) + d2 * zd) * gd[e];
          z0[e] = zb; z1[e] = zbx; z2[e] = zby; z3[e] = zbd;
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
        // column sums of the four features at once; lane pp == e of each 16/32-lane group commits feature e
        {
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
          __builtin_nontemporal_store(z0, plane(Zl, 0) + so); __builtin_nontemporal_store(z1, plane(Zl, 1) + so);
          __builtin_nontemporal_store(z2, plane(Zl, 2) + so); __builtin_nontemporal_store(z3, plane(Zl, 3) + so);
        }
      };

      if (NS == 4 && COLS == 128) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 ga, gx, gy, gd;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            ga[e] = acc[0][4 * g + e]; gx[e] = acc[1][4 * g + e];
            gy[e] = acc[2 % NTL][4 * g + e]; gd[e] = acc[3 % NTL][4 * g + e];
          }
          adj_quad(g, ga, gx, gy, gd);
        }
      } else if (NS == 4) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][q]), __float_as_uint(acc[0][q + 8]), false, false);
          auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[1 % NTL][q]), __float_as_uint(acc[1 % NTL][q + 8]), false, false);
          acc[0][q] = __uint_as_float(s01[0]); acc[0][q + 8] = __uint_as_float(s01[1]);
          acc[1 % NTL][q] = __uint_as_float(s23[0]); acc[1 % NTL][q + 8] = __uint_as_float(s23[1]);
        }
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          f32x4 ga, gx, gy, gd;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int q = 4 * gq + e;
            ga[e] = acc[0][q]; gx[e] = acc[0][q + 8]; gy[e] = acc[1 % NTL][q]; gd[e] = acc[1 % NTL][q + 8];
          }
          adj_quad(gq + 2 * hi, ga, gx, gy, gd);
        }
      } else {
        // value mode: z-bar = (1 - t^2) * g per column; every accumulator tile is 1 or 2 planes
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 zj[NTL];
#pragma unroll
          for (int j = 0; j < NTL; ++j) {
            const int plane = COLS == 128 ? j : 2 * j + hi;
            const f32x4 t4 = *(reinterpret_cast<const f32x4*>(Sl) + ((size_t)plane * (HP / 4) + (ob >> 2) + 2 * g + h) * PPL + pp);
#pragma unroll
            for (int e = 0; e < 4; ++e) zj[j][e] = (1.f - t4[e] * t4[e]) * acc[j][4 * g + e];
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
      if (l == 0) break;
      __syncthreads();
      // ------------- G_{l-1}[i][col] = sum_o W_l[o][i] Zb_l[o][col]  (bf16 MFMA) -------------
      {
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
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
          if (s + PRE < KS) {
            wh[(s + PRE) % RING] = wf[(s + PRE) * 64];
            if (TERMS == 3) wl[(s + PRE) % RING] = wf[(size_t)(HP * HP / 8) + (s + PRE) * 64];
          }
          u32x4 nh[NTL], no[NTL];
          if (s + 1 < KS) {
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

template <int HP, int COLS>
static size_t lds_bytes_t(int L) { return XImg<HP, COLS / 4>::BYTES + ((size_t)4 * COLS + sg_total(HP, L)) * sizeof(float); }

size_t bwd_bf16_lds_bytes(int HP, int L, int cols) {
#define LB(hp) case hp: return cols == 64 ? lds_bytes_t<hp, 64>(L) : lds_bytes_t<hp, 128>(L);
  switch (HP) { LB(32) LB(64) LB(96) LB(128) LB(160) LB(192) LB(224) default: return cols == 64 ? lds_bytes_t<256, 64>(L) : lds_bytes_t<256, 128>(L); }
#undef LB
}

template <int HP, int NS, int TERMS, int COLS>
static int launch_one(const BwdArgs& a, int grid, hipStream_t s) {
  size_t lds = lds_bytes_t<HP, COLS>(a.L);
  if (a.configure) {   // pinn_plan_create: raise the kernel's dynamic-LDS limit on the current device
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_bf16_kernel<HP, NS, TERMS, COLS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PINN_LDS_MAX);
    return e == hipSuccess ? 0 : -(int)e;
  }
  hipLaunchKernelGGL((bwd_bf16_kernel<HP, NS, TERMS, COLS>), dim3(grid), dim3(HP * 2), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

template <int HP, int COLS>
static int launch_hp(int NS, int terms, const BwdArgs& a, int grid, hipStream_t s) {
  if (terms == 3) return NS == 4 ? launch_one<HP, 4, 3, COLS>(a, grid, s) : launch_one<HP, 1, 3, COLS>(a, grid, s);
  return NS == 4 ? launch_one<HP, 4, 1, COLS>(a, grid, s) : launch_one<HP, 1, 1, COLS>(a, grid, s);
}

int launch_bwd_bf16(int HP, int NS, int terms, int cols, const BwdArgs& a, int grid, hipStream_t s) {
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
