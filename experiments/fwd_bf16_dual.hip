// Software-pipelined bf16 forward: TWO 16-point half-tiles (A, B) in flight per workgroup.
//
// Algorithm / reference mapping: fwd.hip, fwd_bf16.hip (NSFnet/net.py:52-54,
// NSFnet/pinn_solver.py:132-163,197-226; ev-NSFnet/pinn_solver.py:290-342,372-428).
//
// Why: in the single-tile kernels all eight waves of the CU's only workgroup march through
// [tanh/chain-rule epilogue + activation spill] -> barrier -> [MFMA phase] -> barrier in lockstep,
// so the matrix pipe idles during the epilogue/spill and the VALU / store path idles during the MFMAs
// (PMC: matrix pipe busy 40 %).  Here every barrier interval holds the GEMM of one half-tile and the
// epilogue of the OTHER one, and the two waves of each SIMD run them in opposite order
// (waves 0..NW/2-1: GEMM then epilogue; waves NW/2..: epilogue then GEMM), so each SIMD always has
// one wave on the matrix pipe and one on VALU/LDS/stores:
//     I(2l-1): GEMM_A(l) || EPI_B(l-1)      I(2l): EPI_A(l) || GEMM_B(l)
// Half-tile geometry (64 columns = 16 points x 4 streams, v_permlane16_swap stream exchange) and the
// S layout are those of fwd_bf16_kernel<.., COLS = 64>, so the reverse sweep / dW kernels for
// 64-column tiles consume its output unchanged.
#include "kernels.h"
#include "bf16_util.h"

template <int HP, int NS, int TERMS>
__global__ __launch_bounds__(HP * 2) void fwd_bf16_dual_kernel(FwdArgs a) {
  constexpr int COLS = 64, PPL = 16, NTL = 2;
  using XI = XImg<HP, PPL>;
  constexpr int NW = HP / 32, NT = HP * 2, KS = HP / 16;
  constexpr int PRE = KS < 4 ? KS : 4, RING = (PRE + 2 < KS) ? PRE + 2 : KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* XA = ldsb;
  unsigned char* XB = ldsb + XI::BYTES;
  float* part = reinterpret_cast<float*>(ldsb + 2 * XI::BYTES);   // [NW][4][128]  (cols 0-63: A, 64-127: B)
  float* outv = part + NW * 4 * 128;                              // [4][128]
  float* biasL = outv + 4 * 128;                                  // [L][HP]
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
  const int hi = col >> 4, pp = col & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool gemm_first = w < (NW + 1) / 2;      // SIMD partners are waves w and w + NW/2
  const int ob = w * 32;
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * (NS == 4 ? PPL : COLS);
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < (a.L - 1) * HP; i += NT) biasL[HP + i] = a.prep[prep_b(HP, 1 + i / HP) + (i % HP)];
  __syncthreads();

  // ---- layer 0 (K = 2) into a half-tile's accumulators ----
  auto init0 = [&](f32x16 (&acc)[NTL], int tile) {
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
        acc[0][r] = hi ? wx : z; acc[1][r] = hi ? 0.f : wy;
      } else {
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[j][r] = fmaf(wx, px[j], fmaf(wy, py[j], b));
      }
    }
  };

  // ---- epilogue of layer l: tanh + chain rule on the accumulators, restage to X (bf16 hi/lo), spill S ----
  auto epilogue = [&](f32x16 (&acc)[NTL], unsigned char* Xb, int tile, int l) {
    float* Sl = (a.S && tile < a.ntiles) ? a.S + ((size_t)tile * L + l) * ((size_t)HP * COLS) : nullptr;
    if (NS == 4) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][q]), __float_as_uint(acc[0][q + 8]), false, false);
        auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[1][q]), __float_as_uint(acc[1][q + 8]), false, false);
        acc[0][q] = __uint_as_float(s01[0]); acc[0][q + 8] = __uint_as_float(s01[1]);
        acc[1][q] = __uint_as_float(s23[0]); acc[1][q + 8] = __uint_as_float(s23[1]);
      }
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        const int g = gq + 2 * hi;
        f32x4 a0, a1, a2, a3, s0, s1, s2, s3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = 4 * gq + e;
          float t = fast_tanh(acc[0][q]);
          float zx = acc[0][q + 8], zy = acc[1][q], zd = acc[1][q + 8];
          float d1 = 1.f - t * t;
          float d2 = -2.f * t * d1;
          a0[e] = t; a1[e] = d1 * zx; a2[e] = d1 * zy; a3[e] = d2 * (zx * zx + zy * zy) + d1 * zd;
          s0[e] = t; s1[e] = zx; s2[e] = zy; s3[e] = zd;
        }
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
          f32x4* Sg = reinterpret_cast<f32x4*>(Sl) + (size_t)((ob >> 2) + 2 * g + h) * PPL + pp;
          Sg[0 * (HP / 4) * PPL] = s0;
          Sg[1 * (HP / 4) * PPL] = s1;
          Sg[2 * (HP / 4) * PPL] = s2;
          Sg[3 * (HP / 4) * PPL] = s3;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        const int plane = 2 * j + hi;
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
  };

  // ---- weight fragments of layer l: the first PRE k-steps.  Requested BEFORE an epilogue's S stores in
  //      program order (vmcnt retires in order: a load behind the store burst waits for its drain) ----
  u32x4 wh[RING], wl[RING];
  auto wprefetch = [&](int l) {
    if (l >= L) return;
    const u32x4* wf = reinterpret_cast<const u32x4*>(P + prep_wf(HP, l)) + (size_t)w * KS * 64 + lane;
#pragma unroll
    for (int s = 0; s < PRE; ++s) {
      wh[s] = wf[s * 64];
      if (TERMS == 3) wl[s] = wf[(size_t)(HP * HP / 8) + s * 64];
    }
    asm volatile("" ::: "memory");
  };

  // ---- hidden GEMM of layer l (1 <= l <= L-1) for one half-tile; ring slots 0..PRE-1 already requested ----
  auto gemm = [&](f32x16 (&acc)[NTL], const unsigned char* Xb, int l) {
    const u32x4* wf = reinterpret_cast<const u32x4*>(P + prep_wf(HP, l)) + (size_t)w * KS * 64 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float b = biasL[l * HP + ob + mfma_row(r, h)];
      if (NS == 4) { acc[0][r] = hi ? 0.f : b; acc[1][r] = 0.f; }
      else { acc[0][r] = b; acc[1][r] = b; }
    }
    const unsigned char* Xl = Xb + hi * XI::PLANE * 2;
    constexpr int TSTR = 2 * XI::PLANE * 2;
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
  };

  const int npairs = (a.ntiles + 1) / 2;
  for (int tp = blockIdx.x; tp < npairs; tp += gridDim.x) {
    const int tileA = 2 * tp, tileB = 2 * tp + 1;
    f32x16 accA[NTL], accB[NTL];
    init0(accA, tileA);
    init0(accB, tileB);
    if (gemm_first) wprefetch(1);
    epilogue(accA, XA, tileA, 0);
    __syncthreads();
    for (int l = 1; l < L; ++l) {
      // interval 2l-1:  GEMM_A(l) || EPI_B(l-1)
      if (gemm_first) { gemm(accA, XA, l); wprefetch(l); epilogue(accB, XB, tileB, l - 1); }
      else            { wprefetch(l); epilogue(accB, XB, tileB, l - 1); gemm(accA, XA, l); }
      __syncthreads();
      // interval 2l:    EPI_A(l) || GEMM_B(l)
      if (gemm_first) { gemm(accB, XB, l); wprefetch(l + 1); epilogue(accA, XA, tileA, l); }
      else            { wprefetch(l); epilogue(accA, XA, tileA, l); gemm(accB, XB, l); }
      __syncthreads();
    }
    epilogue(accB, XB, tileB, L - 1);
    __syncthreads();
    // ---------------- output layer for both half-tiles: VALU, K split over waves ----------------
    {
      float po[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      const float* wo = P + prep_wout(HP, L) + ob;
      const int plane = lane / PPL, cp = lane % PPL;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const unsigned char* Xb = half ? XB : XA;
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
      for (int c3 = 0; c3 < 3; ++c3) {
        part[(w * 4 + c3) * 128 + lane] = po[c3][0];
        part[(w * 4 + c3) * 128 + 64 + lane] = po[c3][1];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 3 * 128; idx += NT) {
      int c3 = idx >> 7, cc = idx & 127;
      float s = (NS == 1 || (cc & 63) < PPL) ? P[prep_bout(HP, L) + c3] : 0.f;
      for (int ww = 0; ww < NW; ++ww) s += part[(ww * 4 + c3) * 128 + cc];
      outv[c3 * 128 + cc] = s;
    }
    __syncthreads();
    // ---------------- per-point stage ----------------
    if (NS == 4) {
      if (tid < 2 * PPL) {
        const int half = tid >> 4, q = tid & 15;
        const int tile = 2 * tp + half;
        const int pt = tile * PPL + q;
        const bool m = pt < a.n;
        const float* ov = outv + 64 * half;
        const float sc = a.scale, sc2 = a.scale * a.scale;
        float u = ov[q], ux = ov[PPL + q] * sc, uy = ov[2 * PPL + q] * sc, ud = ov[3 * PPL + q] * sc2;
        float v = ov[128 + q], vx = ov[128 + PPL + q] * sc, vy = ov[128 + 2 * PPL + q] * sc, vd = ov[128 + 3 * PPL + q] * sc2;
        float p = ov[256 + q], pxx = ov[256 + PPL + q] * sc, pyy = ov[256 + 2 * PPL + q] * sc;
        float vt = 0.f;
        float ev = (a.e && m) ? a.e[pt] : 0.f;
        if (a.vtm && m) {
          vt = fminf(a.vis_t0, a.vtm[pt]);
          a.vtm[pt] = a.alpha_evm * fabsf(ev);
        }
        if (a.vis_used && m) a.vis_used[pt] = vt;
        float nu = a.inv_re + vt;
        float eq1 = (u * ux + v * uy) + pxx - nu * ud;
        float eq2 = (u * vx + v * vy) + pyy - nu * vd;
        float eq3 = ux + vy;
        float eq4 = a.e ? (eq1 * (u - 0.5f) + eq2 * (v - 0.5f)) - ev : 0.f;
        if (tile < a.ntiles) {
          float* f = a.fld + pt;
          f[FLD_U * (size_t)npad] = u; f[FLD_V * (size_t)npad] = v;
          f[FLD_UX * (size_t)npad] = ux; f[FLD_UY * (size_t)npad] = uy;
          f[FLD_VX * (size_t)npad] = vx; f[FLD_VY * (size_t)npad] = vy;
          f[FLD_EQ1 * (size_t)npad] = eq1; f[FLD_EQ2 * (size_t)npad] = eq2;
          f[FLD_EQ3 * (size_t)npad] = eq3; f[FLD_EQ4 * (size_t)npad] = eq4;
          f[FLD_P * (size_t)npad] = p;
        }
        if (m) {
          float ww = a.w ? a.w[pt] : 1.f;
          lsum[0] += ww * eq1 * eq1; lsum[1] += ww * eq2 * eq2;
          lsum[2] += ww * eq3 * eq3; lsum[3] += ww * eq4 * eq4;
        }
      }
    } else {
      for (int idx = tid; idx < 128; idx += NT) {
        const int half = idx >> 6, q = idx & 63;
        const int tile = 2 * tp + half;
        const int pt = tile * COLS + q;
        const bool m = pt < a.n;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (c >= a.n_out) break;
          float pv = outv[c * 128 + idx];
          if (a.pred[c] && m) a.pred[c][pt] = pv;
          float adj = 0.f;
          if (a.tgt[c] && m) {
            float t = a.tgt[c][pt];
            if (t == t && fabsf(t) <= 3.0e38f) {
              float d = pv - t;
              lsum[c] += d * d;
              lsum[3] += (c == 2) ? 1.f : 0.f;
              adj = a.coef[c] * d;
            }
          }
          if (a.oadj && tile < a.ntiles) a.oadj[(size_t)c * npad + pt] = adj;
        }
      }
    }
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
  return 2 * XImg<HP, 16>::BYTES + ((size_t)(HP / 32) * 4 * 128 + 4 * 128 + (size_t)L * HP) * sizeof(float);
}

size_t fwd_bf16_dual_lds_bytes(int HP, int L) { return HP == 128 ? lds_bytes_t<128>(L) : lds_bytes_t<256>(L); }

template <int HP, int NS, int TERMS>
static int launch_one(const FwdArgs& a, int grid, hipStream_t s) {
  size_t lds = lds_bytes_t<HP>(a.L);
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_bf16_dual_kernel<HP, NS, TERMS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
    attr_lds = lds;
  }
  hipLaunchKernelGGL((fwd_bf16_dual_kernel<HP, NS, TERMS>), dim3(grid), dim3(HP * 2), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

template <int HP>
static int launch_hp(int NS, int terms, const FwdArgs& a, int grid, hipStream_t s) {
  if (terms == 3) return NS == 4 ? launch_one<HP, 4, 3>(a, grid, s) : launch_one<HP, 1, 3>(a, grid, s);
  return NS == 4 ? launch_one<HP, 4, 1>(a, grid, s) : launch_one<HP, 1, 1>(a, grid, s);
}

int launch_fwd_bf16_dual(int HP, int NS, int terms, const FwdArgs& a, int grid, hipStream_t s) {
  switch (HP) {
    case 128: return launch_hp<128>(NS, terms, a, grid, s);
    case 256: return launch_hp<256>(NS, terms, a, grid, s);
    default: return -1000;
  }
}
