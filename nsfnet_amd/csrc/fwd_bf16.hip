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
#include "bf16_util.h"

template <int HP, int NS, int TERMS>
__global__ __launch_bounds__(HP * 2) void fwd_bf16_kernel(FwdArgs a) {
  using XI = XImg<HP>;
  constexpr int NW = HP / 32, NT = HP * 2, KS = HP / 16;
  constexpr int PRE = KS < 4 ? KS : 4, RING = (PRE + 2 < KS) ? PRE + 2 : KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* Xb = ldsb;                                   // [2][4][32][RSE] bf16
  float* part = reinterpret_cast<float*>(ldsb + XI::BYTES);   // [NW][4][128]
  float* outv = part + NW * 4 * 128;                          // [4][128]
  float* biasL = outv + 4 * 128;                              // [L][HP] hidden-layer biases (l >= 1)
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ob = w * 32;
  const float* __restrict__ P = a.prep;
  const int L = a.L;
  const int npad = a.ntiles * (NS == 4 ? 32 : 128);
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
    f32x16 acc[4];
    {
      float px[4], py[4];
      if (NS == 4) {
        int pt = tile * 32 + col;
        px[0] = pt < a.n ? a.x[pt] : 0.f;
        py[0] = pt < a.n ? a.y[pt] : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int pt = tile * 128 + 32 * j + col;
          px[j] = pt < a.n ? a.x[pt] : 0.f;
          py[j] = pt < a.n ? a.y[pt] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int o = ob + mfma_row(r, h);
        float wx = P[prep_w0x(HP) + o], wy = P[prep_w0y(HP) + o], b = P[prep_b0(HP) + o];
        if (NS == 4) {
          acc[0][r] = fmaf(wx, px[0], fmaf(wy, py[0], b));
          acc[1][r] = wx; acc[2][r] = wy; acc[3][r] = 0.f;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j][r] = fmaf(wx, px[j], fmaf(wy, py[j], b));
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
      float* Sl = a.S ? a.S + ((size_t)tile * L + l) * act_block(HP) : nullptr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 s0, s1, s2, s3;      // what is saved (t, z_x, z_y, z_D | t_j)
        f32x4 a0, a1, a2, a3;      // what the next layer consumes (a, a_x, a_y, a_D | t_j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          if (NS == 4) {
            float t = fast_tanh(acc[0][r]);
            float zx = acc[1][r], zy = acc[2][r], zd = acc[3][r];
            float d1 = 1.f - t * t;
            float d2 = -2.f * t * d1;
            a0[e] = t; a1[e] = d1 * zx; a2[e] = d1 * zy; a3[e] = d2 * (zx * zx + zy * zy) + d1 * zd;
            s0[e] = t; s1[e] = zx; s2[e] = zy; s3[e] = zd;
          } else {
            a0[e] = s0[e] = fast_tanh(acc[0][r]); a1[e] = s1[e] = fast_tanh(acc[1][r]);
            a2[e] = s2[e] = fast_tanh(acc[2][r]); a3[e] = s3[e] = fast_tanh(acc[3][r]);
          }
        }
        // restage as bf16 hi/lo: 4 consecutive features = 8 bytes at [col][chunk (ob/8+g)] + 8h
        {
          const int off = XI::chunk_off(col, (ob >> 3) + g) + 8 * h;
          u32x2 hi, lo;
          split4(a0[0], a0[1], a0[2], a0[3], hi, lo);
          *reinterpret_cast<u32x2*>(Xb + 0 * XI::PLANE * 2 + off) = hi;
          if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + 0 * XI::PLANE * 2 + off) = lo;
          split4(a1[0], a1[1], a1[2], a1[3], hi, lo);
          *reinterpret_cast<u32x2*>(Xb + 1 * XI::PLANE * 2 + off) = hi;
          if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + 1 * XI::PLANE * 2 + off) = lo;
          split4(a2[0], a2[1], a2[2], a2[3], hi, lo);
          *reinterpret_cast<u32x2*>(Xb + 2 * XI::PLANE * 2 + off) = hi;
          if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + 2 * XI::PLANE * 2 + off) = lo;
          split4(a3[0], a3[1], a3[2], a3[3], hi, lo);
          *reinterpret_cast<u32x2*>(Xb + 3 * XI::PLANE * 2 + off) = hi;
          if (TERMS == 3) *reinterpret_cast<u32x2*>(Xb + XI::HALF * 2 + 3 * XI::PLANE * 2 + off) = lo;
        }
        if (Sl) {
          f32x4* Sg = reinterpret_cast<f32x4*>(Sl) + (size_t)((ob >> 2) + 2 * g + h) * 32 + col;
          Sg[0 * (HP / 4) * 32] = s0;
          Sg[1 * (HP / 4) * 32] = s1;
          Sg[2 * (HP / 4) * 32] = s2;
          Sg[3 * (HP / 4) * 32] = s3;
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the four register quads sequential (VGPR budget)
      }
      __syncthreads();
      if (l == L - 1) break;
      // ------------- hidden GEMM l+1 on bf16 MFMA -------------
      {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float b = biasL[(l + 1) * HP + ob + mfma_row(r, h)];
          acc[0][r] = b;
          if (NS == 4) { acc[1][r] = 0.f; acc[2][r] = 0.f; acc[3][r] = 0.f; }
          else { acc[1][r] = b; acc[2][r] = b; acc[3][r] = b; }
        }
        u32x4 bh[4], bo[4];
        {
          const int off0 = XI::chunk_off(col, h);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bh[j] = *reinterpret_cast<const u32x4*>(Xb + j * XI::PLANE * 2 + off0);
            if (TERMS == 3) bo[j] = *reinterpret_cast<const u32x4*>(Xb + XI::HALF * 2 + j * XI::PLANE * 2 + off0);
          }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          if (s + PRE < KS) {     // stream the weight fragments PRE k-steps ahead
            wh[(s + PRE) % RING] = wf[(s + PRE) * 64];
            if (TERMS == 3) wl[(s + PRE) % RING] = wf[(size_t)(HP * HP / 8) + (s + PRE) * 64];
          }
          u32x4 nh[4], no[4];
          if (s + 1 < KS) {       // next k-step's B fragments in flight during this step's MFMAs
            const int off = XI::chunk_off(col, 2 * (s + 1) + h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              nh[j] = *reinterpret_cast<const u32x4*>(Xb + j * XI::PLANE * 2 + off);
              if (TERMS == 3) no[j] = *reinterpret_cast<const u32x4*>(Xb + XI::HALF * 2 + j * XI::PLANE * 2 + off);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (TERMS == 3) {
              acc[j] = mfma_bf16(wh[s % RING], bo[j], acc[j]);
              acc[j] = mfma_bf16(wl[s % RING], bh[j], acc[j]);
            }
            acc[j] = mfma_bf16(wh[s % RING], bh[j], acc[j]);
          }
          if (s + 1 < KS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { bh[j] = nh[j]; if (TERMS == 3) bo[j] = no[j]; }
          }
        }
      }
      __syncthreads();
    }
    // ---------------- output layer: VALU, K split over waves ----------------
    {
      float po[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      const float* wo = P + prep_wout(HP, L) + ob;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int c2 = lane + 64 * half, j = c2 >> 5, c = c2 & 31;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          const int off = XI::chunk_off(c, 4 * w + ch);
          u32x4 vh = *reinterpret_cast<const u32x4*>(Xb + j * XI::PLANE * 2 + off);
          u32x4 vl = {0u, 0u, 0u, 0u};
          if (TERMS == 3) vl = *reinterpret_cast<const u32x4*>(Xb + XI::HALF * 2 + j * XI::PLANE * 2 + off);
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
      for (int c = 0; c < 3; ++c) {
        part[(w * 4 + c) * 128 + lane] = po[c][0];
        part[(w * 4 + c) * 128 + 64 + lane] = po[c][1];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 3 * 128; idx += NT) {
      int c = idx >> 7, cc = idx & 127;
      float s = (NS == 1 || cc < 32) ? P[prep_bout(HP, L) + c] : 0.f;
      for (int ww = 0; ww < NW; ++ww) s += part[(ww * 4 + c) * 128 + cc];
      outv[c * 128 + cc] = s;
    }
    __syncthreads();
    // ---------------- per-point stage (identical to fwd.hip) ----------------
    if (NS == 4) {
      if (tid < 32) {
        const int pt = tile * 32 + tid;
        const bool m = pt < a.n;
        const float sc = a.scale, sc2 = a.scale * a.scale;
        float u = outv[tid], ux = outv[32 + tid] * sc, uy = outv[64 + tid] * sc, ud = outv[96 + tid] * sc2;
        float v = outv[128 + tid], vx = outv[160 + tid] * sc, vy = outv[192 + tid] * sc, vd = outv[224 + tid] * sc2;
        float p = outv[256 + tid], pxx = outv[288 + tid] * sc, pyy = outv[320 + tid] * sc;
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
        float* f = a.fld + pt;
        f[FLD_U * (size_t)npad] = u; f[FLD_V * (size_t)npad] = v;
        f[FLD_UX * (size_t)npad] = ux; f[FLD_UY * (size_t)npad] = uy;
        f[FLD_VX * (size_t)npad] = vx; f[FLD_VY * (size_t)npad] = vy;
        f[FLD_EQ1 * (size_t)npad] = eq1; f[FLD_EQ2 * (size_t)npad] = eq2;
        f[FLD_EQ3 * (size_t)npad] = eq3; f[FLD_EQ4 * (size_t)npad] = eq4;
        f[FLD_P * (size_t)npad] = p;
        if (m) {
          float ww = a.w ? a.w[pt] : 1.f;
          lsum[0] += ww * eq1 * eq1; lsum[1] += ww * eq2 * eq2;
          lsum[2] += ww * eq3 * eq3; lsum[3] += ww * eq4 * eq4;
        }
      }
    } else {
      for (int idx = tid; idx < 128; idx += NT) {
        const int pt = tile * 128 + idx;
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
          if (a.oadj) a.oadj[(size_t)c * npad + pt] = adj;
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
static size_t lds_bytes_t(int L) { return XImg<HP>::BYTES + ((size_t)(HP / 32) * 4 * 128 + 4 * 128 + (size_t)L * HP) * sizeof(float); }

size_t fwd_bf16_lds_bytes(int HP, int L) {
  switch (HP) {
    case 32: return lds_bytes_t<32>(L); case 64: return lds_bytes_t<64>(L); case 96: return lds_bytes_t<96>(L);
    case 128: return lds_bytes_t<128>(L); case 160: return lds_bytes_t<160>(L); case 192: return lds_bytes_t<192>(L);
    case 224: return lds_bytes_t<224>(L); default: return lds_bytes_t<256>(L);
  }
}

template <int HP, int NS, int TERMS>
static int launch_one(const FwdArgs& a, int grid, hipStream_t s) {
  size_t lds = lds_bytes_t<HP>(a.L);
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_bf16_kernel<HP, NS, TERMS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
    attr_lds = lds;
  }
  hipLaunchKernelGGL((fwd_bf16_kernel<HP, NS, TERMS>), dim3(grid), dim3(HP * 2), lds, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

#define FWD_CASE(hp)                                                                         \
  case hp:                                                                                   \
    if (terms == 3) return NS == 4 ? launch_one<hp, 4, 3>(a, grid, s) : launch_one<hp, 1, 3>(a, grid, s); \
    return NS == 4 ? launch_one<hp, 4, 1>(a, grid, s) : launch_one<hp, 1, 1>(a, grid, s);

int launch_fwd_bf16(int HP, int NS, int terms, const FwdArgs& a, int grid, hipStream_t s) {
  switch (HP) {
    FWD_CASE(32) FWD_CASE(64) FWD_CASE(96) FWD_CASE(128)
    FWD_CASE(160) FWD_CASE(192) FWD_CASE(224) FWD_CASE(256)
    default: return -1000;
  }
}
