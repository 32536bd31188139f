// Small bandwidth-bound kernels around the MFMA pipeline: parameter re-layout,
// fixed-order gradient / loss reductions and the fused Adam update.
#include "kernels.h"

// ---------------------------------------------------------------------------
// flat state_dict-order parameters (NSFnet/net.py:36-46) -> padded, MFMA-fragment
// ordered copies (layout.h).  Wf: lane (i = lane&31, h = lane>>5) of wave w holds
// W[32w+i][2ks+h] for ks = 0..HP/2-1, four ks per 16-byte load.  WTf: same with W^T.
// ---------------------------------------------------------------------------
// fp32 -> bf16, round to nearest even, through the hardware conversion (v_cvt_pk_bf16_f32): unlike the integer
// (u + 0x7FFF + lsb) >> 16 idiom it keeps a NaN a NaN, so a NaN weight still poisons the loss in the bf16 modes
__device__ __forceinline__ unsigned short f2bf_rne(float x) {
  return __builtin_bit_cast(unsigned short, (__bf16)x);
}

// bf16 fragment element `pos` (0 .. HP*HP-1) of the hi (lo_part = 0) or lo (lo_part = 1) array:
// wave w, k-step s, lane (r = lane&31, h = lane>>5), element j holds W[32w+r][16s+8h+j] (or W^T).
__device__ __forceinline__ unsigned short bf16_frag_elem(const float* W, int H, int HP, bool tr, int pos, int lo_part) {
  int j = pos & 7, lane = (pos >> 3) & 63, ws = pos >> 9;
  int KS = HP / 16, s = ws % KS, w = ws / KS;
  int k = 16 * s + 8 * (lane >> 5) + j, o = 32 * w + (lane & 31);
  float v = 0.f;
  if (o < H && k < H) v = tr ? W[(size_t)k * H + o] : W[(size_t)o * H + k];
  unsigned short hi = f2bf_rne(v);
  if (!lo_part) return hi;
  return f2bf_rne(v - __uint_as_float(((unsigned)hi) << 16));
}

__global__ void prep_kernel(const float* __restrict__ params, float* __restrict__ prep,
                            int H, int HP, int L, int n_out, int prec_fwd, int prec_bwd) {
  const size_t total = prep_total(HP, L);
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (idx < (size_t)3 * HP) {
      int which = (int)(idx / HP), o = (int)(idx % HP);
      if (o < H) v = which == 0 ? params[flat_w(H, 0) + 2 * o] : which == 1 ? params[flat_w(H, 0) + 2 * o + 1]
                                                                            : params[flat_b(H, 0, L, n_out) + o];
    } else if (idx < prep_wout(HP, L)) {
      size_t rel = idx - (size_t)3 * HP;
      int l = 1 + (int)(rel / prep_layer_stride(HP));
      size_t q = rel % prep_layer_stride(HP);
      const float* W = params + flat_w(H, l);
      if (q < (size_t)2 * HP * HP) {
        bool tr = q >= (size_t)HP * HP;
        size_t f = tr ? q - (size_t)HP * HP : q;
        if ((tr ? prec_bwd : prec_fwd) != 0) {      // bf16 fragments: [hi HP*HP bf16][lo HP*HP bf16]
          int e0 = (int)(2 * f), half = HP * HP;
          unsigned short b0 = bf16_frag_elem(W, H, HP, tr, e0 % half, e0 / half);
          unsigned short b1 = bf16_frag_elem(W, H, HP, tr, (e0 + 1) % half, (e0 + 1) / half);
          prep[idx] = __uint_as_float((unsigned)b0 | ((unsigned)b1 << 16));
          continue;
        }
        int e = (int)(f & 3), lane = (int)((f >> 2) & 63);
        int wq = (int)(f >> 8);
        int qq = wq % (HP / 8), w = wq / (HP / 8);
        int k = 2 * (4 * qq + e) + (lane >> 5), o = 32 * w + (lane & 31);
        if (o < H && k < H) v = tr ? W[(size_t)k * H + o] : W[(size_t)o * H + k];
      } else {
        int o = (int)(q - (size_t)2 * HP * HP);
        if (o < H) v = params[flat_b(H, l, L, n_out) + o];
      }
    } else if (idx < prep_bout(HP, L)) {
      size_t q = idx - prep_wout(HP, L);
      int c = (int)(q / HP), k = (int)(q % HP);
      if (c < n_out && k < H) v = params[flat_w(H, L) + (size_t)c * H + k];
    } else {
      int c = (int)(idx - prep_bout(HP, L));
      if (c < n_out) v = params[flat_b(H, L, L, n_out) + c];
    }
    prep[idx] = v;
  }
}

int launch_prep(const float* params, float* prep, int H, int HP, int L, int n_out, int prec_fwd, int prec_bwd,
                hipStream_t s) {
  size_t total = prep_total(HP, L);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(prep_kernel, dim3(blocks), dim3(256), 0, s, params, prep, H, HP, L, n_out, prec_fwd, prec_bwd);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------
// gradient assembly: one thread per flat parameter sums that parameter's partials
// (dW slabs of dw.hip, per-workgroup skinny accumulators of bwd.hip) over all
// sources in a fixed order, in fp64.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void reduce_kernel(ReduceArgs a) {
  // 64 consecutive flat parameters per workgroup; wave w of 8 sums sources w, w+8, ... (4 loads in flight),
  // then the 8 wave sums are added in wave order: a fixed order for a given plan, whatever the launch.
  __shared__ double part[8][64];
  const int H = a.H, HP = a.HP, L = a.L;
  const size_t P = flat_total(H, L, a.n_out);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t p = blockIdx.x * (size_t)64 + lane;
  const bool live = p < P;
  int sgi = -1; int layer = -1; size_t slab_off = 0;
  if (live) {
    if (p < (size_t)2 * H) {
      int o = (int)(p / 2), j = (int)(p % 2);
      sgi = (j == 0 ? sg_w0x(HP, L) : sg_w0y(HP, L)) + o;
    } else if (p < (size_t)3 * H) {
      sgi = sg_db(HP, 0) + (int)(p - 2 * H);
    } else if (p < flat_w(H, L)) {
      size_t rel = p - (size_t)3 * H;
      size_t per = (size_t)H * H + H;
      int l = 1 + (int)(rel / per);
      size_t q = rel % per;
      if (q < (size_t)H * H) { layer = l; slab_off = (q / H) * HP + (q % H); }
      else sgi = sg_db(HP, l) + (int)(q - (size_t)H * H);
    } else {
      size_t q = p - flat_w(H, L);
      if (q < (size_t)a.n_out * H) sgi = sg_wout(HP, L) + (int)(q / H) * HP + (int)(q % H);
      else sgi = sg_bout(HP, L) + (int)(q - (size_t)a.n_out * H);
    }
  }
  double s = 0.0;
  if (live) {
    const size_t SG = sg_total(HP, L);
    for (int k = 0; k < a.nsrc; ++k) {
      const ReduceSrc& src = a.src[k];
      const float* base; size_t stride; int n;
      if (layer >= 0) { base = src.slabs + (size_t)(layer - 1) * src.groups * HP * HP + slab_off; stride = (size_t)HP * HP; n = src.groups; }
      else { base = src.sg + sgi; stride = SG; n = src.nwg; }
      int g = w;
      for (; g + 24 < n; g += 32) {
        float v0 = base[(size_t)g * stride], v1 = base[(size_t)(g + 8) * stride];
        float v2 = base[(size_t)(g + 16) * stride], v3 = base[(size_t)(g + 24) * stride];
        s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
      }
      for (; g < n; g += 8) s += (double)base[(size_t)g * stride];
    }
  }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && live) {
    double t = part[0][lane];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += part[i][lane];
    if (a.accumulate) a.grads[p] += (float)t; else a.grads[p] = (float)t;
  }
}

int launch_reduce(const ReduceArgs& a, hipStream_t s) {
  size_t P = flat_total(a.H, a.L, a.n_out);
  hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((P + 63) / 64)), dim3(512), 0, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------
// loss partial sums [nparts][PINN_NLOSS] -> out[PINN_NLOSS] (fp64 accumulate, fixed order:
// 32 interleaved sub-sums per slot, added in index order)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_sums_kernel(const float* __restrict__ partials, int nparts, float* __restrict__ out) {
  __shared__ double part[32][PINN_NLOSS];
  const int k = threadIdx.x % PINN_NLOSS, j = threadIdx.x / PINN_NLOSS;
  double s = 0.0;
  for (int i = j; i < nparts; i += 32) s += (double)partials[(size_t)i * PINN_NLOSS + k];
  part[j][k] = s;
  __syncthreads();
  if (threadIdx.x < PINN_NLOSS) {
    double t = part[0][k];
    for (int i = 1; i < 32; ++i) t += part[i][k];
    out[k] = (float)t;
  }
}

int launch_loss_sums(const float* partials, int nparts, float* out, hipStream_t s) {
  static_assert(PINN_NLOSS * 32 == 256, "loss_sums_kernel layout");
  hipLaunchKernelGGL(loss_sums_kernel, dim3(1), dim3(256), 0, s, partials, nparts, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------
// Adam, torch.optim.Adam single-tensor semantics (weight_decay = 0, amsgrad = False),
// as used at NSFnet/pinn_solver.py:76-79,253 and ev-NSFnet/pinn_solver.py:126-129,472:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// ---------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float step_size, float b1, float b2, float eps,
                            float bc2_sqrt) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i];
    float mi = m[i] + (gi - m[i]) * (1.f - b1);           // torch: exp_avg.lerp_(grad, 1-beta1)
    float vi = v[i] * b2 + (1.f - b2) * gi * gi;           // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi; v[i] = vi;
  }
}

// Same update with the step count kept ON THE DEVICE (bias corrections computed in-kernel), so the whole
// training step can be captured once in a hipGraph and replayed: no host scalar changes between steps.
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                long long* step_counter) {
  const double t = (double)(__atomic_load_n(step_counter, __ATOMIC_RELAXED) + 1);
  const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
  const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i];
    float mi = m[i] + (gi - m[i]) * (1.f - b1);
    float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi; v[i] = vi;
  }
  // the last workgroup to get here has seen every other one read the counter: it advances it
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(step_counter + 1);
    __threadfence();
    if (atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1) {
      *ticket = 0;
      step_counter[0] = step_counter[0] + 1;
    }
  }
}

int launch_adam_dev(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                    float eps, long long* step_counter, hipStream_t s) {
  if (n <= 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, step_counter);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int launch_adam(float* p, const float* g, float* m, float* v, long n, float step_size, float b1, float b2,
                float eps, float bc2_sqrt, hipStream_t s) {
  if (n <= 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, step_size, b1, b2, eps, bc2_sqrt);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
