// Hardware check of the operand layouts the bf16x3 kernels rely on (gfx950):
//  T1  v_mfma_f32_32x32x16_bf16: A lane (r=l&31,h=l>>5) = A[r][8h+j], B lane = B[8h+j][c=l&31], C standard map
//  T2  B operand fetched from an XOR-swizzled LDS image X[col][k] with ds_read_b128
//  T3  ds_read_b64_tr_b16: operand with K along LDS rows ([k][feature] image) for the dW GEMM
//  T4  hi/lo bf16 split, 3-term product accuracy
// Build: hipcc -O3 --offload-arch=gfx950 mfma_bf16_layout.hip -o mfma_bf16_layout ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f2bf(float x) {   // RNE
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

__device__ __forceinline__ int mrow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// C[32][32] = A[32][K] * B[K][32], K = 32.  mode 0: A,B frags straight from global.
// mode 1: B through swizzled LDS image X[col][k].  mode 2: A through tr reads of image [k][row].
__global__ void k_test(const float* A, const float* B, float* C, int mode, int split3) {
  __shared__ __attribute__((aligned(16))) unsigned short Xs[32 * 64];     // [col][k] K<=64, 128 B rows (swizzled 16B chunks)
  __shared__ __attribute__((aligned(16))) unsigned short Ts[32 * (32 + 32)]; // [k][feature + pad], row stride 64 elements = 128 B
  const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
  const int K = 32;
  // stage images
  for (int i = lane; i < 32 * K; i += 64) {
    int col = i / K, k = i % K;
    int chunk = k >> 3;                       // 16-byte chunk (8 bf16) within the row
    int sw = chunk ^ (col & 3);               // K=32 -> 4 chunks per row: swizzle on 2 bits for the test
    Xs[col * 64 + sw * 8 + (k & 7)] = f2bf(B[k * 32 + col]);
    int r = i / K, kk = i % K;                // A[r][kk] -> Ts[kk][r]
    Ts[kk * 64 + r] = f2bf(A[r * K + kk]);
  }
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int s = 0; s < K / 16; ++s) {
    s16x8 a, b, alo, blo;
    for (int j = 0; j < 8; ++j) {
      int k = 16 * s + 8 * h + j;
      float av = A[c * K + k], bv = B[k * 32 + c];
      unsigned short ah = f2bf(av), bh = f2bf(bv);
      a[j] = (short)ah; b[j] = (short)bh;
      alo[j] = (short)f2bf(av - bf2f(ah)); blo[j] = (short)f2bf(bv - bf2f(bh));
    }
    if (mode == 1) {
      int chunk = (16 * s + 8 * h) >> 3;
      b = *reinterpret_cast<const s16x8*>(&Xs[c * 64 + (chunk ^ (c & 3)) * 8]);
    }
    if (mode == 2) {
      // group g = lane>>4: feature block fb = g&1, k-half hh = g>>1 (== h); lane 4q+p of the group
      int li = lane & 15, g = lane >> 4, fb = g & 1, q = li >> 2, p = li & 3;
      const unsigned short* p0 = &Ts[(16 * s + 8 * h + q) * 64 + 16 * fb + 4 * p];
      const unsigned short* p1 = p0 + 4 * 64;
      s16x4 lo4, hi4;
      asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo4) : "v"((unsigned)(size_t)p0) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(hi4) : "v"((unsigned)(size_t)p1) : "memory");
      for (int j = 0; j < 4; ++j) { a[j] = lo4[j]; a[4 + j] = hi4[j]; }
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    if (split3) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, blo), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, alo), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) C[mrow(r, h) * 32 + c] = acc[r];
}

static float bfround(float x) {
  unsigned u; memcpy(&u, &x, 4); u += 0x7FFFu + ((u >> 16) & 1u); u &= 0xFFFF0000u; float y; memcpy(&y, &u, 4); return y;
}

int main() {
  const int K = 32;
  std::vector<float> A(32 * K), B(K * 32), C(32 * 32);
  srand(1);
  for (auto& v : A) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : B) v = (rand() % 2001 - 1000) / 777.f;
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  int fails = 0;
  for (int mode = 0; mode < 3; ++mode)
    for (int split3 = 0; split3 < 2; ++split3) {
      if (mode && split3) continue;
      hipMemset(dC, 0, C.size() * 4);
      hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dA, dB, dC, mode, split3);
      hipDeviceSynchronize();
      hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
      double maxerr = 0, maxref = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double ref = 0;
          for (int k = 0; k < K; ++k) {
            double a = split3 ? A[i * K + k] : bfround(A[i * K + k]);
            double b = split3 ? B[k * 32 + j] : bfround(B[k * 32 + j]);
            ref += a * b;
          }
          maxerr = fmax(maxerr, fabs(ref - C[i * 32 + j])); maxref = fmax(maxref, fabs(ref));
        }
      double tol = split3 ? 3e-5 : 2e-6;
      bool ok = maxerr <= tol * maxref;
      printf("mode %d split3 %d: max err %.3e (rel %.3e) %s\n", mode, split3, maxerr, maxerr / maxref, ok ? "PASS" : "FAIL");
      fails += !ok;
    }
  printf(fails ? "FAILED\n" : "ALL PASS\n");
  return fails;
}
