// Issue cost of the VALU instructions the tanh epilogues are made of, for ONE wave on a SIMD (alone) and beside a SIMD
// partner that issues v_mfma_f32_32x32x16_bf16 back to back (the role-split kernels' situation).  Is packed fp32
// (v_pk_fma_f32: two lanes' worth of work per instruction) cheaper per element than scalar v_fma_f32 there?
// Prints cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(512) void k(int mode, int iters, unsigned long long* out, float* sink) {
  const int w = threadIdx.x >> 6;
  const bool is_mfma = w < 4;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f + 0.01f * i); }
  float v[8];
  f32x2 p[8];
  unsigned u32[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{v[i], v[i] + 0.5f}; u32[i] = threadIdx.x + i; }
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  if (is_mfma) {
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
      }
  } else {
    if (mode & 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
          if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u % 8]) : "v"(v[(u + 3) % 8]));
          if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[u % 8]) : "v"(p[(u + 3) % 8]));
          if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[u % 8]) : "v"(p[(u + 3) % 8]));
          if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[u % 8]) : "v"(p[(u + 3) % 8]));
          if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u32[u % 8]) : "v"(v[(u + 3) % 8]), "v"(v[(u + 5) % 8]));
          if (OP == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[u % 8]) : "v"(v[(u + 3) % 8]));
          if (OP == 6) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u32[u % 8]) : "v"(u32[(u + 3) % 8]));
          if (OP == 7) asm volatile("v_exp_f32 %0, %1" : "=v"(v[u % 8]) : "v"(v[(u + 3) % 8]));
          if (OP == 8) asm volatile("v_rcp_f32 %0, %1" : "=v"(v[u % 8]) : "v"(v[(u + 3) % 8]));
          if (OP == 9) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[0]));      // fully dependent chain
          if (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[0]));   // fully dependent chain
          if (OP == 11) asm volatile("v_lshl_add_u64 %0, %1, 4, %0" : "+v"(p[u % 8]) : "v"(p[(u + 3) % 8]));
        }
      }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + w] = t1 - t0;
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1] + (float)u32[i];
  if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int OP> void run(const char* name) {
  static unsigned long long* d = nullptr; static float* sink = nullptr;
  if (!d) { (void)hipMalloc(&d, 256 * 8 * 8); (void)hipMalloc(&sink, 4096); }
  const int iters = 1000;
  printf("%-34s", name);
  for (int mode = 2; mode <= 3; ++mode) {
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(512), 0, 0, mode, iters, d, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h[8]; (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    double m = (h[0] + h[1] + h[2] + h[3]) / 4.0, v = (h[4] + h[5] + h[6] + h[7]) / 4.0;
    if (mode == 2) printf(" alone: %5.2f cyc/inst |", v / (iters * 64.0));
    if (mode == 3) printf(" beside MFMA partner: %5.2f cyc/inst (MFMA %5.1f cyc)", v / (iters * 64.0), m / (iters * 16.0));
  }
  printf("\n");
}
int main() {
  run<0>("v_fma_f32");
  run<1>("v_pk_fma_f32");
  run<2>("v_pk_mul_f32");
  run<3>("v_pk_add_f32");
  run<5>("v_mul_f32");
  run<4>("v_cvt_pk_bf16_f32");
  run<6>("v_lshlrev_b32");
  run<11>("v_lshl_add_u64");
  run<7>("v_exp_f32");
  run<8>("v_rcp_f32");
  run<9>("v_fma_f32 dependent chain");
  run<10>("v_pk_fma_f32 dependent chain");
  return 0;
}
