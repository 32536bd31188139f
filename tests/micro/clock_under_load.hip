// Shader clock under load: s_memtime (shader cycles) against s_memrealtime (constant 100 MHz) around a busy loop that
// runs MFMAs on waves 0-3, VALU on waves 4-7 (SIMD partners) and, optionally, streams nontemporal loads + stores
// through HBM from waves 4-7 - the three things a role-split sweep does at once.  Prints MHz per mix.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int mode, int iters, unsigned long long* out, float* sink, f32x4* gbuf, size_t per_wave) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (mode & 8) ? (__bf16)0.f : (__bf16)(1.0f + threadIdx.x * 1e-3f + 0.37f * i); b[i] = (mode & 8) ? (__bf16)0.f : (__bf16)(0.5f + 0.01f * i + threadIdx.x * 3e-3f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  f32x4* g = gbuf + ((size_t)blockIdx.x * 4 + (w & 3)) * per_wave + lane;
  __syncthreads();
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  if (w < 4) {
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
      }
  } else {
    for (int it = 0; it < iters; ++it) {
      if (mode & 2) {
#pragma unroll
        for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u % 8]) : "v"(v[(u + 3) % 8]));
      }
      if (mode & 4) {
        size_t o = ((size_t)it * 2 % (per_wave / 64)) * 64;
        f32x4 x = __builtin_nontemporal_load(g + o);
        x[0] += v[0];
        __builtin_nontemporal_store(x, g + o + 64);
      }
    }
  }
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 8 + w) * 2] = c1 - c0; out[(blockIdx.x * 8 + w) * 2 + 1] = r1 - r0; }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) sink[threadIdx.x] = s;
}

int main() {
  unsigned long long* d; float* sink; f32x4* gbuf;
  const size_t per_wave = 64 * 8192;      // f32x4 per wave: 8 MB
  (void)hipMalloc(&d, 256 * 8 * 16); (void)hipMalloc(&sink, 4096); (void)hipMalloc(&gbuf, (size_t)256 * 4 * per_wave * 16);
  const char* names[8] = {"idle", "MFMA", "VALU", "MFMA + VALU", "HBM stream", "MFMA + HBM stream", "VALU + HBM stream", "MFMA + VALU + HBM stream"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 1; mode < 8; ++mode) {
      const int iters = 40000;      // ~20-30 ms per launch: long enough for the power controller to settle
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, d, sink, gbuf, per_wave);
      (void)hipEventRecord(e1);
      (void)hipDeviceSynchronize();
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[16]; (void)hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
      const int wsel = (mode & 1) ? 0 : 4;
      double cyc = (double)h[wsel * 2], rt = (double)h[wsel * 2 + 1];
      printf("%-28s %8.2f ms  wave %d: %12.0f shader cycles in %10.0f x 10 ns  -> %7.1f MHz\n", names[mode], ms, wsel, cyc, rt, cyc / rt * 100.0);
    }
  // the same MFMA stream on all-zero operands: is the clock under MFMA load a power effect?
  for (int mode : {9, 11}) {
    const int iters = 40000;
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, d, sink, gbuf, per_wave);
    (void)hipDeviceSynchronize();
    unsigned long long h[16]; (void)hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    printf("%-28s wave 0: %7.1f MHz\n", mode == 9 ? "MFMA on zeros" : "MFMA on zeros + VALU", (double)h[0] / (double)h[1] * 100.0);
  }
  return 0;
}
