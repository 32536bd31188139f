// Do one wave's back-to-back MFMAs and its SIMD partner's VALU stream overlap on gfx950?
// 512-thread workgroup: waves 0-3 (one per SIMD) issue MFMAs, waves 4-7 (their partners) issue v_fma.
// mode 1: only MFMA waves work; 2: only VALU waves; 3: both.  Prints cycles per wave (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int KIND>   // 0: plain v_fma ; 1: mix with v_exp + cvt_pk + dpp
__global__ __launch_bounds__(512) void k(int mode, int iters, unsigned long long* out, float* sink, int swap, int prio) {
  const int w = threadIdx.x >> 6;
  const bool is_mfma = swap ? (w >= 4) : (w < 4);
  if (!is_mfma && prio == 1) __builtin_amdgcn_s_setprio(3);
  if (is_mfma && prio == 2) __builtin_amdgcn_s_setprio(3);
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  if (is_mfma) {
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
  } else {
    if (mode & 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (KIND == 0) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
            else {
              float e = __builtin_amdgcn_exp2f(v[i] * 0.001f);
              v[i] = __builtin_fmaf(e, 0.999f, v[(i + 1) & 15] * 0.5f);
            }
          }
      }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + w] = t1 - t0;
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int swap, int prio) {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 4096);
  const int iters = 2000;   // MFMA waves: 16 MFMAs / iter ; VALU waves: 96 (or 288) VALU / iter
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, mode, iters, d, sink, swap, prio);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long tm = swap ? h[4] : h[0], tv = swap ? h[0] : h[4];
    printf("%s swap %d prio %d mode %d: mfma wave %llu cycles (%.1f / MFMA)   valu wave %llu cycles\n", name, swap, prio, mode,
           tm, (double)tm / (iters * 16.0), tv);
  }
  hipFree(d); hipFree(sink);
}
int main() {
  for (int swap = 0; swap < 2; ++swap)
    for (int prio = 0; prio < 3; ++prio) run<1>("mix", swap, prio);
  return 0;
}
