// Within ONE wave: how many independent VALU ops fit in the shadow of each v_mfma_f32_32x32x16_bf16?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
template <int NV>
__global__ __launch_bounds__(256) void k(int iters, unsigned long long* out, float* sink) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i % 8]) : "v"(v[(i + 3) % 8]));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) sink[threadIdx.x] = s;
}
template <int NV> void run() {
  unsigned long long* d; float* sink; hipMalloc(&d, 256 * 8); hipMalloc(&sink, 4096);
  const int iters = 4000;
  hipLaunchKernelGGL(k<NV>, dim3(256), dim3(256), 0, 0, iters, d, sink);   // 4 waves: one per SIMD
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%2d v_fma per MFMA: %.1f cycles per MFMA\n", NV, (double)h / (iters * 4.0));
  hipFree(d); hipFree(sink);
}
int main() { run<0>(); run<2>(); run<4>(); run<5>(); run<6>(); run<7>(); run<8>(); run<12>(); run<16>(); return 0; }
