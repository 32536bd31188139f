// Two waves on one SIMD (512-thread workgroup, waves w and w+4 are SIMD partners): wave A issues
// v_mfma_f32_32x32x16_bf16 back to back, optionally with PAD between them; wave B issues v_fma_f32.
// Does padding the MFMA wave's stream (s_nop / s_sleep / independent SALU) let the partner's VALU through?
// Prints cycles per MFMA for A and cycles per VALU for B, alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PAD>
__global__ __launch_bounds__(512) void k(int mode, int iters, unsigned long long* out, float* sink) {
  const int w = threadIdx.x >> 6;
  const bool is_mfma = w < 4;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f + 0.01f * i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  if (is_mfma) {
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
          if (PAD == 1) { asm volatile("s_nop 7\n s_nop 7\n s_nop 5"); }              // 22 idle scalar cycles
          if (PAD == 2) { asm volatile("s_sleep 1"); }                                 // yield ~64 cycles
          if (PAD == 3) { asm volatile("s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3"); }
          if (PAD == 4) { asm volatile("s_setprio 0\n s_nop 7\n s_nop 7\n s_nop 5\n s_setprio 3"); }
        }
      }
  } else {
    if (mode & 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u % 8]) : "v"(v[(u + 3) % 8]));
      }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + w] = t1 - t0;
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int PAD> void run(const char* name) {
  static unsigned long long* d = nullptr; static float* sink = nullptr;
  if (!d) { (void)hipMalloc(&d, 256 * 8 * 8); (void)hipMalloc(&sink, 4096); }
  const int iters = 1000;
  printf("%-44s", name);
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL(k<PAD>, dim3(256), dim3(512), 0, 0, mode, iters, d, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h[8]; (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    double m = (h[0] + h[1] + h[2] + h[3]) / 4.0, v = (h[4] + h[5] + h[6] + h[7]) / 4.0;
    if (mode == 1) printf(" alone: %5.1f cyc/MFMA |", m / (iters * 16.0));
    if (mode == 2) printf(" alone: %4.2f cyc/VALU |", v / (iters * 64.0));
    if (mode == 3) printf(" together: %5.1f cyc/MFMA, %5.2f cyc/VALU (= %.1f VALU per MFMA slot)", m / (iters * 16.0), v / (iters * 64.0),
                          (m / (iters * 16.0)) / (v / (iters * 64.0)));
  }
  printf("\n");
}
int main() {
  run<0>("MFMA back to back");
  run<1>("MFMA + s_nop x3 (22 cycles)");
  run<3>("MFMA + 5 x s_nop 3");
  run<2>("MFMA + s_sleep 1");
  run<4>("MFMA + setprio 0 / nops / setprio 3");
  return 0;
}
