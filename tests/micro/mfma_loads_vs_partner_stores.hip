// Does a SIMD partner's stream of nontemporal stores (HBM-saturating: every CU does it) stall a wave whose MFMAs are fed
// by L2-resident global loads (the weight fragments of the M / G phases: 4 x global_load_dwordx4 per 24 MFMAs,
// requested one k-step ahead)?  Waves 0-3: MFMA (+ loads), waves 4-7: stores.  Prints cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int LOADS, int GAP, int DEPTH = 1>
__global__ __launch_bounds__(512) void k(int mode, int iters, unsigned long long* out, float* sink, const u32x4* wts, f32x4* gbuf, size_t per_wave) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8_t b;
  for (int i = 0; i < 8; ++i) b[i] = (__bf16)(0.5f + 0.01f * i);
  f32x4 gv = {1.f, 2.f, 3.f, (float)threadIdx.x};
  f32x4* g = gbuf + ((size_t)blockIdx.x * 4 + (w & 3)) * per_wave + lane;
  const u32x4* wl = wts + (size_t)(w & 3) * 64 * 64 + lane;
  constexpr int R = DEPTH + 1;
  u32x4 ring[R][4];
  for (int r = 0; r < R; ++r) for (int i = 0; i < 4; ++i) ring[r][i] = wl[i * 64];
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  if (w < 4) {
    if (mode & 1)
      for (int it = 0; it < iters; it += R) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
          if (LOADS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ring[(rr + DEPTH) % R][i] = __builtin_nontemporal_load(wl + (((it + rr) * 4 + i) & 63) * 64);
          }
#pragma unroll
          for (int j = 0; j < 24; ++j)
            acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ring[rr][j & 3]), b, acc[j & 7], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  } else {
    if (mode & 2)
      for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(gv, g + (size_t)(((size_t)it * 4 + u) % (per_wave / 64)) * 64);
        if (GAP) __builtin_amdgcn_s_sleep(GAP);
      }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + w] = t1 - t0;
  float s = 0.f;
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int LOADS, int GAP, int DEPTH = 1> void run(const char* name) {
  static unsigned long long* d = nullptr; static float* sink = nullptr; static f32x4* gbuf = nullptr; static u32x4* wts = nullptr;
  const size_t per_wave = 64 * 16384;
  if (!d) {
    (void)hipMalloc(&d, 256 * 8 * 8); (void)hipMalloc(&sink, 4096); (void)hipMalloc(&gbuf, (size_t)256 * 4 * per_wave * 16);
    (void)hipMalloc(&wts, 4 * 64 * 64 * 16); (void)hipMemset(wts, 0x3c, 4 * 64 * 64 * 16);
  }
  const int iters = 3024;
  printf("%-52s", name);
  for (int mode = 1; mode <= 3; mode += 2) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<LOADS, GAP, DEPTH>), dim3(256), dim3(512), 0, 0, mode, iters, d, sink, wts, gbuf, per_wave);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    double m = (h[0] + h[1] + h[2] + h[3]) / 4.0, st = (h[4] + h[5] + h[6] + h[7]) / 4.0;
    if (mode == 1) printf(" alone: %5.1f cyc/MFMA |", m / (iters * 24.0));
    else printf(" with partner stores: %5.1f cyc/MFMA  (stores: %.0f cyc per 1 KB store, %.2f TB/s device-wide)", m / (iters * 24.0),
                st / (iters * 8.0), 256.0 * 4 * iters * 8 * 1024 / (ms * 1e-3) / 1e12);
  }
  printf("\n");
}
int main() {
  run<0, 0>("MFMA only (operands in registers), stores flat out");
  run<1, 0>("MFMA + 4 loads / 24 MFMAs, stores flat out");
  run<1, 0, 2>("... loads 2 k-steps ahead");
  run<1, 0, 3>("... loads 3 k-steps ahead");
  run<1, 0, 5>("... loads 5 k-steps ahead");
  run<1, 0, 7>("... loads 7 k-steps ahead");
  run<1, 16, 1>("1 k-step ahead, stores + s_sleep 16");
  run<1, 16, 5>("5 k-steps ahead, stores + s_sleep 16");
  return 0;
}
