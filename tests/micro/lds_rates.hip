// LDS service time of the access shapes the fused sweeps use, per CU: N waves issue the same instruction back to back
// (no dependent use until the end), cycles per instruction per CU = elapsed / (instructions issued by all waves).
//   b128 linear     : lane * 16                         (the ideal ds_read_b128)
//   b128 image      : XImg<256>::chunk_off(col, 2 s + h)  (B fragments of the G / M phases)
//   b64 write image : chunk_off(col, quad) + 8 h         (the epilogue's image writes)
//   b64 write linear: lane * 8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int chunk_off(int col, int ch) { return (col * 256 + ((ch ^ (col & 15)) << 3)) * 2; }

template <int OP>
__global__ __launch_bounds__(512) void k(int iters, unsigned long long* out, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  u32x4 acc = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      u32x4 x; u32x2 y = {(unsigned)it, (unsigned)u};
      if (OP == 0) { unsigned ad = ((u * 4 + w) & 63) * 1024 + lane * 16; asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(ad)); }
      if (OP == 1) { unsigned ad = (u & 3) * 16384 + chunk_off(col, 2 * ((u >> 2) + 4 * (w & 1)) + h); asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(ad)); }
      if (OP == 2) { unsigned ad = (u & 3) * 16384 + chunk_off(col, (u >> 2) + 4 * (w & 3)) + 8 * h; asm volatile("ds_write_b64 %0, %1" :: "v"(ad), "v"(y)); }
      if (OP == 3) { unsigned ad = ((u * 8 + w) & 127) * 512 + lane * 8; asm volatile("ds_write_b64 %0, %1" :: "v"(ad), "v"(y)); }
      if (OP == 4) { unsigned ad = ((u * 8 + w) & 127) * 512 + lane * 8; u32x2 z; asm volatile("ds_read_b64 %0, %1" : "=v"(z) : "v"(ad)); }
      if (OP == 5) { unsigned ad = (u & 3) * 16384 + chunk_off(col, 2 * ((u >> 2) + 4 * (w & 1)) + h); u32x2 z; asm volatile("ds_read_b64 %0, %1" : "=v"(z) : "v"(ad)); }
      if (OP == 6) { unsigned ad = (u & 3) * 16384 + chunk_off(col, (u >> 2) + 4 * (w & 3)) + 8 * h; unsigned z1 = it; asm volatile("ds_write_b32 %0, %1" :: "v"(ad), "v"(z1)); }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[threadIdx.x] = acc[0];
}

template <int OP> void run(const char* name) {
  static unsigned long long* d = nullptr; static unsigned* sink = nullptr;
  if (!d) { (void)hipMalloc(&d, 256 * 8 * 8); (void)hipMalloc(&sink, 4096); }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  printf("%-22s", name);
  for (int nw : {1, 4, 8}) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * nw), 131072, 0, iters, d, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h[8]; (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < nw; ++i) m += h[i]; m /= nw;
    printf("  %d waves: %6.2f cyc/inst/CU", nw, m / (iters * 16.0 * nw));
  }
  printf("\n");
}
int main() {
  run<0>("ds_read_b128 linear");
  run<1>("ds_read_b128 image");
  run<4>("ds_read_b64 linear");
  run<2>("ds_write_b64 image");
  run<3>("ds_write_b64 linear");
  run<5>("ds_read_b64 image");
  run<6>("ds_write_b32 image");
  return 0;
}
