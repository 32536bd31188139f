// What in a SIMD partner's instruction stream slows a wave that issues v_mfma_f32_32x32x16_bf16 back to back?
// 512-thread workgroups (waves w and w+4 share a SIMD): waves 0-3 run MFMAs (optionally fed by ds_read_b128 like the
// G phase of the role-split kernels), waves 4-7 run one kind of epilogue instruction.  Prints cycles per MFMA and
// cycles per partner instruction, alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int OP, int FEED, int CHAIN = 1>
__global__ __launch_bounds__(512) void k(int mode, int iters, unsigned long long* out, float* sink, f32x4* gbuf) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool is_mfma = w < 4;
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f + 0.01f * i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  for (int i = threadIdx.x; i < 32768; i += 512) reinterpret_cast<float*>(lds)[i] = 0.f;
  f32x4* g = gbuf + ((size_t)blockIdx.x * 4 + (w & 3)) * 64 * 4096 + lane;
  f32x4 gv = {v[0], v[1], v[2], v[3]};
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  if (is_mfma) {
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (FEED && (j % 6) == 0) {      // one B fragment pair per 6 MFMAs, as in the G phase
            u32x4 x = *reinterpret_cast<const u32x4*>(lds + ((it * 16 + j) & 63) * 1024 + lane * 16);
            u32x4 y = *reinterpret_cast<const u32x4*>(lds + 65536 + ((it * 16 + j) & 63) * 1024 + lane * 16);
            b = __builtin_bit_cast(bf16x8_t, x ^ y);
          }
          acc[(j / CHAIN) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[(j / CHAIN) & 7], 0, 0, 0);
        }
      }
  } else {
    if (mode & 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u % 8]) : "v"(v[(u + 3) % 8]));
          if (OP == 1) *reinterpret_cast<u32x2*>(lds + 98304 + (w & 3) * 8192 + (u & 15) * 512 + lane * 8) = u32x2{(unsigned)it, (unsigned)u};
          if (OP == 2) __builtin_nontemporal_store(gv, g + (size_t)((it * 16 + u) & 4095) * 64);
          if (OP == 3) { f32x4 x = __builtin_nontemporal_load(g + (size_t)((it * 16 + u) & 4095) * 64); asm volatile("" :: "v"(x)); }
          if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[u % 8]) : "v"(v[(u + 3) % 8]), "v"(v[(u + 5) % 8]));
          if (OP == 5) asm volatile("v_add_f32_dpp %0, %1, %1 row_shr:1" : "=v"(v[u % 8]) : "v"(v[(u + 3) % 8]));
          if (OP == 6) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[u % 8]) : "a"(acc[u & 7][u & 15]));
          if (OP == 7) { float* p = reinterpret_cast<float*>(lds + 98304 + (w & 3) * 8192) + lane; __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        }
      }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + w] = t1 - t0;
  float s = 0.f;
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int OP, int FEED, int CHAIN = 1> void run(const char* name) {
  static unsigned long long* d = nullptr; static float* sink = nullptr; static f32x4* gbuf = nullptr;
  if (!d) { (void)hipMalloc(&d, 256 * 8 * 8); (void)hipMalloc(&sink, 4096); (void)hipMalloc(&gbuf, (size_t)256 * 4 * 64 * 4096 * 16); }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP, FEED, CHAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int iters = 500;
  printf("%-44s", name);
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL((k<OP, FEED, CHAIN>), dim3(256), dim3(512), 131072, 0, mode, iters, d, sink, gbuf);
    (void)hipDeviceSynchronize();
    unsigned long long h[8]; (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    double m = (h[0] + h[1] + h[2] + h[3]) / 4.0, v = (h[4] + h[5] + h[6] + h[7]) / 4.0;
    if (mode == 1) printf(" alone: %5.1f cyc/MFMA |", m / (iters * 16.0));
    if (mode == 2) printf(" alone: %6.2f cyc/inst |", v / (iters * 16.0));
    if (mode == 3) printf(" together: %5.1f cyc/MFMA, %6.2f cyc/inst", m / (iters * 16.0), v / (iters * 16.0));
  }
  printf("\n");
}
int main() {
  run<0, 0>("partner v_fma_f32");
  run<0, 0, 3>("partner v_fma_f32, MFMA in dependent triples");
  run<0, 0, 16>("partner v_fma_f32, MFMA fully dependent");
  run<4, 0, 3>("partner v_cvt_pk, MFMA in dependent triples");
  run<0, 1>("partner v_fma_f32, MFMA fed by ds_read_b128");
  run<1, 0>("partner ds_write_b64");
  run<1, 1>("partner ds_write_b64, MFMA fed by ds_read");
  run<7, 1>("partner ds_add_f32, MFMA fed by ds_read");
  run<2, 0>("partner global_store_dwordx4 nt");
  run<2, 1>("partner global_store nt, MFMA fed by ds_read");
  run<3, 0>("partner global_load_dwordx4 nt");
  run<4, 0>("partner v_cvt_pk_bf16_f32");
  run<5, 0>("partner v_add_f32_dpp");
  run<6, 0>("partner v_accvgpr_read_b32");
  return 0;
}
