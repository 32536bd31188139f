// One wave per SIMD: cycles per v_mfma_f32_32x32x16_bf16 when each MFMA gap carries N fillers of one kind.
// Kinds: 0 v_fma_f32, 1 v_accvgpr_read_b32 (of an accumulator no MFMA is writing), 2 v_accvgpr_read_b32 of the tuple
// the NEXT MFMA accumulates into, 3 v_cvt_pk_bf16_f32, 4 v_exp_f32, 5 ds_read_b128, 6 ds_write_b64,
// 7 global_store_dwordx4 (nt), 8 global_load_dwordx4 (L2 hits), 9 v_lshl_add_u64.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NV>
__global__ __launch_bounds__(256, 1) void k(int iters, unsigned long long* out, float* sink, f32x4* gbuf) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  f32x16 acc[4], other;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int r = 0; r < 16; ++r) other[r] = threadIdx.x + r;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f + i * 0.01f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  // keep `other` in AGPRs
  asm volatile("" : "+a"(other));
  f32x4* gp = gbuf + (size_t)blockIdx.x * 256 * 64 + threadIdx.x;
  const unsigned lofs = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
  f32x4 ld = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i % 8]) : "v"(v[(i + 3) % 8]));
        if (KIND == 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[i % 8]) : "a"(other[(i * 5 + j) % 16]));
        if (KIND == 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[i % 8]) : "a"(acc[(j + 1) % 4][(i * 5) % 16]));
        if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[i % 8]) : "v"(v[(i + 1) % 8]), "v"(v[(i + 2) % 8]));
        if (KIND == 4) asm volatile("v_exp_f32 %0, %1" : "=v"(v[i % 8]) : "v"(v[(i + 1) % 8]));
        if (KIND == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(lofs + 1024 * (i & 3)) : "memory");
        if (KIND == 6) asm volatile("ds_write_b64 %0, %1" :: "v"(lofs / 2 + 2048 * (i & 3)), "v"(*(u32x2*)&v[2 * (i % 4)]) : "memory");
        if (KIND == 7) __builtin_nontemporal_store(*(f32x4*)&v[4 * (i & 1)], gp + 256 * ((it * 4 + j + i) & 63));
        if (KIND == 8) { f32x4 t = gp[256 * ((it + j + i) & 63)]; v[i % 8] += t[0]; }
        if (KIND == 9) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(*(unsigned long long*)&v[2 * (i % 4)]) : "s"(0x1234567ull));
      }
    }
    if (KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  float s = ld[0];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) sink[threadIdx.x] = s;
}
static unsigned long long* d; static float* sink; static f32x4* gbuf;
template <int KIND, int NV> double run() {
  const int iters = 2000;
  hipLaunchKernelGGL((k<KIND, NV>), dim3(256), dim3(256), 65536, 0, iters, d, sink, gbuf);
  hipDeviceSynchronize();
  unsigned long long h[256]; hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
  return m / 256 / (iters * 4.0);
}
template <int KIND> void row(const char* name) {
  printf("%-44s 0:%5.1f  1:%5.1f  2:%5.1f  3:%5.1f  4:%5.1f  6:%5.1f  8:%5.1f  cycles per MFMA\n", name, run<KIND, 0>(), run<KIND, 1>(),
         run<KIND, 2>(), run<KIND, 3>(), run<KIND, 4>(), run<KIND, 6>(), run<KIND, 8>());
}
int main() {
  hipMalloc(&d, 256 * 8); hipMalloc(&sink, 4096); hipMalloc(&gbuf, (size_t)256 * 256 * 64 * 16);
  hipMemset(gbuf, 0, (size_t)256 * 256 * 64 * 16);
  hipFuncSetAttribute((const void*)&k<5, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  printf("fillers per MFMA gap ->\n");
  row<0>("v_fma_f32");
  row<1>("v_accvgpr_read (idle accumulator)");
  row<2>("v_accvgpr_read (tuple of the next MFMA)");
  row<3>("v_cvt_pk_bf16_f32");
  row<4>("v_exp_f32");
  row<5>("ds_read_b128");
  row<6>("ds_write_b64");
  row<7>("global_store_dwordx4 nt");
  row<8>("global_load_dwordx4 (L2)");
  row<9>("v_lshl_add_u64");
  return 0;
}
