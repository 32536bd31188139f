// What one MFMA costs at the board's power cap, per operand type: a bare loop of independent MFMAs on every SIMD of the
// chip (256 workgroups x 4 waves, 4 accumulator tiles per wave, 4 x 4 pseudo-random operand fragments), ~3 s per type so
// that an outside `rocm-smi --showpower` loop can sample it.  Prints launch time, the shader clock held (s_memtime against
// the constant 100 MHz s_memrealtime) and the MAC rate; MACs per joule follow from the sampled power.
//   hipcc -O3 --offload-arch=gfx950 mfma_dtype_power.hip -o mfma_dtype_power ; ./mfma_dtype_power [seconds per type]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

enum { BF16_32 = 0, BF16_16 = 1, F16_32 = 2, I8_32 = 3, I8_16 = 4, FP8_32 = 5, NTYPES = 6 };

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int TYPE>
__global__ __launch_bounds__(256) void k(int iters, int zero, unsigned long long* out, float* sink) {
  // operand fragments as raw bits: 8 dwords cover the widest form (f8f6f4: 32 bytes per lane)
  i32x8 a[4], b[4];
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) {
      unsigned ha = hash(threadIdx.x * 977u + blockIdx.x * 131071u + s * 8u + i), hb = hash(ha + 0x9e3779b9u);
      if (TYPE == BF16_32 || TYPE == BF16_16) {          // two bf16 per dword: sign, exponent 120..127 (|x| in [2^-7, 2)), random mantissa
        ha = (ha & 0x807f807fu) | (((ha >> 8) & 0x7u) + 120u) << 7 | (((ha >> 12) & 0x7u) + 120u) << 23;
        hb = (hb & 0x807f807fu) | (((hb >> 8) & 0x7u) + 120u) << 7 | (((hb >> 12) & 0x7u) + 120u) << 23;
      } else if (TYPE == F16_32) {                        // two f16 per dword: exponent 8..15 (bias 15)
        ha = (ha & 0x83ff83ffu) | (((ha >> 10) & 0x7u) + 8u) << 10 | (((ha >> 13) & 0x7u) + 8u) << 26;
        hb = (hb & 0x83ff83ffu) | (((hb >> 10) & 0x7u) + 8u) << 10 | (((hb >> 13) & 0x7u) + 8u) << 26;
      } else if (TYPE == FP8_32) {                        // four e4m3 per dword: clear the top exponent bit (no NaN)
        ha &= 0xbfbfbfbfu; hb &= 0xbfbfbfbfu;
      }                                                   // i8: any byte is a value
      a[s][i] = zero ? 0 : (int)ha;
      b[s][i] = zero ? 0 : (int)hb;
    }
  f32x16 accf[4];
  i32x16 acci[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) { accf[j][r] = 0.f; acci[j][r] = 0; }
  __syncthreads();
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int t = j & 3, sa = (j >> 2) & 3, sb = (j + (j >> 2)) & 3;
      if (TYPE == BF16_32) {
        bf16x8_t x = __builtin_bit_cast(bf16x8_t, i32x4{a[sa][0], a[sa][1], a[sa][2], a[sa][3]});
        bf16x8_t y = __builtin_bit_cast(bf16x8_t, i32x4{b[sb][0], b[sb][1], b[sb][2], b[sb][3]});
        accf[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, accf[t], 0, 0, 0);
      } else if (TYPE == BF16_16) {                       // 16x16x32: two per 32x32x16-equivalent, f32x4 accumulators
        bf16x8_t x = __builtin_bit_cast(bf16x8_t, i32x4{a[sa][0], a[sa][1], a[sa][2], a[sa][3]});
        bf16x8_t y = __builtin_bit_cast(bf16x8_t, i32x4{b[sb][0], b[sb][1], b[sb][2], b[sb][3]});
        f32x4 c = {accf[t][0], accf[t][1], accf[t][2], accf[t][3]};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
        accf[t][0] = c[0]; accf[t][1] = c[1]; accf[t][2] = c[2]; accf[t][3] = c[3];
      } else if (TYPE == F16_32) {
        f16x8_t x = __builtin_bit_cast(f16x8_t, i32x4{a[sa][0], a[sa][1], a[sa][2], a[sa][3]});
        f16x8_t y = __builtin_bit_cast(f16x8_t, i32x4{b[sb][0], b[sb][1], b[sb][2], b[sb][3]});
        accf[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, accf[t], 0, 0, 0);
      } else if (TYPE == I8_32) {
        i32x4 x = {a[sa][0], a[sa][1], a[sa][2], a[sa][3]}, y = {b[sb][0], b[sb][1], b[sb][2], b[sb][3]};
        acci[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, acci[t], 0, 0, 0);
      } else if (TYPE == I8_16) {
        i32x4 x = {a[sa][0], a[sa][1], a[sa][2], a[sa][3]}, y = {b[sb][0], b[sb][1], b[sb][2], b[sb][3]};
        i32x4 c = {acci[t][0], acci[t][1], acci[t][2], acci[t][3]};
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, c, 0, 0, 0);
        acci[t][0] = c[0]; acci[t][1] = c[1]; acci[t][2] = c[2]; acci[t][3] = c[3];
      } else {                                            // fp8 e4m3 x e4m3, 32x32x64, unit scales
        accf[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[sa], b[sb], accf[t], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    }
  }
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += accf[j][r] + (float)acci[j][r];
  if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int TYPE>
static void run(const char* name, double macs_per_mfma, double seconds, int zero, unsigned long long* d, float* sink) {
  const int iters = 20000;                      // x 16 MFMAs per wave and launch
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double ms_sum = 0; int n = 0;
  printf("BEGIN %s%s\n", name, zero ? " zeros" : ""); fflush(stdout);
  auto t0 = std::chrono::steady_clock::now();
  unsigned long long h[2] = {0, 1};
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<TYPE>, dim3(256), dim3(256), 0, 0, iters, zero, d, sink);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    ms_sum += ms; ++n;
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  }
  const double ms = ms_sum / n, mfmas = 256.0 * 4 * iters * 16;
  printf("END   %-22s%s %8.3f ms per launch  %7.1f MHz  %6.2f cycles per MFMA and SIMD  %8.1f TMAC/s\n", name, zero ? " zeros" : "      ", ms,
         (double)h[0] / (double)h[1] * 100.0, (double)h[0] / (iters * 16.0), mfmas * macs_per_mfma / (ms * 1e-3) * 1e-12);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  unsigned long long* d; float* sink;
  (void)hipMalloc(&d, 64); (void)hipMalloc(&sink, 4096);
  for (int zero = 0; zero < 2; ++zero) {
    run<BF16_32>("bf16 32x32x16", 32.0 * 32 * 16, seconds, zero, d, sink);
    run<BF16_16>("bf16 16x16x32", 16.0 * 16 * 32, seconds, zero, d, sink);
    run<F16_32>("f16 32x32x16", 32.0 * 32 * 16, seconds, zero, d, sink);
    run<I8_32>("i8 32x32x32", 32.0 * 32 * 32, seconds, zero, d, sink);
    run<I8_16>("i8 16x16x64", 16.0 * 16 * 64, seconds, zero, d, sink);
    run<FP8_32>("fp8 32x32x64 (f8f6f4)", 32.0 * 32 * 64, seconds, zero, d, sink);
    if (seconds < 1.0) break;
  }
  return 0;
}
