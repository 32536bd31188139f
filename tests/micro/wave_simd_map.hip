// Which SIMD does each wave of a 512-thread workgroup land on (gfx950)?  HW_REG_HW_ID bits [5:4] = SIMD id.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
  unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 8 * 4);
  hipLaunchKernelGGL(k, dim3(64), dim3(512), 0, 0, d);
  unsigned h[64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 64; b += 9) {
    printf("wg %2d: simd of waves 0..7 =", b);
    for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3);
    printf("   (cu %u se %u)\n", (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7);
  }
  return 0;
}
