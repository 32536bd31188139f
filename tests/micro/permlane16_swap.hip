// Which lanes does v_permlane16_swap_b32 exchange?  Expected (used by the wide-net kernels):
// lanes 16-31 of vdst <-> lanes 0-15 of src (and 48-63 <-> 32-47); the other halves stay.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
  float a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  out[threadIdx.x] = __uint_as_float(r[0]);
  out[64 + threadIdx.x] = __uint_as_float(r[1]);
}
int main() {
  float* d; float h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int hi = (l >> 4) & 1;
    float ea = hi ? 100 + (l - 16) : l;       // vdst: upper 16 of each 32 get src's lower 16
    float eb = hi ? 100 + l : (l + 16);       // src : lower 16 of each 32 get vdst's upper 16
    if (h[l] != ea || h[64 + l] != eb) { ++bad; printf("lane %d: a=%g (exp %g) b=%g (exp %g)\n", l, h[l], ea, h[64 + l], eb); }
  }
  printf(bad ? "FAILED\n" : "ALL PASS\n");
  return bad;
}
