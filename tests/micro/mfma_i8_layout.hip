// Hardware check of what an int8-limb dW / reverse sweep would rely on (gfx950):
//  T1  v_mfma_i32_32x32x32_i8 operand layout: which (row, k) byte sits where in the four operand dwords of a lane.
//      Hypothesis H0: lane (r = l & 31, h = l >> 5) holds A[r][16h + j], j = 0..15 (byte j of the 16);
//      hypothesis H1: two K = 16 halves laid out like the bf16 form: byte j < 8 -> k = 8h + j, byte j >= 8 -> k = 16 + 8h + (j - 8).
//      C in the standard 32x32 accumulator map (bf16_util.h mfma_row).
//  T2  ds_read_b64_tr_b8: fills a 16-row x 16-byte LDS block with byte (row, col) = 16 row + col, gives lane i of every
//      16-lane group the address of row (i >> 1), byte 8 (i & 1) (the natural "8 rows x 16 columns per group" guess, rows
//      0..7), and prints the eight bytes every lane of group 0 receives - the block's transposition rule read off directly.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_i8_layout.hip -o mfma_i8_layout ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int mrow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ void k_mfma(const signed char* A, const signed char* B, int* C, int hyp) {      // A[32][32] (row, k), B[32][32] (k, col)
  const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
  i32x4 a, b;
  for (int d = 0; d < 4; ++d) {
    unsigned ua = 0, ub = 0;
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * d + e;
      const int k = hyp == 0 ? 16 * h + j : (j < 8 ? 8 * h + j : 16 + 8 * h + (j - 8));
      ua |= (unsigned)(unsigned char)A[c * 32 + k] << (8 * e);
      ub |= (unsigned)(unsigned char)B[k * 32 + c] << (8 * e);
    }
    a[d] = (int)ua; b[d] = (int)ub;
  }
  i32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0;
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[mrow(r, h) * 32 + c] = acc[r];
}

__global__ void k_tr8(unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char T[16 * 16];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) T[i] = (unsigned char)i;
  __syncthreads();
  const int li = lane & 15;
  typedef __attribute__((address_space(3))) i32x2 lds_i32x2;
  const unsigned char* p = &T[16 * (li >> 1) + 8 * (li & 1)];
  i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)p);
  out[2 * lane] = (unsigned)v[0]; out[2 * lane + 1] = (unsigned)v[1];
}

int main() {
  std::vector<signed char> A(32 * 32), B(32 * 32);
  std::vector<int> C(32 * 32);
  srand(3);
  for (auto& v : A) v = (signed char)(rand() % 255 - 127);
  for (auto& v : B) v = (signed char)(rand() % 255 - 127);
  signed char *dA, *dB; int* dC; unsigned* dO;
  (void)hipMalloc(&dA, A.size()); (void)hipMalloc(&dB, B.size()); (void)hipMalloc(&dC, C.size() * 4); (void)hipMalloc(&dO, 128 * 4);
  (void)hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  int pass = -1;
  for (int hyp = 0; hyp < 2; ++hyp) {
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, hyp);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        int ref = 0;
        for (int k = 0; k < 32; ++k) ref += (int)A[i * 32 + k] * (int)B[k * 32 + j];
        bad += ref != C[i * 32 + j];
      }
    printf("T1 v_mfma_i32_32x32x32_i8, hypothesis H%d: %d of 1024 outputs differ %s\n", hyp, bad, bad ? "" : "-> EXACT");
    if (!bad) pass = hyp;
  }
  hipLaunchKernelGGL(k_tr8, dim3(1), dim3(64), 0, 0, dO);
  (void)hipDeviceSynchronize();
  unsigned o[128];
  (void)hipMemcpy(o, dO, sizeof(o), hipMemcpyDeviceToHost);
  printf("T2 ds_read_b64_tr_b8: lane i of a 16-lane group gave the address of (row i >> 1, byte 8 (i & 1)); received bytes as (row,col):\n");
  for (int l = 0; l < 64; ++l) {
    if (l >= 16 && l < 48) continue;      // groups 1 and 2 are not shown
    if (l == 48) printf("  (group 3, same addresses:)\n");
    printf("  lane %2d:", l);
    for (int e = 0; e < 8; ++e) { unsigned byte = (o[2 * l + (e >> 2)] >> (8 * (e & 3))) & 0xff; printf(" (%2u,%2u)", byte >> 4, byte & 15); }
    printf("\n");
  }
  return pass < 0;
}
