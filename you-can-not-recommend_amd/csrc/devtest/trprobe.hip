// trprobe.hip -- what ds_read_b64_tr_b16 delivers (gfx950): LDS holds 16-bit elements with value
// = their own element index; every lane supplies the address cdna_hip_programming.md T10
// describes (lane 4q+p of a 16-lane group: row q, columns 4p..4p+3 of a [rows][16] block with a
// row stride of STRIDE elements) and the program prints what lane i got in element e.
// Expected: element index (r0 + e) * STRIDE + c0 + i   (column i of row e).
//   hipcc --offload-arch=gfx950 -O2 trprobe.hip -o trprobe && ./trprobe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int STRIDE = 24;
__global__ void k(int *out) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x, grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  // group grp reads rows 8 grp .. 8 grp + 3, columns 4 .. 19 of the image (c0 = 4)
  const int elem = (8 * grp + q) * STRIDE + 4 + 4 * p;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + elem));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = a[e];
}
int main() {
  int *d, h[256];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int e = 0; e < 4; ++e) {
      const int want = (8 * (lane >> 4) + e) * STRIDE + 4 + (lane & 15);
      if (h[lane * 4 + e] != want) {
        if (bad < 8) printf("lane %d elem %d: got %d want %d\n", lane, e, h[lane * 4 + e], want);
        ++bad;
      }
    }
  printf("trprobe: %d mismatches\n", bad);
  return bad != 0;
}
