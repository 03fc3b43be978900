#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned *in, unsigned *out) {
  int l = threadIdx.x;
  unsigned i0 = in[l], i1 = in[64 + l], i2 = in[128 + l], i3 = in[192 + l];
  v2u a = __builtin_amdgcn_permlane16_swap(i0, i1, false, false);
  v2u b = __builtin_amdgcn_permlane16_swap(i2, i3, false, false);
  v2u x = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
  v2u y = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
  out[l] = a[0]; out[64 + l] = a[1]; out[128 + l] = b[0]; out[192 + l] = b[1];
  out[256 + l] = x[0]; out[320 + l] = x[1]; out[384 + l] = y[0]; out[448 + l] = y[1];
}
int main() {
  unsigned *d, *di, h[512], hi[256];
  for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) hi[r * 64 + l] = (r + 1) * 1000 + (l >> 4) * 100 + (l & 15) * 7 % 13;
  (void)hipMalloc(&d, 2048); (void)hipMalloc(&di, 1024);
  (void)hipMemcpy(di, hi, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, d);
  (void)hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
  const char *n[8] = {"a0", "a1", "b0", "b1", "x0(out0)", "x1(out2)", "y0(out1)", "y1(out3)"};
  for (int i = 0; i < 8; ++i) { printf("%-9s rows(c=0):", n[i]); for (int g = 0; g < 4; ++g) printf(" %u", h[i * 64 + g * 16]); printf("   (c=5): "); for (int g = 0; g < 4; ++g) printf(" %u", h[i * 64 + g * 16+5]); printf("\n"); }
}
