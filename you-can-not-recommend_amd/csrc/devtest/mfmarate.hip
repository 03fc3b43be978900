// Back-to-back issue rate of v_mfma_f32_16x16x32_bf16 with 28 accumulator tiles and 7 x 3 operand
// sets per wave (the register pattern of als_gram_slab_x6_kernel), one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int TERMS>
__global__ __launch_bounds__(64, 1) void k(const unsigned *in, float *out, int iters, long long *cyc) {
  const int l = threadIdx.x;
  u4 p[3][7];
  for (int s = 0; s < 3; ++s)
    for (int b = 0; b < 7; ++b) p[s][b] = u4{in[l + 64 * (s * 7 + b)], in[l + 7], in[l + 9], in[l + 11]};
  f4 acc[28];
  for (int t = 0; t < 28; ++t) acc[t] = f4{0, 0, 0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int term = 0; term < TERMS; ++term) {
      int t = 0;
#pragma unroll
      for (int bi = 0; bi < 7; ++bi)
#pragma unroll
        for (int bj = bi; bj < 7; ++bj, ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, p[term % 3][bi]), __builtin_bit_cast(bf16x8, p[(term / 2) % 3][bj]), acc[t], 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < 28; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 64 + l] = s;
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  unsigned *in; float *out; long long *cyc, h[8];
  (void)hipMalloc(&in, 64 * 32 * 4); (void)hipMemset(in, 0x3c, 64 * 32 * 4); (void)hipMalloc(&out, 1024 * 64 * 4); (void)hipMalloc(&cyc, 1024 * 8);
  const int iters = 2000;
  for (int blocks : {1, 1024}) {
    hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64), 0, 0, in, out, iters, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%d waves: %.1f cycles per MFMA (s_memtime)\n", blocks, (double)h[0] / (iters * 6.0 * 28));
  }
}
