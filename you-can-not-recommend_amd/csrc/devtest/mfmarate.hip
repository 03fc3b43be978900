// Back-to-back issue rate of v_mfma_f32_16x16x32_bf16 with 28 accumulator tiles and 7 x 3 operand
// sets per wave (the register pattern of als_gram_slab_x6_kernel), one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
// FILL: independent VALU instructions issued after every MFMA (v_and / v_sub / v_perm on their own registers)
#ifndef WPS
#define WPS 1
#endif
template <int TERMS, int FILL>
__global__ __launch_bounds__(64, WPS) void k(const unsigned *in, float *out, int iters, long long *cyc) {
  const int l = threadIdx.x;
  u4 p[3][7];
  for (int s = 0; s < 3; ++s)
    for (int b = 0; b < 7; ++b) p[s][b] = u4{in[l + 64 * (s * 7 + b)], in[l + 7], in[l + 9], in[l + 11]};
  f4 acc[28];
  for (int t = 0; t < 28; ++t) acc[t] = f4{0, 0, 0, 0};
  float f[8];
  unsigned w[8];
  for (int i = 0; i < 8; ++i) { f[i] = __builtin_bit_cast(float, in[l + i]); w[i] = in[l + 8 + i]; }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int term = 0; term < TERMS; ++term) {
      int t = 0;
#pragma unroll
      for (int bi = 0; bi < 7; ++bi)
#pragma unroll
        for (int bj = bi; bj < 7; ++bj, ++t)
        {
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, p[term % 3][bi]), __builtin_bit_cast(bf16x8, p[(term / 2) % 3][bj]), acc[t], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < FILL; ++i) {
            const int r = (t * FILL + i) & 7;
            if ((i % 3) == 0) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(w[r]));
            else if ((i % 3) == 1) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[r]) : "v"(f[(r + 1) & 7]));
            else asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(w[r]) : "v"(w[(r + 3) & 7]), "s"(0x07060302));
          }
        }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += f[i] + __builtin_bit_cast(float, w[i]);
  for (int t = 0; t < 28; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 64 + l] = s;
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  unsigned *in; float *out; long long *cyc;
  (void)hipMalloc(&in, 64 * 32 * 4); (void)hipMemset(in, 0x3c, 64 * 32 * 4); (void)hipMalloc(&out, 1024 * WPS * 64 * 4); (void)hipMalloc(&cyc, 1024 * WPS * 8);
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  static long long hc[1024 * WPS];
  // wave: mean s_memtime ticks per MFMA inside one wave; SIMD: kernel time x 2.4 GHz / MFMAs issued per SIMD
  auto report = [&](const char *name) {
    (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost);
    double sum = 0; for (long long c : hc) sum += (double)c;
    printf("%s: wave %.1f ticks/MFMA, SIMD %.1f cycles/MFMA\n", name, sum / (1024 * WPS) / (iters * 6.0 * 28),
           ms * 1e-3 * 2.4e9 / (iters * 6.0 * 28 * WPS));
  };
  (void)hipEventRecord(e0, 0); hipLaunchKernelGGL((k<6, 0>), dim3(1024 * WPS), dim3(64), 0, 0, in, out, iters, cyc); report("MFMA only          ");
  (void)hipEventRecord(e0, 0); hipLaunchKernelGGL((k<6, 1>), dim3(1024 * WPS), dim3(64), 0, 0, in, out, iters, cyc); report("MFMA + 1 VALU each ");
  (void)hipEventRecord(e0, 0); hipLaunchKernelGGL((k<6, 2>), dim3(1024 * WPS), dim3(64), 0, 0, in, out, iters, cyc); report("MFMA + 2 VALU each ");
  (void)hipEventRecord(e0, 0); hipLaunchKernelGGL((k<6, 3>), dim3(1024 * WPS), dim3(64), 0, 0, in, out, iters, cyc); report("MFMA + 3 VALU each ");
  (void)hipEventRecord(e0, 0); hipLaunchKernelGGL((k<6, 4>), dim3(1024 * WPS), dim3(64), 0, 0, in, out, iters, cyc); report("MFMA + 4 VALU each ");
  (void)hipEventRecord(e0, 0); hipLaunchKernelGGL((k<6, 6>), dim3(1024 * WPS), dim3(64), 0, 0, in, out, iters, cyc); report("MFMA + 6 VALU each ");
}
