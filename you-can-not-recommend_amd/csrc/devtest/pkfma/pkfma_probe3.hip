// pkfma_probe3.hip -- v_pk_fma_f32 as the FIRST vector instruction behind a scalar write of EXEC (partial -> full), the place where
// hipcc had put the paired right-hand-side updates of the failing dual7 build:
//     s_and_saveexec_b64 s[0:1], mask ; ... (then-branch under the partial mask) ... ; s_or_saveexec_b64 s[0:1], s[4:5] ; v_pk_fma_f32 ...
// Does every lane of the full mask get both halves?  Waves alternate bursts of bf16 MFMAs with bursts of checks, two per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o pkfma_probe3 pkfma_probe3.hip && ./pkfma_probe3 [launches=20] [iters=2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64, 2) void probe(unsigned long long *bad, unsigned long long *lanes, int iters, unsigned seed) {
  const int lane = threadIdx.x;
  f32x4 acc[50];
#pragma unroll
  for (int t = 0; t < 50; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (lane * 40503u);
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(int)(s >> 9) * (1.0f / 4194304.0f) - 1.0f; };
  bf16x8 pa, pb;
#pragma unroll
  for (int e = 0; e < 8; ++e) { pa[e] = (__bf16)rnd(); pb[e] = (__bf16)rnd(); }
  unsigned long long nbad = 0;
  const int shift = (blockIdx.x >> 10) & 1;
  const unsigned long long xmask = 0xffff0000ffff0000ull;  // lane groups 1 and 3, as `xlane` in SolveMfmaF32::solve
  for (int it = 0; it < iters; ++it) {
    if (((it + shift) & 1) == 0) {
#pragma unroll
      for (int t = 0; t < 50; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc[t], 0, 0, 0);
    } else {
#pragma unroll 1
      for (int r = 0; r < 24; ++r) {
        const float x0 = rnd(), x1 = rnd(), z = rnd(), z1 = rnd(), c0 = rnd(), c1 = rnd();
        const float af[2] = {x0, x1}, bf[2] = {z, z1}, cf[2] = {c0, c1};
        double a, b, c;
        __builtin_memcpy(&a, af, 8); __builtin_memcpy(&b, bf, 8); __builtin_memcpy(&c, cf, 8);
        unsigned long long dd;
        float dummy = x0;
        asm volatile(
            "s_and_saveexec_b64 s[20:21], %5\n\t"   // exec &= xmask, old exec saved
            "s_xor_b64 s[22:23], exec, s[20:21]\n\t" // the other lanes
            "v_add_f32 %1, %1, %1\n\t"                // the then-branch: something under the partial mask
            "s_or_saveexec_b64 s[20:21], s[22:23]\n\t"  // exec = all lanes again
            "v_pk_fma_f32 %0, %2, %3, %4 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
            : "=&v"(dd), "+v"(dummy) : "v"(a), "v"(b), "v"(c), "s"(xmask) : "s20", "s21", "s22", "s23", "scc");
        const float d0 = __builtin_bit_cast(float, (unsigned)dd), d1 = __builtin_bit_cast(float, (unsigned)(dd >> 32));
        float e0, e1;
        asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(e0) : "v"(x0), "v"(z), "v"(c0));
        asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(e1) : "v"(x1), "v"(z), "v"(c1));
        const unsigned m = (__builtin_bit_cast(unsigned, d0) != __builtin_bit_cast(unsigned, e0)) + (__builtin_bit_cast(unsigned, d1) != __builtin_bit_cast(unsigned, e1));
        nbad += m;
      }
    }
  }
  float keep = 0.f;
#pragma unroll
  for (int t = 0; t < 50; ++t) keep += acc[t][0] + acc[t][3];
  if (keep == 12345.678f) nbad += 1;
  if (nbad) { atomicAdd(bad, nbad); atomicAdd(&lanes[lane], nbad); }
}

int main(int argc, char **argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20, iters = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned long long *bad, *lanes, h = 0, total = 0, hl[64];
  (void)hipMalloc(&bad, 8);
  (void)hipMalloc(&lanes, 512);
  (void)hipMemset(lanes, 0, 512);
  for (int l = 0; l < launches; ++l) {
    (void)hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(probe, dim3(2048), dim3(64), 0, 0, bad, lanes, iters, 4242u + l);
    (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    total += h;
  }
  (void)hipMemcpy(hl, lanes, 512, hipMemcpyDeviceToHost);
  printf("pkfma_probe3: %d launches x 2048 waves x %d bursts: %llu mismatching products of %.3g\n", launches, iters, total,
         (double)launches * 2048 * 64 * (iters / 2) * 48);
  if (total) { printf("per lane:"); for (int i = 0; i < 64; ++i) printf(" %llu", hl[i]); printf("\n"); }
  return total ? 1 : 0;
}
