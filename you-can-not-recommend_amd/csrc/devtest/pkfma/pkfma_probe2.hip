// pkfma_probe2.hip -- the shape of the dual7 failure, compiler-generated: two float32 MFMA chains ("panel tiles" P, Q), then the
// right-hand-side update of both blocks as ONE float2 multiply-add (hipcc makes it a v_pk_fma_f32 on copies of P[t], Q[t] with z
// broadcast), checked against the same update computed from P, Q long after they were written.  Waves alternate bursts of bf16
// MFMAs with bursts of such steps, two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o pkfma_probe2 pkfma_probe2.hip && ./pkfma_probe2 [launches=20] [iters=2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64, 2) void probe(unsigned long long *bad, int iters, unsigned seed) {
  const int lane = threadIdx.x;
  f32x4 acc[44];
#pragma unroll
  for (int t = 0; t < 44; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (lane * 40503u);
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(int)(s >> 9) * (1.0f / 4194304.0f) - 1.0f; };
  bf16x8 pa, pb;
#pragma unroll
  for (int e = 0; e < 8; ++e) { pa[e] = (__bf16)rnd(); pb[e] = (__bf16)rnd(); }
  unsigned long long nbad = 0;
  const int shift = (blockIdx.x >> 10) & 1;
  for (int it = 0; it < iters; ++it) {
    if (((it + shift) & 1) == 0) {
#pragma unroll
      for (int t = 0; t < 44; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc[t], 0, 0, 0);
    } else {
#pragma unroll 1
      for (int r = 0; r < 8; ++r) {
        float w[4], tp[4], tq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { w[q] = rnd(); tp[q] = rnd(); tq[q] = rnd(); }
        const float z0 = rnd(), z1 = rnd(), z2 = rnd(), z3 = rnd();
        f32x2 b = f32x2{rnd(), rnd()};
        const f32x2 b0 = b;
        // the panel: P = W T_p, Q = W T_q  (four MFMAs of K = 4 each, as SolveMfmaF32::solve step 3)
        f32x4 P = f32x4{0.f, 0.f, 0.f, 0.f}, Q = P;
#pragma unroll
        for (int q = 0; q < 4; ++q) P = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q], tp[q], P, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) Q = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q], tq[q], Q, 0, 0, 0);
        // step 5b for both blocks at once: b -= {P[t], Q[t]} * z[t]   (float2: the vectoriser's pairing, written out)
        b = __builtin_elementwise_fma(-f32x2{P[0], Q[0]}, f32x2{z0, z0}, b);
        b = __builtin_elementwise_fma(-f32x2{P[1], Q[1]}, f32x2{z1, z1}, b);
        b = __builtin_elementwise_fma(-f32x2{P[2], Q[2]}, f32x2{z2, z2}, b);
        b = __builtin_elementwise_fma(-f32x2{P[3], Q[3]}, f32x2{z3, z3}, b);
        const float g0 = b[0], g1 = b[1];
        // the same sums from P, Q long after the MFMAs have retired, one scalar multiply-add at a time
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(P), "+v"(Q));
        float e0 = b0[0], e1 = b0[1];
        e0 = fmaf(-P[0], z0, e0); asm volatile("" : "+v"(e0));
        e1 = fmaf(-Q[0], z0, e1); asm volatile("" : "+v"(e1));
        e0 = fmaf(-P[1], z1, e0); asm volatile("" : "+v"(e0));
        e1 = fmaf(-Q[1], z1, e1); asm volatile("" : "+v"(e1));
        e0 = fmaf(-P[2], z2, e0); asm volatile("" : "+v"(e0));
        e1 = fmaf(-Q[2], z2, e1); asm volatile("" : "+v"(e1));
        e0 = fmaf(-P[3], z3, e0); asm volatile("" : "+v"(e0));
        e1 = fmaf(-Q[3], z3, e1); asm volatile("" : "+v"(e1));
        nbad += (__builtin_bit_cast(unsigned, g0) != __builtin_bit_cast(unsigned, e0)) + (__builtin_bit_cast(unsigned, g1) != __builtin_bit_cast(unsigned, e1));
      }
    }
  }
  float keep = 0.f;
#pragma unroll
  for (int t = 0; t < 44; ++t) keep += acc[t][0] + acc[t][3];
  if (keep == 12345.678f) nbad += 1;
  if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char **argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20, iters = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned long long *bad, h = 0, total = 0;
  (void)hipMalloc(&bad, 8);
  for (int l = 0; l < launches; ++l) {
    (void)hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(probe, dim3(2048), dim3(64), 0, 0, bad, iters, 777u + l);
    (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    total += h;
  }
  printf("pkfma_probe2: %d launches x 2048 waves x %d bursts: %llu mismatching sums of %.3g\n", launches, iters, total,
         (double)launches * 2048 * 64 * (iters / 2) * 16);
  return total ? 1 : 0;
}
