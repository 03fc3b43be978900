// pkfma_probe.hip -- does v_pk_fma_f32 give the results of two v_fma_f32 while another wave of the SIMD runs bf16 MFMAs?
//
// Both open hazards of this library (devtest/stale_b, devtest/dual7) came down to one instruction shape: hipcc's SLP vectoriser
// pairs the multiply-adds of two column blocks into  v_pk_fma_f32 d[0:1], a[0:1], b[0:1], c[0:1] op_sel_hi:[1,0,1]  where a[0:1] are
// kept COPIES (v_mov_b32) of two registers that were produced at different times and b's low half is broadcast.  With two waves per
// SIMD and the other wave in its v_mfma_f32_16x16x32_bf16 phase 1 - 3 % of the rows got one such product wrong; with the pairing
// suppressed (asm barrier, or -fno-slp-vectorize) none did.  This probe issues that shape from inline asm, beside the two scalar
// v_fma_f32, in waves that alternate bursts of bf16 MFMAs with bursts of checks, two waves per SIMD (256 registers each), and counts
// mismatches.      hipcc --offload-arch=gfx950 -O3 -o pkfma_probe pkfma_probe.hip && ./pkfma_probe [launches=20] [iters=2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64, 2) void probe(unsigned long long *bad, int iters, unsigned seed, float *dbg) {
  const int lane = threadIdx.x;
  // 200 live accumulator registers: two waves per SIMD, no more
  f32x4 acc[50];
#pragma unroll
  for (int t = 0; t < 50; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (lane * 40503u);
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(int)(s >> 9) * (1.0f / 4194304.0f) - 1.0f; };
  bf16x8 pa, pb;
#pragma unroll
  for (int e = 0; e < 8; ++e) { pa[e] = (__bf16)rnd(); pb[e] = (__bf16)rnd(); }
  unsigned long long nbad = 0;
  // a per-wave phase shift: the MFMA bursts of one wave fall on the check bursts of the other
  const int shift = (blockIdx.x >> 10) & 1;
  for (int it = 0; it < iters; ++it) {
    if (((it + shift) & 1) == 0) {
#pragma unroll
      for (int t = 0; t < 50; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc[t], 0, 0, 0);
    } else {
#pragma unroll 1
      for (int r = 0; r < 24; ++r) {
        const float x0 = rnd(), x1 = rnd(), z = rnd(), c0 = rnd(), c1 = rnd();
        // (64-bit operands as doubles: element-wise asm outputs of a float2 made hipcc read d[0] twice)
        const f32x2 af = f32x2{x0, x1 + 0.0f * (float)(r & 1)}, bf = f32x2{z, rnd()}, cf = f32x2{c0, c1};
        const double a = __builtin_bit_cast(double, af), b = __builtin_bit_cast(double, bf), c = __builtin_bit_cast(double, cf);
        unsigned long long dd;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(dd) : "v"(a), "v"(b), "v"(c));
        // (scalars: hipcc of ROCm 7.2 reads element 1 of  float2{bit_cast(lo), bit_cast(hi)}  built from the halves of a 64-bit
        //  value as element 0 -- /tmp-sized example in README.md)
        const float d0 = __builtin_bit_cast(float, (unsigned)dd), d1 = __builtin_bit_cast(float, (unsigned)(dd >> 32));
        float e0, e1;
        asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(e0) : "v"(x0), "v"(z), "v"(c0));
        asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(e1) : "v"(x1), "v"(z), "v"(c1));
        const unsigned m = (__builtin_bit_cast(unsigned, d0) != __builtin_bit_cast(unsigned, e0)) +
                           (__builtin_bit_cast(unsigned, d1) != __builtin_bit_cast(unsigned, e1));
        if (m && dbg && atomicAdd(reinterpret_cast<unsigned *>(dbg), 1u) == 0) {
          dbg[1] = x0; dbg[2] = x1; dbg[3] = z; dbg[4] = c0; dbg[5] = c1; dbg[6] = d0; dbg[7] = d1; dbg[8] = e0; dbg[9] = e1; dbg[10] = (float)lane;
        }
        nbad += m;
      }
    }
  }
  float keep = 0.f;
#pragma unroll
  for (int t = 0; t < 50; ++t) keep += acc[t][0] + acc[t][3];
  if (keep == 12345.678f) nbad += 1;  // (keeps the accumulators alive)
  if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char **argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20, iters = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned long long *bad, h = 0, total = 0;
  float *dbg, hd[12];
  (void)hipMalloc(&bad, 8);
  (void)hipMalloc(&dbg, 48);
  (void)hipMemset(dbg, 0, 48);
  for (int l = 0; l < launches; ++l) {
    (void)hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(probe, dim3(2048), dim3(64), 0, 0, bad, iters, 12345u + l, dbg);
    (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    total += h;
  }
  (void)hipMemcpy(hd, dbg, 48, hipMemcpyDeviceToHost);
  if (total) printf("first mismatch: x0 %.9g x1 %.9g z %.9g c0 %.9g c1 %.9g  packed %.9g %.9g  scalar %.9g %.9g  lane %g\n", hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7], hd[8], hd[9], hd[10]);
  printf("pkfma_probe: %d launches x 2048 waves x %d bursts: %llu mismatching products of %.3g\n", launches, iters, total,
         (double)launches * 2048 * 64 * (iters / 2) * 48);
  return total ? 1 : 0;
}
