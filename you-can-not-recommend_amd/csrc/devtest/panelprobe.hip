#include "../als_kernels.hip.h"
#include <cstdio>
#include <vector>
using namespace ycnr;
using S = SolveMfmaF32<2>;
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_panel(const float *W, const float *T, float *P, float *dbgA, float *dbgB) {
  __shared__ float Wt[16 * 20];
  int l = threadIdx.x, g = l >> 4, c = l & 15;
  for (int t = 0; t < 4; ++t) Wt[c * 20 + 4 * g + t] = W[(4 * g + t) * 16 + c];
  __syncthreads();
  float Aop[4];
  for (int q = 0; q < 4; ++q) Aop[q] = Wt[(4 * q + g) * 20 + c];
  f4 Tt = f4{T[(4 * g + 0) * 16 + c], T[(4 * g + 1) * 16 + c], T[(4 * g + 2) * 16 + c], T[(4 * g + 3) * 16 + c]};
  float Bop[4];
  S::transpose_rg(Tt, Bop);
  for (int q = 0; q < 4; ++q) { dbgA[q * 64 + l] = Aop[q]; dbgB[q * 64 + l] = Bop[q]; }
  f4 acc = f4{0, 0, 0, 0};
  for (int q = 0; q < 4; ++q) acc = MfmaTraits<float>::mma(Aop[q], Bop[q], acc);
  for (int t = 0; t < 4; ++t) P[(4 * g + t) * 16 + c] = acc[t];
}
int main() {
  float *d; (void)hipMalloc(&d, 8192 * 4);
  std::vector<float> W(256), T(256), P(256), A(256), B(256);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { W[i * 16 + j] = (j <= i) ? (float)((i * 3 + j * 5) % 7 - 3) : 0.f; T[i * 16 + j] = (float)(i * 100 + j); }
  (void)hipMemcpy(d, W.data(), 1024, hipMemcpyHostToDevice); (void)hipMemcpy(d + 256, T.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_panel, dim3(1), dim3(64), 0, 0, d, d + 256, d + 512, d + 1024, d + 2048);
  (void)hipMemcpy(P.data(), d + 512, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(A.data(), d + 1024, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(B.data(), d + 2048, 1024, hipMemcpyDeviceToHost);
  int fa = 0, fb = 0;
  for (int q = 0; q < 4; ++q) for (int l = 0; l < 64; ++l) { int g = l >> 4, c = l & 15;
    float wa = W[c * 16 + 4 * q + g], wb = T[(4 * q + g) * 16 + c];
    if (A[q * 64 + l] != wa) { if (fa < 6) printf("Aop q=%d lane=%d got %g want %g\n", q, l, A[q * 64 + l], wa); ++fa; }
    if (B[q * 64 + l] != wb) { if (fb < 6) printf("Bop q=%d lane=%d got %g want %g\n", q, l, B[q * 64 + l], wb); ++fb; } }
  printf("Aop %d bad, Bop %d bad\n", fa, fb);
  int f2 = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += W[i * 16 + kk] * T[kk * 16 + j];
      if (P[i * 16 + j] != s) { if (f2 < 6) printf("P[%d][%d] got %g want %g\n", i, j, P[i * 16 + j], s); ++f2; } }
  printf("P %d bad\n", f2);
}
