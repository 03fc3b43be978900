// Device self-test of the lane-level building blocks of als_kernels.hip.h (run on a GPU box):
// the register<->lane-group transpose, the 16x16x4 MFMA operand/result maps it is used with,
// and the DPP / shuffle reductions.  Exact small-integer data, asymmetric operands.
#include "../als_kernels.hip.h"
#include <cstdio>
#include <vector>
using namespace ycnr;
using S = SolveMfmaF32<2>;
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_transpose(float *out) {
  int l = threadIdx.x;
  f4 in = f4{100.f * l + 0, 100.f * l + 1, 100.f * l + 2, 100.f * l + 3};
  float o[4];
  S::transpose_rg(in, o);
  for (int q = 0; q < 4; ++q) out[q * 64 + l] = o[q];
}
// P = W * T with W, T given in C/D layout (lane (g,c), reg t <-> [4g+t][c]); W also as its transposed image
__global__ void k_panel(const float *W, const float *T, float *P, float *U) {
  __shared__ float Wt[16 * 20];
  int l = threadIdx.x, g = l >> 4, c = l & 15;
  // Wt[col][row] = W[row][col]
  for (int t = 0; t < 4; ++t) Wt[c * 20 + 4 * g + t] = W[(4 * g + t) * 16 + c];
  __syncthreads();
  float Aop[4];
  for (int q = 0; q < 4; ++q) Aop[q] = Wt[(4 * q + g) * 20 + c];
  f4 Tt = f4{T[(4 * g + 0) * 16 + c], T[(4 * g + 1) * 16 + c], T[(4 * g + 2) * 16 + c], T[(4 * g + 3) * 16 + c]};
  float Bop[4];
  S::transpose_rg(Tt, Bop);
  f4 acc = f4{0, 0, 0, 0};
  for (int q = 0; q < 4; ++q) acc = MfmaTraits<float>::mma(Aop[q], Bop[q], acc);
  for (int t = 0; t < 4; ++t) P[(4 * g + t) * 16 + c] = acc[t];
  // update form: U = -P^T P via the transposed registers
  float Pt[4];
  S::transpose_rg(acc, Pt);
  f4 u = f4{0, 0, 0, 0};
  for (int q = 0; q < 4; ++q) u = MfmaTraits<float>::mma(-Pt[q], Pt[q], u);
  for (int t = 0; t < 4; ++t) U[(4 * g + t) * 16 + c] = u[t];
}
__global__ void k_reduce(float *out) {
  int l = threadIdx.x;
  out[l] = S::row_sum((float)(l + 1));
  out[64 + l] = S::group_sum((float)(l + 1));
}
int main() {
  float *d;
  hipMalloc(&d, 4096 * 4);
  std::vector<float> h(4096);
  int fails = 0;
  hipLaunchKernelGGL(k_transpose, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost);
  for (int q = 0; q < 4; ++q)
    for (int l = 0; l < 64; ++l) {
      int g = l >> 4, c = l & 15;
      float want = 100.f * (q * 16 + c) + g;
      if (h[q * 64 + l] != want) { if (fails < 8) printf("transpose q=%d lane=%d got %g want %g\n", q, l, h[q * 64 + l], want); ++fails; }
    }
  printf("transpose: %s\n", fails ? "FAIL" : "ok");
  std::vector<float> W(256), T(256), P(256), U(256);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) { W[i * 16 + j] = (j <= i) ? (float)((i * 3 + j * 5) % 7 - 3) : 0.f; T[i * 16 + j] = (float)((i * 7 + j * 11) % 9 - 4); }
  float *dW = d, *dT = d + 256, *dP = d + 512, *dU = d + 768;
  hipMemcpy(dW, W.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(dT, T.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_panel, dim3(1), dim3(64), 0, 0, dW, dT, dP, dU);
  hipMemcpy(P.data(), dP, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(U.data(), dU, 1024, hipMemcpyDeviceToHost);
  int f2 = 0, f3 = 0;
  std::vector<float> Pw(256);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float s = 0;
      for (int kk = 0; kk < 16; ++kk) s += W[i * 16 + kk] * T[kk * 16 + j];
      Pw[i * 16 + j] = s;
      if (P[i * 16 + j] != s) { if (f2 < 8) printf("panel [%d][%d] got %g want %g\n", i, j, P[i * 16 + j], s); ++f2; }
    }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float s = 0;
      for (int kk = 0; kk < 16; ++kk) s -= Pw[kk * 16 + i] * Pw[kk * 16 + j];
      if (U[i * 16 + j] != s) { if (f3 < 8) printf("update [%d][%d] got %g want %g\n", i, j, U[i * 16 + j], s); ++f3; }
    }
  printf("panel W*T: %s\nupdate -P^T P: %s\n", f2 ? "FAIL" : "ok", f3 ? "FAIL" : "ok");
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h.data(), d, 128 * 4, hipMemcpyDeviceToHost);
  int f4_ = 0;
  for (int l = 0; l < 64; ++l) {
    int g = l >> 4, c = l & 15;
    float rs = 0, gs = 0;
    for (int cc = 0; cc < 16; ++cc) rs += g * 16 + cc + 1;
    for (int gg = 0; gg < 4; ++gg) gs += gg * 16 + c + 1;
    if (h[l] != rs || h[64 + l] != gs) { if (f4_ < 8) printf("reduce lane %d: row %g (want %g) group %g (want %g)\n", l, h[l], rs, h[64 + l], gs); ++f4_; }
  }
  printf("reductions: %s\n", f4_ ? "FAIL" : "ok");
  return (fails || f2 || f3 || f4_) ? 1 : 0;
}
