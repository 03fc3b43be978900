// Compares the bf16x6 chunk Gramian against the float32-MFMA one on the same inputs (slabs are
// laid out identically), several runs, to separate arithmetic differences from races.
#include "../als_kernels.hip.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
using namespace ycnr;
template <int NB>
int run(int k, int n, int items) {
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f / std::sqrt((float)k));
  std::vector<float> V((size_t)items * k), vals(n);
  std::vector<int32_t> indx(n);
  for (auto &v : V) v = nd(rng);
  for (int i = 0; i < n; ++i) { indx[i] = (i * 7) % items; vals[i] = 1 + (i % 5); }
  Unit u{0, n, 0, 0};
  float *dV, *dvals, *dslabA, *dslabB, *dz; int32_t *dindx; Unit *du;
  const size_t se = slab_elems(NB);
  hipMalloc(&dV, V.size() * 4); hipMalloc(&dvals, n * 4 + 64); hipMalloc(&dindx, n * 4 + 64); hipMalloc(&du, sizeof(Unit));
  hipMalloc(&dslabA, se * 4); hipMalloc(&dslabB, se * 4); hipMalloc(&dz, 2048); hipMemset(dz, 0, 2048);
  hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dvals, vals.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dindx, indx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(du, &u, sizeof u, hipMemcpyHostToDevice);
  StepArgs<float> a{du, nullptr, dindx, dvals, dV, dz, nullptr, dslabA, nullptr, 0.05, k, 0, 0, (uint32_t)(V.size() * 4)};
  hipLaunchKernelGGL((als_gram_slab_kernel<float, NB, false>), dim3(1), dim3(64), 0, 0, a);
  std::vector<float> A(se), B(se);
  hipMemcpy(A.data(), dslabA, se * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int rep = 0; rep < 4; ++rep) {
    a.slabs = dslabB;
    hipMemset(dslabB, 0xff, se * 4);
    hipLaunchKernelGGL((als_gram_slab_x6_kernel<NB>), dim3(1), dim3(64), 0, 0, a);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(B.data(), dslabB, se * 4, hipMemcpyDeviceToHost);
    double maxrel = 0; int worst = -1;
    const size_t ntile = (size_t)tile_count(NB) * 4 * 64;
    for (size_t i = 0; i < ntile; ++i) { double d = std::fabs((double)A[i] - B[i]) / (std::fabs((double)A[i]) + 1e-3); if (!(d <= maxrel)) { maxrel = d; worst = (int)i; } }
    double maxb = 0;
    for (int cb = 0; cb < NB; ++cb) for (int c = 0; c < 16; ++c) { double sa = 0, sb = 0; for (int g = 0; g < 4; ++g) { sa += A[ntile + cb * 64 + g * 16 + c]; sb += B[ntile + cb * 64 + g * 16 + c]; }
      double d = std::fabs(sa - sb) / (std::fabs(sa) + 1e-3); if (!(d <= maxb)) maxb = d; }
    printf("   b (group sums) max rel diff %.3g\n", maxb);
    if (maxb > 1e-4) ++bad;
    printf("NB=%d k=%d n=%d rep %d: %s max rel diff %.3g at elem %d (tile-reg %d lane %d) A=%g B=%g\n", NB, k, n, rep, hipGetErrorString(e), maxrel, worst,
           worst / 64, worst % 64, worst >= 0 ? A[worst] : 0.f, worst >= 0 ? B[worst] : 0.f);
    if (maxrel > 1e-4) ++bad;
  }
  return bad;
}
int main() {
  int bad = 0;
  bad += run<1>(16, 32, 50);
  bad += run<2>(32, 32, 50);
  bad += run<2>(32, 64, 50);
  bad += run<2>(20, 40, 50);
  bad += run<7>(100, 1000, 500);
  printf("%s\n", bad ? "FAIL" : "ok");
  return bad != 0;
}
