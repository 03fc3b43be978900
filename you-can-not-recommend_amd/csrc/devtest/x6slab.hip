// Compares the bf16x6 chunk Gramian against the float32-MFMA one on the same inputs (slabs are
// laid out identically), several runs, to separate arithmetic differences from races.
#include "../als_kernels.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
using namespace ycnr;
template <int NB, int VARIANT = 0>
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
    if (VARIANT == 0) hipLaunchKernelGGL((als_gram_slab_x6_kernel<NB>), dim3(1), dim3(64), 0, 0, a);
    else if (VARIANT == 3) hipLaunchKernelGGL((als_gram_slab_x6d_kernel<NB, false>), dim3(1), dim3(64), 0, 0, a);
    else hipLaunchKernelGGL((als_gram_slab_x6d_kernel<NB, true>), dim3(1), dim3(64), 0, 0, a);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(B.data(), dslabB, se * 4, hipMemcpyDeviceToHost);
    double maxrel = 0; int worst = -1;
    const size_t ntile = (size_t)tile_count(NB) * 4 * 64;
    for (size_t i = 0; i < ntile; ++i) { double d = std::fabs((double)A[i] - B[i]) / (std::fabs((double)A[i]) + 1e-3); if (!(d <= maxrel)) { maxrel = d; worst = (int)i; } }
    double maxb = 0, normb = 1e-30;  // norm-wise: single entries of b cancel
    for (int cb = 0; cb < NB; ++cb) for (int c = 0; c < 16; ++c) { double sa = 0, sb = 0; for (int g = 0; g < 4; ++g) { sa += A[ntile + cb * 64 + g * 16 + c]; sb += B[ntile + cb * 64 + g * 16 + c]; }
      maxb = std::fmax(maxb, std::fabs(sa - sb)); normb = std::fmax(normb, std::fabs(sa)); if (!(sb == sb)) maxb = 1e30; }
    maxb /= normb;
    printf("   b (group sums) max rel diff %.3g\n", maxb);
    if (getenv("X6_DUMP") && rep == 0 && maxb > 1e-3) {
      for (int cb = 0; cb < NB; ++cb) for (int c = 0; c < 16; ++c) { double sa = 0; for (int g = 0; g < 4; ++g) sa += A[ntile + cb * 64 + g * 16 + c];
        printf("     col %d want %g got g0..3 = %g %g %g %g\n", cb * 16 + c, sa, B[ntile + cb * 64 + c], B[ntile + cb * 64 + 16 + c], B[ntile + cb * 64 + 32 + c], B[ntile + cb * 64 + 48 + c]); }
    }
    if (maxb > 1e-4) ++bad;
    printf("v%d NB=%d k=%d n=%d rep %d: %s max rel diff %.3g at elem %d (tile-reg %d lane %d) A=%g B=%g\n", VARIANT, NB, k, n, rep, hipGetErrorString(e), maxrel, worst,
           worst / 64, worst % 64, worst >= 0 ? A[worst] : 0.f, worst >= 0 ? B[worst] : 0.f);
    if (maxrel > 1e-3) ++bad;
  }
  return bad;
}
int main(int argc, char **argv) {
  if (argc > 1) {  // single configuration, repeated: x6slab <nb> <variant> <k> <n> [times]
    const int nb = atoi(argv[1]), var = atoi(argv[2]), k = atoi(argv[3]), n = atoi(argv[4]), times = argc > 5 ? atoi(argv[5]) : 1;
    int bad = 0;
    for (int i = 0; i < times; ++i) {
      if (nb == 5 && var == 3) bad += run<5, 3>(k, n, 500);
      else if (nb == 5 && var == 4) bad += run<5, 4>(k, n, 500);
      else if (nb == 4 && var == 3) bad += run<4, 3>(k, n, 500);
      else if (nb == 6 && var == 3) bad += run<6, 3>(k, n, 500);
      else if (nb == 7 && var == 3) bad += run<7, 3>(k, n, 500);
      else if (nb == 7 && var == 4) bad += run<7, 4>(k, n, 500);
      else if (nb == 3 && var == 3) bad += run<3, 3>(k, n, 500);
    }
    printf("%s (%d bad)\n", bad ? "FAIL" : "ok", bad);
    return bad != 0;
  }
  int bad = 0;
  bad += run<1>(16, 32, 50);
  bad += run<2>(32, 32, 50);
  bad += run<2>(32, 64, 50);
  bad += run<2>(20, 40, 50);
  bad += run<7>(100, 1000, 500);
  bad += run<1, 3>(16, 32, 50);
  bad += run<1, 4>(12, 33, 50);
  bad += run<2, 3>(32, 70, 50);
  bad += run<2, 4>(20, 40, 50);
  bad += run<4, 3>(64, 333, 500);
  bad += run<5, 3>(80, 333, 500);
  bad += run<5, 4>(68, 333, 500);
  bad += run<6, 3>(96, 1000, 500);
  bad += run<7, 4>(100, 1000, 500);
  bad += run<7, 4>(100, 31, 500);
  bad += run<7, 4>(108, 97, 500);
  bad += run<7, 4>(100, 1, 500);
  bad += run<7, 4>(100, 64, 500);
  bad += run<7, 3>(112, 640, 500);
  printf("%s\n", bad ? "FAIL" : "ok");
  return bad != 0;
}
