// Many short units at once (8 blocks per CU): als_gram_slab_x6d_kernel against the float32-MFMA
// slab kernel, unit by unit.  x6many <nb:1..7> <k> <n> <units>     (exit code 1: some unit differs)
// tests/test_gpu_hazard.py runs the shipped build of it over every block count and both forms of
// the right-hand side (k = 16 nb: VALU accumulators, k < 16 nb: the padded Gramian column; k = 16 (nb - 1) + 4: the
// last block's planes packed into one operand, unless YCNR_NO_PK3 is set).
// Built with -DYCNR_X6D_ALLOW_PK (the right-hand side's multiply-adds packed across blocks) the
// k % 16 == 0 form fails for a few per cent of the units at 8 workgroups per CU.
#include "../als_kernels.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
using namespace ycnr;
template <int NB, bool PAD, bool PK3 = false>
int run(int k, int n, int units, int items) {
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f / std::sqrt((float)k));
  std::vector<float> V((size_t)items * k), vals((size_t)units * n);
  std::vector<int32_t> indx((size_t)units * n);
  for (auto &v : V) v = nd(rng);
  for (size_t i = 0; i < indx.size(); ++i) { indx[i] = rng() % items; vals[i] = 1 + (rng() % 10); }
  std::vector<Unit> us(units);
  for (int u = 0; u < units; ++u) us[u] = Unit{(int64_t)u * n, (int64_t)(u + 1) * n, u, u};
  float *dV, *dvals, *dA, *dB, *dz; int32_t *dindx; Unit *du;
  const size_t se = slab_elems(NB);
  hipMalloc(&dV, V.size() * 4); hipMalloc(&dvals, vals.size() * 4 + 64); hipMalloc(&dindx, indx.size() * 4 + 64); hipMalloc(&du, sizeof(Unit) * units);
  hipMalloc(&dA, se * 4 * units); hipMalloc(&dB, se * 4 * units); hipMalloc(&dz, 2048); hipMemset(dz, 0, 2048);
  hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dvals, vals.data(), vals.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dindx, indx.data(), indx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(du, us.data(), sizeof(Unit) * units, hipMemcpyHostToDevice);
  StepArgs<float> a{du, nullptr, dindx, dvals, dV, dz, nullptr, dA, nullptr, 0.05, k, 0, 0, (uint32_t)(V.size() * 4)};
  hipLaunchKernelGGL((als_gram_slab_kernel<float, NB, false>), dim3(units), dim3(64), 0, 0, a);
  a.slabs = dB;
  hipLaunchKernelGGL((als_gram_slab_x6d_kernel<NB, PAD, PK3>), dim3(units), dim3(64), 0, 0, a);
  hipError_t e = hipDeviceSynchronize();
  std::vector<float> A(se * units), B(se * units);
  hipMemcpy(A.data(), dA, A.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(B.data(), dB, B.size() * 4, hipMemcpyDeviceToHost);
  const size_t ntile = (size_t)tile_count(NB) * 4 * 64;
  if (getenv("YCNR_X6MANY_DUMP")) {  // rows 0 and 1 of the last tile of unit 0, all 16 columns: lane (g, c) holds rows 4 g + t of column c
    const int t = tile_count(NB) - 1;
    for (int r = 0; r < 2; ++r) {
      printf("ref  row %d:", r); for (int c = 0; c < 16; ++c) printf(" %10.6f", A[(size_t)t * 256 + c * 4 + r]); printf("\n");
      printf("x6d  row %d:", r); for (int c = 0; c < 16; ++c) printf(" %10.6f", B[(size_t)t * 256 + c * 4 + r]); printf("\n");
    }
  }
  int badUnits = 0, shown = 0;
  long badBlock[16] = {0};
  for (int u = 0; u < units; ++u) {
    const float *pa = &A[se * u], *pb = &B[se * u];
    double norm = 0, md = 0; int badTiles = 0;
    for (size_t i = 0; i < ntile; ++i) norm = std::fmax(norm, std::fabs((double)pa[i]));
    std::vector<int> tb;
    for (int t = 0; t < tile_count(NB); ++t) { double d = 0; for (int i = 0; i < 256; ++i) d = std::fmax(d, std::fabs((double)pa[t * 256 + i] - pb[t * 256 + i])); if (d > 1e-4 * norm) { ++badTiles; tb.push_back(t); } md = std::fmax(md, d); }
    double bd = 0, bn = 1e-30;
    for (int cb = 0; cb < NB; ++cb) { double worst = 0, nrm = 1e-30; for (int c = 0; c < 16; ++c) { double sa = 0, sb = 0; for (int g = 0; g < 4; ++g) { sa += pa[ntile + cb * 64 + g * 16 + c]; sb += pb[ntile + cb * 64 + g * 16 + c]; } worst = std::fmax(worst, std::fabs(sa - sb)); nrm = std::fmax(nrm, std::fabs(sa)); }
      if (worst > 1e-4 * nrm) ++badBlock[cb]; }
    for (int cb = 0; cb < NB; ++cb) for (int c = 0; c < 16; ++c) { double sa = 0, sb = 0; for (int g = 0; g < 4; ++g) { sa += pa[ntile + cb * 64 + g * 16 + c]; sb += pb[ntile + cb * 64 + g * 16 + c]; } bd = std::fmax(bd, std::fabs(sa - sb)); bn = std::fmax(bn, std::fabs(sa)); }
    if (badTiles || bd > 1e-4 * bn) {
      ++badUnits;
      if (shown++ < 6) { printf("unit %d: %d bad tiles (max diff %.3g of %.3g), b diff %.3g of %.3g; tiles:", u, badTiles, md, norm, bd, bn); for (int t : tb) printf(" %d", t); printf("\n"); }
    }
  }
  printf("%s: NB=%d k=%d n=%d units=%d: %d bad units; units with a wrong b per column block:", hipGetErrorString(e), NB, k, n, units, badUnits);
  for (int cb = 0; cb < NB; ++cb) printf(" %ld", badBlock[cb]);
  printf("\n");
  return badUnits;
}
int main(int argc, char **argv) {
  const int nb = atoi(argv[1]), k = atoi(argv[2]), n = atoi(argv[3]), units = atoi(argv[4]);
  int bad = 0;
  const int items = 20000;
#define YCNR_NB(NBV) \
  if (nb == NBV) bad = k < 16 * NBV ? (k == 16 * (NBV - 1) + 4 && !getenv("YCNR_NO_PK3") ? run<NBV, true, true>(k, n, units, items) : run<NBV, true>(k, n, units, items)) : run<NBV, false>(k, n, units, items);
  YCNR_NB(1) YCNR_NB(2) YCNR_NB(3) YCNR_NB(4) YCNR_NB(5) YCNR_NB(6) YCNR_NB(7)
#undef YCNR_NB
  return bad != 0;
}
