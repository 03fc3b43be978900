// ycnr_als.hip -- host side of libycnr_als.so: the C ABI declared in include/ycnr_als.h.
//
// Owns device memory (CSR shards, factor matrices, work-unit tables, partial slabs),
// builds the work-unit schedule once per rating upload, and launches the kernels of
// als_kernels.hip.h.  There is deliberately no CPU fallback anywhere in this file: without
// a HIP device every entry point fails with YCNR_ERR_HIP.
#include "../../include/ycnr_als.h"
#include "als_kernels.hip.h"
#include "als_wg_kernels.hip.h"
#include "als_pair_kernels.hip.h"
#ifdef YCNR_WITH_G32  // devtest build only (make EXTRA=-DYCNR_WITH_G32): the 32 x 32 Gramian kernel of round 3, measured equal
#include "devtest/als_gram32_kernels.hip.h"
#endif
#include "als_gen_kernels.hip.h"
#include "prep_kernels.hip.h"
#include <hipcub/hipcub.hpp>
#include "prep_kernels.hip.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

using namespace ycnr;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(e_ == hipErrorOutOfMemory ? YCNR_ERR_NOMEM : YCNR_ERR_HIP, "%s failed: %s (%s:%d)", \
                  #expr, hipGetErrorString(e_), __FILE__, __LINE__);                            \
  } while (0)

constexpr int kMaxFactors = 128;     // float64, and the one-wave-per-row float32 kernels
constexpr int kMaxFactorsBig = 256;  // float32 through the workgroup-per-row kernels of als_wg_kernels.hip.h
constexpr int kMaxFactorsAny = 4096; // beyond kMaxFactorsBig (float64: kMaxFactors): the any-k path of als_gen_kernels.hip.h
constexpr int kGenChunk = 4096;      // ratings per unit of the any-k path (als_gen_kernels.hip.h): every row goes through slabs there
constexpr int64_t kGenArenaBytes = (int64_t)2 << 30;  // slab arena of that path: rows are solved in batches that fit it
constexpr int kPairNB = 16;           // block count whose whole rows go Gramian -> slab -> two-wave solve (als_pair_kernels.hip.h): 240 < k <= 256
constexpr int64_t kPairBatchRows = 4096;   // rows per batch of that path (a slab is 140 KB: 0.57 GB of arena).  One GPU's eighth of C5, ms per
                                           // iteration at 512 / 1024 / 2048 / 3072 / 4096 / 6144 / 8192 / 12288 / 24576 / 98304 rows: 141.6 / 138.0 / 136.7 /
                                           // 136.3 / 136.2 / 136.5 / 137.3 / 139.7 / 138.6 / 137.6 -- short batches leave more of a batch's slabs in the
                                           // last-level cache for its solve, shorter ones pay two kernel tails per batch (YCNR_PAIR_BATCH_ROWS overrides)
constexpr int kWgChunk = 8192;       // ratings per chunk of a row that is split over workgroups (k > 128)
#ifndef YCNR_WG_FUSED_MAX
#define YCNR_WG_FUSED_MAX 8192
#endif
constexpr int kWgFusedMax = YCNR_WG_FUSED_MAX;  // longest row one workgroup takes whole (k > 128).  16384 until round 5: a row's Gramian is ONE float32 chain over
                                     // its ratings / 32 steps, and the item rows of the full C5 that missed the flat 1e-5 against float64 (2 of 54 sampled,
                                     // 1.09e-5) were whole rows of 8 - 16 K ratings; at 8192: every sampled row within 7.2e-6, item half-step + 1 %
constexpr int kDefaultChunk = 1024;  // ratings per split unit (and the largest fused row)
constexpr int kMaxSlabsPerRow = 64;  // heavier rows get proportionally longer chunks
constexpr int kMaxSlabsPerRowBig = 4096;  // k > 128 (chunks of kWgChunk ratings): an item of the full C5 has 10 M ratings; with 64 slabs its
                                          // chunks were float32 sums over 156 K ratings each (backward error 0.86 of its gate in round 3);
                                          // 8192-rating chunks keep every accumulation chain short, the workgroup reduce adds the slabs in order
constexpr int64_t kBandBytes = (int64_t)96 << 20;  // slice of the fixed matrix one band of chunks gathers from (cache-sized)
constexpr size_t kErrBytes = 65536;
constexpr size_t kZeroRowBytes = 32768 + 64;  // >= kMaxFactorsAny doubles: the "row" the dual kernels gather for ratings past a row's end
constexpr int kMaxDualBlocks = 12;       // dual-form kernels exist for 1..12 blocks of 16 ratings (12: 340 bytes of scratch per lane, still 1.0 ms per C5-shard iteration cheaper than the primal form for rows of 177..192 ratings; 13: 936 bytes and no gain)
constexpr int kMaxDualBlocksSmallK = 5;  // k <= 128: beyond 80 ratings the row kernel (k x k) is cheaper (MAL scale, k = 100: 6 -> 5 blocks took 0.2 ms off the user half-step once the solve had lost its readlanes and transposes; 4 was slower)

size_t tsize(int dtype) { return dtype == YCNR_F64 ? 8 : 4; }

struct Ratings {
  int64_t rowBegin = 0, rowEnd = 0, nnz = 0;
  int64_t *dRowPtr = nullptr;  // local, rebased to 0 (RMSE only)
  int32_t *dIndx = nullptr;
  void *dVals = nullptr;
  bool loaded = false;
  void release() {
    if (dRowPtr) (void)hipFree(dRowPtr);
    if (dIndx) (void)hipFree(dIndx);
    if (dVals) (void)hipFree(dVals);
    dRowPtr = nullptr;
    dIndx = nullptr;
    dVals = nullptr;
    loaded = false;
    nnz = 0;
  }
};

// a batch of the any-k path: split rows [firstSplit, +nSplit) with their units = slabs [slabBase, +nSlabs)
struct GenBatch {
  int32_t firstSplit, nSplit, slabBase, nSlabs;
};

// the any-k path (als_gen_kernels.hip.h): float32 beyond 256 factors, float64 beyond 128
bool is_gen(int dtype, int k) { return k > (dtype == YCNR_F32 ? kMaxFactorsBig : kMaxFactors); }

// Batches of consecutive split rows whose slabs fit `arenaSlabs` (at least one row per batch).
std::vector<GenBatch> gen_batches(const std::vector<SplitRow> &split, int64_t arenaSlabs) {
  std::vector<GenBatch> out;
  size_t i = 0;
  while (i < split.size()) {
    GenBatch b{(int32_t)i, 0, split[i].slab0, 0};
    while (i < split.size() && (b.nSplit == 0 || b.nSlabs + split[i].nslabs <= arenaSlabs)) {
      b.nSlabs += split[i].nslabs;
      ++b.nSplit;
      ++i;
    }
    out.push_back(b);
  }
  return out;
}

struct Schedule {
  std::vector<GenBatch> genBatches;
  Unit *dUnits = nullptr;
  SplitRow *dSplit = nullptr;
  void *dSlabs = nullptr;
  float *dRowSlabs = nullptr;  // k > 240: the images of a batch of whole rows between their Gramian and their two-wave solve
  int64_t rowSlabRows = 0;
  int64_t nUnits = 0, nSplit = 0, nSlabs = 0, solvedRows = 0, fusedRatings = 0;
  int32_t maxRowSlabs = 0;  // most slabs any split row has
  // whole-row units: [nSlabs, nSlabs + nPrimal) primal form, then dual classes m = kMaxDualBlocks..1
  int64_t nPrimal = 0, dualFirst[kMaxDualBlocks + 1] = {}, dualCount[kMaxDualBlocks + 1] = {};
  int64_t dualRows = 0, dualRatings = 0;
  double dualFlops = 0;  // flops the dual form executes for those rows: G = Y Y^T (symmetric), Cholesky, x = Y^T w
  void release() {
    if (dUnits) (void)hipFree(dUnits);
    if (dSplit) (void)hipFree(dSplit);
    if (dSlabs) (void)hipFree(dSlabs);
    if (dRowSlabs) (void)hipFree(dRowSlabs);
    dRowSlabs = nullptr;
    rowSlabRows = 0;
    dUnits = nullptr;
    dSplit = nullptr;
    dSlabs = nullptr;
    nUnits = nSplit = nSlabs = solvedRows = fusedRatings = nPrimal = dualRows = dualRatings = 0;
    maxRowSlabs = 0;
    dualFlops = 0;
    genBatches.clear();
    for (int m = 0; m <= kMaxDualBlocks; ++m) dualFirst[m] = dualCount[m] = 0;
  }
};

// Ratings per split unit when the caller leaves it to the library: every chunk costs a 30 KB
// slab written and read again (k = 100), so chunks should be long -- but a launch wants >= 16
// units per resident wave (2048 on a 256-CU part at two waves per SIMD) to keep its tail short.
// MAL item side: 121.5 M ratings on one GPU -> 3072; an eighth of it on each of 8 GPUs -> 1024.
// Small uploads (below 2 M ratings: the ML-1M shape) are bound by their longest wave, not by slab traffic:
// there the chunk shrinks, down to 256, so that one round of waves covers the half-step (ML-1M shape, k = 100:
// 0.62 -> 0.54 ms per iteration with 512; 200 K x 20 K with 20 M ratings is fastest at 1024).
// nnz: ratings of the whole side; splitNnz: those of its rows above kDefaultChunk ratings (the rows that are split at all).
// Both describe the side, not the shard: the cuts of a split row must not depend on how the rows are dealt to GPUs.
int auto_chunk(int64_t nnz, int64_t splitNnz) {
  if (nnz < (int64_t)2048 * kDefaultChunk)
    return (int)((std::min<int64_t>(kDefaultChunk, std::max<int64_t>(256, nnz / 2048)) + 3) & ~(int64_t)3);
  const int64_t c = splitNnz / (2048 * 16);
  return (int)((std::min<int64_t>(3072, std::max<int64_t>(kDefaultChunk, c)) + 3) & ~(int64_t)3);
}

// Split rows into wave-level units (the counterpart of EmfLord.splitToPortions,
// lib/emf/EmfLord.js:510-612, at wave instead of worker-process granularity).
// rowPtr: local (length nRows + 1, any base).  Units index ratings relative to rowPtr[0].
void build_schedule(const int64_t *rowPtr, int64_t rowBegin, int64_t nRows, int chunk,
                    std::vector<Unit> &units, std::vector<SplitRow> &split, int64_t &nSlabs,
                    int64_t &solvedRows, int64_t splitAbove = -1, int fusedMax = 0, int maxSlabs = kMaxSlabsPerRow) {
  // fusedMax: longest row that stays one unit (default: chunk)
  if (fusedMax <= 0) fusedMax = chunk;
  // splitAbove >= 0 (big path): every row longer than splitAbove goes through slabs, in row
  // order (no longest-first sort, so that batches are contiguous)
  // on return units = [split chunks (nSlabs of them) | whole rows]
  const int64_t base = rowPtr[0];
  units.clear();
  split.clear();
  nSlabs = 0;
  solvedRows = 0;
  std::vector<Unit> fused;
  fused.reserve((size_t)nRows);
  for (int64_t r = 0; r < nRows; ++r) {
    const int64_t b = rowPtr[r] - base, e = rowPtr[r + 1] - base, n = e - b;
    if (n <= 0) continue;  // rows without ratings are never written
    ++solvedRows;
    const int32_t row = (int32_t)(rowBegin + r);
    if (splitAbove >= 0 ? n <= splitAbove : n <= fusedMax) {
      fused.push_back(Unit{b, e, row, -1});
      continue;
    }
    int64_t ch = chunk;
    if ((n + ch - 1) / ch > maxSlabs) ch = (n + maxSlabs - 1) / maxSlabs;
    ch = (ch + 3) & ~(int64_t)3;
    const int64_t parts = (n + ch - 1) / ch;
    split.push_back(SplitRow{n, row, (int32_t)nSlabs, (int32_t)parts, 0});
    for (int64_t p = 0; p < parts; ++p) {
      const int64_t ub = b + p * ch, ue = std::min(e, ub + ch);
      units.push_back(Unit{ub, ue, row, (int32_t)(nSlabs + p)});
    }
    nSlabs += parts;
  }
  // longest first: split chunks (all ~chunk long, longest chunks first), then fused rows by
  // descending length, so the tail of the launch is made of the cheapest units
  if (splitAbove < 0)
    std::stable_sort(units.begin(), units.end(),
                     [](const Unit &a, const Unit &b) { return (a.end - a.beg) > (b.end - b.beg); });
  std::stable_sort(fused.begin(), fused.end(),
                   [](const Unit &a, const Unit &b) { return (a.end - a.beg) > (b.end - b.beg); });
  units.insert(units.end(), fused.begin(), fused.end());
}

// environment toggles for A/B experiments from unmodified hosts, read once per process
struct EnvFlags {
  bool noDualX6, noX6d, noFusedX6d, noOverlap, ignoreNumeric, noDualQuad, noGraph, noPair, g32, noPk3;
  size_t k1LdsPad;
};
const EnvFlags &env_flags() {
  static const EnvFlags f = {getenv("YCNR_NO_DUAL_X6") != nullptr, getenv("YCNR_NO_X6D") != nullptr, getenv("YCNR_NO_FUSED_X6D") != nullptr,
                             getenv("YCNR_NO_OVERLAP") != nullptr, getenv("YCNR_IGNORE_NUMERIC") != nullptr, getenv("YCNR_NO_DUAL_QUAD") != nullptr,
                             getenv("YCNR_NO_GRAPH") != nullptr, getenv("YCNR_NO_PAIR") != nullptr, getenv("YCNR_G32") != nullptr, getenv("YCNR_NO_PK3") != nullptr,
                             getenv("YCNR_K1_LDSPAD") ? (size_t)atoi(getenv("YCNR_K1_LDSPAD")) : 0};
  return f;
}

// hipFuncAttributeMaxDynamicSharedMemorySize, set once per (kernel, device, size) instead of per half-step
int set_max_lds(const void *fn, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, size_t> done;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find({fn, dev});
  if (it != done.end() && it->second == bytes) return YCNR_OK;
  HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  done[{fn, dev}] = bytes;
  return YCNR_OK;
}

constexpr int kSideStreams = 6;  // most side streams a handle can have; side_streams() of them are in use
// Side streams in use (YCNR_SIDE_STREAMS, read once; experiments).  The runtime maps a process's streams onto
// GPU_MAX_HW_QUEUES hardware queues (4 unless that variable says otherwise, read when the runtime starts): kernels on two
// streams that share a hardware queue run one after the other.
// The runtime reads GPU_MAX_HW_QUEUES once, when it starts.  The library does not touch the environment (a setenv from a dlopen
// constructor races with getenv in the host's other threads, and it cannot know whether the runtime has started already): the
// HOSTS set GPU_MAX_HW_QUEUES=8 before their first HIP call (bench.py, python/ycnr_als/_lib.py -- only while torch has not
// initialised HIP --, lib/ycnr_als.js).  What the variable said when this library was LOADED is what counts
// here: a value that appears later cannot have reached the runtime through one of those hosts.
int g_hw_queues_at_load = 0;
__attribute__((constructor)) void ycnr_note_hw_queues() {
  const char *q = getenv("GPU_MAX_HW_QUEUES");
  g_hw_queues_at_load = q ? atoi(q) : 0;
}
int side_streams() {
  static const int n = [] {
    const char *e = getenv("YCNR_SIDE_STREAMS");
    if (e) return std::max(1, std::min(kSideStreams, atoi(e)));
    // one stream per dual class when every stream gets a hardware queue of its own (the row kernel's stream + five), else two:
    // MAL scale, user half-step, same box: 13.20 ms with 4 queues / 2 side streams, 13.14 with 8 / 2, 12.84 with 8 / 5
    // (five streams on four queues -- streams sharing a queue wait behind each other's event waits -- is the one to avoid)
    return g_hw_queues_at_load >= 8 ? 5 : 2;
  }();
  return n;
}
constexpr int64_t kGraphMaxRatings = 2 * 1024 * 1024;  // uploads below this replay their half-step as a captured hipGraph ...
constexpr int64_t kGraphMinRatings = 256 * 1024;       // ... unless they are so small that the graph launch itself costs more than four
                                                        // kernel launches (ML-100k shape: 0.124 ms per iteration launch by launch, 0.174 as graphs)
constexpr int64_t kMinOverlapDualRows = 1024;  // fewer dual-form rows than this: everything in stream order

struct DualPlan {
  bool noX6 = false;     // YCNR_FLAG_NO_BF16X6: float32-MFMA Gramian in the dual kernels too
  bool fewSlabs = false; // no split row of the piece has more than kFewSlabs slabs (als_reduce_solve_kernel<..., FEW>)
  int64_t nPrimal = -1;  // < 0: no dual classes, every whole row goes through the primal kernel
  const int64_t *first = nullptr, *count = nullptr;
  // Side streams (unless YCNR_FLAG_NO_OVERLAP): the dual classes are independent of the row
  // kernel and of each other (different rows of the solved matrix), so they are launched next
  // to it, alternating over kSideStreams streams that fork from and join the step's stream:
  // their waves fill what the row kernel's two waves per SIMD leave idle, and no kernel waits
  // for the last waves of the one before it (15.5 -> 15.1 ms per MAL-scale user half-step).
  int nSide = 0;
  hipStream_t side[kSideStreams] = {};
  hipEvent_t fork = nullptr, join[kSideStreams] = {};
  mutable int nextSide = 0;
  // Small uploads (a half-step of a few kernels that each fill a fraction of the chip): the chunk Gramians and the
  // reduce that consumes their slabs go to a branch of their own, next to the row kernel and the dual classes
  // -- only the reduce depends on the chunks.  The whole half-step is then captured once into a hipGraph and
  // replayed (ycnr_als_step_async), so the forks and joins cost nothing per half-step.
  hipStream_t slabStream = nullptr;
  hipEvent_t slabJoin = nullptr;
  // The item half-step sharded by user bands (step_banded) borrows the kernel choice of launch_nbe: 1 = the chunk kernel alone
  // over units [0, nSplitUnits) of args.units; 2 = als_band_reduce_solve_kernel alone over nSplit rows of args.split, whose band
  // slabs are found through this pointer table
  int only = 0;
  const void *const *bandSlab = nullptr;
  int nBands = 0;
};

template <int M>
int launch_dual(StepArgs<float> args, const DualPlan &dp, hipStream_t stream) {
  if (dp.count[M] > 0) {
    if (dp.nSide > 0) stream = dp.side[dp.nextSide++ % dp.nSide];
    args.firstDual = (int32_t)dp.first[M];
    if constexpr (M == 1) {
      // rows of at most 16 ratings: four to a wave (needs 32-bit offsets into a fixed matrix below 4 GB, k <= 128)
      if (!dp.noX6 && !env_flags().noDualX6 && !env_flags().noDualQuad && args.k <= 128 && args.k % 4 == 0 && args.fixedBytes != 0) {
        hipLaunchKernelGGL(als_dual_quad_kernel, dim3((unsigned)((dp.count[M] + 3) / 4)), dim3(64), 0, stream, args, (int32_t)dp.count[M]);
        HIP_TRY(hipGetLastError());
        return YCNR_OK;
      }
    }
    // bf16x6 form unless switched off; NBN = 1 has a single tile and too little to gain
    const bool x6 = !dp.noX6 && !env_flags().noDualX6;
    void (*kd)(StepArgs<float>) = als_dual_solve_kernel<M, false>;
    if (x6) kd = als_dual_solve_kernel<M, true>;
    // (YCNR_DUAL_LDSPAD: occupancy experiments -- extra dynamic LDS per workgroup limits the workgroups a CU holds)
    static const size_t ldsPad = getenv("YCNR_DUAL_LDSPAD") ? (size_t)atoi(getenv("YCNR_DUAL_LDSPAD")) : 0;
    if (ldsPad)
      if (int rcl = set_max_lds(reinterpret_cast<const void *>(kd), SolveMfmaF32<M>::lds_bytes() + ldsPad)) return rcl;
    hipLaunchKernelGGL(kd, dim3((unsigned)dp.count[M]), dim3(64), SolveMfmaF32<M>::lds_bytes() + ldsPad, stream, args);
    HIP_TRY(hipGetLastError());
  }
  return YCNR_OK;
}

// the RMSE kernel for this k: chunks of four values per lane of a 16-lane group (als_rmse_kernel)
template <typename T>
void launch_rmse(const RmseArgs<T> &a, int nPieces, hipStream_t stream) {
  const bool vec = ((size_t)a.k * sizeof(T)) % 16 == 0 && a.k <= 512;
  void (*kr)(RmseArgs<T>) = als_rmse_kernel<T, 0>;
  if (vec) kr = a.k <= 64 ? als_rmse_kernel<T, 1> : a.k <= 128 ? als_rmse_kernel<T, 2> : a.k <= 256 ? als_rmse_kernel<T, 4> : als_rmse_kernel<T, 8>;
  hipLaunchKernelGGL(kr, dim3((unsigned)nPieces), dim3(256), 0, stream, a);
}

template <typename T>
int launch_duals(const StepArgs<T> &, const DualPlan &, hipStream_t) { return YCNR_OK; }
template <>
int launch_duals<float>(const StepArgs<float> &args, const DualPlan &dp, hipStream_t stream) {
  int rc = launch_dual<12>(args, dp, stream);
  if (!rc) rc = launch_dual<11>(args, dp, stream);
  if (!rc) rc = launch_dual<10>(args, dp, stream);
  if (!rc) rc = launch_dual<9>(args, dp, stream);
  if (!rc) rc = launch_dual<8>(args, dp, stream);
  if (!rc) rc = launch_dual<7>(args, dp, stream);
  if (!rc) rc = launch_dual<6>(args, dp, stream);
  if (!rc) rc = launch_dual<5>(args, dp, stream);
  if (!rc) rc = launch_dual<4>(args, dp, stream);
  if (!rc) rc = launch_dual<3>(args, dp, stream);
  if (!rc) rc = launch_dual<2>(args, dp, stream);
  if (!rc) rc = launch_dual<1>(args, dp, stream);
  return rc;
}

template <typename T, int NB>
void (*slab_x6_kernel())(StepArgs<T>) { return nullptr; }
#define YCNR_X6(NBV) \
  template <>        \
  void (*slab_x6_kernel<float, NBV>())(StepArgs<float>) { return als_gram_slab_x6_kernel<NBV>; }
YCNR_X6(1) YCNR_X6(2) YCNR_X6(3) YCNR_X6(4) YCNR_X6(5) YCNR_X6(6) YCNR_X6(7) YCNR_X6(8)
#undef YCNR_X6

// k = 16 (NB - 1) + 4: the last block's planes packed into one operand (GramX6D's PK3); YCNR_NO_PK3=1 for A/B runs
template <int NB>
bool pk3_k(int k) { return k == 16 * (NB - 1) + 4 && !env_flags().noPk3; }

// the same Gramian with the gather staged through LDS by LDS-DMA: two waves per SIMD up to k = 112, one from 116 to 128 (round 4:
// these eight-block sizes used to fall to the float32-MFMA kernels -- MAL shape, k = 128: 58.9 ms per iteration against 23.4 at k = 112)
template <typename T, int NB>
void (*slab_x6d_kernel(int))(StepArgs<T>) { return nullptr; }
#define YCNR_X6D(NBV) \
  template <>         \
  void (*slab_x6d_kernel<float, NBV>(int k))(StepArgs<float>) { return k % 4 ? nullptr : k < 16 * NBV ? (pk3_k<NBV>(k) ? als_gram_slab_x6d_kernel<NBV, true, true> : als_gram_slab_x6d_kernel<NBV, true, false>) : als_gram_slab_x6d_kernel<NBV, false>; }
YCNR_X6D(1) YCNR_X6D(2) YCNR_X6D(3) YCNR_X6D(4) YCNR_X6D(5) YCNR_X6D(6) YCNR_X6D(7) YCNR_X6D(8)
#undef YCNR_X6D

// k = 16 (NB - 1) + 4 in float32 with the register solver: the instantiations that eliminate the four edge
// columns first (SolveMfmaF32::solve_edge4)
template <int NB, bool LDS_SOLVER>
constexpr bool edge4_k(int k) { return YCNR_EDGE4_SOLVE && NB >= 2 && !LDS_SOLVER && k == 16 * (NB - 1) + 4; }

template <typename T, int NB, bool LDS_SOLVER>
void (*fused_x6d_kernel(int))(StepArgs<T>) { return nullptr; }
#define YCNR_X6D(NBV, LDSV) \
  template <>               \
  void (*fused_x6d_kernel<float, NBV, LDSV>(int k))(StepArgs<float>) { return k % 4 ? nullptr : k < 16 * NBV ? (edge4_k<NBV, LDSV>(k) ? (pk3_k<NBV>(k) ? als_gram_solve_x6d_kernel<NBV, true, LDSV, true, true> : als_gram_solve_x6d_kernel<NBV, true, LDSV, true, false>) : (pk3_k<NBV>(k) ? als_gram_solve_x6d_kernel<NBV, true, LDSV, false, true> : als_gram_solve_x6d_kernel<NBV, true, LDSV, false, false>)) : als_gram_solve_x6d_kernel<NBV, false, LDSV, false>; }
YCNR_X6D(1, false) YCNR_X6D(2, false) YCNR_X6D(3, false) YCNR_X6D(4, false) YCNR_X6D(5, false) YCNR_X6D(6, false) YCNR_X6D(7, false) YCNR_X6D(8, false)
YCNR_X6D(1, true) YCNR_X6D(2, true) YCNR_X6D(3, true) YCNR_X6D(4, true) YCNR_X6D(5, true) YCNR_X6D(6, true) YCNR_X6D(7, true)
#undef YCNR_X6D

// the fused row kernel on the pre-split planes of the fixed matrix (GramX6P): k % 4 == 0, 16 (NB - 1) < k < 16 NB, register solver;
// k = 16 (NB - 1) + 4: the last block packed into one slot, the four edge columns eliminated first
template <typename T, int NB>
void (*fused_x6p_kernel(int))(StepArgs<T>) { return nullptr; }
#define YCNR_X6P(NBV) \
  template <>         \
  void (*fused_x6p_kernel<float, NBV>(int k))(StepArgs<float>) { return (k % 4 || k >= 16 * NBV || k <= 16 * (NBV - 1)) ? nullptr : (NBV >= 2 && planes_pack(k)) ? als_gram_solve_x6p_kernel<NBV, (NBV >= 2), (NBV >= 2)> : als_gram_solve_x6p_kernel<NBV, false, false>; }
YCNR_X6P(1) YCNR_X6P(2) YCNR_X6P(3) YCNR_X6P(4) YCNR_X6P(5) YCNR_X6P(6) YCNR_X6P(7) YCNR_X6P(8)
#undef YCNR_X6P

// SLABX6: split chunks go through the bf16x6 Gramian kernel (plain slab layout), so the reduce
// kernel reads plain slabs whatever form the fused kernel uses.
template <typename T, int NB, bool LDS_SOLVER, bool EDGE, bool SLABX6>
int launch_nbe(StepArgs<T> args, int64_t nUnits, int64_t nSplitUnits, int64_t nSplit, hipStream_t stream,
               hipEvent_t *ev /* 5 events or null */, const DualPlan &dp) {
  const size_t lds = SolverFor<T, NB, LDS_SOLVER>::type::lds_bytes();
  void (*k0)(StepArgs<T>) = SLABX6 ? slab_x6_kernel<T, NB>() : als_gram_slab_kernel<T, NB, EDGE && !SLABX6>;
  if (SLABX6 && slab_x6d_kernel<T, NB>(args.k) && !env_flags().noX6d) k0 = slab_x6d_kernel<T, NB>(args.k);
  void (*k1)(StepArgs<T>) = als_gram_solve_kernel<T, NB, LDS_SOLVER, EDGE, false>;
  void (*k2)(StepArgs<T>) = als_reduce_solve_kernel<T, NB, LDS_SOLVER, EDGE && !SLABX6, false>;
  if constexpr (std::is_same<T, float>::value && !LDS_SOLVER)
    if (dp.fewSlabs) k2 = als_reduce_solve_kernel<T, NB, LDS_SOLVER, EDGE && !SLABX6, false, true>;
  if constexpr (std::is_same<T, float>::value && NB >= 2 && !LDS_SOLVER) {
    if (edge4_k<NB, LDS_SOLVER>(args.k)) {
      k1 = als_gram_solve_kernel<T, NB, LDS_SOLVER, EDGE, true>;
      k2 = als_reduce_solve_kernel<T, NB, LDS_SOLVER, EDGE && !SLABX6, true>;
      if (dp.fewSlabs) k2 = als_reduce_solve_kernel<T, NB, LDS_SOLVER, EDGE && !SLABX6, true, true>;
    }
  }
  if (SLABX6 && fused_x6d_kernel<T, NB, LDS_SOLVER>(args.k) && !env_flags().noX6d && !env_flags().noFusedX6d)
    k1 = fused_x6d_kernel<T, NB, LDS_SOLVER>(args.k);
  size_t ldsRow = lds;  // dynamic LDS of the row kernel: the solver's image
  if constexpr (SLABX6 && !LDS_SOLVER && NB <= 8) {
    if (args.planes && fused_x6p_kernel<T, NB>(args.k)) {  // the half-step split the fixed matrix into planes
      k1 = fused_x6p_kernel<T, NB>(args.k);
      ldsRow = 0;  // (its solver works in the slots of the Gramian)
    }
  }
  const size_t pad = env_flags().k1LdsPad;  // experiments: limits blocks per CU
  if (int rcl = set_max_lds(reinterpret_cast<const void *>(k1), ldsRow + pad)) return rcl;
  if (int rcl = set_max_lds(reinterpret_cast<const void *>(k2), lds)) return rcl;
  if (dp.only == 1) {
    if (nSplitUnits > 0) {
      hipLaunchKernelGGL(k0, dim3((unsigned)nSplitUnits), dim3(64), 0, stream, args);
      HIP_TRY(hipGetLastError());
    }
    return YCNR_OK;
  }
  if (dp.only == 2) {
    void (*kb)(StepArgs<T>, const T *const *, int32_t) = als_band_reduce_solve_kernel<T, NB, LDS_SOLVER, EDGE && !SLABX6, false>;
    if constexpr (std::is_same<T, float>::value && NB >= 2 && !LDS_SOLVER)
      if (edge4_k<NB, LDS_SOLVER>(args.k)) kb = als_band_reduce_solve_kernel<T, NB, LDS_SOLVER, EDGE && !SLABX6, true>;
    if (int rcl = set_max_lds(reinterpret_cast<const void *>(kb), lds)) return rcl;
    if (nSplit > 0) {
      hipLaunchKernelGGL(kb, dim3((unsigned)nSplit), dim3(64), lds, stream, args, reinterpret_cast<const T *const *>(dp.bandSlab), (int32_t)dp.nBands);
      HIP_TRY(hipGetLastError());
    }
    return YCNR_OK;
  }
  args.firstFused = (int32_t)nSplitUnits;
  if (ev) HIP_TRY(hipEventRecord(ev[0], stream));
  const bool branch = dp.slabStream != nullptr && nSplitUnits > 0;  // chunks -> reduce on their own branch
  const int64_t nPrimal = dp.nPrimal >= 0 ? dp.nPrimal : nUnits - nSplitUnits;
  const bool overlap = dp.nPrimal >= 0 && dp.nSide > 0;
  if (branch || overlap) HIP_TRY(hipEventRecord(dp.fork, stream));
  if (branch) {
    HIP_TRY(hipStreamWaitEvent(dp.slabStream, dp.fork, 0));
    hipLaunchKernelGGL(k0, dim3((unsigned)nSplitUnits), dim3(64), 0, dp.slabStream, args);
    HIP_TRY(hipGetLastError());
    if (nSplit > 0) {
      hipLaunchKernelGGL(k2, dim3((unsigned)nSplit), dim3(64), lds, dp.slabStream, args);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(dp.slabJoin, dp.slabStream));
  } else if (nSplitUnits > 0) {
    hipLaunchKernelGGL(k0, dim3((unsigned)nSplitUnits), dim3(64), 0, stream, args);
    HIP_TRY(hipGetLastError());
  }
  if (ev) HIP_TRY(hipEventRecord(ev[1], stream));
  if (overlap) {  // dual classes first, on the side streams; then the row kernel on this one
    for (int i = 0; i < dp.nSide; ++i) HIP_TRY(hipStreamWaitEvent(dp.side[i], dp.fork, 0));
    dp.nextSide = 0;
    int rc = launch_duals<T>(args, dp, stream);
    if (rc) return rc;
    for (int i = 0; i < dp.nSide; ++i) HIP_TRY(hipEventRecord(dp.join[i], dp.side[i]));
  }
  if (nPrimal > 0) {
    hipLaunchKernelGGL(k1, dim3((unsigned)nPrimal), dim3(64), ldsRow + pad, stream, args);
    HIP_TRY(hipGetLastError());
  }
  if (ev) HIP_TRY(hipEventRecord(ev[2], stream));
  if (overlap) {
    for (int i = 0; i < dp.nSide; ++i) HIP_TRY(hipStreamWaitEvent(stream, dp.join[i], 0));
  } else if (dp.nPrimal >= 0) {
    int rc = launch_duals<T>(args, dp, stream);
    if (rc) return rc;
  }
  if (ev) HIP_TRY(hipEventRecord(ev[3], stream));
  if (branch) {
    HIP_TRY(hipStreamWaitEvent(stream, dp.slabJoin, 0));
  } else if (nSplit > 0) {
    hipLaunchKernelGGL(k2, dim3((unsigned)nSplit), dim3(64), lds, stream, args);
    HIP_TRY(hipGetLastError());
  }
  if (ev) HIP_TRY(hipEventRecord(ev[4], stream));
  return YCNR_OK;
}

// the VALU-edge Gramian exists for float32 and k = 16 (NB-1) + 4 only
template <typename T, int NB, bool LDS_SOLVER>
int launch_nb(const StepArgs<T> &args, int64_t nUnits, int64_t nSplitUnits, int64_t nSplit, hipStream_t stream, hipEvent_t *ev,
              const DualPlan &dp, bool edge, bool x6) {
  if constexpr (std::is_same<T, float>::value) {
    if constexpr (NB >= 2) {
      if (edge && x6) return launch_nbe<T, NB, LDS_SOLVER, true, true>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp);
      if (edge) return launch_nbe<T, NB, LDS_SOLVER, true, false>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp);
    }
    if (x6) return launch_nbe<T, NB, LDS_SOLVER, false, true>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp);
  }
  return launch_nbe<T, NB, LDS_SOLVER, false, false>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp);
}

// k > 128 with k % 4 != 0: the kernels of that path need 16-byte rows, so the half-step runs on copies of
// both matrices with the rows padded to kp = 4 ceil(k / 4) zero columns.  A zero column adds nothing
// to the Gramian or to b, its diagonal entry is lambda n and its solution component exactly 0.
__global__ void pad_rows_kernel(const float *src, float *dst, int64_t rowBegin, int64_t rows, int k, int kp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * kp) return;
  const int64_t r = rowBegin + i / kp;
  const int c = (int)(i % kp);
  dst[r * kp + c] = c < k ? src[r * k + c] : 0.0f;
}
__global__ void unpad_rows_kernel(const float *src, float *dst, int64_t rowBegin, int64_t rows, int k, int kp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * k) return;
  const int64_t r = rowBegin + i / k;
  const int c = (int)(i % k);
  dst[r * k + c] = src[r * kp + c];
}

int device_cus() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
  cus[dev] = n;
  return n;
}

// 128 < k <= 256, float32 (als_wg_kernels.hip.h): one 512-thread workgroup per row or chunk, one
// workgroup per CU, persistent over its share of the units:
//   [chunks of heavy rows -> slabs] [whole rows: Gramian + solve] [dual classes] [slabs -> solve]
template <int NB>
int launch_wg_nb(StepArgs<float> args, int64_t nSplitUnits, int64_t nPrimal, int64_t nSplit, hipStream_t stream, hipEvent_t *ev,
                 const DualPlan &dp, float *rowSlabs, int64_t rowSlabRows) {
  auto k0 = als_wg_gram_slab_kernel<NB>;
  auto k1 = als_wg_gram_solve_kernel<NB>;
  auto k2 = als_wg_reduce_solve_kernel<NB>;
  const size_t lds = WgCfg<NB>::LDS_BYTES;
  if (int rc = set_max_lds(reinterpret_cast<const void *>(k0), lds)) return rc;
  if (int rc = set_max_lds(reinterpret_cast<const void *>(k1), lds)) return rc;
  if (int rc = set_max_lds(reinterpret_cast<const void *>(k2), lds)) return rc;
  const int64_t cus = device_cus();
  args.firstFused = (int32_t)nSplitUnits;
  if (ev) HIP_TRY(hipEventRecord(ev[0], stream));
  // The dual classes go to the side streams, in front of the Gramian / solve batches (round 3).  A Gramian or solve workgroup
  // fills a CU's register file, so the two kinds never share a CU -- but the one-wave dual kernels take the CUs that a
  // batch kernel's tail has already left: the user half-step of one GPU's eighth of C5 76.5 -> 73.5 ms (interleaved A/B,
  // YCNR_NO_OVERLAP=1).  (Tried on top and dropped: the solve of batch i on a second stream beside the Gramians of batch
  // i + 1 on fewer CUs -- 224 / 192 / 160 / 128 CUs for the Gramians: +3.7 / +10 / +20 / +40 ms per iteration.)
  const bool sideDuals = dp.nPrimal >= 0 && dp.nSide > 0 && dp.fork;
  if (sideDuals) {
    HIP_TRY(hipEventRecord(dp.fork, stream));
    for (int i = 0; i < dp.nSide; ++i) HIP_TRY(hipStreamWaitEvent(dp.side[i], dp.fork, 0));
    dp.nextSide = 0;
    int rc = launch_duals<float>(args, dp, stream);
    if (rc) return rc;
    for (int i = 0; i < dp.nSide; ++i) HIP_TRY(hipEventRecord(dp.join[i], dp.side[i]));
  }
  // NB = 16, YCNR_G32=1: the Gramians on 32 x 32 MFMAs, one wave per SIMD (als_gram32_kernels.hip.h) -- measured equal to
  // WgGram on one GPU's eighth of C5 (both are bound by the power the bf16 pipe + the split draw), so off
#ifdef YCNR_WITH_G32
  const bool g32 = NB == kPairNB && env_flags().g32;
  if (g32) {
    if (int rc = set_max_lds(reinterpret_cast<const void *>(als_g32_slab_kernel), (size_t)G32Cfg::LDS_BYTES)) return rc;
    if (int rc = set_max_lds(reinterpret_cast<const void *>(als_g32_rowslab_kernel), (size_t)G32Cfg::LDS_BYTES)) return rc;
  }
#endif
  if (nSplitUnits > 0) {
#ifdef YCNR_WITH_G32
    if (g32)
      hipLaunchKernelGGL(als_g32_slab_kernel, dim3((unsigned)std::min(nSplitUnits, cus)), dim3(kG32Threads), (size_t)G32Cfg::LDS_BYTES, stream, args,
                         (int32_t)nSplitUnits);
    else
#endif
      hipLaunchKernelGGL(k0, dim3((unsigned)std::min(nSplitUnits, cus)), dim3(kWgThreads), lds, stream, args, (int32_t)nSplitUnits);
    HIP_TRY(hipGetLastError());
  }
  if (ev) HIP_TRY(hipEventRecord(ev[1], stream));
  bool pairDone = false;
  if constexpr (NB == kPairNB) {
    // whole rows: Gramian -> slab by the workgroup kernel, then the solve by two waves per row, batch by batch
    // (als_pair_kernels.hip.h); YCNR_NO_PAIR=1 keeps the fused workgroup kernel for A/B runs
    if (nPrimal > 0 && rowSlabs && rowSlabRows > 0 && !env_flags().noPair) {
      auto kg = als_wg_gram_rowslab_kernel<NB>;
      auto ks = als_slab_solve2_kernel<NB>;
      if (int rc = set_max_lds(reinterpret_cast<const void *>(kg), lds)) return rc;
      if (int rc = set_max_lds(reinterpret_cast<const void *>(ks), (size_t)PairCfg<NB>::LDS_BYTES)) return rc;
      for (int64_t b0 = 0; b0 < nPrimal; b0 += rowSlabRows) {
        const int64_t cnt = std::min(rowSlabRows, nPrimal - b0);
        const int32_t first = (int32_t)(nSplitUnits + b0);
#ifdef YCNR_WITH_G32
        if (g32)
          hipLaunchKernelGGL(als_g32_rowslab_kernel, dim3((unsigned)std::min(cnt, cus)), dim3(kG32Threads), (size_t)G32Cfg::LDS_BYTES, stream, args,
                             rowSlabs, first, (int32_t)cnt);
        else
#endif
          hipLaunchKernelGGL(kg, dim3((unsigned)std::min(cnt, cus)), dim3(kWgThreads), lds, stream, args, rowSlabs, first, (int32_t)cnt);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(ks, dim3((unsigned)cnt), dim3(kPairThreads), (size_t)PairCfg<NB>::LDS_BYTES, stream, args, (const float *)rowSlabs, first);
        HIP_TRY(hipGetLastError());
      }
      pairDone = true;
    }
  }
  if (nPrimal > 0 && !pairDone) {
    hipLaunchKernelGGL(k1, dim3((unsigned)std::min(nPrimal, cus)), dim3(kWgThreads), lds, stream, args, (int32_t)nPrimal);
    HIP_TRY(hipGetLastError());
  }
  if (ev) HIP_TRY(hipEventRecord(ev[2], stream));
  if (dp.nPrimal >= 0 && !sideDuals) {  // (YCNR_FLAG_NO_OVERLAP, or too few rows to pay for the fork and the joins)
    DualPlan serial = dp;
    serial.nSide = 0;
    int rc = launch_duals<float>(args, serial, stream);
    if (rc) return rc;
  }
  if (sideDuals)
    for (int i = 0; i < dp.nSide; ++i) HIP_TRY(hipStreamWaitEvent(stream, dp.join[i], 0));
  if (ev) HIP_TRY(hipEventRecord(ev[3], stream));
  if (nSplit > 0) {
    hipLaunchKernelGGL(k2, dim3((unsigned)std::min(nSplit, cus)), dim3(kWgThreads), lds, stream, args, (int32_t)nSplit);
    HIP_TRY(hipGetLastError());
  }
  if (ev) HIP_TRY(hipEventRecord(ev[4], stream));
  return YCNR_OK;
}

int launch_step_big(const StepArgs<float> &args, int64_t nUnits, int64_t nSplitUnits, int64_t nSplit, hipStream_t stream,
                    hipEvent_t *ev, const DualPlan &dp, float *rowSlabs = nullptr, int64_t rowSlabRows = 0) {
  if (nUnits > 0x7fffffffLL) return fail(YCNR_ERR_UNSUPPORTED, "too many work units for one launch (%lld)", (long long)nUnits);
  const int64_t nPrimal = dp.nPrimal >= 0 ? dp.nPrimal : nUnits - nSplitUnits;
  switch ((args.k + 15) / 16) {
    case 9: return launch_wg_nb<9>(args, nSplitUnits, nPrimal, nSplit, stream, ev, dp, rowSlabs, rowSlabRows);
    case 10: return launch_wg_nb<10>(args, nSplitUnits, nPrimal, nSplit, stream, ev, dp, rowSlabs, rowSlabRows);
    case 11: return launch_wg_nb<11>(args, nSplitUnits, nPrimal, nSplit, stream, ev, dp, rowSlabs, rowSlabRows);
    case 12: return launch_wg_nb<12>(args, nSplitUnits, nPrimal, nSplit, stream, ev, dp, rowSlabs, rowSlabRows);
    case 13: return launch_wg_nb<13>(args, nSplitUnits, nPrimal, nSplit, stream, ev, dp, rowSlabs, rowSlabRows);
    case 14: return launch_wg_nb<14>(args, nSplitUnits, nPrimal, nSplit, stream, ev, dp, rowSlabs, rowSlabRows);
    case 15: return launch_wg_nb<15>(args, nSplitUnits, nPrimal, nSplit, stream, ev, dp, rowSlabs, rowSlabRows);
    case 16: return launch_wg_nb<16>(args, nSplitUnits, nPrimal, nSplit, stream, ev, dp, rowSlabs, rowSlabRows);
    default: return fail(YCNR_ERR_UNSUPPORTED, "factorsCount %d is outside the workgroup-per-row path", args.k);
  }
}

// any factorsCount (als_gen_kernels.hip.h): batches of [Gramians -> slabs] [slabs -> solve], then the dual classes
// (float32: rows of at most 176 ratings, whose n x n form does not depend on k)
template <typename T>
int launch_step_gen(const StepArgs<T> &args, const std::vector<GenBatch> &batches, hipStream_t stream, hipEvent_t *ev, const DualPlan &dp) {
  GenArgs<T> ga{args, (args.k + 15) / 16, 0, 0, 0};
  const bool panelLds = (size_t)ga.nb * 256 * sizeof(T) <= kGenPanelLdsMax;
  const size_t lds = gen_solve_lds_bytes(ga.nb, sizeof(T), panelLds);
  if (lds > 160 * 1024) return fail(YCNR_ERR_UNSUPPORTED, "factorsCount %d: the right-hand side does not fit a CU's LDS", args.k);
  const void *solveFn = panelLds ? reinterpret_cast<const void *>(als_gen_solve_kernel<T, true>) : reinterpret_cast<const void *>(als_gen_solve_kernel<T, false>);
  if (int rc = set_max_lds(solveFn, lds)) return rc;
  // the left-looking solve where a block of four rows fits the registers of eight waves: float32 up to k = 512, float64 up to 256 (beyond: the right-looking kernel)
  // (YCNR_GEN_RIGHT_LOOKING: the right-looking kernel everywhere, for A/B runs)
  static const bool rightOnly = getenv("YCNR_GEN_RIGHT_LOOKING") != nullptr;
  const void *leftFn = nullptr;
  int leftWaves = 8, leftPw = 4;
  if (!rightOnly) {
    if constexpr (sizeof(T) == 4) {
      const int maxc = (ga.nb + 7) / 8;
      leftFn = maxc <= 2 ? reinterpret_cast<const void *>(als_gen_solve_left_kernel<T, 2, 4, 8, true, 2>)
               : maxc <= 4 ? reinterpret_cast<const void *>(als_gen_solve_left_kernel<T, 4, 4, 8, true, 2>) : nullptr;
    } else if (ga.nb <= 16) {
      // (k = 256, 200 K x 20 K, per iteration: right-looking 301 ms; 4 rows x 8 waves as in float32, one workgroup per CU: 347;
      // 2 rows x 8 waves, two per CU: 277; 2 rows x 4 waves, three per CU: 253)
      leftFn = reinterpret_cast<const void *>(als_gen_solve_left_kernel<T, 4, 2, 4, false, 3>);
      leftWaves = 4, leftPw = 2;
    }
  }
  const size_t leftLds = gen_solve_left_lds_bytes(ga.nb, sizeof(T), leftPw);
  if (leftFn)
    if (int rc = set_max_lds(leftFn, leftLds)) return rc;
  // Gramian: 16-byte loads when every row of the fixed matrix starts on a 16-byte boundary, single elements otherwise;
  // the waves of a workgroup share the rectangles of a pass evenly; R ratings per panel: as many as two buffers of <= 72 KB
  // (two workgroups per CU) and the loader's slots allow, at least 4
  constexpr int V = 16 / (int)sizeof(T);
  const bool vec = args.k % V == 0 && (reinterpret_cast<uintptr_t>(args.fixed) & 15) == 0;
  const int nSq = gen_items(ga.nb, gen_rect_rows<T>(), gen_sqw<T>() * kGenSq), passes = (nSq + kGenGramMaxWaves - 1) / kGenGramMaxWaves;
  const int waves = std::max(2, (nSq + passes - 1) / passes), nthr = 64 * waves;
  const int P = gen_panel_pitch(ga.nb, sizeof(T));
  int slots = vec ? gen_loader_slots<T, V>() : gen_loader_slots<T, 1>();
  int R = 32;
  // float32 with 16-byte rows whose panel of 32 ratings fits one workgroup's LDS AND its loader slots (8 per thread of at most
  // eight waves: 32 (k / 4) <= 8 * 512, i.e. k <= 512): the products on the bf16 pipe (X6); 516 <= k keeps float32 MFMAs
  static const bool noGenX6 = getenv("YCNR_NO_GEN_X6") != nullptr;  // (A/B runs)
  const bool x6 = sizeof(T) == 4 && vec && !noGenX6 && gen_gram_lds_bytes(32, P, sizeof(T)) <= 150 * 1024 &&
                  (int64_t)32 * (args.k / V) <= (int64_t)8 * nthr;
  if (x6) slots = 8;
  else
    while (R > 4 && (gen_gram_lds_bytes(R, P, sizeof(T)) > 72 * 1024 || (int64_t)R * (args.k / (vec ? V : 1)) > (int64_t)slots * nthr)) R >>= 1;
  const size_t gramLds = gen_gram_lds_bytes(R, P, sizeof(T));
  if (gramLds > 150 * 1024 || (int64_t)R * (args.k / (vec ? V : 1)) > (int64_t)slots * nthr)
    return fail(YCNR_ERR_UNSUPPORTED, "factorsCount %d: a panel of four ratings does not fit a CU's LDS", args.k);
  const void *gramFn = vec ? reinterpret_cast<const void *>(als_gen_gram_kernel<T, V>) : reinterpret_cast<const void *>(als_gen_gram_kernel<T, 1>);
  if constexpr (sizeof(T) == 4)
    if (x6) gramFn = reinterpret_cast<const void *>(als_gen_gram_kernel<T, V, true>);
  if (int rc = set_max_lds(gramFn, gramLds)) return rc;
  if (ev) HIP_TRY(hipEventRecord(ev[0], stream));
  for (const GenBatch &b : batches) {
    ga.slabBase = b.slabBase;
    ga.firstUnit = b.slabBase;  // units of split rows are numbered like their slabs
    ga.firstSplit = b.firstSplit;
    {
      int Rv = R, Pv = P;
      void *gargs[] = {(void *)&ga, (void *)&Rv, (void *)&Pv};
      HIP_TRY(hipLaunchKernel(gramFn, dim3((unsigned)b.nSlabs), dim3(nthr), gargs, gramLds, stream));
    }
    HIP_TRY(hipGetLastError());
    if (leftFn) {
      void *kargs[] = {(void *)&ga};
      HIP_TRY(hipLaunchKernel(leftFn, dim3((unsigned)b.nSplit), dim3(64 * leftWaves), kargs, leftLds, stream));
    } else if (panelLds) {
      hipLaunchKernelGGL((als_gen_solve_kernel<T, true>), dim3((unsigned)b.nSplit), dim3(kGenThreads), lds, stream, ga);
    } else {
      hipLaunchKernelGGL((als_gen_solve_kernel<T, false>), dim3((unsigned)b.nSplit), dim3(kGenThreads), lds, stream, ga);
    }
    HIP_TRY(hipGetLastError());
  }
  if (ev) {
    HIP_TRY(hipEventRecord(ev[1], stream));
    HIP_TRY(hipEventRecord(ev[2], stream));
  }
  if constexpr (std::is_same<T, float>::value) {
    if (dp.nPrimal >= 0) {
      DualPlan serial = dp;
      serial.nSide = 0;
      StepArgs<float> a2 = args;
      a2.firstFused = 0;
      int rc = launch_duals<float>(a2, serial, stream);
      if (rc) return rc;
    }
  }
  if (ev) {
    HIP_TRY(hipEventRecord(ev[3], stream));
    HIP_TRY(hipEventRecord(ev[4], stream));
  }
  return YCNR_OK;
}

template <typename T>
int launch_step(const StepArgs<T> &args, int64_t nUnits, int64_t nSplitUnits, int64_t nSplit, hipStream_t stream,
                hipEvent_t *ev, bool ldsSolver = false, const DualPlan &dp = DualPlan(), bool edge = false, bool x6 = false) {
  if (nUnits > 0x7fffffffLL || nSplit > 0x7fffffffLL)
    return fail(YCNR_ERR_UNSUPPORTED, "too many work units for one launch (%lld)", (long long)nUnits);
  const int nb = (args.k + 15) / 16;
  switch (nb) {
    case 1: return ldsSolver ? launch_nb<T, 1, true>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6)
                             : launch_nb<T, 1, false>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6);
    case 2: return ldsSolver ? launch_nb<T, 2, true>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6)
                             : launch_nb<T, 2, false>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6);
    case 3: return ldsSolver ? launch_nb<T, 3, true>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6)
                             : launch_nb<T, 3, false>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6);
    case 4: return ldsSolver ? launch_nb<T, 4, true>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6)
                             : launch_nb<T, 4, false>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6);
    case 5: return ldsSolver ? launch_nb<T, 5, true>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6)
                             : launch_nb<T, 5, false>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6);
    case 6: return ldsSolver ? launch_nb<T, 6, true>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6)
                             : launch_nb<T, 6, false>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6);
    case 7: return ldsSolver ? launch_nb<T, 7, true>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6)
                             : launch_nb<T, 7, false>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6);
    case 8: return ldsSolver ? launch_nb<T, 8, true>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6)
                             : launch_nb<T, 8, false>(args, nUnits, nSplitUnits, nSplit, stream, ev, dp, edge, x6);
    default:
      return fail(YCNR_ERR_UNSUPPORTED, "factorsCount %d > %d is not supported by this build", args.k,
                  kMaxFactors);
  }
}

int slab_nb(int k) { return (k + 15) / 16; }

// float32, k = 16 m + 4 (m >= 1), k <= 128: the last 4 columns of the Gramian go to the VALU
bool use_valu_edge(const ycnr_als_options &o) {
  return o.dtype == YCNR_F32 && !(o.flags & YCNR_FLAG_NO_VALU_EDGE) && o.factorsCount >= 20 && o.factorsCount <= kMaxFactors &&
         o.factorsCount % 16 == 4;
}

// split chunks on the bf16 matrix pipe (exact 3-way split, six products): float32, one-wave
// kernels, fixed matrix addressable by a 32-bit buffer offset
bool use_slab_x6(const ycnr_als_options &o, int side) {
  const int64_t fixedRows = side == YCNR_BY_USER ? o.totalItemsCount : o.totalUsersCount;
  return o.dtype == YCNR_F32 && !(o.flags & YCNR_FLAG_NO_BF16X6) && o.factorsCount <= kMaxFactors &&
         fixedRows * o.factorsCount * 4 < ((int64_t)1 << 31);
}

// The fused row kernel on pre-split planes (GramX6P): float32 bf16x6 path, k % 4 == 0, k <= 124 with a padded column for
// the right-hand side (k % 16 != 0), and a plane matrix (96 bytes per 16-column block and row) that stays cache-resident.
constexpr int64_t kPlanesMaxBytes = (int64_t)64 << 20;
bool use_planes(const ycnr_als_options &o, int side) {
  static const bool off = getenv("YCNR_NO_X6P") != nullptr;
  const int64_t fixedRows = side == YCNR_BY_USER ? o.totalItemsCount : o.totalUsersCount;
  const int k = o.factorsCount;
  return !off && use_slab_x6(o, side) && !(o.flags & YCNR_FLAG_LDS_SOLVER) && k % 4 == 0 && k <= 124 && k % 16 != 0 &&
         fixedRows * planes_row_bytes(slab_nb(k), planes_pack(k)) <= kPlanesMaxBytes;
}

// registers (x 64 lanes x sizeof(T)) one split unit writes
int64_t slab_regs(const ycnr_als_options &o, int side) {
  const int nb = slab_nb(o.factorsCount);
  if (use_valu_edge(o) && !use_slab_x6(o, side)) {
    const int nbm = nb - 1;
    return tile_count(nbm) * 4 + nbm + nbm * 4 + 8;
  }
  return tile_count(nb) * 4 + nb;
}

// Longest row solved in dual form (0 = never): float32 MFMA solver only, rows of 16-byte
// multiples, and strictly fewer 16-blocks than the primal form would use.
int dual_max_ratings(const ycnr_als_options &o) {
  // (k > 128 with k % 4 != 0 runs on rows padded to a multiple of 4: see kPad)
  if (o.dtype != YCNR_F32 || (o.flags & (YCNR_FLAG_LDS_SOLVER | YCNR_FLAG_NO_DUAL)) || (o.factorsCount % 4 != 0 && o.factorsCount <= kMaxFactors)) return 0;
  const int nb = slab_nb(o.factorsCount);
  // k > 128: every row that is not dual goes through slabs and the 4-wave LDS solve, whose cost
  // grows with k^3; an n x n problem with n <= 176 still fits one wave's registers
  int most = o.factorsCount > kMaxFactors ? kMaxDualBlocks : kMaxDualBlocksSmallK;
  if (const char *e = getenv("YCNR_DUAL_MAX_BLOCKS")) most = std::max(0, std::min(atoi(e), o.factorsCount > kMaxFactors ? kMaxDualBlocks : kMaxDualBlocksSmallK));  // experiments
  return 16 * std::min(most, nb - 1);
}

// copy `bytes` from src (host or device) to a device destination
int copy_in(void *dst, const void *src, size_t bytes, int memKind, hipStream_t stream) {
  if (bytes == 0) return YCNR_OK;
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, memKind == YCNR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                         stream));
  return YCNR_OK;
}

// out[i] = dSrc[pos[i]] for a host list of positions (one small gather kernel + two copies)
// out[q] = first position p in [beg[q], end[q]) with indx[p] >= key[q] (end[q] if none); rows sorted by column id
__global__ void lower_bound_i32_kernel(const int32_t *indx, const int64_t *beg, const int64_t *end, const int32_t *key,
                                       int64_t *out, int64_t n) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  int64_t lo = beg[q], hi = end[q];
  const int32_t kq = key[q];
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (indx[mid] < kq) lo = mid + 1;
    else hi = mid;
  }
  out[q] = lo;
}

int lower_bound_i32(const int32_t *dIndx, const std::vector<int64_t> &beg, const std::vector<int64_t> &end,
                    const std::vector<int32_t> &key, std::vector<int64_t> &out, hipStream_t stream) {
  const size_t n = key.size();
  out.resize(n);
  if (n == 0) return YCNR_OK;
  char *d = nullptr;
  HIP_TRY(hipMalloc(&d, n * 28));
  int64_t *dBeg = (int64_t *)d, *dEnd = dBeg + n, *dOut = dEnd + n;
  int32_t *dKey = (int32_t *)(dOut + n);
  hipError_t e = hipMemcpyAsync(dBeg, beg.data(), n * 8, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dEnd, end.data(), n * 8, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dKey, key.data(), n * 4, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(lower_bound_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dIndx, dBeg, dEnd, dKey, dOut, (int64_t)n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out.data(), dOut, n * 8, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(YCNR_ERR_HIP, "lower_bound_i32: %s", hipGetErrorString(e));
  return YCNR_OK;
}

// validate 0 <= indx[i] < limit on the device
int check_index_range(const int32_t *dIndx, int64_t n, int64_t limit, hipStream_t stream,
                      const char *what) {
  if (n == 0) return YCNR_OK;
  int32_t *dmm = nullptr;
  HIP_TRY(hipMalloc(&dmm, 2 * sizeof(int32_t)));
  int32_t init[2] = {INT32_MIN, INT32_MAX};
  hipError_t e = hipMemcpyAsync(dmm, init, sizeof init, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) {
    const int blocks = (int)std::min<int64_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(max_i32_kernel, dim3(blocks), dim3(256), 0, stream, dIndx, n, dmm, dmm + 1);
    e = hipGetLastError();
  }
  int32_t got[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(got, dmm, sizeof got, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(dmm);
  if (e != hipSuccess) return fail(YCNR_ERR_HIP, "index range check failed: %s", hipGetErrorString(e));
  if (got[1] < 0 || (int64_t)got[0] >= limit)
    return fail(YCNR_ERR_INVALID, "%s: column id out of range (min %d, max %d, rows of the opposite side %lld)",
                what, got[1], got[0], (long long)limit);
  return YCNR_OK;
}

#include "comm_impl.hip.h"

// one pipelined piece of a side's row shard: its ratings, its work-unit schedule and the events
// that time its kernels and its exchange
struct Part {
  Ratings R;
  Schedule S;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ready = nullptr, x0 = nullptr, x1 = nullptr;
  // fork / join of the piece's dual-class launches on the handle's side streams, and "all kernels of the piece done":
  // per piece, because two pieces are in flight at a time (on the handle's two piece streams)
  hipEvent_t fork = nullptr, join[kSideStreams] = {}, done = nullptr, slabJoin = nullptr;
  hipError_t create_events() {
    hipError_t e = hipSuccess;
    for (int i = 0; i < 5 && e == hipSuccess; ++i) e = hipEventCreate(&ev[i]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&fork, hipEventDisableTiming);
    for (int i = 0; i < kSideStreams && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&join[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&slabJoin, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&x0);
    if (e == hipSuccess) e = hipEventCreate(&x1);
    return e;
  }
  void release() {
    R.release();
    S.release();
    for (int i = 0; i < 5; ++i)
      if (ev[i]) (void)hipEventDestroy(ev[i]);
    if (ready) (void)hipEventDestroy(ready);
    if (x0) (void)hipEventDestroy(x0);
    if (x1) (void)hipEventDestroy(x1);
    if (fork) (void)hipEventDestroy(fork);
    if (done) (void)hipEventDestroy(done);
    if (slabJoin) (void)hipEventDestroy(slabJoin);
    for (int i = 0; i < kSideStreams; ++i) {
      if (join[i]) (void)hipEventDestroy(join[i]);
      join[i] = nullptr;
    }
    for (int i = 0; i < 5; ++i) ev[i] = nullptr;
    ready = x0 = x1 = fork = done = slabJoin = nullptr;
  }
};

// A side whose half-step is sharded by BANDS OF COLUMNS (ycnr_als_set_ratings_banded): what the upload leaves besides the
// side's single Part (local ratings, chunk units, the arena of band + temporary slabs, the owned rows as SplitRow list)
struct BandedSide {
  bool on = false;
  int nBands = 0, world = 1, rank = 0;
  std::vector<int64_t> rankBands, ownerBounds;    // world + 1 band indices / row ids
  std::vector<int64_t> groupUnit0, groupUnits;     // per owner group: its chunk units [unit0, +n) of the Part's unit list
  std::vector<int64_t> groupSum0, groupSums;       // ... and its multi-chunk (row, band) segments [sum0, +n) of dSums
  std::vector<int64_t> groupSlab0;                 // first band slab of group g in the arena (rows of g x local bands)
  std::vector<int64_t> recvSlab0;                  // first slab of source rank r in dRecv (owned rows x bands of r; own rank: unused)
  SlabSum *dSums = nullptr;
  void *dRecv = nullptr;
  const void **dTable = nullptr;                   // owned rows x nBands slab pointers (null: no rating of the row in that band)
  int64_t bandSlabs = 0, tmpSlabs = 0, recvSlabs = 0, ownedSolved = 0, nSums = 0;
  int64_t slabElems = 0;
  hipEvent_t evRecv = nullptr, evStage[kFewSlabs] = {};  // stage s: the group computed in it is complete (on its piece stream)
  void release() {
    if (dSums) (void)hipFree(dSums);
    if (dRecv) (void)hipFree(dRecv);
    if (dTable) (void)hipFree(dTable);
    if (evRecv) (void)hipEventDestroy(evRecv);
    for (hipEvent_t &e : evStage) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
    dSums = nullptr;
    dRecv = nullptr;
    dTable = nullptr;
    evRecv = nullptr;
    on = false;
  }
};

}  // namespace

struct ycnr_als {
  ycnr_als_options opt{};
  hipStream_t ownStream = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t sideStream[kSideStreams] = {};  // dual classes next to the row kernel (DualPlan)
  // A sharded side is solved in pieces (the exchange of one travels while the next is solved).  The pieces are
  // independent (different rows of the solved matrix), so they alternate over two streams that fork from and join the
  // step's stream: the kernels of piece c + 1 fill the tails of piece c's (one GPU's eighth of the MAL-scale user side,
  // 4 pieces in stream order: 2.5 ms against 2.0 ms in one piece).
  hipStream_t pieceStream[2] = {};
  hipEvent_t evStepStart = nullptr;
  // Small uploads: the launches of a half-step captured once and replayed (hipGraph).  state: 0 = not run yet (the
  // first half-step runs launch by launch: it sets function attributes), 1 = capture at the next one, 2 = exec is
  // valid, -1 = capture failed once, stay with launches.  Dropped whenever something a kernel argument points
  // at changes (upload, bound matrix, stream).
  struct GraphSlot {
    hipGraphExec_t exec = nullptr;
    int state = 0;
  } graph[2];
  bool graphRun = false;  // the half-step in flight was a graph launch: only ev[0] / ev[4] of its piece are recorded
  void drop_graphs() {
    for (GraphSlot &g : graph) {
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
      g.state = 0;
    }
  }
  void *factors[2] = {nullptr, nullptr};
  bool ownFactors[2] = {false, false};
  // GramX6P: the fixed matrix of a half-step split into bf16 planes once (als_split_planes_kernel), where it is small
  // enough to stay cache-resident (the user half-step: the item matrix); planes[s] belongs to factors[s]
  unsigned short *planes[2] = {nullptr, nullptr};
  bool planesValid[2] = {false, false};  // the half-step in flight split factors[s] into planes[s]
  int kPad = 0;                          // != 0: float32, factorsCount % 4 != 0: the kernels work on copies padded to a multiple of 4
  ycnr_als_options copt{};               // opt as the kernels see it: factorsCount = kPad when padded (schedules, kernel choice, slab sizes)
  float *padded[2] = {nullptr, nullptr};  // [rows x kPad] copies the kernels of that case work on
  bool autoChunk = false;  // options.chunkRatings was 0: sized per upload (auto_chunk)
  std::vector<Part> parts[2];      // the side's local row shard, cut into pipelined pieces (usually one)
  BandedSide banded[2];            // ycnr_als_set_ratings_banded: the side's half-step is sharded by bands of columns
  bool deferExchange[2] = {false, false};  // ycnr_als_defer_exchange: the side's half-steps leave the solved rows where they are
  std::vector<int64_t> bounds[2];   // sharded upload: row bounds of every rank's pieces, world x (nParts + 1)
  Comm comm;                        // exchange step of the multi-GPU path (comm_impl.hip.h)
  hipEvent_t evComputeEnd = nullptr;
  bool exchangedInStep = false;
  Ratings rmse[2];
  double lastRmseMs = -1.0;  // device time of the last ycnr_als_rmse (its kernel, HIP events on the handle's stream)
  ErrInfo *dErr = nullptr;
  // The error record is never reset on the device: its count only grows, the host remembers what it has seen
  // (errSeen) and gets the record through an 8-byte copy into page-locked memory at the end of every half-step
  // (before: a memset kernel in front of every half-step and a synchronous copy behind it -- 30 us of a 90 us
  // half-step at the ML-100k shape).
  ErrInfo *hErr = nullptr;
  int32_t errSeen = 0;
  void *dZeros = nullptr;  // the zero "factor row" read for ratings past a unit's end
  ycnr_als_step_info info{};  // of the half-step being enqueued / completed
  bool infoPending = false;   // some half-step has been enqueued and not completed by ycnr_als_sync
  int infoSide = 0;           // side of the last half-step enqueued
  // Half-steps of BOTH sides may be in flight (ycnr_als_step_async twice, then ycnr_als_sync: a whole iteration without the
  // host in between).  What ycnr_als_sync needs of each, in the order they were enqueued:
  struct Pending {
    ycnr_als_step_info info{};
    bool graphRun = false, exchanged = false;
  } pend[2];
  int pendOrder[2] = {0, 0}, nPend = 0;
  ycnr_als_step_info infoOf[2] = {};  // the last completed half-step of each side

  int64_t rows(int side) const { return side == YCNR_BY_USER ? opt.totalUsersCount : opt.totalItemsCount; }
  size_t ts() const { return tsize(opt.dtype); }
};

extern "C" {

const char *ycnr_last_error(void) { return g_last_error.c_str(); }

int ycnr_version(void) { return YCNR_ALS_ABI_VERSION; }

int ycnr_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(YCNR_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  return n;
}

// ---- N1: split + stats (prep_kernels.hip.h) ----
namespace {
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};
// a pair of timing events that is destroyed on every return path
struct EvPair {
  hipEvent_t a = nullptr, b = nullptr;
  hipError_t create() {
    hipError_t e = hipEventCreate(&a);
    return e == hipSuccess ? hipEventCreate(&b) : e;
  }
  ~EvPair() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
};
int timed(hipEvent_t e0, hipEvent_t e1, double *ms) {
  HIP_TRY(hipEventRecord(e1, nullptr));
  HIP_TRY(hipEventSynchronize(e1));
  float f = 0;
  HIP_TRY(hipEventElapsedTime(&f, e0, e1));
  if (ms) *ms = f;
  return YCNR_OK;
}
}  // namespace

int ycnr_split_to_sets(int64_t rows, const int64_t *rowPtr, int8_t *types, const int32_t pcts[3], uint32_t seed,
                       double *deviceMs) {
  if (rows < 0 || !rowPtr || !pcts) return fail(YCNR_ERR_INVALID, "ycnr_split_to_sets: null argument");
  if (pcts[0] < 0 || pcts[1] < 0 || pcts[2] < 0 || pcts[0] + pcts[1] + pcts[2] != 100)
    return fail(YCNR_ERR_INVALID, "ycnr_split_to_sets: dataSetDistr %d/%d/%d does not sum to 100", pcts[0], pcts[1], pcts[2]);
  if (rows == 0) return YCNR_OK;
  if (rowPtr[0] != 0) return fail(YCNR_ERR_INVALID, "ycnr_split_to_sets: rowPtr[0] != 0");
  const int64_t nnz = rowPtr[rows];
  if (nnz < 0 || (nnz > 0 && !types)) return fail(YCNR_ERR_INVALID, "ycnr_split_to_sets: bad nnz / null types");
  if (rows >= ((int64_t)1 << 31)) return fail(YCNR_ERR_UNSUPPORTED, "ycnr_split_to_sets: more than 2^31 rows");
  for (int64_t r = 0; r < rows; ++r) {
    if (rowPtr[r + 1] < rowPtr[r]) return fail(YCNR_ERR_INVALID, "ycnr_split_to_sets: rowPtr decreases at row %lld", (long long)r);
    if (rowPtr[r + 1] - rowPtr[r] >= ((int64_t)1 << 31)) return fail(YCNR_ERR_UNSUPPORTED, "row %lld too long", (long long)r);
  }
  if (nnz == 0) return YCNR_OK;
  DevBuf dPtr, dTypes;
  HIP_TRY(hipMalloc(&dPtr.p, (size_t)(rows + 1) * 8));
  HIP_TRY(hipMalloc(&dTypes.p, (size_t)nnz));
  HIP_TRY(hipMemcpy(dPtr.p, rowPtr, (size_t)(rows + 1) * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dTypes.p, types, (size_t)nnz, hipMemcpyHostToDevice));
  EvPair evp;
  HIP_TRY(evp.create());
  const hipEvent_t e0 = evp.a, e1 = evp.b;
  HIP_TRY(hipEventRecord(e0, nullptr));
  // three classes of rows: up to kSplitRankRow ratings one wave ranks every rating (O(n^2 / 64), cheapest for short rows);
  // longer ones find the two thresholds by bisection instead -- one wave up to kSplitShortRow ratings, a 1024-thread
  // workgroup beyond (whose 60 KB of LDS would leave two waves per CU for everybody, hence the separate launch)
  std::vector<int32_t> lists[3];
  for (int64_t r = 0; r < rows; ++r) {
    const int64_t n = rowPtr[r + 1] - rowPtr[r];
    if (n > 0) lists[n > kSplitShortRow ? 1 : n > kSplitRankRow ? 2 : 0].push_back((int32_t)r);
  }
  // longest first within the long classes (they set the tail)
  for (int q = 1; q < 3; ++q)
    std::stable_sort(lists[q].begin(), lists[q].end(),
                     [&](int32_t x, int32_t y) { return rowPtr[x + 1] - rowPtr[x] > rowPtr[y + 1] - rowPtr[y]; });
  DevBuf dList[3];
  for (int q = 0; q < 3; ++q) {
    if (lists[q].empty()) continue;
    HIP_TRY(hipMalloc(&dList[q].p, lists[q].size() * 4));
    HIP_TRY(hipMemcpy(dList[q].p, lists[q].data(), lists[q].size() * 4, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipEventRecord(e0, nullptr));  // the kernels only (lists are part of the upload)
  if (!lists[1].empty())
    hipLaunchKernelGGL((split_to_sets_select_kernel<kSplitLdsKeys, 1024>), dim3((unsigned)lists[1].size()), dim3(1024), 0, nullptr, (const int64_t *)dPtr.p,
                       (const int32_t *)dList[1].p, (int64_t)lists[1].size(), (int8_t *)dTypes.p, (int)pcts[0], (int)pcts[1], seed);
  if (!lists[2].empty())
    hipLaunchKernelGGL((split_to_sets_select_kernel<kSplitShortRow, 64>), dim3((unsigned)std::min<size_t>(lists[2].size(), (size_t)1 << 20)), dim3(64), 0,
                       nullptr, (const int64_t *)dPtr.p, (const int32_t *)dList[2].p, (int64_t)lists[2].size(), (int8_t *)dTypes.p,
                       (int)pcts[0], (int)pcts[1], seed);
  if (!lists[0].empty())
    hipLaunchKernelGGL((split_to_sets_kernel<kSplitRankRow, 64>), dim3((unsigned)std::min<size_t>(lists[0].size(), (size_t)1 << 20)), dim3(64), 0,
                       nullptr, (const int64_t *)dPtr.p, (const int32_t *)dList[0].p, (int64_t)lists[0].size(), (int8_t *)dTypes.p,
                       (int)pcts[0], (int)pcts[1], seed);
  hipError_t le = hipGetLastError();
  int rc = le == hipSuccess ? timed(e0, e1, deviceMs) : fail(YCNR_ERR_HIP, "split_to_sets launch: %s", hipGetErrorString(le));
  if (rc) return rc;
  HIP_TRY(hipMemcpy(types, dTypes.p, (size_t)nnz, hipMemcpyDeviceToHost));
  return YCNR_OK;
}

int ycnr_rating_stats(int dtype, int64_t rows, const int64_t *rowPtr, const void *vals, const int8_t *types,
                      int32_t *cnt, double *sum, double *deviceMs) {
  if (rows < 0 || !rowPtr || !cnt || !sum) return fail(YCNR_ERR_INVALID, "ycnr_rating_stats: null argument");
  if (dtype != YCNR_F32 && dtype != YCNR_F64) return fail(YCNR_ERR_INVALID, "bad dtype %d", dtype);
  if (rows == 0) return YCNR_OK;
  if (rowPtr[0] != 0) return fail(YCNR_ERR_INVALID, "ycnr_rating_stats: rowPtr[0] != 0");
  const int64_t nnz = rowPtr[rows];
  if (nnz < 0 || (nnz > 0 && !vals)) return fail(YCNR_ERR_INVALID, "ycnr_rating_stats: bad nnz / null vals");
  const size_t ts = tsize(dtype);
  DevBuf dPtr, dVals, dTypes, dCnt, dSum;
  HIP_TRY(hipMalloc(&dPtr.p, (size_t)(rows + 1) * 8));
  HIP_TRY(hipMalloc(&dVals.p, std::max<size_t>((size_t)nnz * ts, 8)));
  HIP_TRY(hipMalloc(&dCnt.p, (size_t)rows * 4));
  HIP_TRY(hipMalloc(&dSum.p, (size_t)rows * 8));
  HIP_TRY(hipMemcpy(dPtr.p, rowPtr, (size_t)(rows + 1) * 8, hipMemcpyHostToDevice));
  if (nnz) HIP_TRY(hipMemcpy(dVals.p, vals, (size_t)nnz * ts, hipMemcpyHostToDevice));
  if (types && nnz) {
    HIP_TRY(hipMalloc(&dTypes.p, (size_t)nnz));
    HIP_TRY(hipMemcpy(dTypes.p, types, (size_t)nnz, hipMemcpyHostToDevice));
  }
  EvPair evp;
  HIP_TRY(evp.create());
  const hipEvent_t e0 = evp.a, e1 = evp.b;
  HIP_TRY(hipEventRecord(e0, nullptr));
  constexpr int64_t kLongRow = 2048;  // longer rows: a workgroup per segment of kStatsSegment ratings, then the segments in order
  std::vector<int32_t> longRows, segRow;
  std::vector<int64_t> segBeg, firstSeg;
  for (int64_t r = 0; r < rows; ++r) {
    const int64_t n = rowPtr[r + 1] - rowPtr[r];
    if (n <= kLongRow) continue;
    longRows.push_back((int32_t)r);
    firstSeg.push_back((int64_t)segRow.size());
    for (int64_t b = rowPtr[r]; b < rowPtr[r + 1]; b += kStatsSegment) {
      segRow.push_back((int32_t)r);
      segBeg.push_back(b);
    }
  }
  firstSeg.push_back((int64_t)segRow.size());
  DevBuf dLong, dSegRow, dSegBeg, dFirst, dPartCnt, dPartSum;
  if (!longRows.empty()) {
    HIP_TRY(hipMalloc(&dLong.p, longRows.size() * 4));
    HIP_TRY(hipMalloc(&dSegRow.p, segRow.size() * 4));
    HIP_TRY(hipMalloc(&dSegBeg.p, segBeg.size() * 8));
    HIP_TRY(hipMalloc(&dFirst.p, firstSeg.size() * 8));
    HIP_TRY(hipMalloc(&dPartCnt.p, segRow.size() * 4));
    HIP_TRY(hipMalloc(&dPartSum.p, segRow.size() * 8));
    HIP_TRY(hipMemcpy(dLong.p, longRows.data(), longRows.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dSegRow.p, segRow.data(), segRow.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dSegBeg.p, segBeg.data(), segBeg.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dFirst.p, firstSeg.data(), firstSeg.size() * 8, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipEventRecord(e0, nullptr));
  const unsigned blocks = (unsigned)((rows * 16 + 255) / 256);
  const unsigned nSeg = (unsigned)segRow.size(), nLong = (unsigned)longRows.size();
  if (dtype == YCNR_F32) {
    if (nLong)  // (first: its workgroups are the long ones)
      hipLaunchKernelGGL(rating_stats_long_kernel<float>, dim3(nSeg), dim3(256), 0, nullptr, (const int64_t *)dPtr.p, (const int32_t *)dSegRow.p,
                         (const int64_t *)dSegBeg.p, (const float *)dVals.p, (const int8_t *)dTypes.p, (int32_t *)dPartCnt.p, (double *)dPartSum.p);
    hipLaunchKernelGGL(rating_stats_kernel<float>, dim3(blocks), dim3(256), 0, nullptr, (const int64_t *)dPtr.p, rows,
                       (const float *)dVals.p, (const int8_t *)dTypes.p, (int32_t *)dCnt.p, (double *)dSum.p, kLongRow);
  } else {
    if (nLong)
      hipLaunchKernelGGL(rating_stats_long_kernel<double>, dim3(nSeg), dim3(256), 0, nullptr, (const int64_t *)dPtr.p, (const int32_t *)dSegRow.p,
                         (const int64_t *)dSegBeg.p, (const double *)dVals.p, (const int8_t *)dTypes.p, (int32_t *)dPartCnt.p, (double *)dPartSum.p);
    hipLaunchKernelGGL(rating_stats_kernel<double>, dim3(blocks), dim3(256), 0, nullptr, (const int64_t *)dPtr.p, rows,
                       (const double *)dVals.p, (const int8_t *)dTypes.p, (int32_t *)dCnt.p, (double *)dSum.p, kLongRow);
  }
  if (nLong)
    hipLaunchKernelGGL(rating_stats_combine_kernel, dim3((nLong + 255) / 256), dim3(256), 0, nullptr, (const int32_t *)dLong.p, (const int64_t *)dFirst.p,
                       (int64_t)nLong, (const int32_t *)dPartCnt.p, (const double *)dPartSum.p, (int32_t *)dCnt.p, (double *)dSum.p);
  hipError_t le = hipGetLastError();
  int rc = le == hipSuccess ? timed(e0, e1, deviceMs) : fail(YCNR_ERR_HIP, "rating_stats launch: %s", hipGetErrorString(le));
  if (rc) return rc;
  HIP_TRY(hipMemcpy(cnt, dCnt.p, (size_t)rows * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(sum, dSum.p, (size_t)rows * 8, hipMemcpyDeviceToHost));
  return YCNR_OK;
}

// ---- N2: CSR from triplets / transpose ----
namespace {
int bits_for(int64_t n) {
  int b = 1;
  while (b < 32 && ((int64_t)1 << b) < n) ++b;
  return b;
}
// makeKeys launches the kernel that writes keys / pos on the device; then sorts and unpacks into host arrays.  Every
// buffer is allocated BEFORE the first timing event: the events bracket kernels only.  (Round 3 allocated six buffers
// between the two events; the host time of those hipMalloc calls -- 10 ... 110 ms depending on the box -- was reported as
// device time of the sort.)
int sort_and_unpack(int dtype, int64_t n, int64_t outRows, int64_t outCols, uint64_t *dKeys, uint32_t *dPos, const void *dVals,
                    int64_t *rowPtr, int32_t *indx, void *outVals, const std::function<void()> &makeKeys, double *deviceMs) {
  const size_t ts = tsize(dtype);
  DevBuf dKeys2, dPos2, dTmp, dIndx, dOutVals, dPtr;
  HIP_TRY(hipMalloc(&dKeys2.p, (size_t)n * 8));
  HIP_TRY(hipMalloc(&dPos2.p, (size_t)n * 4));
  HIP_TRY(hipMalloc(&dIndx.p, (size_t)n * 4));
  HIP_TRY(hipMalloc(&dOutVals.p, (size_t)n * ts));
  HIP_TRY(hipMalloc(&dPtr.p, (size_t)(outRows + 1) * 8));
  size_t tmpBytes = 0;
  const int colBits = bits_for(outCols), endBit = colBits + bits_for(outRows);
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, dKeys, (uint64_t *)dKeys2.p, dPos, (uint32_t *)dPos2.p, (int)n, 0, endBit));
  HIP_TRY(hipMalloc(&dTmp.p, std::max<size_t>(tmpBytes, 8)));
  EvPair evp;
  HIP_TRY(evp.create());
  const hipEvent_t e0 = evp.a, e1 = evp.b;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipEventRecord(e0, nullptr));
  makeKeys();
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(dTmp.p, tmpBytes, dKeys, (uint64_t *)dKeys2.p, dPos, (uint32_t *)dPos2.p, (int)n, 0, endBit));
  const unsigned b256 = (unsigned)((n + 255) / 256);
  if (dtype == YCNR_F32)
    hipLaunchKernelGGL(unpack_sorted_kernel<float>, dim3(b256), dim3(256), 0, nullptr, (const uint64_t *)dKeys2.p, (const uint32_t *)dPos2.p,
                       (const float *)dVals, n, colBits, (int32_t *)dIndx.p, (float *)dOutVals.p);
  else
    hipLaunchKernelGGL(unpack_sorted_kernel<double>, dim3(b256), dim3(256), 0, nullptr, (const uint64_t *)dKeys2.p, (const uint32_t *)dPos2.p,
                       (const double *)dVals, n, colBits, (int32_t *)dIndx.p, (double *)dOutVals.p);
  hipLaunchKernelGGL(row_ptr_kernel, dim3((unsigned)((outRows + 1 + 255) / 256)), dim3(256), 0, nullptr, (const uint64_t *)dKeys2.p, n, outRows, colBits,
                     (int64_t *)dPtr.p);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) return fail(YCNR_ERR_HIP, "csr build: %s", hipGetErrorString(le));
  int rc = timed(e0, e1, deviceMs);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(rowPtr, dPtr.p, (size_t)(outRows + 1) * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(indx, dIndx.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(outVals, dOutVals.p, (size_t)n * ts, hipMemcpyDeviceToHost));
  return YCNR_OK;
}
}  // namespace

int ycnr_csr_from_triplets(int dtype, int64_t n, const int32_t *rowIdx, const int32_t *colIdx, const void *vals,
                           int64_t rows, int64_t cols, int64_t *rowPtr, int32_t *indx, void *outVals, double *deviceMs) {
  if (dtype != YCNR_F32 && dtype != YCNR_F64) return fail(YCNR_ERR_INVALID, "bad dtype %d", dtype);
  if (n < 0 || rows < 0 || cols < 0 || !rowPtr) return fail(YCNR_ERR_INVALID, "ycnr_csr_from_triplets: bad sizes / null rowPtr");
  if (rows >= ((int64_t)1 << 31) || cols >= ((int64_t)1 << 31) || n >= ((int64_t)1 << 31))
    return fail(YCNR_ERR_UNSUPPORTED, "ycnr_csr_from_triplets: more than 2^31 rows, columns or ratings");
  if (n == 0) {
    for (int64_t r = 0; r <= rows; ++r) rowPtr[r] = 0;
    return YCNR_OK;
  }
  if (!rowIdx || !colIdx || !vals || !indx || !outVals) return fail(YCNR_ERR_INVALID, "ycnr_csr_from_triplets: null array");
  for (int64_t q = 0; q < n; ++q)
    if (rowIdx[q] < 0 || rowIdx[q] >= rows || colIdx[q] < 0 || colIdx[q] >= cols)
      return fail(YCNR_ERR_INVALID, "triplet %lld: (%d, %d) outside %lld x %lld", (long long)q, rowIdx[q], colIdx[q], (long long)rows,
                  (long long)cols);
  const size_t ts = tsize(dtype);
  DevBuf dR, dC, dV, dKeys, dPos;
  HIP_TRY(hipMalloc(&dR.p, (size_t)n * 4));
  HIP_TRY(hipMalloc(&dC.p, (size_t)n * 4));
  HIP_TRY(hipMalloc(&dV.p, (size_t)n * ts));
  HIP_TRY(hipMalloc(&dKeys.p, (size_t)n * 8));
  HIP_TRY(hipMalloc(&dPos.p, (size_t)n * 4));
  HIP_TRY(hipMemcpy(dR.p, rowIdx, (size_t)n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dC.p, colIdx, (size_t)n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dV.p, vals, (size_t)n * ts, hipMemcpyHostToDevice));
  auto makeKeys = [&] {
    hipLaunchKernelGGL(make_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const int32_t *)dR.p, (const int32_t *)dC.p, n,
                       bits_for(cols), (uint64_t *)dKeys.p, (uint32_t *)dPos.p);
  };
  return sort_and_unpack(dtype, n, rows, cols, (uint64_t *)dKeys.p, (uint32_t *)dPos.p, dV.p, rowPtr, indx, outVals, makeKeys, deviceMs);
}

int ycnr_csr_transpose(int dtype, int64_t rows, int64_t cols, const int64_t *rowPtr, const int32_t *indx, const void *vals,
                       int64_t *outPtr, int32_t *outIndx, void *outVals, double *deviceMs) {
  if (dtype != YCNR_F32 && dtype != YCNR_F64) return fail(YCNR_ERR_INVALID, "bad dtype %d", dtype);
  if (rows < 0 || cols < 0 || !rowPtr || !outPtr) return fail(YCNR_ERR_INVALID, "ycnr_csr_transpose: bad sizes / null rowPtr");
  if (rowPtr[0] != 0) return fail(YCNR_ERR_INVALID, "ycnr_csr_transpose: rowPtr[0] != 0");
  const int64_t n = rowPtr[rows];
  if (rows >= ((int64_t)1 << 31) || cols >= ((int64_t)1 << 31) || n >= ((int64_t)1 << 31))
    return fail(YCNR_ERR_UNSUPPORTED, "ycnr_csr_transpose: more than 2^31 rows, columns or ratings");
  for (int64_t r = 0; r < rows; ++r)
    if (rowPtr[r + 1] < rowPtr[r]) return fail(YCNR_ERR_INVALID, "ycnr_csr_transpose: rowPtr decreases at row %lld", (long long)r);
  if (n == 0) {
    for (int64_t c = 0; c <= cols; ++c) outPtr[c] = 0;
    return YCNR_OK;
  }
  if (!indx || !vals || !outIndx || !outVals) return fail(YCNR_ERR_INVALID, "ycnr_csr_transpose: null array");
  for (int64_t q = 0; q < n; ++q)
    if (indx[q] < 0 || indx[q] >= cols) return fail(YCNR_ERR_INVALID, "entry %lld: column %d outside %lld", (long long)q, indx[q], (long long)cols);
  const size_t ts = tsize(dtype);
  DevBuf dP, dI, dV, dPos, dRowOf, dKeys2, dPos2, dTmp, dIndx, dOutVals, dPtr;
  HIP_TRY(hipMalloc(&dP.p, (size_t)(rows + 1) * 8));
  HIP_TRY(hipMalloc(&dI.p, (size_t)n * 4));
  HIP_TRY(hipMalloc(&dV.p, (size_t)n * ts));
  HIP_TRY(hipMalloc(&dPos.p, (size_t)n * 4));
  HIP_TRY(hipMalloc(&dRowOf.p, (size_t)n * 4));
  HIP_TRY(hipMalloc(&dKeys2.p, (size_t)n * 4));
  HIP_TRY(hipMalloc(&dPos2.p, (size_t)n * 4));
  HIP_TRY(hipMalloc(&dIndx.p, (size_t)n * 4));
  HIP_TRY(hipMalloc(&dOutVals.p, (size_t)n * ts));
  HIP_TRY(hipMalloc(&dPtr.p, (size_t)(cols + 1) * 8));
  HIP_TRY(hipMemcpy(dP.p, rowPtr, (size_t)(rows + 1) * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dI.p, indx, (size_t)n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dV.p, vals, (size_t)n * ts, hipMemcpyHostToDevice));
  // a stable sort of the positions by column id alone (prep_kernels.hip.h): the ids are non-negative, so they sort as uint32
  size_t tmpBytes = 0;
  const int endBit = bits_for(cols);
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, (const uint32_t *)dI.p, (uint32_t *)dKeys2.p, (const uint32_t *)dPos.p,
                                             (uint32_t *)dPos2.p, (int)n, 0, endBit));
  HIP_TRY(hipMalloc(&dTmp.p, std::max<size_t>(tmpBytes, 8)));
  EvPair evp;
  HIP_TRY(evp.create());
  const hipEvent_t e0 = evp.a, e1 = evp.b;
  HIP_TRY(hipEventRecord(e0, nullptr));
  const unsigned b256 = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(iota_rows_kernel, dim3(b256), dim3(256), 0, nullptr, (const int64_t *)dP.p, rows, n, (uint32_t *)dPos.p, (int32_t *)dRowOf.p);
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(dTmp.p, tmpBytes, (const uint32_t *)dI.p, (uint32_t *)dKeys2.p, (const uint32_t *)dPos.p,
                                             (uint32_t *)dPos2.p, (int)n, 0, endBit));
  if (dtype == YCNR_F32)
    hipLaunchKernelGGL(unpack_transposed_kernel<float>, dim3(b256), dim3(256), 0, nullptr, (const uint32_t *)dPos2.p, (const int32_t *)dRowOf.p,
                       (const float *)dV.p, n, (int32_t *)dIndx.p, (float *)dOutVals.p);
  else
    hipLaunchKernelGGL(unpack_transposed_kernel<double>, dim3(b256), dim3(256), 0, nullptr, (const uint32_t *)dPos2.p, (const int32_t *)dRowOf.p,
                       (const double *)dV.p, n, (int32_t *)dIndx.p, (double *)dOutVals.p);
  hipLaunchKernelGGL(row_ptr32_kernel, dim3((unsigned)((cols + 1 + 255) / 256)), dim3(256), 0, nullptr, (const uint32_t *)dKeys2.p, n, cols,
                     (int64_t *)dPtr.p);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) return fail(YCNR_ERR_HIP, "csr transpose: %s", hipGetErrorString(le));
  int rc = timed(e0, e1, deviceMs);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(outPtr, dPtr.p, (size_t)(cols + 1) * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(outIndx, dIndx.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(outVals, dOutVals.p, (size_t)n * ts, hipMemcpyDeviceToHost));
  return YCNR_OK;
}

// ---- N3: top-N recommend ----
int ycnr_recommend_items(int dtype, int32_t k, int64_t nUsers, const void *userRows, int64_t totalItems, const void *itemFactors,
                         const int64_t *skipPtr, const int32_t *skipIds, double globalAvgShift, double minRecommendRating,
                         int32_t limit, int32_t *outIds, double *outPredict, int32_t *outCount, double *deviceMs) {
  if (dtype != YCNR_F32 && dtype != YCNR_F64) return fail(YCNR_ERR_INVALID, "bad dtype %d", dtype);
  if (k < 1 || k > 4096 || nUsers < 0 || totalItems < 0 || limit < 1 || limit > 4096)
    return fail(YCNR_ERR_INVALID, "ycnr_recommend_items: bad k / sizes / limit");
  if (totalItems >= ((int64_t)1 << 31)) return fail(YCNR_ERR_UNSUPPORTED, "ycnr_recommend_items: more than 2^31 items");
  if (nUsers == 0) return YCNR_OK;
  if (!userRows || !skipPtr || !outIds || !outPredict || !outCount || (totalItems > 0 && !itemFactors))
    return fail(YCNR_ERR_INVALID, "ycnr_recommend_items: null argument");
  if (skipPtr[0] != 0) return fail(YCNR_ERR_INVALID, "ycnr_recommend_items: skipPtr[0] != 0");
  for (int64_t u = 0; u < nUsers; ++u) {
    if (skipPtr[u + 1] > skipPtr[u] && !skipIds) return fail(YCNR_ERR_INVALID, "ycnr_recommend_items: null skipIds");
    if (skipPtr[u + 1] < skipPtr[u]) return fail(YCNR_ERR_INVALID, "ycnr_recommend_items: skipPtr decreases at user %lld", (long long)u);
    for (int64_t q = skipPtr[u]; q < skipPtr[u + 1]; ++q)
      if (q > skipPtr[u] && skipIds[q] <= skipIds[q - 1])
        return fail(YCNR_ERR_INVALID, "ycnr_recommend_items: skip ids of user %lld are not strictly ascending", (long long)u);
  }
  const int64_t nSkip = skipPtr[nUsers];
  if (nSkip > 0 && !skipIds) return fail(YCNR_ERR_INVALID, "ycnr_recommend_items: null skipIds");
  const size_t ts = tsize(dtype);
  const int take = limit - 1;  // lib/YcnrController.js:268-269
  // users in batches whose score matrix stays below 1 GB
  const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(nUsers, ((int64_t)1 << 27) / std::max<int64_t>(1, totalItems)));
  DevBuf dItems, dUsers, dSkipPtr, dSkip, dScores, dIds, dPred, dCnt;
  HIP_TRY(hipMalloc(&dItems.p, std::max<size_t>((size_t)totalItems * k * ts, 8)));
  HIP_TRY(hipMalloc(&dUsers.p, (size_t)batch * k * ts));
  HIP_TRY(hipMalloc(&dSkipPtr.p, (size_t)(batch + 1) * 8));
  HIP_TRY(hipMalloc(&dSkip.p, std::max<size_t>((size_t)nSkip * 4, 8)));
  HIP_TRY(hipMalloc(&dScores.p, std::max<size_t>((size_t)batch * totalItems * 8, 8)));
  HIP_TRY(hipMalloc(&dIds.p, (size_t)batch * limit * 4));
  HIP_TRY(hipMalloc(&dPred.p, (size_t)batch * limit * 8));
  HIP_TRY(hipMalloc(&dCnt.p, (size_t)batch * 4));
  if (totalItems) HIP_TRY(hipMemcpy(dItems.p, itemFactors, (size_t)totalItems * k * ts, hipMemcpyHostToDevice));
  if (nSkip) HIP_TRY(hipMemcpy(dSkip.p, skipIds, (size_t)nSkip * 4, hipMemcpyHostToDevice));
  EvPair evp;
  HIP_TRY(evp.create());
  const hipEvent_t e0 = evp.a, e1 = evp.b;
  double total = 0.0;
  int rc = YCNR_OK;
  std::vector<int64_t> localPtr;
  for (int64_t u0 = 0; u0 < nUsers && !rc; u0 += batch) {
    const int64_t nb = std::min(batch, nUsers - u0);
    localPtr.assign(skipPtr + u0, skipPtr + u0 + nb + 1);  // absolute offsets into dSkip
    HIP_TRY(hipMemcpy(dUsers.p, (const char *)userRows + (size_t)u0 * k * ts, (size_t)nb * k * ts, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dSkipPtr.p, localPtr.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipEventRecord(e0, nullptr));
    // scores on the matrix cores: 16 users per workgroup, the item tiles dealt over the waves of gridDim.y workgroups
    // (YCNR_RECOMMEND_VALU=1: the 16-lane-group kernel, one user per workgroup, for A/B runs)
    const size_t kp = (size_t)((k + 15) & ~15);
    const unsigned ublk = (unsigned)((nb + 15) / 16);
    const unsigned yblk = (unsigned)std::max<int64_t>(1, std::min<int64_t>((totalItems + 63) / 64, (8 * device_cus() + ublk - 1) / ublk));
    const bool valu = getenv("YCNR_RECOMMEND_VALU") != nullptr || 16 * kp * ts > 48 * 1024;  // (16 users' factors must fit the default LDS limit)
    if (dtype == YCNR_F32) {
      if (valu)
        hipLaunchKernelGGL((recommend_scores_kernel<float, 1>), dim3((unsigned)nb), dim3(256), (size_t)k * ts, nullptr, (const float *)dUsers.p, nb,
                           (const float *)dItems.p, totalItems, (int)k, globalAvgShift, minRecommendRating, (double *)dScores.p);
      else
        hipLaunchKernelGGL(recommend_scores_mfma_kernel<float>, dim3(ublk, yblk), dim3(256), 16 * kp * ts, nullptr, (const float *)dUsers.p, nb,
                           (const float *)dItems.p, totalItems, (int)k, globalAvgShift, minRecommendRating, (double *)dScores.p);
    } else {
      if (valu)
        hipLaunchKernelGGL((recommend_scores_kernel<double, 1>), dim3((unsigned)nb), dim3(256), (size_t)k * ts, nullptr, (const double *)dUsers.p, nb,
                           (const double *)dItems.p, totalItems, (int)k, globalAvgShift, minRecommendRating, (double *)dScores.p);
      else
        hipLaunchKernelGGL(recommend_scores_mfma_kernel<double>, dim3(ublk, yblk), dim3(256), 16 * kp * ts, nullptr, (const double *)dUsers.p, nb,
                           (const double *)dItems.p, totalItems, (int)k, globalAvgShift, minRecommendRating, (double *)dScores.p);
    }
    if (nSkip)
      hipLaunchKernelGGL(recommend_skip_kernel, dim3((unsigned)nb), dim3(256), 0, nullptr, (const int64_t *)dSkipPtr.p, (const int32_t *)dSkip.p, totalItems,
                         (double *)dScores.p);
    hipLaunchKernelGGL(recommend_select_kernel, dim3((unsigned)nb), dim3(256), 0, nullptr, (double *)dScores.p, totalItems, take, (int)limit,
                       (int32_t *)dIds.p, (double *)dPred.p, (int32_t *)dCnt.p);
    hipError_t le = hipGetLastError();
    double ms = 0.0;
    rc = le == hipSuccess ? timed(e0, e1, &ms) : fail(YCNR_ERR_HIP, "recommend launch: %s", hipGetErrorString(le));
    total += ms;
    if (rc) break;
    HIP_TRY(hipMemcpy(outIds + u0 * limit, dIds.p, (size_t)nb * limit * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(outPredict + u0 * limit, dPred.p, (size_t)nb * limit * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(outCount + u0, dCnt.p, (size_t)nb * 4, hipMemcpyDeviceToHost));
  }
  if (deviceMs) *deviceMs = total;
  return rc;
}

int ycnr_als_create(const ycnr_als_options *o, ycnr_als **out) {
  if (!o || !out) return fail(YCNR_ERR_INVALID, "ycnr_als_create: null argument");
  if (o->struct_size != (int32_t)sizeof(ycnr_als_options))
    return fail(YCNR_ERR_INVALID, "ycnr_als_create: options.struct_size %d != %zu", o->struct_size,
                sizeof(ycnr_als_options));
  if (o->dtype != YCNR_F32 && o->dtype != YCNR_F64) return fail(YCNR_ERR_INVALID, "bad dtype %d", o->dtype);
  if (o->factorsCount < 1) return fail(YCNR_ERR_INVALID, "factorsCount must be >= 1");
  if (o->factorsCount > kMaxFactorsAny)
    return fail(YCNR_ERR_UNSUPPORTED, "factorsCount %d > %d is not supported by this build", o->factorsCount, kMaxFactorsAny);
  if (o->totalUsersCount < 1 || o->totalItemsCount < 1 || o->totalUsersCount > 0x7fffffffLL ||
      o->totalItemsCount > 0x7fffffffLL)
    return fail(YCNR_ERR_INVALID, "totalUsersCount / totalItemsCount must be in [1, 2^31)");
  if (!(o->userFactReg >= 0) || !(o->itemFactReg >= 0)) return fail(YCNR_ERR_INVALID, "negative regularisation");
  if (o->chunkRatings < 0) return fail(YCNR_ERR_INVALID, "chunkRatings < 0");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (o->device < 0 || o->device >= ndev)
    return fail(YCNR_ERR_INVALID, "device %d out of range (%d visible)", o->device, ndev);
  HIP_TRY(hipSetDevice(o->device));
  ycnr_als *h = new (std::nothrow) ycnr_als();
  if (!h) return fail(YCNR_ERR_NOMEM, "out of host memory");
  h->opt = *o;
  if (const char *e = getenv("YCNR_EXTRA_FLAGS")) h->opt.flags |= (uint32_t)atoi(e);  // experiments from unmodified hosts
  h->autoChunk = h->opt.chunkRatings == 0;
  if (h->opt.chunkRatings == 0) h->opt.chunkRatings = kDefaultChunk;
  h->opt.chunkRatings = (h->opt.chunkRatings + 3) & ~3;
  hipError_t e = hipStreamCreateWithFlags(&h->ownStream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&h->evComputeEnd);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->evStepStart, hipEventDisableTiming);
  // (the side streams of the one-piece half-step are created when a half-step first needs them: every stream a process creates
  // takes a share of the runtime's 4 or 8 hardware queues, and a sharded run -- step's stream, two piece streams, the
  // communicator's stream -- must not find one of its streams behind another's event waits on a shared queue)
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipStreamCreateWithFlags(&h->pieceStream[i], hipStreamNonBlocking);
  for (int s = 0; s < 2 && e == hipSuccess; ++s) {
    e = hipMalloc(&h->factors[s], (size_t)h->rows(s) * o->factorsCount * h->ts());
    if (e == hipSuccess) {
      h->ownFactors[s] = true;
      e = hipMemsetAsync(h->factors[s], 0, (size_t)h->rows(s) * o->factorsCount * h->ts(), h->ownStream);
    }
  }
  // float32 with factorsCount % 4 != 0: the kernels run on copies of the matrices padded to a multiple of 4 columns (zero columns:
  // lambda n on their diagonal, x = 0 there, the other columns' arithmetic unchanged) -- rows are then 16-byte aligned and every
  // such size takes the bf16x6 / LDS-DMA / dual-form kernels of its padded size instead of the plain float32-MFMA ones
  // (round 3: k > 128 only; round 4: every k).  Costs a pad of the fixed side and an unpad of the solved rows per half-step.
  if (o->dtype == YCNR_F32 && o->factorsCount % 4 != 0 && (o->factorsCount > kMaxFactors || !getenv("YCNR_NO_KPAD_SMALL"))) {  // (the variable: A/B runs)
    h->kPad = (o->factorsCount + 3) & ~3;
    for (int s = 0; s < 2 && e == hipSuccess; ++s) e = hipMalloc(&h->padded[s], (size_t)h->rows(s) * h->kPad * sizeof(float));
  }
  h->copt = h->opt;
  if (h->kPad) h->copt.factorsCount = h->kPad;
  if (e == hipSuccess) e = hipMalloc(&h->dErr, kErrBytes);  // ErrInfo + room for in-kernel stamps of diagnostic builds
  if (e == hipSuccess) e = hipMemsetAsync(h->dErr, 0, kErrBytes, h->ownStream);
  if (e == hipSuccess) e = hipHostMalloc((void **)&h->hErr, sizeof(ErrInfo), hipHostMallocDefault);
  if (e == hipSuccess) memset(h->hErr, 0, sizeof(ErrInfo));
  if (e == hipSuccess) e = hipMalloc(&h->dZeros, kZeroRowBytes);
  if (e == hipSuccess) e = hipMemsetAsync(h->dZeros, 0, kZeroRowBytes, h->ownStream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->ownStream);
  if (e != hipSuccess) {
    int code = fail(e == hipErrorOutOfMemory ? YCNR_ERR_NOMEM : YCNR_ERR_HIP, "ycnr_als_create: %s",
                    hipGetErrorString(e));
    ycnr_als_destroy(h);
    return code;
  }
  h->stream = h->ownStream;
  *out = h;
  return YCNR_OK;
}

int ycnr_als_destroy(ycnr_als *h) {
  if (!h) return YCNR_OK;
  (void)hipSetDevice(h->opt.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  h->drop_graphs();
  comm_release(h->comm);
  for (int s = 0; s < 2; ++s) {
    for (Part &p : h->parts[s]) p.release();
    h->parts[s].clear();
    h->banded[s].release();
    h->rmse[s].release();
    if (h->ownFactors[s] && h->factors[s]) (void)hipFree(h->factors[s]);
    if (h->padded[s]) (void)hipFree(h->padded[s]);
    if (h->planes[s]) (void)hipFree(h->planes[s]);
  }
  if (h->dErr) (void)hipFree(h->dErr);
  if (h->hErr) (void)hipHostFree(h->hErr);
  if (h->dZeros) (void)hipFree(h->dZeros);
  if (h->evComputeEnd) (void)hipEventDestroy(h->evComputeEnd);
  if (h->evStepStart) (void)hipEventDestroy(h->evStepStart);
  for (int i = 0; i < kSideStreams; ++i) {
    if (h->sideStream[i]) {
      (void)hipStreamSynchronize(h->sideStream[i]);
      (void)hipStreamDestroy(h->sideStream[i]);
    }
  }
  for (int i = 0; i < 2; ++i) {
    if (h->pieceStream[i]) {
      (void)hipStreamSynchronize(h->pieceStream[i]);
      (void)hipStreamDestroy(h->pieceStream[i]);
    }
  }
  if (h->ownStream) (void)hipStreamDestroy(h->ownStream);
  delete h;
  return YCNR_OK;
}

int ycnr_als_set_stream(ycnr_als *h, void *s) {
  if (!h) return fail(YCNR_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->opt.device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->drop_graphs();
  h->stream = s == YCNR_OWN_STREAM ? h->ownStream : (hipStream_t)s;  // NULL = the null (default) stream
  return YCNR_OK;
}

static int upload_ratings(ycnr_als *h, Ratings &R, int64_t totalRows, int64_t oppositeRows,
                          const int64_t *rowPtr, const int32_t *indx, const void *vals, int64_t rowBegin,
                          int64_t rowEnd, int memKind, std::vector<int64_t> &hostPtr, const char *what) {
  if (!rowPtr || !indx || !vals) return fail(YCNR_ERR_INVALID, "%s: null array", what);
  if (rowBegin < 0 || rowEnd < rowBegin || rowEnd > totalRows)
    return fail(YCNR_ERR_INVALID, "%s: shard [%lld, %lld) outside [0, %lld)", what, (long long)rowBegin,
                (long long)rowEnd, (long long)totalRows);
  if (memKind != YCNR_MEM_HOST && memKind != YCNR_MEM_DEVICE) return fail(YCNR_ERR_INVALID, "bad memKind");
  HIP_TRY(hipSetDevice(h->opt.device));
  const int64_t nRows = rowEnd - rowBegin;
  hostPtr.resize((size_t)nRows + 1);
  if (memKind == YCNR_MEM_HOST) {
    memcpy(hostPtr.data(), rowPtr + rowBegin, sizeof(int64_t) * ((size_t)nRows + 1));
  } else {
    HIP_TRY(hipMemcpy(hostPtr.data(), rowPtr + rowBegin, sizeof(int64_t) * ((size_t)nRows + 1),
                      hipMemcpyDeviceToHost));
  }
  for (int64_t r = 0; r < nRows; ++r)
    if (hostPtr[r + 1] < hostPtr[r]) return fail(YCNR_ERR_INVALID, "%s: rowPtr not ascending at row %lld", what,
                                                 (long long)(rowBegin + r));
  if (hostPtr[0] < 0) return fail(YCNR_ERR_INVALID, "%s: negative rowPtr", what);
  const int64_t base = hostPtr[0], nnz = hostPtr[nRows] - base;
  R.release();
  R.rowBegin = rowBegin;
  R.rowEnd = rowEnd;
  R.nnz = nnz;
  const size_t ts = h->ts();
  // 64 bytes of slack: the bf16x6 kernel reads ids and ratings with 16-byte vector loads
  HIP_TRY(hipMalloc(&R.dIndx, (size_t)nnz * 4 + 64));
  HIP_TRY(hipMalloc(&R.dVals, (size_t)nnz * ts + 64));
  int rc = copy_in(R.dIndx, indx + base, (size_t)nnz * 4, memKind, h->stream);
  if (rc) return rc;
  rc = copy_in(R.dVals, (const char *)vals + (size_t)base * ts, (size_t)nnz * ts, memKind, h->stream);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  rc = check_index_range(R.dIndx, nnz, oppositeRows, h->stream, what);
  if (rc) {
    R.release();
    return rc;
  }
  R.loaded = true;
  return YCNR_OK;
}

// Upload + schedule of one piece [rowBegin, rowEnd) of a side's rows into newR / S (both released
// by the caller when this fails).
static int build_part(ycnr_als *h, int side, const int64_t *rowPtr, const int32_t *indx, const void *vals, int64_t rowBegin,
                      int64_t rowEnd, int memKind, Ratings &newR, Schedule &S, int64_t sideNnz, int64_t sideSplitNnz) {
  std::vector<int64_t> hp;
  int rc = upload_ratings(h, newR, h->rows(side), h->rows(1 - side), rowPtr, indx, vals, rowBegin,
                          rowEnd, memKind, hp, side == YCNR_BY_USER ? "set_ratings(byUser)" : "set_ratings(byItem)");
  if (rc) return rc;
  std::vector<Unit> units;
  std::vector<SplitRow> split;
  int64_t nSlabs = 0, solved = 0;
  const bool gen = is_gen(h->copt.dtype, h->copt.factorsCount);  // any-k path: every primal row through slabs, in row order
  const bool big = !gen && h->copt.factorsCount > kMaxFactors;
  // k > 128: a unit is a whole workgroup's work, so chunks are long (a slab is 140 KB at k = 256)
  // (an explicit options.chunkRatings is honoured there too: tests cut short rows into chunks with it)
  // The automatic chunk length follows the split rows of the WHOLE side (sideNnz: their ratings), not of this piece: where a
  // split row is cut decides the order its partial sums are added in, so a length that depended on the piece would make the
  // factors depend on how the rows are cut into shards and pieces (and a feedback re-cut would change them: round-3 review).
  const int chunkRatings = h->autoChunk ? (gen ? kGenChunk : big ? kWgChunk : auto_chunk(sideNnz, sideSplitNnz)) : h->copt.chunkRatings;
  if (gen)
    build_schedule(hp.data(), rowBegin, rowEnd - rowBegin, chunkRatings, units, split, nSlabs, solved, dual_max_ratings(h->copt), 0);
  else
    build_schedule(hp.data(), rowBegin, rowEnd - rowBegin, chunkRatings, units, split, nSlabs, solved, -1,
                   big && h->autoChunk ? kWgFusedMax : std::min(chunkRatings, h->copt.chunkRatings), big ? kMaxSlabsPerRowBig : kMaxSlabsPerRow);
  int64_t arenaSlabs = nSlabs;
  // Band-major chunks (unless YCNR_FLAG_NO_BANDS): when the fixed matrix is far larger than the
  // last-level cache, cut every split row at the same column-id boundaries ("bands" of
  // kBandBytes of the fixed matrix) instead of every `chunk` ratings, and run the chunks band
  // by band.  All waves in flight then gather from one band, which stays cache-resident while
  // each of its rows is used once per rating that refers to it.  Rows are assumed sorted by
  // column id (they are when they come from a CSR transpose); unsorted rows only lose locality.
  {
    const int64_t fixedRows = h->rows(1 - side);
    const int64_t rowBytes = (int64_t)h->copt.factorsCount * (int64_t)h->ts();
    int64_t bandBytes = kBandBytes;
    if (const char *e = getenv("YCNR_BAND_MB")) bandBytes = (int64_t)atoi(e) << 20;  // per upload, not per half-step: read live (tests set it)
    if (!big && !gen && nSlabs > 1 && !(h->copt.flags & YCNR_FLAG_NO_BANDS) && bandBytes > 0 && fixedRows * rowBytes > 2 * bandBytes) {
      // at most kMaxSlabsPerRow / 2 bands, so that a row present in every band still has chunks to
      // spare (very large fixed matrices get wider bands instead of a failed upload)
      const int64_t W = std::max<int64_t>(std::max<int64_t>(1, bandBytes / rowBytes), (fixedRows + kMaxSlabsPerRow / 2 - 1) / (kMaxSlabsPerRow / 2));
      const int nBands = (int)((fixedRows + W - 1) / W);
      const int64_t base = hp[0];
      std::vector<int64_t> qBeg, qEnd, cuts;
      std::vector<int32_t> qKey;
      for (const SplitRow &sr : split) {
        const int64_t b = hp[sr.row - rowBegin] - base, e = hp[sr.row - rowBegin + 1] - base;
        for (int j = 1; j < nBands; ++j) {
          qBeg.push_back(b);
          qEnd.push_back(e);
          qKey.push_back((int32_t)(j * W));
        }
      }
      int rc2 = lower_bound_i32(newR.dIndx, qBeg, qEnd, qKey, cuts, h->stream);
      if (rc2) return rc2;
      struct BandUnit { Unit u; int band; };
      std::vector<BandUnit> bu;
      const int64_t chunk = chunkRatings, minSeg = std::max<int64_t>(64, chunk / 4);
      int64_t slabs = 0;
      for (size_t r = 0; r < split.size(); ++r) {
        SplitRow &sr = split[r];
        const int64_t b = hp[sr.row - rowBegin] - base, e = hp[sr.row - rowBegin + 1] - base;
        // band segments, short ones merged into their successor (the last into its predecessor)
        std::vector<std::pair<int64_t, int>> seg;  // (start, band); ends at the next start / e
        int64_t p = b;
        for (int j = 0; j < nBands; ++j) {
          const int64_t q = j + 1 < nBands ? std::min(e, std::max(p, cuts[r * (size_t)(nBands - 1) + j])) : e;
          if (q - p >= minSeg) {
            seg.push_back({p, j});
            p = q;
          }
        }
        if (seg.empty()) seg.push_back({b, 0});
        seg[0].first = b;  // ratings left over before the first kept boundary join the first segment
        int64_t ch = chunk;
        const int64_t room = kMaxSlabsPerRow - (int64_t)seg.size();
        if (room < 1 || (sr.n + ch - 1) / ch > room) ch = std::max(ch, (sr.n + std::max<int64_t>(1, room) - 1) / std::max<int64_t>(1, room));
        sr.slab0 = (int32_t)slabs;
        int32_t parts = 0;
        for (size_t i = 0; i < seg.size(); ++i) {
          const int64_t sb = seg[i].first, se = i + 1 < seg.size() ? seg[i + 1].first : e, len = se - sb;
          const int64_t np = (len + ch - 1) / ch;
          const int64_t pl = (((len + np - 1) / np) + 3) & ~(int64_t)3;
          for (int64_t q = 0; q < np; ++q) {
            const int64_t ub = sb + q * pl, ue = std::min(se, ub + pl);
            if (ue <= ub) break;
            bu.push_back({Unit{ub, ue, sr.row, (int32_t)(slabs + parts)}, seg[i].second});
            ++parts;
          }
        }
        if (parts > kMaxSlabsPerRow) return fail(YCNR_ERR_STATE, "band schedule: %d chunks for row %d", parts, sr.row);
        sr.nslabs = parts;
        slabs += parts;
      }
      std::stable_sort(bu.begin(), bu.end(), [](const BandUnit &x, const BandUnit &y) {
        return x.band != y.band ? x.band < y.band : (x.u.end - x.u.beg) > (y.u.end - y.u.beg);
      });
      std::vector<Unit> all;
      all.reserve(bu.size() + units.size() - (size_t)nSlabs);
      for (const BandUnit &x : bu) all.push_back(x.u);
      all.insert(all.end(), units.begin() + nSlabs, units.end());
      units.swap(all);
      nSlabs = slabs;
      arenaSlabs = slabs;
    }
  }
  S.nUnits = (int64_t)units.size();
  S.nSplit = (int64_t)split.size();
  S.maxRowSlabs = 0;
  for (const SplitRow &sr : split) S.maxRowSlabs = std::max(S.maxRowSlabs, sr.nslabs);
  S.nSlabs = nSlabs;
  S.solvedRows = solved;
  S.fusedRatings = 0;
  for (size_t i = (size_t)nSlabs; i < units.size(); ++i) S.fusedRatings += units[i].end - units[i].beg;
  // whole rows are sorted by descending length: rows short enough for the dual form are the
  // tail of the list, grouped by their number of 16-rating blocks
  const int dualMax = dual_max_ratings(h->copt);
  S.nPrimal = 0;
  for (size_t i = (size_t)nSlabs; i < units.size(); ++i) {
    const int64_t n = units[i].end - units[i].beg;
    if (n > dualMax) {
      ++S.nPrimal;
      continue;
    }
    const int m = (int)((n + 15) / 16);
    if (S.dualCount[m] == 0) S.dualFirst[m] = (int64_t)i;
    ++S.dualCount[m];
    ++S.dualRows;
    S.dualRatings += n;
    const double nd = (double)n, kd = (double)h->opt.factorsCount;
    S.dualFlops += nd * (nd + 1) * kd + nd * nd * nd / 3.0 + 2.0 * nd * nd + 2.0 * nd * kd;
  }
  // Launch order of the whole rows in primal form (k <= 128: one wave per row).  Sorted by descending length, the waves that
  // share a SIMD at any moment work on rows of the same length that started together and go through their Gramian phase (matrix
  // pipe) and their solve phase (vector ALU) side by side.  A large launch therefore takes its rows in a fixed pseudo-random
  // order -- neighbours in launch order differ in length and drift apart -- with the shortest rows kept, sorted, for the tail:
  // MAL-scale user half-step 12.18 -> 11.95 ms (interleaved A/B, two runs each; profiles/r05_ab.sh).  Every row is solved by
  // its own wave: the order cannot change a result.  YCNR_ROW_ORDER=0: sorted (A/B).  Riffles of the sorted list (position q i + j
  // takes row i + j N / q) were measured much SLOWER for q = 2 ... 32 (12.5 ... 19.4 ms): the dispatcher deals workgroups to
  // the XCDs round-robin, so part j of the list -- and all the long rows -- landed on XCD j mod 8.
  constexpr int64_t kShuffleMinRows = 16384;
  if (!big && !gen && S.nPrimal >= kShuffleMinRows) {
    static const int order = getenv("YCNR_ROW_ORDER") ? atoi(getenv("YCNR_ROW_ORDER")) : -1;
    if (order < 0) {
      const int64_t tail = std::min<int64_t>(S.nPrimal / 8, 8192);  // the shortest rows stay where they are
      uint64_t x = 0x9E3779B97F4A7C15ull;
      for (int64_t i = S.nPrimal - tail - 1; i > 0; --i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        std::swap(units[(size_t)(nSlabs + i)], units[(size_t)(nSlabs + (int64_t)(x % (uint64_t)(i + 1)))]);
      }
    }
  }
  if (big && slab_nb(h->copt.factorsCount) == kPairNB && S.nPrimal > 0 && !env_flags().noPair) {
    int64_t batchRows = kPairBatchRows;
    if (const char *e = getenv("YCNR_PAIR_BATCH_ROWS")) batchRows = std::max(256, atoi(e));  // read per upload
    S.rowSlabRows = std::min<int64_t>(S.nPrimal, batchRows);
    HIP_TRY(hipMalloc(&S.dRowSlabs, (size_t)S.rowSlabRows * (size_t)wg_slab_floats(kPairNB) * sizeof(float)));
  }
  if (S.nUnits) {
    HIP_TRY(hipMalloc(&S.dUnits, sizeof(Unit) * units.size()));
    HIP_TRY(hipMemcpy(S.dUnits, units.data(), sizeof(Unit) * units.size(), hipMemcpyHostToDevice));
  }
  if (S.nSplit) {
    // the reduce kernel takes the rows in this order, one wave each: rows with the most slabs first, so that
    // the 64-slab rows of the most popular items do not start when everything else has finished
    if (!big && !gen)
      std::stable_sort(split.begin(), split.end(), [](const SplitRow &x, const SplitRow &y) { return x.nslabs > y.nslabs; });
    HIP_TRY(hipMalloc(&S.dSplit, sizeof(SplitRow) * split.size()));
    HIP_TRY(hipMemcpy(S.dSplit, split.data(), sizeof(SplitRow) * split.size(), hipMemcpyHostToDevice));
    const size_t slabElems = gen ? (size_t)gen_slab_elems(slab_nb(h->copt.factorsCount))
                                 : big ? (size_t)wg_slab_floats(slab_nb(h->copt.factorsCount)) : (size_t)slab_regs(h->copt, side) * 64;
    if (gen) {
      // rows are solved in batches whose slabs fit the arena (a k = 512 float32 image is 541 KB)
      int64_t arenaBytes = kGenArenaBytes;
      if (const char *e = getenv("YCNR_GEN_ARENA_MB")) arenaBytes = (int64_t)std::max(1, atoi(e)) << 20;  // tests force several batches
      const int64_t fit = std::max<int64_t>(1, arenaBytes / (int64_t)(slabElems * h->ts()));
      S.genBatches = gen_batches(split, fit);
      arenaSlabs = 0;
      for (const GenBatch &b : S.genBatches) arenaSlabs = std::max<int64_t>(arenaSlabs, b.nSlabs);
    }
    HIP_TRY(hipMalloc(&S.dSlabs, (size_t)arenaSlabs * slabElems * h->ts()));
  }
  return YCNR_OK;
}

// nParts pieces [b[i], b[i + 1]) of this rank's rows.  Everything is built in locals and committed
// together at the end: a failure on the way leaves the handle's previous upload (or none) intact,
// never new ratings with an old schedule.
static int set_ratings_parts(ycnr_als *h, int side, const int64_t *rowPtr, const int32_t *indx, const void *vals, int memKind,
                             int nParts, const int64_t *b) {
  struct Pending {
    std::vector<Part> parts;
    bool keep = false;
    ~Pending() {
      if (!keep)
        for (Part &p : parts) p.release();
    }
  } pend;
  pend.parts.resize((size_t)nParts);
  // The ratings of the WHOLE side that sit in rows long enough to be split (rowPtr describes every row of the side, whatever
  // shard this handle solves): what the automatic chunk length is sized for.  MAL scale: nearly all of the item side's 121 M
  // ratings (-> 3072), a few per cent of the user side's (-> 1024: its chunk kernel is a few hundred waves whose length is
  // the kernel's; with the item side's 3072 it was a 300 us chain in front of every piece's row kernel).
  if (!rowPtr) return fail(YCNR_ERR_INVALID, "set_ratings: null rowPtr");
  int64_t sideNnz = 0, sideSplitNnz = 0;
  if (h->autoChunk) {
    const int64_t nr = h->rows(side);
    std::vector<int64_t> whole;
    const int64_t *rp = rowPtr;
    if (memKind == YCNR_MEM_DEVICE) {
      whole.resize((size_t)nr + 1);
      HIP_TRY(hipMemcpy(whole.data(), rowPtr, sizeof(int64_t) * ((size_t)nr + 1), hipMemcpyDeviceToHost));
      rp = whole.data();
    }
    for (int64_t r = 0; r < nr; ++r) {
      const int64_t n = rp[r + 1] - rp[r];
      sideNnz += n > 0 ? n : 0;
      if (n > kDefaultChunk) sideSplitNnz += n;
    }
  }
  for (int i = 0; i < nParts; ++i) {
    Part &p = pend.parts[(size_t)i];
    hipError_t e = p.create_events();
    if (e != hipSuccess) return fail(YCNR_ERR_HIP, "set_ratings: hipEventCreate: %s", hipGetErrorString(e));
    int rc = build_part(h, side, rowPtr, indx, vals, b[i], b[i + 1], memKind, p.R, p.S, sideNnz, sideSplitNnz);
    if (rc) return rc;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));  // nothing in flight still reads the previous upload
  for (Part &p : h->parts[side]) p.release();
  h->banded[side].release();
  h->deferExchange[side] = false;  // (a new upload exchanges after every half-step again until the host says otherwise)
  h->drop_graphs();
  h->parts[side].swap(pend.parts);
  pend.parts.clear();
  pend.keep = true;
  return YCNR_OK;
}

int ycnr_als_set_ratings(ycnr_als *h, int side, const int64_t *rowPtr, const int32_t *indx, const void *vals,
                         int64_t rowBegin, int64_t rowEnd, int memKind) {
  if (!h) return fail(YCNR_ERR_INVALID, "null handle");
  if (side != YCNR_BY_USER && side != YCNR_BY_ITEM) return fail(YCNR_ERR_INVALID, "bad side %d", side);
  const int64_t b[2] = {rowBegin, rowEnd};
  int rc = set_ratings_parts(h, side, rowPtr, indx, vals, memKind, 1, b);
  if (rc) return rc;
  h->bounds[side].clear();  // no exchange ranges: a plain shard (or the whole matrix)
  return YCNR_OK;
}

int ycnr_als_set_ratings_sharded(ycnr_als *h, int side, const int64_t *rowPtr, const int32_t *indx, const void *vals, int memKind,
                                 int nChunks, int boundsWorld, const int64_t *bounds) {
  if (!h || !bounds) return fail(YCNR_ERR_INVALID, "null argument");
  if (side != YCNR_BY_USER && side != YCNR_BY_ITEM) return fail(YCNR_ERR_INVALID, "bad side %d", side);
  if (nChunks < 1 || nChunks > 64) return fail(YCNR_ERR_INVALID, "set_ratings_sharded: nChunks %d outside [1, 64]", nChunks);
  const int world = h->comm.world, rank = h->comm.rank;
  if (boundsWorld != world)
    return fail(YCNR_ERR_INVALID, "set_ratings_sharded: bounds describe %d rank(s), the handle's communicator has %d", boundsWorld, world);
  const size_t n = (size_t)world * (size_t)(nChunks + 1);
  if (bounds[0] != 0 || bounds[n - 1] != h->rows(side))
    return fail(YCNR_ERR_INVALID, "set_ratings_sharded: the shards must tile rows [0, %lld) (got [%lld, %lld))", (long long)h->rows(side),
                (long long)bounds[0], (long long)bounds[n - 1]);
  int64_t prev = 0;
  for (size_t i = 0; i < n; ++i) {
    if (bounds[i] < prev || bounds[i] > h->rows(side))
      return fail(YCNR_ERR_INVALID, "set_ratings_sharded: bounds must ascend within [0, %lld]", (long long)h->rows(side));
    prev = bounds[i];
  }
  for (int r = 0; r + 1 < world; ++r)
    if (bounds[(size_t)r * (nChunks + 1) + nChunks] != bounds[(size_t)(r + 1) * (nChunks + 1)])
      return fail(YCNR_ERR_INVALID, "set_ratings_sharded: the shards of ranks %d and %d do not meet", r, r + 1);
  int rc = set_ratings_parts(h, side, rowPtr, indx, vals, memKind, nChunks, bounds + (size_t)rank * (nChunks + 1));
  if (!rc) h->bounds[side].assign(bounds, bounds + n);
  // IPC: the peers map this rank's matrices; a matrix bound (ycnr_als_bind_factors) since the communicator was made is
  // published here -- this call is collective, every rank passes through, also one whose upload has just failed
  // (it says so in its slot: the peers return an error instead of waiting for it at the barrier)
  if (h->comm.transport == YCNR_COMM_IPC && h->comm.world > 1) {
    const std::string uploadError = g_last_error;
    bool failed = rc != YCNR_OK;
    if (!failed && hipStreamSynchronize(h->stream) != hipSuccess) {
      (void)hipGetLastError();
      failed = true;
    }
    const int rcp = ipc_publish(h->comm, h->factors, failed);
    if (rc) g_last_error = uploadError;  // this rank's own failure is the message to keep
    else rc = rcp;
  }
  return rc;
}

// min / max of the column ids against a half-open range (banded uploads: only this rank's bands may appear)
static int check_index_bounds(const int32_t *dIndx, int64_t n, int64_t lo, int64_t hi, hipStream_t stream, const char *what) {
  if (n == 0) return YCNR_OK;
  int32_t *dmm = nullptr;
  HIP_TRY(hipMalloc(&dmm, 2 * sizeof(int32_t)));
  int32_t init[2] = {INT32_MIN, INT32_MAX};
  hipError_t e = hipMemcpyAsync(dmm, init, sizeof init, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) {
    const int blocks = (int)std::min<int64_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(max_i32_kernel, dim3(blocks), dim3(256), 0, stream, dIndx, n, dmm, dmm + 1);
    e = hipGetLastError();
  }
  int32_t got[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(got, dmm, sizeof got, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(dmm);
  if (e != hipSuccess) return fail(YCNR_ERR_HIP, "index range check failed: %s", hipGetErrorString(e));
  if ((int64_t)got[1] < lo || (int64_t)got[0] >= hi)
    return fail(YCNR_ERR_INVALID, "%s: column id outside this rank's bands (min %d, max %d, bands cover [%lld, %lld))", what, got[1], got[0],
                (long long)lo, (long long)hi);
  return YCNR_OK;
}

int ycnr_als_set_ratings_banded(ycnr_als *h, int side, const int64_t *rowPtr, const int32_t *indx, const void *vals, int memKind,
                                int nBands, const int64_t *bandBounds, const int64_t *rankBands, const int64_t *ownerBounds) {
  if (!h || !bandBounds || !rankBands || !ownerBounds) return fail(YCNR_ERR_INVALID, "null argument");
  if (side != YCNR_BY_USER && side != YCNR_BY_ITEM) return fail(YCNR_ERR_INVALID, "bad side %d", side);
  if (nBands < 1 || nBands > kFewSlabs)
    return fail(YCNR_ERR_INVALID, "set_ratings_banded: nBands %d outside [1, %d] (a row's band slabs are added as one short float32 chain)", nBands, kFewSlabs);
  if (h->copt.factorsCount > kMaxFactors || h->kPad)
    return fail(YCNR_ERR_UNSUPPORTED, "set_ratings_banded: factorsCount %d (the banded half-step is built for the one-wave kernels: k <= %d, float32 k %% 4 == 0)",
                h->opt.factorsCount, kMaxFactors);
  const int world = h->comm.active() ? h->comm.world : 1, rank = h->comm.active() ? h->comm.rank : 0;
  if (world > kFewSlabs) return fail(YCNR_ERR_UNSUPPORTED, "set_ratings_banded: at most %d ranks", kFewSlabs);
  if (world > 1 && h->comm.transport == YCNR_COMM_SHM)
    return fail(YCNR_ERR_UNSUPPORTED, "set_ratings_banded: the host-staged shm stand-in does not carry band slabs (use ipc or rccl)");
  const int64_t rows = h->rows(side), cols = h->rows(1 - side);
  for (int b = 0; b <= nBands; ++b)
    if (bandBounds[b] < 0 || bandBounds[b] > cols || (b > 0 && bandBounds[b] < bandBounds[b - 1]))
      return fail(YCNR_ERR_INVALID, "set_ratings_banded: bandBounds must ascend within [0, %lld]", (long long)cols);
  if (bandBounds[0] != 0 || bandBounds[nBands] != cols) return fail(YCNR_ERR_INVALID, "set_ratings_banded: the bands must tile the columns [0, %lld)", (long long)cols);
  if (rankBands[0] != 0 || rankBands[world] != nBands || ownerBounds[0] != 0 || ownerBounds[world] != rows)
    return fail(YCNR_ERR_INVALID, "set_ratings_banded: rankBands must tile the %d bands and ownerBounds the %lld rows over the %d rank(s)", nBands, (long long)rows, world);
  for (int r = 0; r < world; ++r)
    if (rankBands[r + 1] < rankBands[r] || ownerBounds[r + 1] < ownerBounds[r]) return fail(YCNR_ERR_INVALID, "set_ratings_banded: rankBands / ownerBounds must ascend");
  HIP_TRY(hipSetDevice(h->opt.device));
  struct Pending {
    Part part;
    BandedSide B;
    bool keep = false;
    ~Pending() {
      if (!keep) {
        part.release();
        B.release();
      }
    }
  } pend;
  Part &part = pend.part;
  BandedSide &B = pend.B;
  int rc = YCNR_OK;
  const size_t ts = h->ts();
  const int bl = (int)rankBands[rank], bh = (int)rankBands[rank + 1], bpr = bh - bl;
  std::vector<int64_t> hp;
  std::vector<double> counts((size_t)rows * (size_t)nBands, 0.0);
  auto build = [&]() -> int {
    hipError_t e = part.create_events();
    if (e == hipSuccess) e = hipEventCreateWithFlags(&B.evRecv, hipEventDisableTiming);
    for (int i = 0; i < kFewSlabs && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&B.evStage[i], hipEventDisableTiming);
    if (e != hipSuccess) return fail(YCNR_ERR_HIP, "set_ratings_banded: hipEventCreate: %s", hipGetErrorString(e));
    int r1 = upload_ratings(h, part.R, rows, cols, rowPtr, indx, vals, 0, rows, memKind, hp, "set_ratings_banded");
    if (r1) return r1;
    if (bpr <= 0 && part.R.nnz > 0) return fail(YCNR_ERR_INVALID, "set_ratings_banded: rank %d has no band but %lld ratings", rank, (long long)part.R.nnz);
    if (bpr > 0)
      if (int r2 = check_index_bounds(part.R.dIndx, part.R.nnz, bandBounds[bl], bandBounds[bh], h->stream, "set_ratings_banded")) return r2;
    // where every row's ratings cross this rank's inner band boundaries (rows are sorted by column id)
    const int64_t base = hp[0];
    std::vector<int64_t> qBeg, qEnd, cuts;
    std::vector<int32_t> qKey;
    std::vector<int64_t> qRow;
    for (int64_t r = 0; r < rows; ++r) {
      if (hp[r + 1] == hp[r]) continue;
      qRow.push_back(r);
      for (int b = bl + 1; b < bh; ++b) {
        qBeg.push_back(hp[r] - base);
        qEnd.push_back(hp[r + 1] - base);
        qKey.push_back((int32_t)bandBounds[b]);
      }
    }
    if (int r3 = lower_bound_i32(part.R.dIndx, qBeg, qEnd, qKey, cuts, h->stream)) return r3;
    for (size_t i = 0; i < qRow.size(); ++i) {
      const int64_t r = qRow[i];
      int64_t p = hp[r] - base;
      for (int j = 0; j < bpr; ++j) {
        const int64_t q = j + 1 < bpr ? cuts[i * (size_t)(bpr - 1) + j] : hp[r + 1] - base;
        if (q < p) return fail(YCNR_ERR_INVALID, "set_ratings_banded: the column ids of row %lld are not ascending", (long long)r);
        counts[(size_t)r * nBands + (size_t)(bl + j)] = (double)(q - p);
        p = q;
      }
    }
    return YCNR_OK;
  };
  rc = build();
  // the ratings of every (row, band) on every rank: the owners skip empty bands, lambda n needs the row's total, and the chunk
  // length follows the ratings of the WHOLE side -- collective, a rank whose upload failed takes part with what it has
  if (world > 1) {
    const std::string why = g_last_error;
    const int rca = comm_allreduce_sum(h->comm, counts.data(), (int64_t)counts.size());
    if (rc) g_last_error = why;
    else rc = rca;
  }
  auto finish = [&]() -> int {
    int64_t sideNnz = 0, sideSplitNnz = 0;
    std::vector<int64_t> rowTotal((size_t)rows, 0);
    for (int64_t r = 0; r < rows; ++r) {
      int64_t t = 0;
      for (int b = 0; b < nBands; ++b) t += (int64_t)counts[(size_t)r * nBands + b];
      rowTotal[(size_t)r] = t;
      sideNnz += t;
      if (t > kDefaultChunk) sideSplitNnz += t;
    }
    if (h->comm.active() && h->comm.transport == YCNR_COMM_STUB) {
      // (one rank of an emulated world: the counts of the other ranks' bands never arrive; size the chunks for a side `world`
      // times this rank's share, as the real run would)
      sideNnz = 0;
      sideSplitNnz = 0;
      for (int64_t r = 0; r < rows; ++r) {
        const int64_t t = rowTotal[(size_t)r] * world;
        sideNnz += t;
        if (t > kDefaultChunk) sideSplitNnz += t;
      }
    }
    const int64_t chunk = h->autoChunk ? auto_chunk(sideNnz, sideSplitNnz) : h->copt.chunkRatings;
    B.nBands = nBands;
    B.world = world;
    B.rank = rank;
    B.rankBands.assign(rankBands, rankBands + world + 1);
    B.ownerBounds.assign(ownerBounds, ownerBounds + world + 1);
    B.slabElems = (int64_t)slab_regs(h->copt, side) * 64;
    B.groupSlab0.assign((size_t)world + 1, 0);
    for (int g = 0; g < world; ++g) B.groupSlab0[(size_t)g + 1] = B.groupSlab0[(size_t)g] + (ownerBounds[g + 1] - ownerBounds[g]) * bpr;
    B.bandSlabs = B.groupSlab0[(size_t)world];
    // units: group by group (the order they are computed and sent in), longest first inside a group
    const int64_t base = hp[0];
    std::vector<Unit> units;
    std::vector<SlabSum> sums;
    B.groupUnit0.assign((size_t)world, 0);
    B.groupUnits.assign((size_t)world, 0);
    B.groupSum0.assign((size_t)world, 0);
    B.groupSums.assign((size_t)world, 0);
    int64_t tmp = 0;
    for (int st = 1; st <= world; ++st) {  // in the order the groups are computed and sent: consecutive stages are contiguous
      const int g = (rank + st) % world;
      B.groupUnit0[(size_t)g] = (int64_t)units.size();
      B.groupSum0[(size_t)g] = (int64_t)sums.size();
      const size_t u0 = units.size();
      for (int64_t r = ownerBounds[g]; r < ownerBounds[g + 1]; ++r) {
        if (hp[r + 1] == hp[r]) continue;
        int64_t p = hp[r] - base;
        for (int j = 0; j < bpr; ++j) {
          const int64_t len = (int64_t)counts[(size_t)r * nBands + (size_t)(bl + j)];
          if (len <= 0) continue;
          const int64_t bandSlab = B.groupSlab0[(size_t)g] + (r - ownerBounds[g]) * bpr + j;
          const int64_t np = (len + chunk - 1) / chunk, pl = (((len + np - 1) / np) + 3) & ~(int64_t)3;
          if (np == 1) {
            units.push_back(Unit{p, p + len, (int32_t)r, (int32_t)bandSlab});
          } else {
            int64_t made = 0;
            for (int64_t q = 0; q < np; ++q) {
              const int64_t ub = p + q * pl, ue = std::min(p + len, ub + pl);
              if (ue <= ub) break;
              units.push_back(Unit{ub, ue, (int32_t)r, (int32_t)(B.bandSlabs + tmp + made)});
              ++made;
            }
            sums.push_back(SlabSum{bandSlab * B.slabElems, (B.bandSlabs + tmp) * B.slabElems, (int32_t)made, 0});
            tmp += made;
          }
          p += len;
        }
      }
      std::stable_sort(units.begin() + (ptrdiff_t)u0, units.end(), [](const Unit &x, const Unit &y) { return (x.end - x.beg) > (y.end - y.beg); });
      B.groupUnits[(size_t)g] = (int64_t)units.size() - B.groupUnit0[(size_t)g];
      B.groupSums[(size_t)g] = (int64_t)sums.size() - B.groupSum0[(size_t)g];
    }
    B.tmpSlabs = tmp;
    B.nSums = (int64_t)sums.size();
    if (B.bandSlabs + B.tmpSlabs > 0x7fffffffLL) return fail(YCNR_ERR_UNSUPPORTED, "set_ratings_banded: too many slabs");
    Schedule &S = part.S;
    S.nUnits = (int64_t)units.size();
    S.nSlabs = S.nUnits;
    if (S.nUnits) {
      HIP_TRY(hipMalloc(&S.dUnits, sizeof(Unit) * units.size()));
      HIP_TRY(hipMemcpy(S.dUnits, units.data(), sizeof(Unit) * units.size(), hipMemcpyHostToDevice));
    }
    if (B.nSums) {
      HIP_TRY(hipMalloc(&B.dSums, sizeof(SlabSum) * sums.size()));
      HIP_TRY(hipMemcpy(B.dSums, sums.data(), sizeof(SlabSum) * sums.size(), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&S.dSlabs, std::max<size_t>((size_t)(B.bandSlabs + B.tmpSlabs) * (size_t)B.slabElems * ts, 64)));
    // what this rank owns: its rows' band slabs from every rank, and the rows to solve
    const int64_t own0 = ownerBounds[rank], ownN = ownerBounds[rank + 1] - own0;
    B.recvSlab0.assign((size_t)world + 1, 0);
    for (int r = 0; r < world; ++r)
      B.recvSlab0[(size_t)r + 1] = B.recvSlab0[(size_t)r] + (r == rank ? 0 : ownN * (rankBands[r + 1] - rankBands[r]));
    B.recvSlabs = B.recvSlab0[(size_t)world];
    HIP_TRY(hipMalloc(&B.dRecv, std::max<size_t>((size_t)B.recvSlabs * (size_t)B.slabElems * ts, 64)));
    HIP_TRY(hipMemset(B.dRecv, 0, std::max<size_t>((size_t)B.recvSlabs * (size_t)B.slabElems * ts, 64)));
    std::vector<const void *> table((size_t)ownN * (size_t)nBands, nullptr);
    std::vector<SplitRow> owned;
    const size_t slabBytes = (size_t)B.slabElems * ts;
    for (int64_t i = 0; i < ownN; ++i) {
      const int64_t r = own0 + i;
      if (rowTotal[(size_t)r] <= 0) continue;
      owned.push_back(SplitRow{rowTotal[(size_t)r], (int32_t)r, (int32_t)(i * nBands), (int32_t)nBands, 0});
      for (int b = 0; b < nBands; ++b) {
        if (counts[(size_t)r * nBands + b] <= 0) continue;
        int src = 0;
        while (src + 1 < world && b >= rankBands[src + 1]) ++src;
        const char *p = src == rank ? (const char *)S.dSlabs + (size_t)(B.groupSlab0[(size_t)rank] + i * bpr + (b - bl)) * slabBytes
                                    : (const char *)B.dRecv + (size_t)(B.recvSlab0[(size_t)src] + i * (rankBands[src + 1] - rankBands[src]) + (b - rankBands[src])) * slabBytes;
        table[(size_t)i * nBands + b] = p;
      }
    }
    if (owned.size() > 0x7fffffffULL / (size_t)std::max(nBands, 1)) return fail(YCNR_ERR_UNSUPPORTED, "set_ratings_banded: too many owned rows");
    S.nSplit = (int64_t)owned.size();
    S.solvedRows = S.nSplit;
    S.maxRowSlabs = nBands;
    B.ownedSolved = S.nSplit;
    if (S.nSplit) {
      HIP_TRY(hipMalloc(&S.dSplit, sizeof(SplitRow) * owned.size()));
      HIP_TRY(hipMemcpy(S.dSplit, owned.data(), sizeof(SplitRow) * owned.size(), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc((void **)&B.dTable, std::max<size_t>(sizeof(void *) * table.size(), 64)));
    if (!table.empty()) HIP_TRY(hipMemcpy((void *)B.dTable, table.data(), sizeof(void *) * table.size(), hipMemcpyHostToDevice));
    B.on = true;
    return YCNR_OK;
  };
  if (!rc) rc = finish();
  // IPC: the peers push their band slabs into dRecv -- published here, collectively (a rank that failed says so in its slot)
  if (h->comm.active() && h->comm.transport == YCNR_COMM_IPC) {
    const std::string why = g_last_error;
    const int rcp = ipc_publish_extra(h->comm, side, rc ? nullptr : B.dRecv, rc != YCNR_OK);
    if (rc) g_last_error = why;
    else rc = rcp;
  }
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (Part &p : h->parts[side]) p.release();
  h->parts[side].clear();
  h->banded[side].release();
  h->drop_graphs();
  h->parts[side].push_back(part);
  h->banded[side] = B;
  pend.keep = true;
  // the solved rows travel like a sharded side's: one piece per rank (ycnr_als_exchange, the exchange behind the half-step)
  h->bounds[side].clear();
  for (int r = 0; r < world; ++r) {
    h->bounds[side].push_back(ownerBounds[r]);
    h->bounds[side].push_back(ownerBounds[r + 1]);
  }
  return YCNR_OK;
}

int ycnr_als_set_rmse_ratings(ycnr_als *h, int which, const int64_t *rowPtr, const int32_t *indx,
                              const void *vals, int64_t rowBegin, int64_t rowEnd, int memKind) {
  if (!h) return fail(YCNR_ERR_INVALID, "null handle");
  if (which != YCNR_RMSE_VALIDATE && which != YCNR_RMSE_TEST) return fail(YCNR_ERR_INVALID, "bad rmse set %d", which);
  std::vector<int64_t> hp;
  Ratings &R = h->rmse[which];
  int rc = upload_ratings(h, R, h->opt.totalUsersCount, h->opt.totalItemsCount, rowPtr, indx, vals, rowBegin,
                          rowEnd, memKind, hp, "set_rmse_ratings");
  if (rc) return rc;
  const int64_t base = hp[0];
  for (auto &v : hp) v -= base;
  hipError_t e = hipMalloc(&R.dRowPtr, sizeof(int64_t) * hp.size());
  if (e == hipSuccess) e = hipMemcpy(R.dRowPtr, hp.data(), sizeof(int64_t) * hp.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    R.release();  // never "loaded" without its row pointers
    return fail(e == hipErrorOutOfMemory ? YCNR_ERR_NOMEM : YCNR_ERR_HIP, "set_rmse_ratings: %s", hipGetErrorString(e));
  }
  return YCNR_OK;
}

int ycnr_als_set_factors(ycnr_als *h, int side, const void *src, int memKind) {
  if (!h || !src) return fail(YCNR_ERR_INVALID, "null argument");
  if (side != 0 && side != 1) return fail(YCNR_ERR_INVALID, "bad side %d", side);
  HIP_TRY(hipSetDevice(h->opt.device));
  int rc = copy_in(h->factors[side], src, (size_t)h->rows(side) * h->opt.factorsCount * h->ts(), memKind, h->stream);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return YCNR_OK;
}

int ycnr_als_get_factors(ycnr_als *h, int side, void *dst, int64_t rowBegin, int64_t rowCount, int memKind) {
  if (!h || !dst) return fail(YCNR_ERR_INVALID, "null argument");
  if (side != 0 && side != 1) return fail(YCNR_ERR_INVALID, "bad side %d", side);
  if (rowBegin < 0 || rowCount < 0 || rowBegin + rowCount > h->rows(side))
    return fail(YCNR_ERR_INVALID, "get_factors: rows [%lld, +%lld) outside the matrix", (long long)rowBegin,
                (long long)rowCount);
  HIP_TRY(hipSetDevice(h->opt.device));
  const size_t rowBytes = (size_t)h->opt.factorsCount * h->ts();
  HIP_TRY(hipMemcpyAsync(dst, (const char *)h->factors[side] + (size_t)rowBegin * rowBytes, (size_t)rowCount * rowBytes,
                         memKind == YCNR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return YCNR_OK;
}

int ycnr_als_factors_ptr(ycnr_als *h, int side, void **p) {
  if (!h || !p) return fail(YCNR_ERR_INVALID, "null argument");
  if (side != 0 && side != 1) return fail(YCNR_ERR_INVALID, "bad side %d", side);
  *p = h->factors[side];
  return YCNR_OK;
}

int ycnr_als_bind_factors(ycnr_als *h, int side, void *p) {
  if (!h || !p) return fail(YCNR_ERR_INVALID, "null argument");
  if (side != 0 && side != 1) return fail(YCNR_ERR_INVALID, "bad side %d", side);
  HIP_TRY(hipSetDevice(h->opt.device));
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess || attr.type != hipMemoryTypeDevice) {
    (void)hipGetLastError();
    return fail(YCNR_ERR_INVALID, "bind_factors: not a device pointer");
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->ownFactors[side] && h->factors[side]) (void)hipFree(h->factors[side]);
  h->drop_graphs();
  h->factors[side] = p;
  h->ownFactors[side] = false;
  return YCNR_OK;
}

// kernels of one piece of the shard, on the handle's stream, timed by the piece's events
// branches: the small-upload form (chunks -> reduce, row kernel and dual classes as parallel branches, no per-kernel
// timing events) that ycnr_als_step_async captures into a hipGraph
// inOrder: every kernel of the piece on `stream`, one after the other (a piece of several: two pieces are in flight on the two
// piece streams and fill each other's tails; side streams per piece cost a fork and a join per side stream, and a wait on an
// event of ANOTHER hardware queue costs the step's stream tens of microseconds -- in the trace of one GPU's eighth of the
// MAL-scale user side, 4 pieces, the joins were 100 - 300 us of gaps in a 2.2 ms half-step)
static int launch_part(ycnr_als *h, int side, Part &part, hipStream_t stream, bool branches = false, bool inOrder = false) {
  const Ratings &R = part.R;
  const Schedule &S = part.S;
  hipEvent_t *ev = branches ? nullptr : part.ev;
  const double lambda = side == YCNR_BY_USER ? h->copt.userFactReg : h->copt.itemFactReg;
  if (h->copt.dtype == YCNR_F32) {
    const int kk = h->copt.factorsCount;  // (= kPad when the matrices are padded)
    const float *fixedM = h->kPad ? h->padded[1 - side] : (const float *)h->factors[1 - side];
    float *solvedM = h->kPad ? h->padded[side] : (float *)h->factors[side];
    StepArgs<float> a{S.dUnits, S.dSplit, R.dIndx, (const float *)R.dVals, fixedM,
                      (const float *)h->dZeros, solvedM, (float *)S.dSlabs, h->dErr, lambda, kk, 0, 0,
                      use_slab_x6(h->copt, side) ? (uint32_t)(h->rows(1 - side) * h->copt.factorsCount * 4) : 0u};
    if (h->kPad) a.kReal = h->opt.factorsCount;
    if (h->planes[1 - side] && h->planesValid[1 - side] && use_planes(h->copt, side)) {
      a.planes = h->planes[1 - side];
      a.planesBytes = (uint32_t)(h->rows(1 - side) * planes_row_bytes(slab_nb(h->copt.factorsCount), planes_pack(h->copt.factorsCount)));
    }
    DualPlan dp;
    dp.noX6 = (h->copt.flags & YCNR_FLAG_NO_BF16X6) != 0;
    dp.fewSlabs = S.maxRowSlabs <= kFewSlabs;
    if (dual_max_ratings(h->copt) > 0) {
      dp.nPrimal = S.nPrimal;
      dp.first = S.dualFirst;
      dp.count = S.dualCount;
      // (the fork and join cost nine more runtime calls per half-step: with a few hundred rows,
      // where the half-step is bound by the launches themselves, they made it slower)
      if ((S.dualRows >= kMinOverlapDualRows || (branches && S.dualRows > 0)) && !inOrder && !(h->copt.flags & YCNR_FLAG_NO_OVERLAP) && !env_flags().noOverlap) {
        // (a captured small half-step keeps two branches for its dual classes: every branch is a join, 30 - 60 us each)
        dp.nSide = branches ? std::min(2, side_streams()) : side_streams();
        for (int i = 0; i < dp.nSide; ++i) {
          if (!h->sideStream[i]) HIP_TRY(hipStreamCreateWithFlags(&h->sideStream[i], hipStreamNonBlocking));
          dp.side[i] = h->sideStream[i];
          dp.join[i] = part.join[i];
        }
      }
    }
    dp.fork = part.fork;
    if (branches) {
      dp.slabStream = h->pieceStream[0];
      dp.slabJoin = part.slabJoin;
    }
    int rc;
    if (h->copt.factorsCount > kMaxFactors)
      rc = is_gen(YCNR_F32, h->copt.factorsCount) ? launch_step_gen<float>(a, S.genBatches, stream, ev, dp)
                                                  : launch_step_big(a, S.nUnits, S.nSlabs, S.nSplit, stream, ev, dp, S.dRowSlabs, S.rowSlabRows);
    else
      rc = launch_step<float>(a, S.nUnits, S.nSlabs, S.nSplit, stream, ev, (h->copt.flags & YCNR_FLAG_LDS_SOLVER) != 0, dp,
                              use_valu_edge(h->copt), use_slab_x6(h->copt, side));
    if (rc || !h->kPad) return rc;
    // the piece's solved rows back into the caller's matrix (before its exchange)
    const int64_t nr = R.rowEnd - R.rowBegin, n = nr * h->opt.factorsCount;
    if (n > 0) {
      hipLaunchKernelGGL(unpad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float *)h->padded[side],
                         (float *)h->factors[side], R.rowBegin, nr, h->opt.factorsCount, h->kPad);
      HIP_TRY(hipGetLastError());
    }
    return YCNR_OK;
  }
  StepArgs<double> a{S.dUnits, S.dSplit, R.dIndx, (const double *)R.dVals, (const double *)h->factors[1 - side],
                     (const double *)h->dZeros, (double *)h->factors[side], (double *)S.dSlabs, h->dErr, lambda, h->copt.factorsCount, 0, 0, 0u};
  if (is_gen(YCNR_F64, h->copt.factorsCount)) return launch_step_gen<double>(a, S.genBatches, stream, ev, DualPlan());
  DualPlan dpd;  // float64 has no dual classes; the chunk branch of the small-upload form applies
  dpd.fork = part.fork;
  if (branches) {
    dpd.slabStream = h->pieceStream[0];
    dpd.slabJoin = part.slabJoin;
  }
  return launch_step<double>(a, S.nUnits, S.nSlabs, S.nSplit, stream, ev, (h->copt.flags & YCNR_FLAG_LDS_SOLVER) != 0, dpd);
}

// row ranges of piece c of every rank (sharded upload)
static void part_ranges(const ycnr_als *h, int side, int c, std::vector<int64_t> &begin, std::vector<int64_t> &end) {
  const int world = h->comm.world, np = (int)h->parts[side].size();
  begin.resize((size_t)world);
  end.resize((size_t)world);
  for (int r = 0; r < world; ++r) {
    begin[(size_t)r] = h->bounds[side][(size_t)r * (np + 1) + c];
    end[(size_t)r] = h->bounds[side][(size_t)r * (np + 1) + c + 1];
  }
}

}  // extern "C"

// bookkeeping of a half-step that has just been enqueued: what the step info reports besides the times, and its place among
// the half-steps ycnr_als_sync will complete
static void note_enqueued(ycnr_als *h, int side, const std::vector<Part> &parts) {
  h->info.struct_size = (int32_t)sizeof(ycnr_als_step_info);
  h->info.side = side;
  h->info.parts = (int32_t)parts.size();
  const bool dual = h->opt.dtype == YCNR_F32 && dual_max_ratings(h->copt) > 0;
  for (const Part &p : parts) {
    const Schedule &S = p.S;
    h->info.rows += S.solvedRows;
    h->info.ratings += p.R.nnz;
    h->info.units += S.nUnits;
    h->info.splitRows += S.nSplit;
    h->info.fusedRows += S.solvedRows - S.nSplit;
    h->info.fusedRatings += S.fusedRatings;
    if (dual) {
      h->info.dualRows += S.dualRows;
      h->info.dualRatings += S.dualRatings;
      h->info.dualFlops += S.dualFlops;
      if (S.dualRows >= kMinOverlapDualRows && !is_gen(YCNR_F32, h->copt.factorsCount) && !(h->opt.flags & YCNR_FLAG_NO_OVERLAP) &&
          !env_flags().noOverlap)
        h->info.dualOverlapped = 1;
    }
  }
  h->infoPending = true;
  h->infoSide = side;
  // (the same side twice without a sync: its events have been re-recorded, the earlier half-step's times are gone)
  int np = 0;
  for (int i = 0; i < h->nPend; ++i)
    if (h->pendOrder[i] != side) h->pendOrder[np++] = h->pendOrder[i];
  h->pendOrder[np++] = side;
  h->nPend = np;
  h->pend[side].info = h->info;
  h->pend[side].graphRun = h->graphRun;
  h->pend[side].exchanged = h->exchangedInStep;
}

// One half-step of a side sharded by bands of columns (ycnr_als_set_ratings_banded): owner group by owner group the chunk
// Gramians of this rank's ratings (+ the sums of the segments that were cut into several chunks), every group's band slabs
// sent to its owner while the next group is computed; then the owned rows' band slabs added in band order + solve, and the
// solved rows exchanged like a sharded side's.  RCCL: everything stream-ordered.  IPC (the functional path of the one-GPU
// tests): the pushes are completed with a host barrier before the owners reduce.
template <typename T>
static int step_banded_t(ycnr_als *h, int side) {
  BandedSide &B = h->banded[side];
  Part &part = h->parts[side][0];
  Schedule &S = part.S;
  Comm &c = h->comm;
  const bool comm = c.active() && B.world > 1;
  if (comm && (c.world != B.world || c.rank != B.rank)) return fail(YCNR_ERR_STATE, "step: the banded upload of this side was made for another communicator");
  hipStream_t stream = h->stream;
  const size_t ts = sizeof(T), slabBytes = (size_t)B.slabElems * ts;
  const double lambda = side == YCNR_BY_USER ? h->copt.userFactReg : h->copt.itemFactReg;
  if (comm && c.transport == YCNR_COMM_IPC) {
    if (c.pendingFinish) {  // (as ycnr_als_step_async: a half-step the peers were still pushing must be complete first)
      HIP_TRY(hipStreamSynchronize(stream));
      if (int rcf = ipc_finish(c)) return rcf;
    }
    if (int rcb = ipc_enter(c)) return rcb;
  }
  StepArgs<T> a{S.dUnits, S.dSplit, part.R.dIndx, (const T *)part.R.dVals, (const T *)h->factors[1 - side], (const T *)h->dZeros, (T *)h->factors[side],
                (T *)S.dSlabs, h->dErr, lambda, h->copt.factorsCount, 0, 0,
                use_slab_x6(h->copt, side) ? (uint32_t)(h->rows(1 - side) * h->copt.factorsCount * 4) : 0u};
  DualPlan dp;
  dp.noX6 = (h->copt.flags & YCNR_FLAG_NO_BF16X6) != 0;
  const bool lds = (h->copt.flags & YCNR_FLAG_LDS_SOLVER) != 0, edge = use_valu_edge(h->copt), x6 = use_slab_x6(h->copt, side);
  HIP_TRY(hipEventRecord(part.ev[0], stream));
  const int bpr = (int)(B.rankBands[(size_t)B.rank + 1] - B.rankBands[(size_t)B.rank]);
  // The groups are dealt over up to four streams (the handle's two piece streams and two of its side streams): a group is an
  // eighth of the rows -- about one round of waves -- and its longest chunk sets its kernel's length, so the kernels of several
  // groups have to be in flight for the chip to be full (eight launches in stream order: 2.8 ms per rank and item half-step at
  // MAL scale; over two streams 1.8; one launch over all groups, which could not signal a group's completion: 1.2).
  int nStreams = 1;
  hipStream_t gs[kFewSlabs] = {stream, stream, stream, stream, stream, stream, stream, stream};
  if (B.world > 1 && !(h->opt.flags & YCNR_FLAG_NO_OVERLAP) && !env_flags().noOverlap) {
    static const int want = getenv("YCNR_BAND_STREAMS") ? std::max(1, std::min(2 + kSideStreams, atoi(getenv("YCNR_BAND_STREAMS")))) : 4;
    nStreams = std::min(want, B.world);
    for (int i = 0; i < nStreams; ++i) {
      if (i < 2) {
        gs[i] = h->pieceStream[i];
      } else {
        if (!h->sideStream[i - 2]) HIP_TRY(hipStreamCreateWithFlags(&h->sideStream[i - 2], hipStreamNonBlocking));
        gs[i] = h->sideStream[i - 2];
      }
    }
    HIP_TRY(hipEventRecord(h->evStepStart, stream));
    for (int i = 0; i < nStreams; ++i) HIP_TRY(hipStreamWaitEvent(gs[i], h->evStepStart, 0));
  }
  const bool twoStreams = nStreams > 1;
  // stages per launch: consecutive stages' units (and sums) are contiguous, one launch covers `batch` of them -- enough units to
  // fill the chip -- and their band slabs leave together when it has finished
  static const int batchEnv = getenv("YCNR_BAND_BATCH") ? std::max(1, atoi(getenv("YCNR_BAND_BATCH"))) : 0;
  const int batch = batchEnv ? batchEnv : 1;
  for (int s = 1; s <= B.world; ++s) {
    const int g = (B.rank + s) % B.world;  // the own group last: its slabs do not travel
    const int b0 = ((s - 1) / batch) * batch + 1, b1 = std::min(B.world, b0 + batch - 1);  // the stages launched together with s
    hipStream_t ps = gs[((s - 1) / batch) % nStreams];
    if (s == b0) {
      int64_t nu = 0, ns = 0;
      const int g0 = (B.rank + b0) % B.world;
      for (int t = b0; t <= b1; ++t) {
        nu += B.groupUnits[(size_t)((B.rank + t) % B.world)];
        ns += B.groupSums[(size_t)((B.rank + t) % B.world)];
      }
      if (nu > 0) {
        StepArgs<T> ag = a;
        ag.units = S.dUnits + B.groupUnit0[(size_t)g0];
        dp.only = 1;
        int rc = launch_step<T>(ag, nu, nu, 0, ps, nullptr, lds, dp, edge, x6);
        if (rc) return rc;
      }
      if (ns > 0) {
        hipLaunchKernelGGL(als_slab_sum_kernel<T>, dim3((unsigned)ns), dim3(256), 0, ps, (T *)S.dSlabs, (const SlabSum *)B.dSums + B.groupSum0[(size_t)g0], B.slabElems);
        HIP_TRY(hipGetLastError());
      }
    }
    HIP_TRY(hipEventRecord(B.evStage[s - 1], ps));
    if (!comm || g == B.rank) continue;
    const int from = (B.rank - s + B.world) % B.world;  // the rank whose stage-s group this rank is
    const int64_t sendSlabs = (B.ownerBounds[(size_t)g + 1] - B.ownerBounds[(size_t)g]) * bpr;
    const int64_t ownN = B.ownerBounds[(size_t)B.rank + 1] - B.ownerBounds[(size_t)B.rank];
    const int64_t recvSlabs = ownN * (B.rankBands[(size_t)from + 1] - B.rankBands[(size_t)from]);
    h->info.exchangeBytes += (sendSlabs + recvSlabs) * (int64_t)slabBytes;
    if (c.transport == YCNR_COMM_STUB) continue;  // (one rank of an emulated world: the bytes are counted, nothing travels)
    HIP_TRY(hipStreamWaitEvent(c.stream, B.evStage[s - 1], 0));
    const char *src = (const char *)S.dSlabs + (size_t)B.groupSlab0[(size_t)g] * slabBytes;
    if (c.transport == YCNR_COMM_RCCL) {
      const ncclDataType_t dt = ts == 8 ? ncclDouble : ncclFloat;
      NCCL_TRY(c.api, c.api->GroupStart());
      if (sendSlabs > 0) NCCL_TRY(c.api, c.api->Send(src, (size_t)sendSlabs * (size_t)B.slabElems, dt, g, c.nccl, c.stream));
      if (recvSlabs > 0)
        NCCL_TRY(c.api, c.api->Recv((char *)B.dRecv + (size_t)B.recvSlab0[(size_t)from] * slabBytes, (size_t)recvSlabs * (size_t)B.slabElems, dt, from, c.nccl, c.stream));
      NCCL_TRY(c.api, c.api->GroupEnd());
    } else if (c.transport == YCNR_COMM_IPC) {
      if (c.extra[side].size() != (size_t)c.world || (sendSlabs > 0 && !c.extra[side][(size_t)g].ptr))
        return fail(YCNR_ERR_STATE, "ipc: the band-slab buffer of rank %d is not mapped", g);
      // where this rank's slabs start in g's receive buffer: behind those of the ranks before it (g itself sends nothing)
      int64_t off = 0;
      const int64_t gRows = B.ownerBounds[(size_t)g + 1] - B.ownerBounds[(size_t)g];
      for (int r = 0; r < B.rank; ++r)
        if (r != g) off += gRows * (B.rankBands[(size_t)r + 1] - B.rankBands[(size_t)r]);
      if (sendSlabs > 0)
        HIP_TRY(hipMemcpyAsync(c.extra[side][(size_t)g].ptr + (size_t)off * slabBytes, src, (size_t)sendSlabs * slabBytes, hipMemcpyDeviceToDevice, c.stream));
    } else {
      return fail(YCNR_ERR_UNSUPPORTED, "step: this transport does not carry band slabs");
    }
  }
  // the step's stream joins the piece streams: the last stage of each (enqueued behind every launch, as in ycnr_als_step_async)
  if (twoStreams) {
    const int nb = (B.world + batch - 1) / batch;  // the last launch of every stream in use
    for (int bi = std::max(0, nb - nStreams); bi < nb; ++bi) HIP_TRY(hipStreamWaitEvent(stream, B.evStage[std::min(B.world, (bi + 1) * batch) - 1], 0));
  }
  for (int i = 1; i <= 3; ++i) HIP_TRY(hipEventRecord(part.ev[i], stream));  // (no row kernel, no dual classes)
  if (comm && c.transport == YCNR_COMM_RCCL) {
    HIP_TRY(hipEventRecord(B.evRecv, c.stream));
    HIP_TRY(hipStreamWaitEvent(stream, B.evRecv, 0));
  } else if (comm && c.transport == YCNR_COMM_IPC) {
    HIP_TRY(hipStreamSynchronize(c.stream));  // this rank's pushes have landed ...
    if (int rcb = shm_barrier(c)) return rcb;  // ... and everybody's
  }
  if (S.nSplit > 0) {
    dp.only = 2;
    dp.bandSlab = B.dTable;
    dp.nBands = B.nBands;
    int rc = launch_step<T>(a, 0, 0, S.nSplit, stream, nullptr, lds, dp, edge, x6);
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(part.ev[4], stream));
  h->exchangedInStep = comm;
  if (comm) {
    std::vector<int64_t> xb((size_t)B.world), xe((size_t)B.world);
    for (int r = 0; r < B.world; ++r) {
      xb[(size_t)r] = B.ownerBounds[(size_t)r];
      xe[(size_t)r] = B.ownerBounds[(size_t)r + 1];
    }
    int rc = comm_exchange(c, h->factors[side], side, h->opt.factorsCount, ts, xb.data(), xe.data(), stream, part.ready, part.x0, part.x1, &h->info.exchangeBytes);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->evComputeEnd, stream));
    if (c.transport != YCNR_COMM_SHM) HIP_TRY(hipStreamWaitEvent(stream, part.x1, 0));
  }
  HIP_TRY(hipMemcpyAsync(h->hErr, h->dErr, sizeof(ErrInfo), hipMemcpyDeviceToHost, stream));
  return YCNR_OK;
}

extern "C" {

int ycnr_als_step_async(ycnr_als *h, int side) {
  if (!h) return fail(YCNR_ERR_INVALID, "null handle");
  if (side != YCNR_BY_USER && side != YCNR_BY_ITEM) return fail(YCNR_ERR_INVALID, "bad side %d", side);
  std::vector<Part> &parts = h->parts[side];
  if (parts.empty()) return fail(YCNR_ERR_STATE, "step: set_ratings was not called for this side");
  HIP_TRY(hipSetDevice(h->opt.device));
  if (h->banded[side].on) {
    memset(&h->info, 0, sizeof h->info);
    h->graphRun = false;
    h->planesValid[1 - side] = false;
    const int rcb = h->opt.dtype == YCNR_F64 ? step_banded_t<double>(h, side) : step_banded_t<float>(h, side);
    if (rcb) return rcb;
    note_enqueued(h, side, parts);
    return YCNR_OK;
  }
  // With a communicator and a sharded upload the half-step includes its exchange: the rows of piece
  // c travel (on the communicator's stream) while piece c + 1 is being solved, and the step's
  // stream waits for the last piece to land -- the next half-step reads the whole matrix.
  const bool exchange = h->comm.active() && !h->bounds[side].empty() && !h->deferExchange[side];
  if (exchange && h->bounds[side].size() != (size_t)h->comm.world * (parts.size() + 1))
    return fail(YCNR_ERR_STATE, "step: the sharded upload of this side was made for another communicator (bounds of %zu values, world %d x %zu pieces)",
                h->bounds[side].size(), h->comm.world, parts.size());
  h->exchangedInStep = exchange;
  if (exchange) {
    // IPC, a half-step enqueued behind one that has not been completed by ycnr_als_sync (back-to-back
    // ycnr_als_step_async calls): the peers' pushes of THAT half-step land in this rank's replica at the peers' pace, and
    // this half-step reads that matrix as its fixed side -- so the pending half-step is completed here first (this rank's
    // stream drained, then the end-of-step barrier: every rank's pushes have landed everywhere).  Receiver-posted
    // transports (RCCL, SHM) order this on the stream.  Collective like the call itself: every rank sees the same flag.
    if (h->comm.transport == YCNR_COMM_IPC && h->comm.pendingFinish) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (int rcf = ipc_finish(h->comm)) return rcf;
    }
    if (int rcb = ipc_enter(h->comm)) return rcb;  // push transport: no peer is still preparing its replica
  }
  // (behind the completion of a pending IPC half-step, like the planes below: the pad reads the matrix the peers were pushing
  // into -- in front of that guard, back-to-back half-steps of a padded upload could pad rows that were still on their way)
  if (h->kPad) {
    // the FIXED side in full; of the solved side only this rank's rows (launch_part copies them back piece by
    // piece, the rows without ratings among them unchanged)
    const int s = 1 - side;
    const int64_t n = h->rows(s) * h->kPad;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (const float *)h->factors[s],
                       h->padded[s], (int64_t)0, h->rows(s), h->opt.factorsCount, h->kPad);
    HIP_TRY(hipGetLastError());
    for (const Part &p : parts) {
      const int64_t nr = p.R.rowEnd - p.R.rowBegin, m = nr * h->kPad;
      if (m <= 0) continue;
      hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream, (const float *)h->factors[side],
                         h->padded[side], p.R.rowBegin, nr, h->opt.factorsCount, h->kPad);
      HIP_TRY(hipGetLastError());
    }
  }
  // (behind the completion of a pending IPC half-step above: the planes are read from the matrix the peers were pushing into)
  // the fixed matrix of this half-step as bf16 planes, once for all its waves (12.7 K x 100 floats at MAL scale: microseconds);
  // a replayed graph holds the launch itself (it was captured in front of the piece's kernels)
  // (not for uploads below kGraphMinRatings, whose half-step is a handful of launches: one more launch costs what the planes save)
  int64_t sideRatings = 0;
  for (const Part &p : parts) sideRatings += p.R.nnz;
  const bool planesNow = use_planes(h->copt, side) && sideRatings >= kGraphMinRatings;
  if (!planesNow) h->planesValid[1 - side] = false;
  auto split_planes = [&]() -> int {
    if (!planesNow) return YCNR_OK;
    h->planesValid[1 - side] = true;
    const int s = 1 - side, nb = slab_nb(h->copt.factorsCount);
    const bool pack = planes_pack(h->copt.factorsCount);
    const int64_t n = h->rows(s) * nb * 4;
    hipLaunchKernelGGL(als_split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->kPad ? (const float *)h->padded[s] : (const float *)h->factors[s], h->planes[s],
                       h->rows(s), h->copt.factorsCount, nb, pack ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return YCNR_OK;
  };
  if (planesNow && !h->planes[1 - side])
    HIP_TRY(hipMalloc(&h->planes[1 - side], (size_t)h->rows(1 - side) * (size_t)planes_row_bytes(slab_nb(h->copt.factorsCount), planes_pack(h->copt.factorsCount))));
  memset(&h->info, 0, sizeof h->info);
  // Small uploads (the ML-100k / ML-1M shapes): ~15 launches, forks and joins of a half-step whose kernels each fill a
  // fraction of the chip.  Captured once in the branch form (launch_part) and replayed: one launch per half-step.
  h->graphRun = false;
  {
    ycnr_als::GraphSlot &gs = h->graph[side];
    // (a padded upload -- kPad -- is captured too: the pads above are plain launches in front of the graph, the unpad of the piece's rows is part of it)
    const bool graphable = !exchange && parts.size() == 1 && h->copt.factorsCount <= kMaxFactors &&
                           parts[0].R.nnz < kGraphMaxRatings && parts[0].R.nnz >= kGraphMinRatings && h->stream == h->ownStream &&
                           !(h->opt.flags & (YCNR_FLAG_NO_OVERLAP | YCNR_FLAG_NO_GRAPH)) && !env_flags().noOverlap && !env_flags().noGraph;
    if (graphable && gs.state == 1) {
      for (int i = 0; i < side_streams(); ++i)  // (not inside the capture)
        if (!h->sideStream[i]) HIP_TRY(hipStreamCreateWithFlags(&h->sideStream[i], hipStreamNonBlocking));
      gs.state = -1;
      hipGraph_t g = nullptr;
      if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
        int rc = split_planes();
        if (rc == YCNR_OK) rc = launch_part(h, side, parts[0], h->stream, true);
        hipError_t ce = hipMemcpyAsync(h->hErr, h->dErr, sizeof(ErrInfo), hipMemcpyDeviceToHost, h->stream);
        hipError_t ee = hipStreamEndCapture(h->stream, &g);
        if (rc == YCNR_OK && ce == hipSuccess && ee == hipSuccess && g && hipGraphInstantiate(&gs.exec, g, nullptr, nullptr, 0) == hipSuccess)
          gs.state = 2;
        if (g) (void)hipGraphDestroy(g);
      }
      (void)hipGetLastError();
    } else if (graphable && gs.state == 0) {
      gs.state = 1;
    }
    if (graphable && gs.state == 2) {
      Part &p = parts[0];
      HIP_TRY(hipEventRecord(p.ev[0], h->stream));
      HIP_TRY(hipGraphLaunch(gs.exec, h->stream));
      HIP_TRY(hipEventRecord(p.ev[4], h->stream));
      h->graphRun = true;
    }
  }
  if (!h->graphRun)
    if (int rcs = split_planes()) return rcs;
  std::vector<int64_t> xb, xe;
  // several pieces: alternating over the two piece streams (unless YCNR_FLAG_NO_OVERLAP / the staged SHM stand-in,
  // whose exchange blocks the host); one piece: on the step's stream itself
  const bool twoStreams = parts.size() > 1 && !(h->opt.flags & YCNR_FLAG_NO_OVERLAP) && !env_flags().noOverlap &&
                          !(exchange && h->comm.transport == YCNR_COMM_SHM);
  if (twoStreams) {
    HIP_TRY(hipEventRecord(h->evStepStart, h->stream));
    for (int i = 0; i < 2; ++i) HIP_TRY(hipStreamWaitEvent(h->pieceStream[i], h->evStepStart, 0));
  }
  for (size_t c = 0; c < parts.size() && !h->graphRun; ++c) {
    hipStream_t ps = twoStreams ? h->pieceStream[c & 1] : h->stream;
    static const bool pieceSides = getenv("YCNR_PIECE_SIDE_STREAMS") != nullptr;  // A/B: round 3's form (side streams inside every piece)
    int rc = launch_part(h, side, parts[c], ps, false, twoStreams && !pieceSides);
    if (rc) return rc;
    if (exchange) {
      part_ranges(h, side, (int)c, xb, xe);
      rc = comm_exchange(h->comm, h->factors[side], side, h->opt.factorsCount, h->ts(), xb.data(), xe.data(), ps, parts[c].ready,
                         parts[c].x0, parts[c].x1, &h->info.exchangeBytes);
      if (rc) return rc;
    }
    if (twoStreams && c + 2 >= parts.size()) HIP_TRY(hipEventRecord(parts[c].done, ps));  // the last piece of each stream
  }
  // ... joins the step's stream -- enqueued only now, behind every piece's launches: streams share hardware queues (the
  // runtime has 4 or 8 for all the streams of a process), a hardware queue is served in order, and a wait for piece c enqueued
  // on the step's stream BEFORE the launches of piece c + 1 held those launches back whenever the two streams shared a queue
  // (kernel trace of one GPU's eighth of the MAL-scale user side: the fourth piece started when the third had ended).
  if (twoStreams && !h->graphRun)
    for (size_t c = parts.size() >= 2 ? parts.size() - 2 : 0; c < parts.size(); ++c) HIP_TRY(hipStreamWaitEvent(h->stream, parts[c].done, 0));
  if (exchange) {
    HIP_TRY(hipEventRecord(h->evComputeEnd, h->stream));
    if (h->comm.transport != YCNR_COMM_SHM) HIP_TRY(hipStreamWaitEvent(h->stream, parts.back().x1, 0));  // (SHM is synchronous)
  }
  if (!h->graphRun) HIP_TRY(hipMemcpyAsync(h->hErr, h->dErr, sizeof(ErrInfo), hipMemcpyDeviceToHost, h->stream));
  note_enqueued(h, side, parts);
  return YCNR_OK;
}

int ycnr_als_sync(ycnr_als *h) {
  if (!h) return fail(YCNR_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->opt.device));
  HIP_TRY(hipStreamSynchronize(h->stream));  // (polling hipStreamQuery instead was measured: no difference, the runtime's wait spins)
  if (int rcf = ipc_finish(h->comm)) return rcf;  // IPC: every rank's pushes have landed everywhere
  const int nPend = h->infoPending ? h->nPend : 0;
  h->infoPending = false;
  h->nPend = 0;
  for (int pi = 0; pi < nPend; ++pi) {
    const int pside = h->pendOrder[pi];
    const bool last = pi + 1 == nPend;
    h->info = h->pend[pside].info;
    h->graphRun = h->pend[pside].graphRun;
    h->exchangedInStep = h->pend[pside].exchanged;
    const std::vector<Part> &parts = h->parts[pside];
    float ms = 0;
    if (h->graphRun && !parts.empty()) {
      // one interval for the whole half-step: chunks -> reduce, the row kernel and the dual classes ran as branches of one graph
      HIP_TRY(hipEventElapsedTime(&ms, parts[0].ev[0], parts[0].ev[4]));
      h->info.gramSolveMs = ms;
      h->info.dualOverlapped = 1;
    }
    for (const Part &p : parts) {
      if (h->graphRun) break;
      HIP_TRY(hipEventElapsedTime(&ms, p.ev[0], p.ev[1]));
      h->info.gramSlabMs += ms;
      HIP_TRY(hipEventElapsedTime(&ms, p.ev[1], p.ev[2]));
      h->info.gramSolveMs += ms;
      HIP_TRY(hipEventElapsedTime(&ms, p.ev[2], p.ev[3]));
      h->info.dualSolveMs += ms;
      HIP_TRY(hipEventElapsedTime(&ms, p.ev[3], p.ev[4]));
      h->info.reduceSolveMs += ms;
      if (h->exchangedInStep) {
        HIP_TRY(hipEventElapsedTime(&ms, p.x0, p.x1));
        h->info.exchangeMs += ms;
      }
    }
    if (!parts.empty()) {
      h->info.totalMs = 0;
      for (const Part &p : parts) {  // (pieces run two at a time: the last one need not end last)
        HIP_TRY(hipEventElapsedTime(&ms, parts.front().ev[0], p.ev[4]));
        h->info.totalMs = std::max(h->info.totalMs, ms);
      }
      if (h->exchangedInStep) {
        if (h->comm.transport != YCNR_COMM_SHM) {
          // what the step's stream still had to wait for after its own last kernel
          HIP_TRY(hipEventElapsedTime(&ms, h->evComputeEnd, parts.back().x1));
          h->info.exposedExchangeMs = ms > 0 ? ms : 0;
        } else {
          h->info.exposedExchangeMs = h->info.exchangeMs;  // the staged stand-in blocks the stream
          HIP_TRY(hipEventElapsedTime(&ms, parts.back().x0, parts.back().x1));
          h->info.totalMs -= h->info.exchangeMs - ms;  // kernels only (the last exchange lies behind the last kernel)
        }
      }
    }
    // (the error counter is cumulative and copied at the end of every half-step: with two half-steps in flight the rows
    // counted since the last sync are reported with the last one)
    ErrInfo ei = *h->hErr;  // copied by the step's stream, which has drained
    ei.count = last ? ei.count - h->errSeen : 0;
    h->errSeen += ei.count;
    h->info.numericErrors = ei.count;
    h->infoOf[pside] = h->info;
#ifdef YCNR_WG_STAMPS
    if (const char *path = getenv("YCNR_DUMP_STAMPS")) {
      std::vector<unsigned char> buf(kErrBytes);
      HIP_TRY(hipMemcpy(buf.data(), h->dErr, kErrBytes, hipMemcpyDeviceToHost));
      if (FILE *f = fopen(path, "wb")) {
        fwrite(buf.data(), 1, buf.size(), f);
        fclose(f);
      }
    }
#endif
    if (ei.count > 0 && !env_flags().ignoreNumeric)  // the env var exists for timing experiments with ablated kernels
      return fail(YCNR_ERR_NUMERIC, "%d row(s) had a normal matrix that is not positive definite (e.g. row %d)",
                  ei.count, ei.firstRow);
  }
  return YCNR_OK;
}

int ycnr_als_step(ycnr_als *h, int side) {
  int rc = ycnr_als_step_async(h, side);
  if (rc) return rc;
  return ycnr_als_sync(h);
}

int ycnr_als_last_step_info(ycnr_als *h, ycnr_als_step_info *info) {
  if (!h || !info) return fail(YCNR_ERR_INVALID, "null argument");
  if (h->infoPending) return fail(YCNR_ERR_STATE, "last_step_info: call ycnr_als_sync first");
  *info = h->info;
  return YCNR_OK;
}

int ycnr_als_step_info_of(ycnr_als *h, int side, ycnr_als_step_info *info) {
  if (!h || !info) return fail(YCNR_ERR_INVALID, "null argument");
  if (side != YCNR_BY_USER && side != YCNR_BY_ITEM) return fail(YCNR_ERR_INVALID, "bad side %d", side);
  if (h->infoPending) return fail(YCNR_ERR_STATE, "step_info_of: call ycnr_als_sync first");
  if (h->infoOf[side].struct_size == 0) return fail(YCNR_ERR_STATE, "step_info_of: no half-step of this side has been completed");
  *info = h->infoOf[side];
  return YCNR_OK;
}

int ycnr_als_rmse(ycnr_als *h, int which, double shift, int nPortions, const int64_t *portionRowEnd, double *out) {
  if (!h || !out) return fail(YCNR_ERR_INVALID, "null argument");
  if (which != YCNR_RMSE_VALIDATE && which != YCNR_RMSE_TEST) return fail(YCNR_ERR_INVALID, "bad rmse set %d", which);
  const Ratings &R = h->rmse[which];
  if (!R.loaded) return fail(YCNR_ERR_STATE, "rmse: set_rmse_ratings was not called for this set");
  if (nPortions < 0 || (nPortions > 0 && !portionRowEnd)) return fail(YCNR_ERR_INVALID, "bad portions");
  HIP_TRY(hipSetDevice(h->opt.device));
  const int64_t nRows = R.rowEnd - R.rowBegin;
  std::vector<int64_t> ends;
  if (nPortions == 0) {
    ends.push_back(nRows);
  } else {
    int64_t prev = 0;
    for (int p = 0; p < nPortions; ++p) {
      int64_t e = portionRowEnd[p] - R.rowBegin;  // global -> local, clipped to the shard
      e = std::max<int64_t>(0, std::min(nRows, e));
      if (e < prev) return fail(YCNR_ERR_INVALID, "rmse: portionRowEnd must be ascending");
      ends.push_back(e);
      prev = e;
    }
  }
  // One workgroup per piece of at most ceil(nRows / 4096) rows, so that a caller's portioning
  // (one portion = the whole set when nPortions == 0; the reference's default is 10 000 ratings)
  // does not decide how much of the GPU works; the pieces of a portion are added in row order.
  const int nPort = (int)ends.size();
  const int64_t pieceRows = std::max<int64_t>(1, (nRows + 4095) / 4096);
  std::vector<int64_t> pieceEnds;
  std::vector<int> pieceOf;
  {
    int64_t prev = 0;
    for (int p = 0; p < nPort; ++p) {
      if (ends[p] == prev) {
        pieceEnds.push_back(prev);
        pieceOf.push_back(p);
      }
      for (int64_t x = prev; x < ends[p]; x += pieceRows) {
        pieceEnds.push_back(std::min(ends[p], x + pieceRows));
        pieceOf.push_back(p);
      }
      prev = ends[p];
    }
  }
  const int np = (int)pieceEnds.size();
  std::vector<double> part((size_t)3 * np);
  int64_t *dEnds = nullptr;
  double *dOut = nullptr;
  HIP_TRY(hipMalloc(&dEnds, sizeof(int64_t) * np));
  hipError_t e = hipMalloc(&dOut, sizeof(double) * 3 * np);
  if (e == hipSuccess) e = hipMemcpyAsync(dEnds, pieceEnds.data(), sizeof(int64_t) * np, hipMemcpyHostToDevice, h->stream);
  EvPair evp;  // the kernel alone (ycnr_als_last_rmse_ms)
  if (e == hipSuccess) e = evp.create();
  if (e == hipSuccess) e = hipEventRecord(evp.a, h->stream);
  if (e == hipSuccess) {
    if (h->opt.dtype == YCNR_F32) {
      RmseArgs<float> a{R.dRowPtr, R.dIndx, (const float *)R.dVals, (const float *)h->factors[0],
                        (const float *)h->factors[1], dEnds, dOut, shift, R.rowBegin, h->opt.factorsCount};
      launch_rmse(a, np, h->stream);
    } else {
      RmseArgs<double> a{R.dRowPtr, R.dIndx, (const double *)R.dVals, (const double *)h->factors[0],
                         (const double *)h->factors[1], dEnds, dOut, shift, R.rowBegin, h->opt.factorsCount};
      launch_rmse(a, np, h->stream);
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipEventRecord(evp.b, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(part.data(), dOut, sizeof(double) * 3 * np, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) {
    float ms = 0;
    h->lastRmseMs = hipEventElapsedTime(&ms, evp.a, evp.b) == hipSuccess ? (double)ms : -1.0;
  }
  if (e == hipSuccess) {
    for (int i = 0; i < 3 * nPort; ++i) out[i] = 0.0;
    for (int j = 0; j < np; ++j)
      for (int t = 0; t < 3; ++t) out[3 * pieceOf[j] + t] += part[(size_t)3 * j + t];
  }
  if (dEnds) (void)hipFree(dEnds);
  if (dOut) (void)hipFree(dOut);
  if (e != hipSuccess) return fail(YCNR_ERR_HIP, "rmse: %s", hipGetErrorString(e));
  return YCNR_OK;
}

int ycnr_als_last_rmse_ms(ycnr_als *h, double *ms) {
  if (!h || !ms) return fail(YCNR_ERR_INVALID, "null argument");
  if (h->lastRmseMs < 0) return fail(YCNR_ERR_STATE, "last_rmse_ms: no RMSE pass has run on this handle");
  *ms = h->lastRmseMs;
  return YCNR_OK;
}

// ---- multi-GPU exchange (comm_impl.hip.h) ----

int ycnr_comm_unique_id(int transport, void *id) {
  if (!id) return fail(YCNR_ERR_INVALID, "ycnr_comm_unique_id: null id");
  memset(id, 0, YCNR_COMM_ID_BYTES);
  if (transport == YCNR_COMM_RCCL) {
    const RcclApi *api = nullptr;
    int rc = rccl_api(&api);
    if (rc) return rc;
    ncclUniqueId uid;
    NCCL_TRY(api, api->GetUniqueId(&uid));
    memcpy(id, &uid, sizeof uid);
    return YCNR_OK;
  }
  if (transport == YCNR_COMM_STUB) return YCNR_OK;
  if (transport == YCNR_COMM_SHM || transport == YCNR_COMM_IPC) {
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f || fread(id, 1, 16, f) != 16) {
      if (f) fclose(f);
      return fail(YCNR_ERR_HIP, "ycnr_comm_unique_id: cannot read /dev/urandom");
    }
    fclose(f);
    return YCNR_OK;
  }
  return fail(YCNR_ERR_INVALID, "ycnr_comm_unique_id: unknown transport %d", transport);
}

int ycnr_als_comm_init(ycnr_als *h, int transport, const void *id, int rank, int world) {
  if (!h) return fail(YCNR_ERR_INVALID, "null handle");
  if (h->comm.transport != YCNR_COMM_NONE) return fail(YCNR_ERR_STATE, "comm_init: the handle already has a communicator");
  HIP_TRY(hipSetDevice(h->opt.device));
  // the staged stand-in needs room for the larger matrix (an exchange never stages more than one side)
  const size_t big = (size_t)std::max(h->opt.totalUsersCount, h->opt.totalItemsCount) * h->opt.factorsCount * h->ts();
  int rc = comm_setup(h->comm, id, transport, rank, world, std::max<size_t>(big, (size_t)1 << 20));
  if (!rc && transport == YCNR_COMM_IPC) rc = ipc_publish(h->comm, h->factors);
  if (rc) comm_release(h->comm);
  return rc;
}

int ycnr_als_comm_info(ycnr_als *h, int32_t out[4]) {
  if (!h || !out) return fail(YCNR_ERR_INVALID, "null argument");
  out[0] = h->comm.transport;
  out[1] = h->comm.rank;
  out[2] = h->comm.world;
  out[3] = -1;
  if (h->comm.transport == YCNR_COMM_RCCL && h->comm.nccl && h->comm.api && h->comm.api->CommCount) {
    int n = -1;
    NCCL_TRY(h->comm.api, h->comm.api->CommCount(h->comm.nccl, &n));
    out[3] = n;
  }
  return YCNR_OK;
}

int ycnr_als_comm_destroy(ycnr_als *h) {
  if (!h) return fail(YCNR_ERR_INVALID, "null handle");
  (void)hipSetDevice(h->opt.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  comm_release(h->comm);
  // exchange ranges belong to the communicator they were made for: a later ycnr_als_set_ratings[_sharded] renews them
  h->bounds[0].clear();
  h->bounds[1].clear();
  return YCNR_OK;
}

int ycnr_als_defer_exchange(ycnr_als *h, int side, int deferred) {
  if (!h) return fail(YCNR_ERR_INVALID, "null handle");
  if (side != 0 && side != 1) return fail(YCNR_ERR_INVALID, "bad side %d", side);
  h->deferExchange[side] = deferred != 0;
  return YCNR_OK;
}

int ycnr_als_exchange(ycnr_als *h, int side) {
  if (!h) return fail(YCNR_ERR_INVALID, "null handle");
  if (side != 0 && side != 1) return fail(YCNR_ERR_INVALID, "bad side %d", side);
  if (!h->comm.active()) return YCNR_OK;
  if (h->bounds[side].empty() || h->parts[side].empty()) return fail(YCNR_ERR_STATE, "exchange: no sharded upload for this side");
  HIP_TRY(hipSetDevice(h->opt.device));
  const int world = h->comm.world, np = (int)h->parts[side].size();
  if (h->bounds[side].size() != (size_t)world * (size_t)(np + 1)) return fail(YCNR_ERR_STATE, "exchange: the sharded upload was made for another communicator");
  std::vector<int64_t> b((size_t)world), e((size_t)world);
  for (int r = 0; r < world; ++r) {
    b[(size_t)r] = h->bounds[side][(size_t)r * (np + 1)];
    e[(size_t)r] = h->bounds[side][(size_t)r * (np + 1) + np];
  }
  Part &p0 = h->parts[side][0];
  if (int rcb = ipc_enter(h->comm)) return rcb;
  int rc = comm_exchange(h->comm, h->factors[side], side, h->opt.factorsCount, h->ts(), b.data(), e.data(), h->stream, p0.ready, p0.x0, p0.x1, nullptr);
  if (rc) return rc;
  if (h->comm.transport != YCNR_COMM_SHM) HIP_TRY(hipStreamSynchronize(h->comm.stream));
  return ipc_finish(h->comm);
}

int ycnr_als_allreduce_sum(ycnr_als *h, double *vals, int64_t n) {
  if (!h || (n > 0 && !vals)) return fail(YCNR_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(h->opt.device));
  return comm_allreduce_sum(h->comm, vals, n);
}

int ycnr_als_broadcast_factors(ycnr_als *h, int side, int root) {
  if (!h) return fail(YCNR_ERR_INVALID, "null handle");
  if (side != 0 && side != 1) return fail(YCNR_ERR_INVALID, "bad side %d", side);
  HIP_TRY(hipSetDevice(h->opt.device));
  return comm_broadcast(h->comm, h->factors[side], side, (size_t)h->rows(side) * h->opt.factorsCount * h->ts(), root, h->stream);
}

// One self-addressed send / receive pair and one all-reduce through the handle's communicator:
// lets a single-GPU box prove that the RCCL calls of the exchange run (tests/test_gpu_comm.py).
int ycnr_als_comm_selftest(ycnr_als *h, int64_t nFloats) {
  if (!h) return fail(YCNR_ERR_INVALID, "null handle");
  if (h->comm.transport == YCNR_COMM_NONE) return fail(YCNR_ERR_STATE, "comm_selftest: no communicator");
  if (nFloats < 1) return fail(YCNR_ERR_INVALID, "comm_selftest: nFloats < 1");
  HIP_TRY(hipSetDevice(h->opt.device));
  Comm &c = h->comm;
  std::vector<float> src((size_t)nFloats), got((size_t)nFloats);
  for (int64_t i = 0; i < nFloats; ++i) src[(size_t)i] = (float)((i * 2654435761u) % 65521) + 0.25f * (float)c.rank;
  double sum[3] = {1.0 + c.rank, 2.0, -0.5 * c.rank};
  if (c.transport == YCNR_COMM_RCCL) {
    DevBuf a, b;
    HIP_TRY(hipMalloc(&a.p, sizeof(float) * (size_t)nFloats));
    HIP_TRY(hipMalloc(&b.p, sizeof(float) * (size_t)nFloats));
    HIP_TRY(hipMemcpy(a.p, src.data(), sizeof(float) * (size_t)nFloats, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(b.p, 0, sizeof(float) * (size_t)nFloats));
    NCCL_TRY(c.api, c.api->GroupStart());
    NCCL_TRY(c.api, c.api->Send(a.p, (size_t)nFloats, ncclFloat, c.rank, c.nccl, c.stream));
    NCCL_TRY(c.api, c.api->Recv(b.p, (size_t)nFloats, ncclFloat, c.rank, c.nccl, c.stream));
    NCCL_TRY(c.api, c.api->GroupEnd());
    HIP_TRY(hipStreamSynchronize(c.stream));
    HIP_TRY(hipMemcpy(got.data(), b.p, sizeof(float) * (size_t)nFloats, hipMemcpyDeviceToHost));
    if (memcmp(got.data(), src.data(), sizeof(float) * (size_t)nFloats) != 0)
      return fail(YCNR_ERR_STATE, "comm_selftest: the self-addressed ncclSend / ncclRecv pair did not return the bytes sent");
  }
  int rc = comm_allreduce_sum(c, sum, 3);
  if (rc) return rc;
  const double w = c.world;
  const double want[3] = {w + w * (w - 1) / 2, 2.0 * w, -0.5 * w * (w - 1) / 2};
  for (int i = 0; i < 3; ++i)
    if (!(fabs(sum[i] - want[i]) <= 1e-12 * (1 + fabs(want[i])))) return fail(YCNR_ERR_STATE, "comm_selftest: all-reduce gave %g, expected %g", sum[i], want[i]);
  return YCNR_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------ level 1

namespace {

// Per-thread state of the level-1 portion ops: the reference calls mw_calcTrainAlsPortion ~11 500
// times per MAL half-step (EmfBase.js:99-103: 10 000 ratings per portion), so stream, device
// buffers and -- when the host pins it -- the fixed factor matrix stay alive between calls.
struct DevArena {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 4096);
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct L1Ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  DevArena io, slabs, misc;  // io: [solved | ErrInfo | units | split | indx | vals | compacted fixed rows]; misc: zero row
  // page-locked host image of io: ONE copy in (everything behind the solved rows) and ONE copy out (solved rows +
  // ErrInfo) per portion instead of five pageable copies, two memsets and two copies back
  void *hostIo = nullptr;
  size_t hostIoCap = 0;
  hipError_t reserve_host(size_t bytes) {
    if (bytes <= hostIoCap) return hipSuccess;
    if (hostIo) (void)hipHostFree(hostIo);
    hostIo = nullptr;
    hostIoCap = 0;
    const size_t cap = bytes + bytes / 2;
    hipError_t e = hipHostMalloc(&hostIo, cap, hipHostMallocDefault);
    if (e == hipSuccess) hostIoCap = cap;
    return e;
  }
  // fixed matrix pinned by ycnr_{s,d}AlsPinFixedFactors
  const void *pinnedHost = nullptr;
  int64_t pinnedRows = 0;
  int pinnedK = 0, pinnedDtype = -1;
  DevArena pinned;
  void release() {
    if (device >= 0) (void)hipSetDevice(device);
    if (stream) {
      (void)hipStreamSynchronize(stream);
      (void)hipStreamDestroy(stream);
    }
    stream = nullptr;
    for (DevArena *a : {&io, &slabs, &misc, &pinned}) a->release();
    if (hostIo) (void)hipHostFree(hostIo);
    hostIo = nullptr;
    hostIoCap = 0;
    pinnedHost = nullptr;
    device = -1;
  }
  ~L1Ctx() { release(); }
};

L1Ctx &l1ctx() {
  static thread_local L1Ctx c;
  return c;
}

// stream + the small fixed buffers, on the calling thread's current device
int l1_prepare(L1Ctx &C) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (C.device != dev) {
    C.release();
    C.device = dev;
  }
  if (!C.stream) HIP_TRY(hipStreamCreateWithFlags(&C.stream, hipStreamNonBlocking));
  if (!C.misc.p) {
    HIP_TRY(C.misc.reserve(kZeroRowBytes));
    HIP_TRY(hipMemsetAsync(C.misc.p, 0, kZeroRowBytes, C.stream));
  }
  return YCNR_OK;
}

template <typename T>
int pin_fixed(const T *fixed, int64_t rows, int k, int dtype) {
  if (!fixed || rows < 1 || k < 1) return fail(YCNR_ERR_INVALID, "AlsPinFixedFactors: bad argument");
  L1Ctx &C = l1ctx();
  int rc = l1_prepare(C);
  if (rc) return rc;
  const size_t bytes = (size_t)rows * k * sizeof(T);
  HIP_TRY(C.pinned.reserve(bytes));
  HIP_TRY(hipMemcpyAsync(C.pinned.p, fixed, bytes, hipMemcpyHostToDevice, C.stream));
  HIP_TRY(hipStreamSynchronize(C.stream));
  C.pinnedHost = fixed;
  C.pinnedRows = rows;
  C.pinnedK = k;
  C.pinnedDtype = dtype;
  return YCNR_OK;
}

// Shared body of ycnr_{s,d}AlsCalcPortion.  Unless the caller has pinned this fixed matrix, the portion's
// column ids are compacted on the host (a 10 000-rating portion never needs more than 10 000 fixed
// rows) and only those rows are uploaded; the solved rows are scattered back into the caller's matrix.
template <typename T>
int64_t als_calc_portion(double lambda, int k, const int32_t *alsRows, const int32_t *alsIndx, const T *alsVals,
                         const T *fixedFactors, int64_t fixedRows, T *solvedFactors, int64_t solvedRows, int dtype) {
  if (!alsRows || !alsIndx || !alsVals || !fixedFactors || !solvedFactors)
    return fail(YCNR_ERR_INVALID, "AlsCalcPortion: null argument");
  if (k < 1) return fail(YCNR_ERR_INVALID, "AlsCalcPortion: k < 1");
  if (k > kMaxFactorsAny) return fail(YCNR_ERR_UNSUPPORTED, "factorsCount %d > %d is not supported by this build", k, kMaxFactorsAny);
  // the any-k path also takes float32 rows that are not 16-byte multiples above 128 factors (the resident trainer pads those)
  const bool gen = is_gen(dtype, k) || (k > kMaxFactors && k % 4 != 0);
  const bool big = !gen && k > kMaxFactors;
  if (!(lambda >= 0)) return fail(YCNR_ERR_INVALID, "AlsCalcPortion: negative lambda");
  const int nRows = alsRows[0];
  if (nRows < 0) return fail(YCNR_ERR_INVALID, "AlsCalcPortion: alsRows[0] < 0");
  if (nRows == 0) return 0;
  std::vector<int64_t> rowPtr((size_t)nRows + 1, 0);
  for (int r = 0; r < nRows; ++r) {
    const int rowId = alsRows[1 + 2 * r], cols = alsRows[2 + 2 * r];
    if (rowId < 0 || rowId >= solvedRows) return fail(YCNR_ERR_INVALID, "AlsCalcPortion: rowId %d outside [0, %lld)", rowId, (long long)solvedRows);
    if (cols < 0) return fail(YCNR_ERR_INVALID, "AlsCalcPortion: negative cols");
    rowPtr[r + 1] = rowPtr[r] + cols;
  }
  const int64_t total = rowPtr[nRows];
  if (total == 0) return 0;
  int32_t lo = alsIndx[0], hi = alsIndx[0];
  for (int64_t i = 1; i < total; ++i) {
    lo = std::min(lo, alsIndx[i]);
    hi = std::max(hi, alsIndx[i]);
  }
  if (lo < 0 || hi >= fixedRows)
    return fail(YCNR_ERR_INVALID, "AlsCalcPortion: column id out of range (min %d, max %d, fixedRows %lld)", lo, hi, (long long)fixedRows);
  L1Ctx &C = l1ctx();
  int rc = l1_prepare(C);
  if (rc) return rc;
  // The reference's host updates its matrices in place: the matrix pinned for 'byUser' is the one 'byItem' writes.
  // A portion that SOLVES into the pinned host range makes the device copy stale: drop the pin (the host pins
  // again at its next 'startTrainStep'; until then portions upload the rows they need).
  if (C.pinnedHost) {
    const char *pb = (const char *)C.pinnedHost, *pe = pb + (size_t)C.pinnedRows * C.pinnedK * tsize(C.pinnedDtype);
    const char *sb = (const char *)solvedFactors, *se = sb + (size_t)solvedRows * k * sizeof(T);
    if (sb < pe && pb < se) C.pinnedHost = nullptr;
  }
  const bool pinned = C.pinnedHost == fixedFactors && C.pinnedRows == fixedRows && C.pinnedK == k && C.pinnedDtype == dtype;
  // compact the referenced fixed rows unless the whole matrix is resident
  std::vector<int32_t> uniq, remap;  // remap: fixed row -> compacted row when the fixed side is small enough to mark
  if (!pinned) {
    if (fixedRows <= 16 * total) {
      remap.assign((size_t)fixedRows, -1);
      for (int64_t i = 0; i < total; ++i) remap[alsIndx[i]] = 0;
      for (int64_t r = 0; r < fixedRows; ++r)
        if (remap[r] == 0) {
          remap[r] = (int32_t)uniq.size();
          uniq.push_back((int32_t)r);
        }
    } else {
      uniq.assign(alsIndx, alsIndx + total);
      std::sort(uniq.begin(), uniq.end());
      uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    }
  }
  std::vector<Unit> units;
  std::vector<SplitRow> split;
  int64_t nSlabs = 0, solved = 0;
  build_schedule(rowPtr.data(), 0, nRows, gen ? kGenChunk : big ? kWgChunk : kDefaultChunk, units, split, nSlabs, solved, gen ? 0 : -1,
                 big ? kWgFusedMax : 0);

  const int nb = slab_nb(k);
  const size_t slabElems = gen ? (size_t)gen_slab_elems(nb) : big ? (size_t)wg_slab_floats(nb) : (size_t)slab_elems(nb);
#define L1_TRY(expr)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return fail(e_ == hipErrorOutOfMemory ? YCNR_ERR_NOMEM : YCNR_ERR_HIP, "%s failed: %s", #expr,  \
                  hipGetErrorString(e_));                                                             \
  } while (0)
  // layout of the io arena (device) and of its page-locked host image, 256-byte aligned parts
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t oSolved = 0, bSolved = sizeof(T) * (size_t)nRows * k;
  const size_t oErr = al(oSolved + bSolved);
  const size_t oUnits = al(oErr + sizeof(ErrInfo)), oSplit = al(oUnits + sizeof(Unit) * std::max<size_t>(1, units.size()));
  const size_t oIndx = al(oSplit + sizeof(SplitRow) * std::max<size_t>(1, split.size()));
  const size_t oVals = al(oIndx + sizeof(int32_t) * (size_t)total + 64);
  const size_t oFixed = al(oVals + sizeof(T) * (size_t)total + 64);
  const size_t ioBytes = al(oFixed + (pinned ? 0 : sizeof(T) * uniq.size() * (size_t)k));
  L1_TRY(C.io.reserve(ioBytes));
  L1_TRY(C.reserve_host(ioBytes));
  // any-k path: rows are solved in batches whose images fit the arena, as in the resident trainer (build_part) -- one
  // image is 541 KB at k = 512 and several MB beyond 1024 factors, a portion of 10 K short rows would ask for gigabytes
  std::vector<GenBatch> genBatches;
  int64_t arenaSlabs = nSlabs;
  if (gen) {
    int64_t arenaBytes = kGenArenaBytes;
    if (const char *e = getenv("YCNR_GEN_ARENA_MB")) arenaBytes = (int64_t)std::max(1, atoi(e)) << 20;  // tests force several batches
    genBatches = gen_batches(split, std::max<int64_t>(1, arenaBytes / (int64_t)(slabElems * sizeof(T))));
    arenaSlabs = 0;
    for (const GenBatch &b : genBatches) arenaSlabs = std::max<int64_t>(arenaSlabs, b.nSlabs);
  }
  L1_TRY(C.slabs.reserve(sizeof(T) * std::max<size_t>(1, (size_t)arenaSlabs * slabElems)));
  char *hb = (char *)C.hostIo, *db = (char *)C.io.p;
  memset(hb + oErr, 0, sizeof(ErrInfo));
  if (!units.empty()) memcpy(hb + oUnits, units.data(), sizeof(Unit) * units.size());
  if (!split.empty()) memcpy(hb + oSplit, split.data(), sizeof(SplitRow) * split.size());
  if (pinned) {
    memcpy(hb + oIndx, alsIndx, sizeof(int32_t) * (size_t)total);
  } else {
    int32_t *cidx = (int32_t *)(hb + oIndx);
    if (!remap.empty()) {
      for (int64_t i = 0; i < total; ++i) cidx[i] = remap[alsIndx[i]];
    } else {
      for (int64_t i = 0; i < total; ++i) cidx[i] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), alsIndx[i]) - uniq.begin());
    }
    T *cfix = (T *)(hb + oFixed);
    for (size_t u = 0; u < uniq.size(); ++u) memcpy(cfix + u * k, fixedFactors + (size_t)uniq[u] * k, sizeof(T) * k);
  }
  memcpy(hb + oVals, alsVals, sizeof(T) * (size_t)total);
  hipStream_t stream = C.stream;
  ErrInfo *dErr = (ErrInfo *)(db + oErr);
  T *dZeros = (T *)C.misc.p;
  L1_TRY(hipMemcpyAsync(db + oErr, hb + oErr, ioBytes - oErr, hipMemcpyHostToDevice, stream));
  {
    // rows are numbered 0..nRows-1 on the device and scattered to rowId on the host
    const T *dFixed = pinned ? (const T *)C.pinned.p : (const T *)(db + oFixed);
    StepArgs<T> a{(const Unit *)(db + oUnits), (const SplitRow *)(db + oSplit), (const int32_t *)(db + oIndx), (const T *)(db + oVals), dFixed, dZeros,
                  (T *)(db + oSolved), (T *)C.slabs.p, dErr, lambda, k, 0, 0, 0u};
    if (gen) {
      rc = launch_step_gen<T>(a, genBatches, stream, nullptr, DualPlan());
    } else if constexpr (std::is_same<T, float>::value) {
      if (big) rc = launch_step_big(a, (int64_t)units.size(), nSlabs, (int64_t)split.size(), stream, nullptr, DualPlan());
      else rc = launch_step<T>(a, (int64_t)units.size(), nSlabs, (int64_t)split.size(), stream, nullptr);
    } else {
      rc = launch_step<T>(a, (int64_t)units.size(), nSlabs, (int64_t)split.size(), stream, nullptr);
    }
    if (rc) return rc;
  }
  // rows without ratings are never written by the kernels and never copied back below
  L1_TRY(hipMemcpyAsync(hb + oSolved, db + oSolved, oErr + sizeof(ErrInfo), hipMemcpyDeviceToHost, stream));
  L1_TRY(hipStreamSynchronize(stream));
#undef L1_TRY
  const ErrInfo ei = *(const ErrInfo *)(hb + oErr);
  const T *hostSolved = (const T *)(hb + oSolved);
  if (ei.count > 0)
    return fail(YCNR_ERR_NUMERIC, "%d row(s) of the portion had a normal matrix that is not positive definite", ei.count);
  for (int r = 0; r < nRows; ++r) {
    if (rowPtr[r + 1] == rowPtr[r]) continue;  // cols == 0: left untouched
    memcpy(solvedFactors + (size_t)alsRows[1 + 2 * r] * k, hostSolved + (size_t)r * k, sizeof(T) * k);
  }
  return total;
}

template <typename T>
int rmse_portion(int k, const int32_t *rows, const int32_t *indx, const T *vals, const T *uF, int64_t usersRows,
                 const T *iF, int64_t itemsRows, double shift, double *out3, int dtype) {
  if (!rows || !indx || !vals || !uF || !iF || !out3) return fail(YCNR_ERR_INVALID, "RmsePortion: null argument");
  if (k < 1) return fail(YCNR_ERR_INVALID, "RmsePortion: k < 1");
  const int nRows = rows[0];
  out3[0] = out3[1] = out3[2] = 0;
  if (nRows < 0) return fail(YCNR_ERR_INVALID, "RmsePortion: rmseRows[0] < 0");
  if (nRows == 0) return YCNR_OK;
  // compact users and items, then run the resident kernel on the compacted problem
  std::vector<int64_t> rowPtr((size_t)nRows + 1, 0);
  for (int r = 0; r < nRows; ++r) {
    const int u = rows[1 + 2 * r], cols = rows[2 + 2 * r];
    if (u < 0 || u >= usersRows || cols < 0) return fail(YCNR_ERR_INVALID, "RmsePortion: bad row entry %d", r);
    rowPtr[r + 1] = rowPtr[r] + cols;
  }
  const int64_t total = rowPtr[nRows];
  if (total == 0) return YCNR_OK;
  std::vector<int32_t> uniq(indx, indx + total);
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  if (uniq.front() < 0 || uniq.back() >= itemsRows) return fail(YCNR_ERR_INVALID, "RmsePortion: item id out of range");
  std::vector<int32_t> cidx((size_t)total);
  for (int64_t i = 0; i < total; ++i)
    cidx[i] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), indx[i]) - uniq.begin());
  std::vector<T> cI(uniq.size() * (size_t)k), cU((size_t)nRows * k);
  for (size_t j = 0; j < uniq.size(); ++j) memcpy(&cI[j * k], iF + (size_t)uniq[j] * k, sizeof(T) * k);
  for (int r = 0; r < nRows; ++r) memcpy(&cU[(size_t)r * k], uF + (size_t)rows[1 + 2 * r] * k, sizeof(T) * k);

  ycnr_als_options o{};
  o.struct_size = (int32_t)sizeof o;
  o.device = 0;
  o.dtype = dtype;
  o.factorsCount = k;
  o.totalUsersCount = nRows;
  o.totalItemsCount = (int64_t)uniq.size();
  ycnr_als *h = nullptr;
  int rc = ycnr_als_create(&o, &h);
  if (rc) return rc;
  rc = ycnr_als_set_factors(h, YCNR_BY_USER, cU.data(), YCNR_MEM_HOST);
  if (!rc) rc = ycnr_als_set_factors(h, YCNR_BY_ITEM, cI.data(), YCNR_MEM_HOST);
  if (!rc) rc = ycnr_als_set_rmse_ratings(h, YCNR_RMSE_VALIDATE, rowPtr.data(), cidx.data(), vals, 0, nRows, YCNR_MEM_HOST);
  if (!rc) rc = ycnr_als_rmse(h, YCNR_RMSE_VALIDATE, shift, 0, nullptr, out3);
  ycnr_als_destroy(h);
  return rc;
}

}  // namespace

extern "C" {

int64_t ycnr_sAlsCalcPortion(double lambda, int k, const int32_t *alsRows, const int32_t *alsIndx, const float *alsVals,
                             const float *fixedFactors, int64_t fixedRows, float *solvedFactors, int64_t solvedRows) {
  return als_calc_portion<float>(lambda, k, alsRows, alsIndx, alsVals, fixedFactors, fixedRows, solvedFactors,
                                 solvedRows, YCNR_F32);
}

int64_t ycnr_dAlsCalcPortion(double lambda, int k, const int32_t *alsRows, const int32_t *alsIndx, const double *alsVals,
                             const double *fixedFactors, int64_t fixedRows, double *solvedFactors, int64_t solvedRows) {
  return als_calc_portion<double>(lambda, k, alsRows, alsIndx, alsVals, fixedFactors, fixedRows, solvedFactors,
                                  solvedRows, YCNR_F64);
}

int ycnr_sAlsPinFixedFactors(const float *fixedFactors, int64_t fixedRows, int k) {
  return pin_fixed<float>(fixedFactors, fixedRows, k, YCNR_F32);
}
int ycnr_dAlsPinFixedFactors(const double *fixedFactors, int64_t fixedRows, int k) {
  return pin_fixed<double>(fixedFactors, fixedRows, k, YCNR_F64);
}
int ycnr_AlsUnpinFixedFactors(void) {
  l1ctx().pinnedHost = nullptr;
  return YCNR_OK;
}
int ycnr_AlsReleasePortionState(void) {
  l1ctx().release();
  return YCNR_OK;
}

int ycnr_sRmsePortion(int k, const int32_t *rmseRows, const int32_t *rmseIndx, const float *rmseVals,
                      const float *userFactors, int64_t usersRows, const float *itemFactors, int64_t itemsRows,
                      double globalAvgShift, double *out3) {
  return rmse_portion<float>(k, rmseRows, rmseIndx, rmseVals, userFactors, usersRows, itemFactors, itemsRows,
                             globalAvgShift, out3, YCNR_F32);
}

int ycnr_dRmsePortion(int k, const int32_t *rmseRows, const int32_t *rmseIndx, const double *rmseVals,
                      const double *userFactors, int64_t usersRows, const double *itemFactors, int64_t itemsRows,
                      double globalAvgShift, double *out3) {
  return rmse_portion<double>(k, rmseRows, rmseIndx, rmseVals, userFactors, usersRows, itemFactors, itemsRows,
                              globalAvgShift, out3, YCNR_F64);
}

}  // extern "C"
