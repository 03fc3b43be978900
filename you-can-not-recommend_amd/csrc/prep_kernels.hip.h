// Preprocessing kernels for the step directly before the ALS path:
// the per-user train / validate / test split and the per-row rating statistics.
//
// Reference: EmfLord.doSplitToSets (lib/emf/EmfLord.js:402-505) shuffles each user's unassigned
// ratings and cuts the shuffle into the counts its formula gives; updateUsersStats /
// updateItemsStats (EmfLord.js:252-396) count and average each row's ratings with
// dataset_type IN (1, 2, 3).  Both go through PostgreSQL there (1 h 05 m on MAL, README.md:127).
//
// The shuffle is defined without a sequential generator so that every implementation (this
// kernel, oracle/als_oracle.c, hosts) produces the same split: unassigned rating j of row r gets
// the key ycnr_split_key(seed, r, j) and the "shuffled order" is ascending (key, j).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ycnr {

__host__ __device__ inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
// the key of include/ycnr_als.h (ycnr_split_to_sets)
__host__ __device__ inline uint32_t split_key(uint32_t seed, uint32_t row, uint32_t j) {
  return fmix32(fmix32(seed + 0x9e3779b9u * row) ^ j);
}

// EmfLord.js:447-457: how many of a row's free ratings go to each set
__host__ __device__ inline void split_counts(int64_t freeCnt, int64_t c1, int64_t c2, int64_t c3, int p0, int p1,
                                             int64_t (&nw)[3]) {
  const int64_t total = freeCnt + c1 + c2 + c3;
  int64_t t[3];
  t[0] = (total * p0 + 99) / 100;                // Math.ceil(totalCnt * pcts[0] / 100)
  t[1] = (total * (p0 + p1) + 99) / 100 - t[0];  // Math.ceil(totalCnt * (pcts[0] + pcts[1]) / 100) - targetCnts[0]
  t[2] = total - (t[0] + t[1]);
  nw[0] = t[0] > c1 ? t[0] - c1 : 0;
  nw[1] = t[1] > c2 ? t[1] - c2 : 0;
  nw[2] = t[2] > c3 ? t[2] - c3 : 0;
  if (nw[0] + nw[1] + nw[2] < freeCnt) nw[0] += freeCnt - (nw[0] + nw[1] + nw[2]);
}

constexpr int kSplitRankRow = 40;      // rows up to this length: every rating ranked by one wave; longer ones find the two thresholds by bisection
                                       // (MAL scale, ms for 16 / 40 / 80 / 192 ratings: 3.04 / 2.80 / 3.10 / 3.95)
constexpr int kSplitShortRow = 1024;   // rows up to this length: one wave, 5 KB of LDS, many waves per CU
constexpr int kSplitLdsKeys = 12288;   // longer rows: 60 KB per wave; beyond this keys are recomputed in the inner loop

// One wave per row.  Rank of every free rating among the row's free ratings by (key, j): O(n^2 / 64)
// compares per row with keys and free flags kept in LDS -- the ratings themselves are never read.
// rowList: the rows this launch handles (the host sorts rows into a short and a long class so
// that the many short rows do not pay for the long rows' LDS).  BLOCK threads share one row:
// a wave for short rows, 16 waves for long ones (each wave ranks its own 64-blocks of j).
template <int CAP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void split_to_sets_kernel(const int64_t *rowPtr, const int32_t *rowList, int64_t nList, int8_t *types,
                                                             int p0, int p1, uint32_t seed) {
  __shared__ uint32_t keys[CAP];
  __shared__ uint8_t isFree[CAP];
  __shared__ int cshared[4];
  const int tid = threadIdx.x;
  for (int64_t li = blockIdx.x; li < nList; li += gridDim.x) {
    const int64_t r = rowList[li];
    const int64_t b = rowPtr[r], n = rowPtr[r + 1] - b;
    if (n <= 0) continue;
    int8_t *t = types + b;
    __syncthreads();  // the previous row's LDS contents are dead
    if (tid < 4) cshared[tid] = 0;
    __syncthreads();
    // existing assignments and the free count
    int c[4] = {0, 0, 0, 0};
    for (int64_t j = tid; j < n; j += BLOCK) {
      const int v = t[j];
      if (v >= 0 && v <= 3) ++c[v];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      for (int m = 32; m >= 1; m >>= 1) c[q] += __shfl_xor(c[q], m, 64);
      if ((tid & 63) == 0 && c[q]) atomicAdd(&cshared[q], c[q]);
    }
    __syncthreads();
    const int64_t c0 = cshared[0], c1 = cshared[1], c2 = cshared[2], c3 = cshared[3];
    if (c0 == 0) continue;
    int64_t nw[3];
    split_counts(c0, c1, c2, c3, p0, p1, nw);
    const bool cached = n <= CAP;
    if (cached) {
      for (int64_t j = tid; j < n; j += BLOCK) {
        keys[j] = split_key(seed, (uint32_t)r, (uint32_t)j);
        isFree[j] = t[j] == 0;
      }
    }
    __syncthreads();
    // Results are written after the whole row has been ranked when it is cached; a longer row
    // marks what it has assigned with bit 6 so that later threads still count it as free.
    for (int64_t j0 = 0; j0 < n; j0 += BLOCK) {
      const int64_t j = j0 + tid;
      const bool mine = j < n && (cached ? isFree[j] != 0 : t[j] == 0);
      const uint32_t kj = j < n ? (cached ? keys[j] : split_key(seed, (uint32_t)r, (uint32_t)j)) : 0u;
      int64_t rank = 0;
      if (cached) {
        for (int64_t i = 0; i < n; ++i) {
          const uint32_t ki = keys[i];
          rank += (isFree[i] && (ki < kj || (ki == kj && i < j))) ? 1 : 0;
        }
      } else {
        for (int64_t i = 0; i < n; ++i) {
          const int v = t[i];
          const uint32_t ki = split_key(seed, (uint32_t)r, (uint32_t)i);
          rank += ((v == 0 || (v & 0x40)) && (ki < kj || (ki == kj && i < j))) ? 1 : 0;
        }
        __syncthreads();  // every thread has finished reading t[] for this block of j
      }
      if (mine) {
        const int8_t set = rank < nw[0] ? 1 : rank < nw[0] + nw[1] ? 2 : rank < nw[0] + nw[1] + nw[2] ? 3 : 0;
        t[j] = cached ? set : (int8_t)(set | 0x40);
      }
      if (!cached) __syncthreads();
    }
    if (!cached) {
      __syncthreads();
      for (int64_t j = tid; j < n; j += BLOCK)
        if (t[j] & 0x40) t[j] = (int8_t)(t[j] & 0x3f);
    }
  }
}

// Long rows (round 3): the sets only ask on which side of two thresholds a rating's (key, j) lies -- the pairs of the free
// ratings of rank nw[0] and nw[0] + nw[1] -- so the ranks themselves are never formed: both thresholds are found by bisection
// over the 64-bit pair, one count of "free and below" per step with the keys (and free flags) of the row in LDS.
// O(n / BLOCK) per thread and step, about log2(n) + 2 steps per row, instead of O(n^2 / BLOCK) compares: the longest MAL-scale user
// (13 K ratings) used to take 2 ms of one CU.  Same sets bit for bit as split_to_sets_kernel and the oracle (the order is
// total: (key, j) pairs are unique).  Rows longer than CAP recompute keys and re-read flags in every step.
template <int CAP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void split_to_sets_select_kernel(const int64_t *rowPtr, const int32_t *rowList, int64_t nList,
                                                                    int8_t *types, int p0, int p1, uint32_t seed) {
  __shared__ uint32_t keys[CAP];
  __shared__ uint8_t isFree[CAP];
  __shared__ int cshared[4];
  __shared__ unsigned cnt[2];
  const int tid = threadIdx.x;
  for (int64_t li = blockIdx.x; li < nList; li += gridDim.x) {
    const int64_t r = rowList[li];
    const int64_t b = rowPtr[r], n = rowPtr[r + 1] - b;
    if (n <= 0) continue;
    int8_t *t = types + b;
    __syncthreads();  // the previous row's LDS contents are dead
    if (tid < 4) cshared[tid] = 0;
    __syncthreads();
    int c[4] = {0, 0, 0, 0};
    for (int64_t j = tid; j < n; j += BLOCK) {
      const int v = t[j];
      if (v >= 0 && v <= 3) ++c[v];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      for (int m = 32; m >= 1; m >>= 1) c[q] += __shfl_xor(c[q], m, 64);
      if ((tid & 63) == 0 && c[q]) atomicAdd(&cshared[q], c[q]);
    }
    __syncthreads();
    const int64_t c0 = cshared[0], c1 = cshared[1], c2 = cshared[2], c3 = cshared[3];
    if (c0 == 0) continue;
    int64_t nw[3];
    split_counts(c0, c1, c2, c3, p0, p1, nw);
    const bool cached = n <= CAP;
    if (cached) {
      for (int64_t j = tid; j < n; j += BLOCK) {
        keys[j] = split_key(seed, (uint32_t)r, (uint32_t)j);
        isFree[j] = t[j] == 0;
      }
    }
    __syncthreads();
    // A threshold theta with below(theta) == R, where below(v) = the number of free ratings whose pair (key << 32 | j) is < v:
    // then exactly the R first free ratings of the keyed order lie below it.  For R = nw[0] and nw[0] + nw[1] at once, by
    // bisection with the invariant below(lo) <= R < below(hi) (below(0) = 0; below(2^64 - 1) = c0: j < n keeps every pair
    // under it), stopping at the first midpoint whose count IS R -- about log2(n) + 2 steps, since the keys are hashes --
    // and at lo when the interval has closed (then below(lo) == R).  R >= c0: every free rating lies below the threshold.
    // One pass over the row per step; a one-wave block needs neither LDS nor barriers for the counts.
    const int64_t R[2] = {nw[0], nw[0] + nw[1]};
    unsigned long long lo[2] = {0ull, 0ull}, hi[2] = {~0ull, ~0ull}, theta[2] = {0ull, 0ull};
    bool found[2] = {R[0] >= c0 || R[0] == 0, R[1] >= c0 || R[1] == 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) theta[q] = R[q] >= c0 ? ~0ull : 0ull;
    for (int step = 0; step < 64 && !(found[0] && found[1]); ++step) {
      unsigned long long mid[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) mid[q] = lo[q] + ((hi[q] - lo[q]) >> 1);
      if constexpr (BLOCK > 64) {
        if (tid < 2) cnt[tid] = 0;
        __syncthreads();
      }
      unsigned mine[2] = {0u, 0u};
      for (int64_t j = tid; j < n; j += BLOCK) {
        const bool fr = cached ? isFree[j] != 0 : t[j] == 0;
        const uint32_t k = cached ? keys[j] : split_key(seed, (uint32_t)r, (uint32_t)j);
        const unsigned long long pj = ((unsigned long long)k << 32) | (unsigned long long)j;
        mine[0] += (fr && pj < mid[0]) ? 1u : 0u;
        mine[1] += (fr && pj < mid[1]) ? 1u : 0u;
      }
      int64_t total[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        for (int m = 32; m >= 1; m >>= 1) mine[q] += __shfl_xor(mine[q], m, 64);
        if constexpr (BLOCK > 64) {
          if ((tid & 63) == 0 && mine[q]) atomicAdd(&cnt[q], mine[q]);
        } else {
          total[q] = mine[q];
        }
      }
      if constexpr (BLOCK > 64) {
        __syncthreads();
        total[0] = cnt[0];
        total[1] = cnt[1];
        __syncthreads();  // cnt[] is reset at the top of the next step
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (found[q]) continue;
        if (total[q] == R[q]) {
          theta[q] = mid[q];
          found[q] = true;
        } else if (total[q] > R[q]) {
          hi[q] = mid[q];
        } else {
          lo[q] = mid[q];
        }
        if (!found[q] && hi[q] - lo[q] <= 1) {  // closed: below(lo) <= R < below(lo + 1) <= below(lo) + 1
          theta[q] = lo[q];
          found[q] = true;
        }
      }
    }
    // (pairs below theta[q] are exactly the ratings of rank < R[q])
    for (int64_t j = tid; j < n; j += BLOCK) {
      if (t[j] != 0) continue;
      const uint32_t k = cached ? keys[j] : split_key(seed, (uint32_t)r, (uint32_t)j);
      const unsigned long long pj = ((unsigned long long)k << 32) | (unsigned long long)j;
      t[j] = pj < theta[0] ? 1 : pj < theta[1] ? 2 : 3;
    }
  }
}

// One 16-lane group per row: count and double sum of the ratings whose set is 1, 2 or 3
// (every rating when types == nullptr), fixed summation tree per row.
template <typename T>
__global__ __launch_bounds__(256) void rating_stats_kernel(const int64_t *rowPtr, int64_t rows, const T *vals, const int8_t *types,
                                                          int32_t *cnt, double *sum, int64_t longRow) {
  const int sub = threadIdx.x & 15;
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  if (r >= rows) return;
  const int64_t b = rowPtr[r], e = rowPtr[r + 1];
  if (e - b > longRow) return;  // rating_stats_long_kernel's
  double s = 0.0;
  int32_t c = 0;
  for (int64_t q = b + sub; q < e; q += 16) {
    const bool in = types == nullptr || (types[q] >= 1 && types[q] <= 3);
    if (in) {
      s += (double)vals[q];
      ++c;
    }
  }
  for (int m = 8; m >= 1; m >>= 1) {
    s += __shfl_xor(s, m, 16);
    c += __shfl_xor(c, m, 16);
  }
  if (sub == 0) {
    cnt[r] = c;
    sum[r] = s;
  }
}

// The same for long rows (item rows of a popular title have 10^5 ratings), cut into segments of kStatsSegment ratings
// (round 3: one workgroup per ROW left the longest row's 2400 dependent iterations as the tail of the launch): one
// 256-thread workgroup per segment, strided partial sums, fixed tree -> a partial per segment; rating_stats_combine_kernel
// adds a row's partials in segment order.  The summation tree depends on the row's length only.
constexpr int64_t kStatsSegment = 8192;
template <typename T>
__global__ __launch_bounds__(256) void rating_stats_long_kernel(const int64_t *rowPtr, const int32_t *segRow, const int64_t *segBeg, const T *vals,
                                                               const int8_t *types, int32_t *partCnt, double *partSum) {
  __shared__ double ss[256];
  __shared__ int32_t sc[256];
  const int64_t r = segRow[blockIdx.x];
  const int64_t b = segBeg[blockIdx.x], rowEnd = rowPtr[r + 1];
  const int64_t e = b + kStatsSegment < rowEnd ? b + kStatsSegment : rowEnd;
  double s = 0.0;
  int32_t c = 0;
  for (int64_t q = b + threadIdx.x; q < e; q += 256) {
    const bool in = types == nullptr || (types[q] >= 1 && types[q] <= 3);
    if (in) {
      s += (double)vals[q];
      ++c;
    }
  }
  ss[threadIdx.x] = s;
  sc[threadIdx.x] = c;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) {
      ss[threadIdx.x] += ss[threadIdx.x + m];
      sc[threadIdx.x] += sc[threadIdx.x + m];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partCnt[blockIdx.x] = sc[0];
    partSum[blockIdx.x] = ss[0];
  }
}
// long row i of the list: its segments' partials [first[i], first[i + 1]) in order
__global__ void rating_stats_combine_kernel(const int32_t *rowList, const int64_t *first, int64_t nLong, const int32_t *partCnt,
                                            const double *partSum, int32_t *cnt, double *sum) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nLong) return;
  double s = 0.0;
  int32_t c = 0;
  for (int64_t u = first[i]; u < first[i + 1]; ++u) {
    s += partSum[u];
    c += partCnt[u];
  }
  cnt[rowList[i]] = c;
  sum[rowList[i]] = s;
}

// ---- N2: CSR construction (sort by a (row << colBits | col) key with rocPRIM's radix sort; these
// kernels build the keys and unpack the result).  colBits = the bits the column ids need (round 3: the key was
// row << 32 | col, whose 18 always-zero bits between the two ids at MAL scale each radix pass still had to read) ----
__global__ void make_keys_kernel(const int32_t *rowIdx, const int32_t *colIdx, int64_t n, int colBits, uint64_t *keys, uint32_t *pos) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  keys[q] = ((uint64_t)(uint32_t)rowIdx[q] << colBits) | (uint32_t)colIdx[q];
  pos[q] = (uint32_t)q;
}
template <typename T>
__global__ void unpack_sorted_kernel(const uint64_t *keys, const uint32_t *pos, const T *vals, int64_t n, int colBits, int32_t *indx, T *outVals) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  indx[q] = (int32_t)(keys[q] & (((uint64_t)1 << colBits) - 1));
  outVals[q] = vals[pos[q]];
}
// rowPtr[r] = first sorted position whose row (high key half) is >= r, r = 0..rows
__global__ void row_ptr_kernel(const uint64_t *keys, int64_t n, int64_t rows, int colBits, int64_t *rowPtr) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > rows) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)(keys[mid] >> colBits) < r) lo = mid + 1;
    else hi = mid;
  }
  rowPtr[r] = lo;
}

// Transposing a CSR needs no (col, row) key: the entries are already in (row, col-position) order, so a STABLE sort by
// column id alone (a 32-bit key of bits_for(cols) bits: two radix passes at MAL scale instead of five) leaves every new row
// in ascending order of the old row ids (round 3).  The old row of every entry is found before the sort.
// position q -> itself, and the row it belongs to (first r with rowPtr[r + 1] > q); consecutive threads search
// neighbouring ranges, so rowPtr stays in cache (the same search from the SORTED positions, scattered over the array, cost
// three times as much)
__global__ void iota_rows_kernel(const int64_t *rowPtr, int64_t rows, int64_t n, uint32_t *pos, int32_t *rowOf) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  int64_t lo = 0, hi = rows;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowPtr[mid + 1] <= q) lo = mid + 1;
    else hi = mid;
  }
  pos[q] = (uint32_t)q;
  rowOf[q] = (int32_t)lo;
}
template <typename T>
__global__ void unpack_transposed_kernel(const uint32_t *pos, const int32_t *rowOf, const T *vals, int64_t n, int32_t *indx, T *outVals) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int64_t p = pos[q];
  indx[q] = rowOf[p];
  outVals[q] = vals[p];
}
// rowPtr[r] = first sorted position whose key is >= r, r = 0..rows
__global__ void row_ptr32_kernel(const uint32_t *keys, int64_t n, int64_t rows, int64_t *rowPtr) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > rows) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)keys[mid] < r) lo = mid + 1;
    else hi = mid;
  }
  rowPtr[r] = lo;
}

// ---- N3: top-N recommend (YcnrController.recommendItemsForUser, lib/YcnrController.js:227-284) ----
// predict = u . I[item] + globalAvgShift for every item that is not in the user's skip list;
// items below minRecommendRating score -inf.  One 256-thread workgroup per user, a 16-lane group
// per item row (float4 / double2 loads, fixed summation tree), 16 items per pass.
template <typename T, int UB>
__global__ __launch_bounds__(256) void recommend_scores_kernel(const T *userRows, int64_t nUsers, const T *items, int64_t totalItems, int k,
                                                              double shift, double minRating, double *scores) {
  // UB users per workgroup (round 3: one user per workgroup re-read the item matrix from L2 once per user): an item row is
  // loaded once and multiplied into UB accumulators; every (user, item) sum keeps its order, so the scores do not change.
  extern __shared__ __attribute__((aligned(16))) unsigned char smemRec[];
  T *u = reinterpret_cast<T *>(smemRec);  // [UB][k]
  const int64_t user0 = (int64_t)blockIdx.x * UB;
  const int nu = (int)(nUsers - user0 < UB ? nUsers - user0 : UB);
  for (int f = threadIdx.x; f < UB * k; f += 256) u[f] = f / k < nu ? userRows[user0 * k + f] : T(0);
  __syncthreads();
  const int sub = threadIdx.x & 15, slot = threadIdx.x >> 4;
  for (int64_t it0 = 0; it0 < totalItems; it0 += 16) {
    const int64_t it = it0 + slot;
    double s[UB];
#pragma unroll
    for (int q = 0; q < UB; ++q) s[q] = 0.0;
    if (it < totalItems) {
      const T *row = items + it * k;
      T acc[UB];
#pragma unroll
      for (int q = 0; q < UB; ++q) acc[q] = T(0);
      for (int f = sub; f < k; f += 16) {
        const T rv = row[f];
#pragma unroll
        for (int q = 0; q < UB; ++q) acc[q] = fma(u[q * k + f], rv, acc[q]);  // uF.dot(iF) in the factors' precision
      }
#pragma unroll
      for (int q = 0; q < UB; ++q) s[q] = (double)acc[q];
    }
#pragma unroll
    for (int q = 0; q < UB; ++q)
      for (int m = 8; m >= 1; m >>= 1) s[q] += __shfl_xor(s[q], m, 16);
    // lane q of the group finishes user q (the skip lists are applied by recommend_skip_kernel afterwards: a binary search
    // per (user, item) here was six dependent global loads per item pass -- most of this kernel's time)
    if (it < totalItems && sub < nu) {
      double mine = s[0];
#pragma unroll
      for (int q = 1; q < UB; ++q) mine = sub == q ? s[q] : mine;
      const double predict = mine + shift;
      scores[(user0 + sub) * totalItems + it] = predict >= minRating ? predict : -INFINITY;
    }
  }
}
// The same scores on the matrix cores (round 3): predict = U V^T is a GEMM -- 16 users x 16 items per MFMA tile, the
// users' factors of the workgroup in LDS (zero-padded to a multiple of 16), a wave per item tile; lane (g, j) feeds
// V[item j][16 m + 4 g + t] (a 16-byte load) and U[user j][16 m + 4 g + t] to MFMA (m, t) -- operands of
// v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 carry contraction index g, so any assignment of factors to (MFMA, g) that
// is the same on both sides is a dot product.  The dot stays in the factors' precision (float32 / float64 accumulators),
// shift and threshold in double as before.  Needs als_kernels.hip.h's MfmaTraits in front of this header.
template <typename T>
__global__ __launch_bounds__(256) void recommend_scores_mfma_kernel(const T *userRows, int64_t nUsers, const T *items, int64_t totalItems, int k,
                                                                   double shift, double minRating, double *scores) {
  using Tr = MfmaTraits<T>;
  using acc_t = typename Tr::acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smemRec[];
  T *u = reinterpret_cast<T *>(smemRec);  // [16][kp]
  const int kp = (k + 15) & ~15;
  const int64_t user0 = (int64_t)blockIdx.x * 16;
  const int nu = (int)(nUsers - user0 < 16 ? nUsers - user0 : 16);
  for (int f = threadIdx.x; f < 16 * kp; f += 256) {
    const int i = f / kp, c = f - i * kp;
    u[f] = (i < nu && c < k) ? userRows[(user0 + i) * k + c] : T(0);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  const int64_t ntiles = (totalItems + 15) >> 4;
  const T *urow = u + j * kp + 4 * g;
  for (int64_t tile = (int64_t)blockIdx.y * 4 + wave; tile < ntiles; tile += 4 * (int64_t)gridDim.y) {
    const int64_t item = tile * 16 + j;
    const bool valid = item < totalItems;
    const T *row = items + (valid ? item : 0) * (int64_t)k + 4 * g;
    acc_t acc = {T(0), T(0), T(0), T(0)};
    for (int c0 = 0; c0 < kp; c0 += 16) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = c0 + 4 * g + t;
        const T b = (valid && c < k) ? row[c0 + t] : T(0);
        acc = Tr::mma(urow[c0 + t], b, acc);
      }
    }
    if (valid) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int64_t user = user0 + Tr::cd_row(lane, t);
        if (user < nUsers) {
          const double predict = (double)acc[t] + shift;
          scores[user * totalItems + item] = predict >= minRating ? predict : -INFINITY;
        }
      }
    }
  }
}

// the items a user must not be offered (already rated): -inf into their scores; one workgroup per user
__global__ __launch_bounds__(256) void recommend_skip_kernel(const int64_t *skipPtr, const int32_t *skipIds, int64_t totalItems, double *scores) {
  const int64_t user = blockIdx.x;
  for (int64_t q = skipPtr[user] + threadIdx.x; q < skipPtr[user + 1]; q += 256) {
    const int64_t id = skipIds[q];
    if (id >= 0 && id < totalItems) scores[user * totalItems + id] = -INFINITY;  // (ids outside the catalogue match nothing, as before)
  }
}

// picks the `take` best scores of a user, best first, ties by the lower item id; -1 pads
__global__ __launch_bounds__(256) void recommend_select_kernel(double *scores, int64_t totalItems, int take, int stride, int32_t *outIds,
                                                              double *outPredict, int32_t *outCount) {
  // Every thread keeps the best of its own slice (items tid, tid + 256, ...); a round is a workgroup arg-max over those and a
  // re-scan of the winner's slice by its owner only (round 3: every thread re-read its slice in every round -- the 208 MB
  // of scores of 2048 users crossed the memory system `limit` times).
  __shared__ double bestV[256];
  __shared__ int64_t bestI[256];
  __shared__ int64_t winner;
  const int64_t user = blockIdx.x;
  double *sc = scores + user * totalItems;
  auto scan = [&](double &v, int64_t &idx) {
    v = -INFINITY;
    idx = -1;
    for (int64_t it = threadIdx.x; it < totalItems; it += 256) {
      const double x = sc[it];
      if (x > v) {  // strict: the lowest id among equal scores of this thread stays
        v = x;
        idx = it;
      }
    }
  };
  double myV;
  int64_t myI;
  scan(myV, myI);
  int found = 0;
  for (int round = 0; round < stride; ++round) {
    bestV[threadIdx.x] = round < take ? myV : -INFINITY;
    bestI[threadIdx.x] = round < take ? myI : -1;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
      if ((int)threadIdx.x < m) {
        const double ov = bestV[threadIdx.x + m];
        const int64_t oi = bestI[threadIdx.x + m];
        const bool better = oi >= 0 && (bestI[threadIdx.x] < 0 || ov > bestV[threadIdx.x] || (ov == bestV[threadIdx.x] && oi < bestI[threadIdx.x]));
        if (better) {
          bestV[threadIdx.x] = ov;
          bestI[threadIdx.x] = oi;
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const bool got = bestI[0] >= 0 && bestV[0] > -INFINITY;
      outIds[user * stride + round] = got ? (int32_t)bestI[0] : -1;
      outPredict[user * stride + round] = got ? bestV[0] : 0.0;
      winner = got ? bestI[0] : -1;
      if (got) ++found;
    }
    __syncthreads();
    const int64_t w = winner;
    if (w >= 0 && (int)(w & 255) == (int)threadIdx.x) {  // the owner of the winner takes it out and looks for its next best
      sc[w] = -INFINITY;
      scan(myV, myI);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) outCount[user] = found;
}

}  // namespace ycnr
