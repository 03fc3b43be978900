// als_gen_kernels.hip.h -- the half-step for ANY factorsCount in either precision: float32 beyond 256 factors,
// float64 beyond 128 (the reference accepts every factorsCount, lib/emf/EmfBase.js:112, config/config-base.js:31;
// EmfWorker.js:200-246 allocates its k x k matrices per portion whatever k is).
//
// The register / LDS kernels stop where a row's normal matrix no longer fits a wave's registers (k <= 128) or a
// CU's LDS (float32 k <= 256: 136 KB of tiles; a 256 x 256 double image would be 272 KB).  Here the matrix lives in
// GLOBAL memory -- one image per row or chunk of a row in a slab arena, small enough to stay in L2 / the Infinity
// Cache while it is worked on (k = 512, float32: 541 KB) -- and the same right-looking block Cholesky runs on it
// with the same 16 x 16 MFMA tiles (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64: exact IEEE fma chains):
//
//   als_gen_gram_kernel   one 256-thread workgroup per unit (a row, or a chunk of a long row): upper tiles of
//                         Y^T Y and b = Y^T r into the unit's slab.  A wave takes a square of 4 x 4 tiles at a time
//                         (round 4; strips of four tiles before): operands straight from global memory in MFMA layout
//                         (lane (g, c): factor 16 cb + c of rating n0 + g -- one element, any k, no alignment demands).
//   als_gen_solve_kernel  one workgroup per row: slabs summed in slab order into the first (fixed order: results do
//                         not depend on launch geometry), + lambda n I, then per block step
//                            wave 0: diagonal tile -> L (16 pivots) and W = L^-1 through a small LDS image
//                            panel   U[J][bj] = W T[J][bj]               tiles dealt over the waves
//                            update  T[bi][bj] -= U[J][bi]^T U[J][bj]     one block row per wave at a time
//                         with the right-hand side as vectors in LDS, workgroup barriers between the phases
//                         (global-memory tiles written by one wave are read by another only across a barrier:
//                         workgroup scope, the waves of a workgroup share their CU's vector cache).
//
// Rows are processed in batches whose slabs fit the arena.  Rows of at most 176 ratings never come here in
// float32: als_dual_solve_kernel's n x n form does not depend on k.  This path is built for coverage and
// correctness (parity tests at k = 129 ... 512 in both precisions); its speed is that of the float32 / float64
// MFMA pipe with every operand re-read from cache, far from the bf16 kernels of the sizes the benchmarks use.
#pragma once
#include "als_kernels.hip.h"

namespace ycnr {

constexpr int kGenWaves = 4;
constexpr int kGenThreads = kGenWaves * 64;

__host__ __device__ constexpr int64_t gen_slab_elems(int nb) { return (int64_t)tile_count(nb) * 256 + (int64_t)nb * 16; }

template <typename T>
struct GenArgs {
  StepArgs<T> a;
  int32_t nb;         // 16-column blocks of the (padded) matrix
  int32_t slabBase;   // slab number of the batch's first slab (slabs are numbered per upload, the arena holds a batch)
  int32_t firstUnit;  // gram kernel: first unit of the batch
  int32_t firstSplit; // solve kernel: first split row of the batch
};

template <typename T>
__device__ __forceinline__ T gen_shfl_xor(T v, int m) { return __shfl_xor(v, m, 64); }

// Element of tile-register t of this lane in a tile stored in register order: [t][lane]
template <typename T>
__device__ __forceinline__ typename MfmaTraits<T>::acc_t gen_ld_tile(const T *tile, int lane) {
  typename MfmaTraits<T>::acc_t v;
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = tile[t * 64 + lane];
  return v;
}
template <typename T>
__device__ __forceinline__ void gen_st_tile(T *tile, int lane, const typename MfmaTraits<T>::acc_t &v) {
#pragma unroll
  for (int t = 0; t < 4; ++t) tile[t * 64 + lane] = v[t];
}

// Gramian of a unit (round 4, second form).  The first form let every wave gather its operands itself, one element per lane and
// MFMA operand, a square of 4 x 4 tiles at a time: each step waited for an index and then for the values behind it (two trips
// to the cache per 16 MFMAs, nothing requested ahead), and a k = 512 unit was walked 36 times.  Here the workgroup stages
// PANELS of R ratings x k factors in LDS (whole 16-byte loads where every row of the fixed matrix is 16-byte aligned; two LDS
// buffers, one barrier per panel; while panel p is multiplied the values of panels p + 1 and p + 2 are in registers or in
// flight and the ids of panel p + 3 requested: one panel ahead left a CU with 32 KB in flight -- 250 against 241 ms per
// k = 512 iteration) and every wave keeps ONE rectangle of SR x CW tiles in registers per
// pass over the unit's ratings, its operands read from the panel in MFMA layout (lane (g, c): factor 16 cb + c of rating
// 4 s + g; the row pitch P puts the four lane groups on different banks).  float32: 4 x 4 tiles, five passes of eight
// waves at k = 512; float64: 2 x 4, three passes of seven at k = 256.
// 200 K x 20 K, 20 M ratings, per iteration: k = 512 float32 902 (strips) -> 606 (squares from cache) -> see DESIGN.md.
constexpr int kGenSq = 4;
constexpr int kGenGramMaxWaves = 8;
#ifndef YCNR_GEN_GRAM_WAVES_PER_SIMD
#define YCNR_GEN_GRAM_WAVES_PER_SIMD 2  // one workgroup of eight waves per CU
#endif

template <typename T, int V>
struct alignas(sizeof(T) * V) GenVec {
  T e[V];
};

// rectangles of sr x cw tiles over a matrix of nb block columns: block rows sr at a time, from the diagonal to the right in steps of cw
__host__ __device__ constexpr int gen_items(int nb, int sr, int cw) {
  int n = 0;
  for (int x = 0; x < nb; x += sr) n += (nb - x + cw - 1) / cw;
  return n;
}
// block rows of a wave's rectangle: float64 tiles take twice the registers, and 2 x 4 tiles leave room for two waves per SIMD
// of an eight-wave workgroup (k = 256: 20 rectangles, three passes of seven waves; 4 x 4 ran five waves on four SIMDs)
template <typename T>
constexpr int gen_rect_rows() { return sizeof(T) == 4 ? 4 : 2; }
#ifndef YCNR_GEN_SQW_F32
#define YCNR_GEN_SQW_F32 1  // (2 -- 4 x 8 tiles, three passes of seven waves at k = 512 -- measured slower: 269 against 241 ms per iteration)
#endif
template <typename T>
constexpr int gen_sqw() { return sizeof(T) == 4 ? YCNR_GEN_SQW_F32 : 1; }
// row pitch of a panel in elements: a multiple of 16, P * sizeof(T) = 64 bytes modulo 256 (float32) / 128 modulo 256 (float64)
__host__ __device__ constexpr int gen_panel_pitch(int nb, size_t ts) {
  const int mod = (int)(256 / ts);
  return nb * 16 + ((16 - (nb * 16) % mod) + mod) % mod;
}
template <typename T, int V>
constexpr int gen_loader_slots() { return V == 1 ? 8 : 4; }
__host__ __device__ constexpr size_t gen_gram_lds_bytes(int R, int P, size_t ts) { return (size_t)2 * R * ((size_t)P + 1) * ts; }

typedef __bf16 gen_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gen_u32x4 __attribute__((ext_vector_type(4)));
// eight float32 values -> their three bf16 planes (exact: x = h + m + l, each the top 16 bits of what is left), two values per dword
__device__ __forceinline__ void gen_split8(const float (&x)[8], gen_u32x4 &o1, gen_u32x4 &o2, gen_u32x4 &o3) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const float x0 = x[2 * jj], x1 = x[2 * jj + 1];
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    h[jj] = __builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float s0 = x0 - __builtin_bit_cast(float, u0 & 0xFFFF0000u);
    const float s1 = x1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
    const unsigned v0 = __builtin_bit_cast(unsigned, s0), v1 = __builtin_bit_cast(unsigned, s1);
    m[jj] = __builtin_amdgcn_perm(v1, v0, 0x07060302);
    const float t0 = s0 - __builtin_bit_cast(float, v0 & 0xFFFF0000u);
    const float t1 = s1 - __builtin_bit_cast(float, v1 & 0xFFFF0000u);
    l[jj] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, t1), __builtin_bit_cast(unsigned, t0), 0x07060302);
  }
  o1 = gen_u32x4{h[0], h[1], h[2], h[3]};
  o2 = gen_u32x4{m[0], m[1], m[2], m[3]};
  o3 = gen_u32x4{l[0], l[1], l[2], l[3]};
}

// X6 (float32 only, R % 32 == 0): the products on the bf16 pipe -- every wave splits the operands it reads from the panel into three
// bf16 planes and multiplies six of the nine plane pairs (v_mfma_f32_16x16x32_bf16: 32 ratings per instruction; the dropped
// pairs are below 2^-24 of the product), the arithmetic of the bf16x6 kernels of k <= 256.
template <typename T, int V, bool X6 = false>
__global__ __launch_bounds__(64 * kGenGramMaxWaves, YCNR_GEN_GRAM_WAVES_PER_SIMD) void als_gen_gram_kernel(GenArgs<T> ga, int R, int P) {
  using Tr = MfmaTraits<T>;
  using acc_t = typename Tr::acc_t;
  using Vec = GenVec<T, V>;
  constexpr int MAXI = X6 ? 8 : gen_loader_slots<T, V>();  // (X6: panels of 32 ratings)
  constexpr int SR = gen_rect_rows<T>(), CW = gen_sqw<T>() * kGenSq;  // a wave's rectangle: SR x CW tiles
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const StepArgs<T> &a = ga.a;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int nthr = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), W = nthr >> 6;
  const int NB = ga.nb, k = a.k;
  const Unit u = a.units[ga.firstUnit + blockIdx.x];
  const int64_t n = u.end - u.beg;
  T *slab = a.slabs + (int64_t)(u.slab - ga.slabBase) * gen_slab_elems(NB);
  T *bout = slab + (int64_t)tile_count(NB) * 256;
  T *buf0 = reinterpret_cast<T *>(smem), *buf1 = buf0 + (int64_t)R * P, *rb0 = buf1 + (int64_t)R * P, *rb1 = rb0 + R;
  // (columns k .. P - 1 of both buffers stay zero: the loader never writes them)
  for (int i = tid; i < 2 * R * (P + 1); i += nthr) buf0[i] = T(0);
  // loader slots of this thread: (rating of the panel) << 16 | (vector of the row); the same in every panel.  Every load is
  // unconditional, from a clamped (valid) address -- a load under a lane condition is followed by the select that merges its
  // result, and the wait for it, right where it is issued: four trips to memory per panel one after the other -- and what
  // a slot without a rating loaded is replaced by zero when it is written to LDS.
  const int kv = k / V, total = R * kv;
  int rc[MAXI];
  unsigned slotOk = 0;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int v = tid + i * nthr;
    rc[i] = v < total ? (((v / kv) << 16) | (v % kv)) : 0;
    slotOk |= v < total ? 1u << i : 0u;
  }
  auto ld_idx = [&](int64_t p, int i) -> int32_t {
    const int64_t q = p * R + (rc[i] >> 16);
    return a.indx[u.beg + (q < n ? q : n - 1)];
  };
  auto live_mask = [&](int64_t p) -> unsigned {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) m |= (p * R + (rc[i] >> 16) < n) ? 1u << i : 0u;
    return m & slotOk;
  };
  auto ld_val = [&](int32_t id, int i) -> Vec {
    return *reinterpret_cast<const Vec *>(a.fixed + (int64_t)id * k + (int64_t)(rc[i] & 0xffff) * V);
  };
  auto st_val = [&](T *B, Vec v, int i, unsigned live) {
    if (!((live >> i) & 1u)) {
#pragma unroll
      for (int e = 0; e < V; ++e) v.e[e] = T(0);
    }
    if ((slotOk >> i) & 1u) *reinterpret_cast<Vec *>(B + (int64_t)(rc[i] >> 16) * P + (rc[i] & 0xffff) * V) = v;
  };
  const int rslot = tid < R ? tid : R - 1;
  auto ld_r = [&](int64_t p) -> T {
    const int64_t q = p * R + rslot;
    return a.vals[u.beg + (q < n ? q : n - 1)];
  };
  auto st_r = [&](T *rbuf, T r, int64_t p) {
    if (tid < R) rbuf[tid] = (p * R + tid < n) ? r : T(0);
  };
  const int64_t np = (n + R - 1) / R;
  const int nItems = gen_items(NB, SR, CW);
  __syncthreads();
  for (int item0 = 0; item0 < nItems; item0 += W) {
    // this wave's rectangle of the pass (wave-uniform)
    const int item = item0 + wave;
    const bool active = item < nItems;
    int bi0 = 0, bj0 = 0;
    {
      int it = 0;
      for (int x = 0; x < NB; x += SR)
        for (int y = x; y < NB; y += CW, ++it)
          if (it == item) {
            bi0 = x;
            bj0 = y;
          }
    }
    const bool diag = bj0 == bi0;
    int ca[SR], cb[CW];  // operand columns, clamped to the matrix: tiles beyond its edge are computed and not stored
#pragma unroll
    for (int i = 0; i < SR; ++i) ca[i] = 16 * (bi0 + i < NB ? bi0 + i : NB - 1);
#pragma unroll
    for (int j = 0; j < CW; ++j) cb[j] = 16 * (bj0 + j < NB ? bj0 + j : NB - 1);
    acc_t acc[SR][CW];
#pragma unroll
    for (int i = 0; i < SR; ++i)
#pragma unroll
      for (int j = 0; j < CW; ++j) acc[i][j] = acc_t{T(0), T(0), T(0), T(0)};
    T bacc[SR];
#pragma unroll
    for (int i = 0; i < SR; ++i) bacc[i] = T(0);
    // the products of one panel (B: its values, rb: its ratings)
    auto multiply = [&](const T *B, const T *rb) {
      // Straight-line code, the same MFMAs per step for every rectangle: tiles beyond the matrix edge are computed from clamped
      // (valid) columns and the lower tiles of a rectangle on the diagonal from their transposes' operands -- none of them
      // is stored.  With a condition per tile every MFMA sat in a basic block of its own behind a scalar branch and a
      // wait; with one loop for whole rectangles and one for cut ones the accumulators were allocated twice.
      if (active) {
        if constexpr (X6 && sizeof(T) == 4) {
          // lane (g, c) of an operand: factor column 16 cb + c, ratings 8 g ... 8 g + 7 of the 32-rating step
          for (int s32 = 0; s32 < R; s32 += 32) {
            const T *rowp = B + (int64_t)(s32 + 8 * g) * P + c;
            gen_u32x4 ah[SR], am[SR], al[SR];
#pragma unroll
            for (int i = 0; i < SR; ++i) {
              float v[8];
#pragma unroll
              for (int r8 = 0; r8 < 8; ++r8) v[r8] = rowp[(int64_t)r8 * P + ca[i]];
              float bs = bacc[i];  // b rides with the rectangles on the diagonal (the others do not store it)
#pragma unroll
              for (int r8 = 0; r8 < 8; ++r8) bs = fmaf(v[r8], rb[s32 + 8 * g + r8], bs);
              bacc[i] = bs;
              gen_split8(v, ah[i], am[i], al[i]);
            }
#pragma unroll
            for (int j = 0; j < CW; ++j) {
              float v[8];
#pragma unroll
              for (int r8 = 0; r8 < 8; ++r8) v[r8] = rowp[(int64_t)r8 * P + cb[j]];
              gen_u32x4 bh, bm, bl;
              gen_split8(v, bh, bm, bl);
              // smallest terms first: m m, h l, l h, h m, m h, h h
#pragma unroll
              for (int term = 0; term < 6; ++term) {
#pragma unroll
                for (int i = 0; i < SR; ++i) {
                  const gen_u32x4 &pa = term == 0 ? am[i] : (term == 1 || term == 3 || term == 5) ? ah[i] : (term == 2 ? al[i] : am[i]);
                  const gen_u32x4 &pb = term == 0 ? bm : term == 1 ? bl : term == 2 ? bh : term == 3 ? bm : bh;
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gen_bf16x8, pa), __builtin_bit_cast(gen_bf16x8, pb), acc[i][j], 0, 0, 0);
                }
              }
            }
          }
        } else if constexpr (sizeof(T) == 4) {
          // (the operands of step s + 1 are read from LDS before the MFMAs of step s are issued; float64 has no registers for it)
          const T *rowp = B + (int64_t)g * P + c;
          T ya[SR], yb[CW], r;
#pragma unroll
          for (int i = 0; i < SR; ++i) ya[i] = rowp[ca[i]];
#pragma unroll
          for (int j = 0; j < CW; ++j) yb[j] = rowp[cb[j]];
          r = rb[g];  // b rides with the rectangles on the diagonal (the others do not store it)
          for (int s4 = 0; s4 < R; s4 += 4) {
            const int sn = s4 + 4 < R ? s4 + 4 : s4;  // (the last step reads its own operands again)
            const T *rown = B + (int64_t)(sn + g) * P + c;
            T yan[SR], ybn[CW];
#pragma unroll
            for (int i = 0; i < SR; ++i) yan[i] = rown[ca[i]];
#pragma unroll
            for (int j = 0; j < CW; ++j) ybn[j] = rown[cb[j]];
            const T rn = rb[sn + g];
#pragma unroll
            for (int i = 0; i < SR; ++i) bacc[i] = fma(ya[i], r, bacc[i]);
#pragma unroll
            for (int i = 0; i < SR; ++i)
#pragma unroll
              for (int j = 0; j < CW; ++j) acc[i][j] = Tr::mma(ya[i], yb[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < SR; ++i) ya[i] = yan[i];
#pragma unroll
            for (int j = 0; j < CW; ++j) yb[j] = ybn[j];
            r = rn;
          }
        } else {
          for (int s4 = 0; s4 < R; s4 += 4) {
            const T *rowp = B + (int64_t)(s4 + g) * P + c;
            T ya[SR], yb[CW];
#pragma unroll
            for (int i = 0; i < SR; ++i) ya[i] = rowp[ca[i]];
#pragma unroll
            for (int j = 0; j < CW; ++j) yb[j] = rowp[cb[j]];
            const T r = rb[s4 + g];
#pragma unroll
            for (int i = 0; i < SR; ++i) bacc[i] = fma(ya[i], r, bacc[i]);
#pragma unroll
            for (int i = 0; i < SR; ++i)
#pragma unroll
              for (int j = 0; j < CW; ++j) acc[i][j] = Tr::mma(ya[i], yb[j], acc[i][j]);
          }
        }
      }
    };
    int32_t idn[MAXI];
    Vec vA[MAXI];
    T rA = T(0);
    // panel 0 into buffer 0
#pragma unroll
    for (int i = 0; i < MAXI; ++i) idn[i] = ld_idx(0, i);
#pragma unroll
    for (int i = 0; i < MAXI; ++i) vA[i] = ld_val(idn[i], i);
    rA = ld_r(0);
#pragma unroll
    for (int i = 0; i < MAXI; ++i) idn[i] = ld_idx(1, i);
    {
      const unsigned live0 = live_mask(0);
#pragma unroll
      for (int i = 0; i < MAXI; ++i) st_val(buf0, vA[i], i, live0);
      st_r(rb0, rA, 0);
    }
    if constexpr (X6) {
      // panels of 32 ratings: one panel of values in flight while the one before it is multiplied (twice the MFMA time per panel
      // of the float32 form, and no registers for a second set)
      __syncthreads();
      for (int64_t p = 0; p < np; ++p) {
        const T *B = (p & 1) ? buf1 : buf0, *rb = (p & 1) ? rb1 : rb0;
        T *Bn = (p & 1) ? buf0 : buf1, *rbn = (p & 1) ? rb0 : rb1;
        const bool more = p + 1 < np;  // workgroup-uniform
        if (more) {
#pragma unroll
          for (int i = 0; i < MAXI; ++i) vA[i] = ld_val(idn[i], i);
          rA = ld_r(p + 1);
#pragma unroll
          for (int i = 0; i < MAXI; ++i) idn[i] = ld_idx(p + 2, i);
        }
        multiply(B, rb);
        if (more) {
          const unsigned liveN = live_mask(p + 1);
#pragma unroll
          for (int i = 0; i < MAXI; ++i) st_val(Bn, vA[i], i, liveN);
          st_r(rbn, rA, p + 1);
        }
        __syncthreads();
      }
    } else {
      // panel 1 into registers, the ids of panel 2 in flight
      Vec vB[MAXI];
      T rB = T(0);
#pragma unroll
      for (int i = 0; i < MAXI; ++i) vB[i] = ld_val(idn[i], i);
      rB = ld_r(1);
#pragma unroll
      for (int i = 0; i < MAXI; ++i) idn[i] = ld_idx(2, i);
      __syncthreads();
      // one panel: request panel p + 2 into (vX, rX), multiply panel p, write panel p + 1 from (vY, rY) into the other buffer
      auto panel = [&](int64_t p, Vec(&vX)[MAXI], T &rX, Vec(&vY)[MAXI], T &rY) {
        const T *B = (p & 1) ? buf1 : buf0, *rb = (p & 1) ? rb1 : rb0;
        T *Bn = (p & 1) ? buf0 : buf1, *rbn = (p & 1) ? rb0 : rb1;
        if (p + 2 < np) {  // workgroup-uniform
#pragma unroll
          for (int i = 0; i < MAXI; ++i) vX[i] = ld_val(idn[i], i);
          rX = ld_r(p + 2);
#pragma unroll
          for (int i = 0; i < MAXI; ++i) idn[i] = ld_idx(p + 3, i);
        }
        multiply(B, rb);
        if (p + 1 < np) {
          const unsigned liveN = live_mask(p + 1);
#pragma unroll
          for (int i = 0; i < MAXI; ++i) st_val(Bn, vY[i], i, liveN);
          st_r(rbn, rY, p + 1);
        }
        __syncthreads();
      };
      for (int64_t p = 0; p < np; p += 2) {
        panel(p, vA, rA, vB, rB);
        if (p + 1 < np) panel(p + 1, vB, rB, vA, rA);
      }
    }
    if (active) {
#pragma unroll
      for (int i = 0; i < SR; ++i) {
#pragma unroll
        for (int j = 0; j < CW; ++j)
          if (bi0 + i < NB && bj0 + j < NB && bj0 + j >= bi0 + i)
            gen_st_tile<T>(slab + (int64_t)tile_index(bi0 + i, bj0 + j, NB) * 256, lane, acc[i][j]);
        if (diag && bi0 + i < NB) {
          T bs = bacc[i];
          bs += gen_shfl_xor<T>(bs, 16);
          bs += gen_shfl_xor<T>(bs, 32);
          if (g == 0) bout[16 * (bi0 + i) + c] = bs;
        }
      }
    }
  }
}

template <typename T>
struct GenSolve {
  using Tr = MfmaTraits<T>;
  using acc_t = typename Tr::acc_t;
  static constexpr int LDW = sizeof(T) == 8 ? 18 : 20;  // elements per row of the 16 x 16 LDS images (16-byte aligned rows)

  static __device__ __forceinline__ T row_sum(T v) {  // over the 16 lanes of a lane group
    v += gen_shfl_xor<T>(v, 8);
    v += gen_shfl_xor<T>(v, 4);
    v += gen_shfl_xor<T>(v, 2);
    v += gen_shfl_xor<T>(v, 1);
    return v;
  }
  static __device__ __forceinline__ T group_sum(T v) {  // over the 4 lane groups
    v += gen_shfl_xor<T>(v, 16);
    v += gen_shfl_xor<T>(v, 32);
    return v;
  }
  static __device__ __forceinline__ T readlane(T v, int l) {
    if constexpr (sizeof(T) == 8) {
      const long long b = __builtin_bit_cast(long long, v);
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
      return __builtin_bit_cast(T, ((long long)hi << 32) | (long long)lo);
    } else {
      return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
    }
  }

  // One wave.  d: the diagonal tile (C/D layout).  On return Wt (LDS, [col][row], stride LDW) holds W = L^-1 with
  // D = L L^T, and the return value is W in C/D layout.  (Padded pivots are rows of the
  // identity).  dmin collects the smallest real pivot.
  static __device__ __forceinline__ acc_t diag_invert(const acc_t &d, T *Dt, T *Wt, int lane, T &dmin) {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) Dt[Tr::cd_row(lane, t) * LDW + c] = d[t];
    // (LDS operations of one wave complete in order: the reads below see the writes above)
    T R[16];
    const bool xlane = (g & 1) != 0;  // groups 1 / 3: the identity, which the same column operations turn into L^-1
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const T v = Dt[c * LDW + m];
      R[m] = xlane ? (c == m ? T(1) : T(0)) : v;
    }
    // (no branch on nreal: a padded pivot is a row of the identity -- dp = 1, every s = 0, the step changes nothing -- and a
    // branch per pivot made hipcc carry R as one 16-wide vector copied at every join: 860 spilled registers in float32)
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      T dp = readlane(R[p], p);
      dmin = dp < dmin ? dp : dmin;
      if (!(dp > T(0))) dp = T(1);
      const T rs = T(1) / sqrt(dp);
      R[p] *= rs;
#pragma unroll
      for (int j = p + 1; j < 16; ++j) {
        const T s = readlane(R[p], j);  // L[j][p]
        R[j] = fma(-R[p], s, R[j]);
      }
    }
    if (g == 1) {
#pragma unroll
      for (int m = 0; m < 16; ++m) Wt[c * LDW + m] = R[m];  // Wt[col c][row m] = W[m][c]
    }
    acc_t W;
#pragma unroll
    for (int t = 0; t < 4; ++t) W[t] = Wt[c * LDW + Tr::cd_row(lane, t)];
    return W;
  }
};

// PANEL_LDS: the panel U[J][.] of the current block step is kept in LDS too (NB KB in float32), so that the trailing update
// reads only the tiles it changes from global memory.
template <typename T, bool PANEL_LDS>
__global__ __launch_bounds__(kGenThreads, 3) void als_gen_solve_kernel(GenArgs<T> ga) {
  using GS = GenSolve<T>;
  using Tr = MfmaTraits<T>;
  using acc_t = typename Tr::acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const StepArgs<T> &a = ga.a;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NB = ga.nb, k = a.k, NT = tile_count(NB);
  const int kDiag = a.kReal > 0 ? a.kReal : k;
  constexpr int LDW = GS::LDW;
  T *Dt = reinterpret_cast<T *>(smem), *Wt = Dt + 16 * LDW;
  T *bvec = Wt + 16 * LDW, *zvec = bvec + NB * 16, *xvec = zvec + NB * 16;
  int *flag = reinterpret_cast<int *>(xvec + NB * 16);
  T *panL = xvec + NB * 16 + 16;  // PANEL_LDS: tile bj of the panel at panL + 256 bj, in register order
  const SplitRow sr = a.split[ga.firstSplit + blockIdx.x];
  const int64_t se = gen_slab_elems(NB);
  T *S = a.slabs + (int64_t)(sr.slab0 - ga.slabBase) * se;
  // ---- slabs summed in slab order into the first; the right-hand side into LDS
  const int64_t ne = (int64_t)NT * 256;
  {
    // (16-byte vectors, four of them requested before the first is stored: the loop used to wait for every element behind
    // the store of the one before it -- a round trip to HBM per element, about 1 ms of a k = 512 row)
    constexpr int VE = 16 / (int)sizeof(T), UN = 4;
    using V4 = GenVec<T, VE>;
    V4 *S4 = reinterpret_cast<V4 *>(S);  // slabs are 16-byte aligned: gen_slab_elems is a multiple of 16 elements
    const int64_t nv = ne / VE, sev = se / VE;
    for (int64_t i0 = tid; i0 < nv; i0 += (int64_t)UN * kGenThreads) {
      V4 v[UN];
#pragma unroll
      for (int x = 0; x < UN; ++x) {
        const int64_t i = i0 + (int64_t)x * kGenThreads;
        v[x] = S4[i < nv ? i : nv - 1];
      }
      for (int sl = 1; sl < sr.nslabs; ++sl) {
        V4 w[UN];
#pragma unroll
        for (int x = 0; x < UN; ++x) {
          const int64_t i = i0 + (int64_t)x * kGenThreads;
          w[x] = S4[(int64_t)sl * sev + (i < nv ? i : nv - 1)];
        }
#pragma unroll
        for (int x = 0; x < UN; ++x)
#pragma unroll
          for (int e = 0; e < VE; ++e) v[x].e[e] += w[x].e[e];
      }
      if (sr.nslabs > 1) {
#pragma unroll
        for (int x = 0; x < UN; ++x) {
          const int64_t i = i0 + (int64_t)x * kGenThreads;
          if (i < nv) S4[i] = v[x];
        }
      }
    }
  }
  for (int i = tid; i < NB * 16; i += kGenThreads) {
    T v = S[ne + i];
    for (int sl = 1; sl < sr.nslabs; ++sl) v += S[(int64_t)sl * se + ne + i];
    bvec[i] = v;
  }
  if (tid == 0) *flag = 0;
  __syncthreads();
  // ---- + lambda n on the real diagonal, 1 on the padded one.  Element (r, r) of a diagonal tile: the lane and
  // register whose C/D row and column are both r
  const T lam = (T)(a.lambda * (double)sr.n);
  for (int bi = wave; bi < NB; bi += kGenWaves) {
    T *tl = S + (int64_t)tile_index(bi, bi, NB) * 256;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (Tr::cd_row(lane, t) == c) tl[t * 64 + lane] += (16 * bi + c < kDiag) ? lam : T(1);
  }
  __syncthreads();
  T dmin = T(3.0e38);
  const bool c0 = c == 0;
  auto ld_rhs = [&](const T *vec, int blk) {  // block `blk` of a vector as a tile with the vector in column 0
    acc_t v;
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = c0 ? vec[blk * 16 + Tr::cd_row(lane, t)] : T(0);
    return v;
  };
  auto st_rhs = [&](T *vec, int blk, const acc_t &v) {
    if (c0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) vec[blk * 16 + Tr::cd_row(lane, t)] = v[t];
    }
  };
  for (int J = 0; J < NB; ++J) {
    T *TJJ = S + (int64_t)tile_index(J, J, NB) * 256;
    if (wave == 0) {
      const acc_t d = gen_ld_tile<T>(TJJ, lane);
      const acc_t W = GS::diag_invert(d, Dt, Wt, lane, dmin);
      gen_st_tile<T>(TJJ, lane, W);
    }
    __syncthreads();
    // ---- panel: U[J][bj] = W T[J][bj], bj = J+1 .. NB-1, and z_J = W b_J (item bj = NB)
    {
      T Aop[4];  // A operand of MFMA q: W[i = c][kk], kk = the C/D row of register q
#pragma unroll
      for (int q = 0; q < 4; ++q) Aop[q] = Wt[Tr::cd_row(lane, q) * LDW + c];
      // (this wave's tiles PB at a time, all requested before the first is multiplied)
      constexpr int PB = 4;
      for (int bj0 = J + 1 + wave; bj0 < NB; bj0 += PB * kGenWaves) {
        acc_t B[PB];
#pragma unroll
        for (int x = 0; x < PB; ++x) {
          const int bj = bj0 + x * kGenWaves;
          B[x] = gen_ld_tile<T>(S + (int64_t)tile_index(J, bj < NB ? bj : NB - 1, NB) * 256, lane);
        }
#pragma unroll
        for (int x = 0; x < PB; ++x) {
          const int bj = bj0 + x * kGenWaves;
          acc_t P = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
          for (int q = 0; q < 4; ++q) P = Tr::mma(Aop[q], B[x][q], P);
          if (bj < NB) {  // wave-uniform
            gen_st_tile<T>(S + (int64_t)tile_index(J, bj, NB) * 256, lane, P);  // (kept for the back substitution)
            if constexpr (PANEL_LDS) gen_st_tile<T>(panL + bj * 256, lane, P);
          }
        }
      }
      if (wave == (NB - J - 1) % kGenWaves) {  // z_J = W b_J
        const acc_t B = ld_rhs(bvec, J);
        acc_t P = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int q = 0; q < 4; ++q) P = Tr::mma(Aop[q], B[q], P);
        st_rhs(zvec, J, P);
      }
    }
    __syncthreads();
    if (J + 1 == NB) break;
    // ---- trailing update: T[bi][bj] -= U[J][bi]^T U[J][bj] (bj >= bi > J), b_bi -= U[J][bi]^T z_J.  The tiles of a block row
    // go to the waves GB at a time, all GB requested before the first is multiplied (round 3 loaded, multiplied and stored
    // one tile at a time, every tile a round trip to the cache behind the one before it: 4.4 ms per k = 512 row).
    {
      constexpr int GB = sizeof(T) == 8 ? 4 : 8;
      const acc_t Z = ld_rhs(zvec, J);
      const T *pan;
      if constexpr (PANEL_LDS) pan = panL;
      else pan = S + ((int64_t)tile_index(J, J, NB) - J) * 256;  // tile (J, bj) at pan + 256 bj
      int chunk = 0;
      for (int bi = J + 1; bi < NB; ++bi) {
        T *trow = S + ((int64_t)tile_index(bi, bi, NB) - bi) * 256;  // tile (bi, bj) at trow + 256 bj
        for (int bj0 = bi; bj0 < NB; bj0 += GB, ++chunk) {
          if ((chunk % kGenWaves) != wave) continue;  // wave-uniform
          acc_t nPi = gen_ld_tile<T>(pan + bi * 256, lane);
#pragma unroll
          for (int q = 0; q < 4; ++q) nPi[q] = -nPi[q];
          acc_t t[GB], Pj[GB];
#pragma unroll
          for (int x = 0; x < GB; ++x) t[x] = gen_ld_tile<T>(trow + (bj0 + x < NB ? bj0 + x : NB - 1) * 256, lane);
#pragma unroll
          for (int x = 0; x < GB; ++x) Pj[x] = gen_ld_tile<T>(pan + (bj0 + x < NB ? bj0 + x : NB - 1) * 256, lane);
#pragma unroll
          for (int x = 0; x < GB; ++x) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[x] = Tr::mma(nPi[q], Pj[x][q], t[x]);
          }
#pragma unroll
          for (int x = 0; x < GB; ++x)
            if (bj0 + x < NB) gen_st_tile<T>(trow + (bj0 + x) * 256, lane, t[x]);
          if (bj0 == bi) {  // the right-hand side of block row bi goes with the row's first tiles
            acc_t tb = ld_rhs(bvec, bi);
#pragma unroll
            for (int q = 0; q < 4; ++q) tb = Tr::mma(nPi[q], Z[q], tb);
            st_rhs(bvec, bi, tb);
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- back substitution, right-looking: x_J = W_J^T z_J; z_bi -= U[bi][J] x_J for bi < J
  for (int J = NB - 1; J >= 0; --J) {
    if (wave == 0) {
      const acc_t W = gen_ld_tile<T>(S + (int64_t)tile_index(J, J, NB) * 256, lane);
      T s = T(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) s = fma(W[t], zvec[J * 16 + Tr::cd_row(lane, t)], s);  // sum_r W[r][c] z[r], this group's rows
      s = GS::group_sum(s);
      if (g == 0) xvec[J * 16 + c] = s;
    }
    __syncthreads();
    if (J == 0) break;
    {
      const T xc = xvec[J * 16 + c];
      for (int bi = wave; bi < J; bi += kGenWaves) {
        const acc_t U = gen_ld_tile<T>(S + (int64_t)tile_index(bi, J, NB) * 256, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const T dsum = GS::row_sum(U[t] * xc);  // (U x)[row of register t]
          if (c0) zvec[bi * 16 + Tr::cd_row(lane, t)] -= dsum;
        }
      }
    }
    __syncthreads();
  }
  T *out = a.solved + (int64_t)sr.row * k;
  T chk = T(0);
  for (int i = tid; i < k; i += kGenThreads) {
    const T x = xvec[i];
    out[i] = x;
    chk = fma(x, T(0), chk);
  }
  if (!(chk == T(0))) atomicOr(flag, 1);                       // NaN / Inf in the input ends up in x
  if (wave == 0 && lane == 0 && !(dmin > T(0))) atomicOr(flag, 1);  // a real pivot was not positive
  __syncthreads();
  if (tid == 0 && *flag) {
    atomicAdd(&a.err->count, 1);
    a.err->firstRow = sr.row;
  }
}

// ---- The solve, LEFT-looking by blocks of PW block rows (round 4).  The right-looking kernel above reads and writes the
// whole trailing matrix once per block step: 11 MB of traffic per k = 512 row, and with several hundred rows in flight (0.5 MB
// each) none of it stays in a cache -- 490 GB per iteration of the 200 K x 20 K benchmark, the kernel ran at the speed of HBM.
// Here a block of PW rows of the factor is computed at a time, its tiles in REGISTERS (wave w of LW owns the block
// columns Jb + w, Jb + w + LW, ...: MAXC tiles per row):
//   1. acc[r][x] = A[Jb + r][bj_x], the slabs of the row summed in slab order while they are loaded, + lambda n on the diagonal
//   2. for every finished row I < Jb:  acc[r][x] -= U[I][Jb + r]^T U[I][bj_x]   (U streamed from global memory, the next
//      row's tiles requested before this one's are multiplied) -- the matrix is read ~ NB / (2 PW) times instead of NB / 3
//      times read AND written, and nothing of it is written twice
//   3. the PW rows of the block one after the other: the owner of the diagonal tile factors it (16 pivots) and publishes
//      W = L^-1; every wave turns its tiles of the row into U[J][.] = W acc, stores them (final) and, the PW - 1 tiles next
//      to the diagonal, publishes them in LDS for the update of the block's remaining rows.  Two barriers per row.
// The right-hand side rides on the last wave as a tile with the vector in column 0.  Tiles that lie left of the diagonal
// or beyond the matrix edge are loaded from clamped addresses, multiplied like the others and never stored.
// PW: block rows per block; LW: waves per workgroup (MAXC = ceil(NB / LW) column tiles per wave and row); DB: the next finished
// row's tiles are requested before this one's are multiplied (twice the operand registers); WPS: waves per SIMD the
// registers are bounded for.  float32: 4 rows, 8 waves, one workgroup per CU; float64 (twice the registers per tile): 2 rows,
// 4 waves, three workgroups per CU.
template <typename T, int MAXC, int PW, int LW, bool DB, int WPS>
__global__ __launch_bounds__(64 * LW, WPS) void als_gen_solve_left_kernel(GenArgs<T> ga) {
  using GS = GenSolve<T>;
  using Tr = MfmaTraits<T>;
  using acc_t = typename Tr::acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const StepArgs<T> &a = ga.a;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NB = ga.nb, k = a.k, NT = tile_count(NB);
  const int kDiag = a.kReal > 0 ? a.kReal : k;
  constexpr int LDW = GS::LDW, WB = LW - 1, NTHR = 64 * LW;
  T *Dt = reinterpret_cast<T *>(smem), *Wt = Dt + 16 * LDW;
  T *bvec = Wt + 16 * LDW, *zvec = bvec + NB * 16, *xvec = zvec + NB * 16;
  int *flag = reinterpret_cast<int *>(xvec + NB * 16);
  T *blk = xvec + NB * 16 + 16;  // blk[r][r2]: tile U[Jb + r][Jb + r2] of the current block, in register order
  const SplitRow sr = a.split[ga.firstSplit + blockIdx.x];
  const int64_t se = gen_slab_elems(NB);
  T *S = a.slabs + (int64_t)(sr.slab0 - ga.slabBase) * se;
  const int64_t ne = (int64_t)NT * 256;
  for (int i = tid; i < NB * 16; i += NTHR) {
    T v = S[ne + i];
    for (int sl = 1; sl < sr.nslabs; ++sl) v += S[(int64_t)sl * se + ne + i];
    bvec[i] = v;
  }
  if (tid == 0) *flag = 0;
  __syncthreads();
  const T lam = (T)(a.lambda * (double)sr.n);
  T dmin = T(3.0e38);
  const bool c0 = c == 0;
  auto ld_rhs = [&](const T *vec, int b) {  // block b of a vector as a tile with the vector in column 0
    acc_t v;
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = c0 ? vec[b * 16 + Tr::cd_row(lane, t)] : T(0);
    return v;
  };
  auto st_rhs = [&](T *vec, int b, const acc_t &v) {
    if (c0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) vec[b * 16 + Tr::cd_row(lane, t)] = v[t];
    }
  };
  auto neg = [](acc_t v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = -v[q];
    return v;
  };
  for (int Jb = 0; Jb < NB; Jb += PW) {
    int bjx[MAXC];  // this wave's block columns, clamped to the matrix
#pragma unroll
    for (int x = 0; x < MAXC; ++x) bjx[x] = Jb + wave + LW * x < NB ? Jb + wave + LW * x : NB - 1;
    // ---- 1. the block's rows of A
    acc_t acc[PW][MAXC], tb[PW];
#pragma unroll
    for (int r = 0; r < PW; ++r) {
      const int bi = Jb + r < NB ? Jb + r : NB - 1;
#pragma unroll
      for (int x = 0; x < MAXC; ++x) {
        const int bj = bjx[x] > bi ? bjx[x] : bi;
        const T *tl = S + (int64_t)tile_index(bi, bj, NB) * 256;
        acc_t v = gen_ld_tile<T>(tl, lane);
        for (int sl = 1; sl < sr.nslabs; ++sl) {
          const acc_t w = gen_ld_tile<T>(tl + (int64_t)sl * se, lane);
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] += w[t];
        }
        if (bj == bi) {  // (wave-uniform) + lambda n on the real diagonal, 1 on the padded one
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (Tr::cd_row(lane, t) == c) v[t] += (16 * bi + c < kDiag) ? lam : T(1);
        }
        acc[r][x] = v;
      }
      tb[r] = ld_rhs(bvec, bi);
    }
    // ---- 2. minus the finished rows
    if (Jb > 0) {
      acc_t Pi[PW], Pj[MAXC], Pin[DB ? PW : 1], Pjn[DB ? MAXC : 1];
      auto fetch = [&](int I, acc_t(&pi)[PW], acc_t(&pj)[MAXC]) {
        const T *rowI = S + ((int64_t)tile_index(I, I, NB) - I) * 256;  // tile (I, bj) at rowI + 256 bj
#pragma unroll
        for (int r = 0; r < PW; ++r) pi[r] = gen_ld_tile<T>(rowI + (Jb + r < NB ? Jb + r : NB - 1) * 256, lane);
#pragma unroll
        for (int x = 0; x < MAXC; ++x) pj[x] = gen_ld_tile<T>(rowI + bjx[x] * 256, lane);
      };
      const int nx = (NB - Jb + LW - 1) / LW;  // column slots in use in this block (the same for every wave: a wave whose last one lies beyond the edge multiplies a clamped tile)
      auto apply = [&](int I, acc_t(&pi)[PW], acc_t(&pj)[MAXC]) {
        acc_t nP[PW];
#pragma unroll
        for (int r = 0; r < PW; ++r) nP[r] = neg(pi[r]);
#pragma unroll
        for (int x = 0; x < MAXC; ++x) {
          if (x < nx) {  // workgroup-uniform
#pragma unroll
            for (int r = 0; r < PW; ++r)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[r][x] = Tr::mma(nP[r][q], pj[x][q], acc[r][x]);
          }
        }
        if (wave == WB) {
          const acc_t Z = ld_rhs(zvec, I);
#pragma unroll
          for (int r = 0; r < PW; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) tb[r] = Tr::mma(nP[r][q], Z[q], tb[r]);
        }
      };
      if constexpr (DB) {
        fetch(0, Pi, Pj);
        for (int I = 0; I < Jb; I += 2) {
          if (I + 1 < Jb) fetch(I + 1, Pin, Pjn);
          apply(I, Pi, Pj);
          if (I + 1 < Jb) {
            if (I + 2 < Jb) fetch(I + 2, Pi, Pj);
            apply(I + 1, Pin, Pjn);
          }
        }
      } else {
        for (int I = 0; I < Jb; ++I) {
          fetch(I, Pi, Pj);
          apply(I, Pi, Pj);
        }
      }
    }
    // ---- 3. the rows of the block
#pragma unroll
    for (int r = 0; r < PW; ++r) {
      const int J = Jb + r;
      if (J < NB) {  // workgroup-uniform
        if (wave == r) {  // (the diagonal tile of row J is this wave's first tile of the row)
          const acc_t W = GS::diag_invert(acc[r][0], Dt, Wt, lane, dmin);
          gen_st_tile<T>(S + (int64_t)tile_index(J, J, NB) * 256, lane, W);
        }
        __syncthreads();
        T Aop[4];  // A operand of MFMA q: W[i = c][kk], kk = the C/D row of register q
#pragma unroll
        for (int q = 0; q < 4; ++q) Aop[q] = Wt[Tr::cd_row(lane, q) * LDW + c];
#pragma unroll
        for (int x = 0; x < MAXC; ++x) {
          const acc_t B = acc[r][x];
          acc_t P = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
          for (int q = 0; q < 4; ++q) P = Tr::mma(Aop[q], B[q], P);
          acc[r][x] = P;
          const int bj = Jb + wave + LW * x;
          if (bj > J && bj < NB) {  // wave-uniform
            gen_st_tile<T>(S + (int64_t)tile_index(J, bj, NB) * 256, lane, P);
            if (x == 0 && wave < PW) gen_st_tile<T>(blk + (r * PW + wave) * 256, lane, P);
          }
        }
        acc_t ZJ = acc_t{T(0), T(0), T(0), T(0)};
        if (wave == WB) {  // z_J = W b_J
#pragma unroll
          for (int q = 0; q < 4; ++q) ZJ = Tr::mma(Aop[q], tb[r][q], ZJ);
          st_rhs(zvec, J, ZJ);
        }
        __syncthreads();
#pragma unroll
        for (int r2 = r + 1; r2 < PW; ++r2) {
          if (Jb + r2 < NB) {
            const acc_t nP = neg(gen_ld_tile<T>(blk + (r * PW + r2) * 256, lane));
#pragma unroll
            for (int x = 0; x < MAXC; ++x)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[r2][x] = Tr::mma(nP[q], acc[r][x][q], acc[r2][x]);
            if (wave == WB) {
#pragma unroll
              for (int q = 0; q < 4; ++q) tb[r2] = Tr::mma(nP[q], ZJ[q], tb[r2]);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- back substitution, right-looking: x_J = W_J^T z_J; z_bi -= U[bi][J] x_J for bi < J
  for (int J = NB - 1; J >= 0; --J) {
    if (wave == 0) {
      const acc_t W = gen_ld_tile<T>(S + (int64_t)tile_index(J, J, NB) * 256, lane);
      T s = T(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) s = fma(W[t], zvec[J * 16 + Tr::cd_row(lane, t)], s);  // sum_r W[r][c] z[r], this group's rows
      s = GS::group_sum(s);
      if (g == 0) xvec[J * 16 + c] = s;
    }
    __syncthreads();
    if (J == 0) break;
    {
      const T xc = xvec[J * 16 + c];
      for (int bi = wave; bi < J; bi += LW) {
        const acc_t U = gen_ld_tile<T>(S + (int64_t)tile_index(bi, J, NB) * 256, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const T dsum = GS::row_sum(U[t] * xc);  // (U x)[row of register t]
          if (c0) zvec[bi * 16 + Tr::cd_row(lane, t)] -= dsum;
        }
      }
    }
    __syncthreads();
  }
  T *out = a.solved + (int64_t)sr.row * k;
  T chk = T(0);
  for (int i = tid; i < k; i += NTHR) {
    const T x = xvec[i];
    out[i] = x;
    chk = fma(x, T(0), chk);
  }
  if (!(chk == T(0))) atomicOr(flag, 1);                       // NaN / Inf in the input ends up in x
  if (lane == 0 && !(dmin > T(0))) atomicOr(flag, 1);          // a real pivot was not positive
  __syncthreads();
  if (tid == 0 && *flag) {
    atomicAdd(&a.err->count, 1);
    a.err->firstRow = sr.row;
  }
}
__host__ __device__ constexpr size_t gen_solve_left_lds_bytes(int nb, size_t ts, int pw) {
  return (2 * 16 * (ts == 8 ? 18 : 20) + 3 * (size_t)nb * 16 + 16 + (size_t)pw * pw * 256) * ts + 64;
}

__host__ __device__ constexpr size_t gen_solve_lds_bytes(int nb, size_t ts, bool panelLds) {
  return (2 * 16 * (ts == 8 ? 18 : 20) + 3 * (size_t)nb * 16 + 16 + (panelLds ? (size_t)nb * 256 : 0)) * ts + 64;
}
constexpr size_t kGenPanelLdsMax = 36 * 1024;  // panels up to this size live in LDS (k = 512 float32, k = 256 float64: four workgroups per CU)

}  // namespace ycnr
