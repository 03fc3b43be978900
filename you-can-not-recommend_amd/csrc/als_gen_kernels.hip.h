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

// Work item of a wave: a SQUARE of kGenSq x kGenSq tiles (block rows bi0.., block columns bj0.., bi0 <= bj0; on the diagonal only
// the upper tiles).  Per 4-rating step it loads kGenSq A values and kGenSq B values for kGenSq^2 MFMAs -- round 3's strips of
// four tiles of ONE block row loaded 5 values per 4 MFMAs and walked the unit's ratings 3.7 x as often (k = 512: 132 strips
// against 36 squares), every walk re-reading the gathered rows from L2: the kernel was bound by that traffic.  200 K x 20 K,
// 20 M ratings: k = 512 float32 902 -> 606 ms per iteration, k = 256 float64 513 -> 460.  (Operands requested one step
// ahead by hand: 702 ms -- the registers cost a wave per SIMD.)
constexpr int kGenSq = 4;

template <typename T>
__global__ __launch_bounds__(kGenThreads) void als_gen_gram_kernel(GenArgs<T> ga) {
  using Tr = MfmaTraits<T>;
  using acc_t = typename Tr::acc_t;
  const StepArgs<T> &a = ga.a;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NB = ga.nb, k = a.k;
  const Unit u = a.units[ga.firstUnit + blockIdx.x];
  const int64_t n = u.end - u.beg;
  T *slab = a.slabs + (int64_t)(u.slab - ga.slabBase) * gen_slab_elems(NB);
  T *bout = slab + (int64_t)tile_count(NB) * 256;
  const int64_t nsteps = (n + 3) >> 2;
  int item = 0;
  for (int bi0 = 0; bi0 < NB; bi0 += kGenSq) {
    for (int bj0 = bi0; bj0 < NB; bj0 += kGenSq, ++item) {
      if ((item % kGenWaves) != wave) continue;
      const bool diag = bj0 == bi0;
      acc_t acc[kGenSq][kGenSq];
#pragma unroll
      for (int i = 0; i < kGenSq; ++i)
#pragma unroll
        for (int j = 0; j < kGenSq; ++j) acc[i][j] = acc_t{T(0), T(0), T(0), T(0)};
      T bacc[kGenSq];
#pragma unroll
      for (int i = 0; i < kGenSq; ++i) bacc[i] = T(0);
      for (int64_t st = 0; st < nsteps; ++st) {
        const int64_t q = (st << 2) + g;
        const bool live = q < n;
        const int64_t qc = u.beg + (live ? q : n - 1);
        const T *row = a.fixed + (int64_t)a.indx[qc] * k;
        const T r = live ? a.vals[qc] : T(0);
        T ya[kGenSq], yb[kGenSq];
#pragma unroll
        for (int i = 0; i < kGenSq; ++i) {
          const int colA = 16 * (bi0 + i) + c, colB = 16 * (bj0 + i) + c;
          ya[i] = (live && colA < k) ? row[colA] : T(0);
          yb[i] = diag ? ya[i] : ((live && colB < k) ? row[colB] : T(0));
        }
        if (diag) {  // b rides with the squares on the diagonal (wave-uniform)
#pragma unroll
          for (int i = 0; i < kGenSq; ++i) bacc[i] = fma(ya[i], r, bacc[i]);
        }
#pragma unroll
        for (int i = 0; i < kGenSq; ++i) {
#pragma unroll
          for (int j = 0; j < kGenSq; ++j) {
            if (bi0 + i < NB && bj0 + j < NB && (!diag || j >= i))  // wave-uniform
              acc[i][j] = Tr::mma(ya[i], yb[j], acc[i][j]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < kGenSq; ++i) {
#pragma unroll
        for (int j = 0; j < kGenSq; ++j)
          if (bi0 + i < NB && bj0 + j < NB && (!diag || j >= i))
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
  // D = L L^T, and the return value is W in C/D layout.  nreal: pivots that are real (the rest are rows of the
  // identity).  dmin collects the smallest real pivot.
  static __device__ __forceinline__ acc_t diag_invert(const acc_t &d, T *Dt, T *Wt, int lane, int nreal, T &dmin) {
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
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      if (p < nreal) {  // wave-uniform
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

template <typename T>
__global__ __launch_bounds__(kGenThreads) void als_gen_solve_kernel(GenArgs<T> ga) {
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
  const SplitRow sr = a.split[ga.firstSplit + blockIdx.x];
  const int64_t se = gen_slab_elems(NB);
  T *S = a.slabs + (int64_t)(sr.slab0 - ga.slabBase) * se;
  // ---- slabs summed in slab order into the first; the right-hand side into LDS
  const int64_t ne = (int64_t)NT * 256;
  for (int64_t i = tid; i < ne; i += kGenThreads) {
    T v = S[i];
    for (int sl = 1; sl < sr.nslabs; ++sl) v += S[(int64_t)sl * se + i];
    S[i] = v;
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
      int nreal = k - 16 * J;
      nreal = nreal > 16 ? 16 : nreal;
      const acc_t W = GS::diag_invert(d, Dt, Wt, lane, nreal, dmin);
      gen_st_tile<T>(TJJ, lane, W);
    }
    __syncthreads();
    // ---- panel: U[J][bj] = W T[J][bj], bj = J+1 .. NB-1, and z_J = W b_J (item bj = NB)
    {
      T Aop[4];  // A operand of MFMA q: W[i = c][kk], kk = the C/D row of register q
#pragma unroll
      for (int q = 0; q < 4; ++q) Aop[q] = Wt[Tr::cd_row(lane, q) * LDW + c];
      for (int bj = J + 1 + wave; bj <= NB; bj += kGenWaves) {
        const bool rhs = bj == NB;
        T *tl = S + (int64_t)tile_index(J, rhs ? J : bj, NB) * 256;
        const acc_t B = rhs ? ld_rhs(bvec, J) : gen_ld_tile<T>(tl, lane);
        acc_t P = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int q = 0; q < 4; ++q) P = Tr::mma(Aop[q], B[q], P);
        if (rhs) st_rhs(zvec, J, P);
        else gen_st_tile<T>(tl, lane, P);
      }
    }
    __syncthreads();
    if (J + 1 == NB) break;
    // ---- trailing update: T[bi][bj] -= U[J][bi]^T U[J][bj] (bj >= bi > J), b_bi -= U[J][bi]^T z_J; a block row per wave
    {
      const acc_t Z = ld_rhs(zvec, J);
      for (int bi = J + 1 + wave; bi < NB; bi += kGenWaves) {
        acc_t nPi = gen_ld_tile<T>(S + (int64_t)tile_index(J, bi, NB) * 256, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) nPi[q] = -nPi[q];
        for (int bj = bi; bj < NB; ++bj) {
          const acc_t Pj = gen_ld_tile<T>(S + (int64_t)tile_index(J, bj, NB) * 256, lane);
          T *tl = S + (int64_t)tile_index(bi, bj, NB) * 256;
          acc_t t = gen_ld_tile<T>(tl, lane);
#pragma unroll
          for (int q = 0; q < 4; ++q) t = Tr::mma(nPi[q], Pj[q], t);
          gen_st_tile<T>(tl, lane, t);
        }
        acc_t t = ld_rhs(bvec, bi);
#pragma unroll
        for (int q = 0; q < 4; ++q) t = Tr::mma(nPi[q], Z[q], t);
        st_rhs(bvec, bi, t);
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

__host__ __device__ constexpr size_t gen_solve_lds_bytes(int nb, size_t ts) {
  return (2 * 16 * (ts == 8 ? 18 : 20) + 3 * (size_t)nb * 16) * ts + 64;
}

}  // namespace ycnr
