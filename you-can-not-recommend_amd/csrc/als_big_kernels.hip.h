// als_big_kernels.hip.h -- the float32 path for 128 < factorsCount <= 256.
//
// At k = 256 the upper triangle of A is 136 tiles of 16x16 = 544 accumulator registers, more
// than one wave owns, so a 4-wave workgroup shares a unit:
//   als_gram_big_kernel   each wave keeps the tiles of its block rows (rows paired r, NB-1-r
//                         so every wave gets the same count) and runs the same gather + MFMA
//                         loop as the small path over all NB operand blocks; tiles go to a
//                         slab in global memory (every row of this path is "split", a short
//                         row simply has one slab);
//   als_solve_big_kernel  sums a row's slabs into an LDS image of the upper tiles (136 KB at
//                         k = 256, one workgroup per CU), then the same right-looking block
//                         Cholesky as SolveMfmaF32 with the tiles in LDS instead of
//                         registers: wave 0 factors + inverts the diagonal tile, all waves
//                         share the panel and the trailing update (MFMA operands are read
//                         from LDS directly in operand layout, so no lane transposes), three
//                         barriers per block step; wave 0 finishes with the two triangular
//                         solves.
// Rows with fewer than 97 ratings never come here: they take the dual form
// (als_dual_solve_kernel), whose cost does not grow with k^3.
#pragma once
#include "als_kernels.hip.h"

namespace ycnr {

constexpr int kBigWaves = 4;

// block row r of an NB-block matrix is owned by this wave; rows r and NB-1-r are paired so
// the tile counts balance (NB = 16: 34 tiles per wave)
__host__ __device__ constexpr int big_owner(int r, int nb) {
  return ((r < nb - 1 - r) ? r : nb - 1 - r) % kBigWaves;
}
__host__ __device__ constexpr int big_tiles_of(int w, int nb) {
  int n = 0;
  for (int r = 0; r < nb; ++r)
    if (big_owner(r, nb) == w) n += nb - r;
  return n;
}
__host__ __device__ constexpr int big_max_tiles(int nb) {
  int m = 0;
  for (int w = 0; w < kBigWaves; ++w) m = big_tiles_of(w, nb) > m ? big_tiles_of(w, nb) : m;
  return m;
}
// index of tile (bi, bj) among the tiles of its owner, in (bi, bj) order
__host__ __device__ constexpr int big_local_index(int bi, int bj, int nb) {
  const int w = big_owner(bi, nb);
  int n = 0;
  for (int r = 0; r < bi; ++r)
    if (big_owner(r, nb) == w) n += nb - r;
  return n + (bj - bi);
}

template <int NB, int W>
__device__ __forceinline__ void gram_big_body(const StepArgs<float> &a, const Unit &u, int lane) {
  using G = Gram<float, NB>;
  using Tr = MfmaTraits<float>;
  using acc_t = typename Tr::acc_t;
  constexpr int MT = big_max_tiles(NB);
  const int g = lane >> 4, c = lane & 15;
  acc_t acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
  float bacc[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) bacc[cb] = 0.0f;
  const int k = a.k;
  const int64_t beg = u.beg, end = u.end, last = end - 1;
  const int64_t nsteps = (end - beg + 3) >> 2;
  int64_t n = beg + g;
  int64_t nc = n < end ? n : last;
  const float *row0 = n < end ? a.fixed + (int64_t)a.indx[nc] * k : a.zeros;
  float rv = a.vals[nc];
  float r0 = n < end ? rv : 0.0f;
  n += 4;
  nc = n < end ? n : last;
  int32_t id1 = a.indx[nc];
  rv = a.vals[nc];
  float r1 = n < end ? rv : 0.0f;
  bool v1 = n < end;
  float yA[NB], yB[NB];
  G::load_y(yA, row0, a.zeros, k, c);
  for (int64_t i = 0; i < nsteps; ++i) {
    const float *row1 = v1 ? a.fixed + (int64_t)id1 * k : a.zeros;
    G::load_y(yB, row1, a.zeros, k, c);
    const float ra = r0;
    r0 = r1;
    n += 4;
    nc = n < end ? n : last;
    v1 = n < end;
    id1 = a.indx[nc];
    rv = a.vals[nc];
    r1 = v1 ? rv : 0.0f;
    if (W == 0) {
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) bacc[cb] = fmaf(yA[cb], ra, bacc[cb]);
    }
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
      if (big_owner(bi, NB) != W) continue;
#pragma unroll
      for (int bj = bi; bj < NB; ++bj)
        acc[big_local_index(bi, bj, NB)] = Tr::mma(yA[bi], yA[bj], acc[big_local_index(bi, bj, NB)]);
    }
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) yA[cb] = yB[cb];
  }
  // slab layout = the small path's: [tile_index(bi, bj)][reg][lane], then NB b-partials
  float *s = a.slabs + (int64_t)u.slab * slab_elems(NB) + lane;
#pragma unroll
  for (int bi = 0; bi < NB; ++bi) {
    if (big_owner(bi, NB) != W) continue;
#pragma unroll
    for (int bj = bi; bj < NB; ++bj) {
      const acc_t t = acc[big_local_index(bi, bj, NB)];
#pragma unroll
      for (int r = 0; r < 4; ++r) s[(tile_index(bi, bj, NB) * 4 + r) * 64] = t[r];
    }
  }
  if (W == 0) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) s[(tile_count(NB) * 4 + cb) * 64] = bacc[cb];
  }
}

template <int NB>
__global__ __launch_bounds__(256) void als_gram_big_kernel(StepArgs<float> a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Unit u = a.units[a.firstFused + blockIdx.x];
  switch (wave) {
    case 0: gram_big_body<NB, 0>(a, u, lane); break;
    case 1: gram_big_body<NB, 1>(a, u, lane); break;
    case 2: gram_big_body<NB, 2>(a, u, lane); break;
    default: gram_big_body<NB, 3>(a, u, lane); break;
  }
}

// LDS-resident block Cholesky for one row.  LDS: NT tiles of 16x16 floats ([row][col], row
// stride 16), then the two 16 x 20 images of the diagonal step, then b / z / x vectors.
template <int NB>
struct SolveBig {
  using Tr = MfmaTraits<float>;
  using acc_t = typename Tr::acc_t;
  using Sm = SolveMfmaF32<1>;
  static constexpr int NT = tile_count(NB);
  static constexpr int LDW = 20;
  static constexpr size_t lds_floats() { return (size_t)NT * 256 + 2 * 16 * LDW + 3 * NB * 16; }
  static constexpr size_t lds_bytes() { return lds_floats() * sizeof(float); }

  static __device__ __forceinline__ float *tile(float *S, int bi, int bj) { return S + tile_index(bi, bj, NB) * 256; }
  // C/D-layout access: lane (g, c), reg t <-> [4g+t][c]
  static __device__ __forceinline__ acc_t load_cd(const float *T, int g, int c) {
    return acc_t{T[(4 * g + 0) * 16 + c], T[(4 * g + 1) * 16 + c], T[(4 * g + 2) * 16 + c], T[(4 * g + 3) * 16 + c]};
  }
  static __device__ __forceinline__ void store_cd(float *T, const acc_t &v, int g, int c) {
#pragma unroll
    for (int t = 0; t < 4; ++t) T[(4 * g + t) * 16 + c] = v[t];
  }

  static __device__ void run(const StepArgs<float> &a, const SplitRow &sr, float *S) {
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = a.k;
    float *Dt = S + NT * 256, *Wt = Dt + 16 * LDW;
    float *bvec = Wt + 16 * LDW, *zvec = bvec + NB * 16, *xvec = zvec + NB * 16;
    const float lam = (float)(a.lambda * (double)sr.n);
    // ---- 0. slabs -> LDS tiles (tile ti handled by wave ti % 4), b -> bvec (wave 0)
    // SUMW tiles per wave and slab at a time: their 4 SUMW loads are in flight together.  (One tile at
    // a time was one dependent global-memory round trip per tile and slab, 34+ of them per wave with
    // nothing else resident on the CU to hide them: most of this kernel's time.)
    constexpr int SUMW = 8;
    for (int t0 = wave; t0 < NT; t0 += kBigWaves * SUMW) {
      acc_t v[SUMW];
#pragma unroll
      for (int u = 0; u < SUMW; ++u) v[u] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
      for (int sl = 0; sl < sr.nslabs; ++sl) {
        const float *s = a.slabs + (int64_t)(sr.slab0 + sl) * slab_elems(NB) + lane;
#pragma unroll
        for (int u = 0; u < SUMW; ++u) {
          const int ti = t0 + u * kBigWaves;
          if (ti < NT) {  // wave-uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) v[u][r] += s[(ti * 4 + r) * 64];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < SUMW; ++u) {
        const int ti = t0 + u * kBigWaves;
        if (ti < NT) store_cd(S + ti * 256, v[u], g, c);
      }
    }
    if (wave == 0) {
      float v[NB];
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) v[cb] = 0.0f;
      for (int sl = 0; sl < sr.nslabs; ++sl) {
        const float *s = a.slabs + (int64_t)(sr.slab0 + sl) * slab_elems(NB) + NT * 4 * 64 + lane;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) v[cb] += s[cb * 64];
      }
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        const float t = Sm::group_sum(v[cb]);
        if (g == 0) bvec[cb * 16 + c] = t;
      }
    }
    __syncthreads();
    // diagonal: + lam on real indices, 1 on padded ones (tile bi handled by thread group bi)
    for (int i = tid; i < NB * 16; i += 256) tile(S, i >> 4, i >> 4)[(i & 15) * 17] += (i < k) ? lam : 1.0f;
    __syncthreads();
    bool bad = false;
    for (int J = 0; J < NB; ++J) {
      // ---- 1. wave 0: factor + invert the diagonal tile (same routine as SolveMfmaF32)
      if (wave == 0) {
        const acc_t d = load_cd(tile(S, J, J), g, c);
#pragma unroll
        for (int t = 0; t < 4; ++t) Dt[(4 * g + t) * LDW + c] = d[t];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
        float R[16];
        {
          const bool xlane = (g & 1) != 0;
          const float4 *src = reinterpret_cast<const float4 *>(Dt + c * LDW);
#pragma unroll
          for (int m4 = 0; m4 < 4; ++m4) {
            const float4 v = src[m4];
            R[4 * m4 + 0] = xlane ? (c == 4 * m4 + 0 ? 1.0f : 0.0f) : v.x;
            R[4 * m4 + 1] = xlane ? (c == 4 * m4 + 1 ? 1.0f : 0.0f) : v.y;
            R[4 * m4 + 2] = xlane ? (c == 4 * m4 + 2 ? 1.0f : 0.0f) : v.z;
            R[4 * m4 + 3] = xlane ? (c == 4 * m4 + 3 ? 1.0f : 0.0f) : v.w;
          }
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          float d0 = Sm::readlane(R[p], p);
          if (!(d0 > 0.0f)) {
            bad = bad || (J * 16 + p < k);
            d0 = 1.0f;
          }
          float rs = __builtin_amdgcn_rsqf(d0);
          rs = rs * (1.5f - 0.5f * d0 * rs * rs);
          R[p] *= rs;
#pragma unroll
          for (int j = p + 1; j < 16; ++j) {
            const float s = Sm::readlane(R[p], j);
            R[j] = fmaf(-R[p], s, R[j]);
          }
        }
        if (g == 1) {
          float4 *dst = reinterpret_cast<float4 *>(Wt + c * LDW);
#pragma unroll
          for (int m4 = 0; m4 < 4; ++m4) dst[m4] = float4{R[4 * m4], R[4 * m4 + 1], R[4 * m4 + 2], R[4 * m4 + 3]};
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // keep W = L^-1 in the diagonal tile, C/D layout
        const float4 w = *reinterpret_cast<const float4 *>(Wt + c * LDW + 4 * g);
        store_cd(tile(S, J, J), acc_t{w.x, w.y, w.z, w.w}, g, c);
      }
      __syncthreads();
      // ---- 2. panel: U[J][bj] = W * T[J][bj]
      float Aop[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) Aop[q] = Wt[(4 * q + g) * LDW + c];
      for (int bj = J + 1 + wave; bj < NB; bj += kBigWaves) {
        float *T = tile(S, J, bj);
        acc_t P = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 4; ++q) P = Tr::mma(Aop[q], T[(4 * q + g) * 16 + c], P);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // operands are in registers before the tile is overwritten
        store_cd(T, P, g, c);
      }
      __syncthreads();
      // ---- 3. trailing update: T[bi][bj] -= U[J][bi]^T U[J][bj], tiles dealt round-robin
      {
        int e = 0;
        for (int bi = J + 1; bi < NB; ++bi) {
          for (int bj = bi; bj < NB; ++bj, ++e) {
            if ((e & (kBigWaves - 1)) != wave) continue;
            const float *Pi = tile(S, J, bi), *Pj = tile(S, J, bj);
            float *T = tile(S, bi, bj);
            acc_t t = load_cd(T, g, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) t = Tr::mma(-Pi[(4 * q + g) * 16 + c], Pj[(4 * q + g) * 16 + c], t);
            store_cd(T, t, g, c);
          }
        }
      }
      __syncthreads();
    }
    // ---- 4. wave 0: z = U^-T b (left-looking), then x = U^-1 z
    if (wave == 0) {
      for (int J = 0; J < NB; ++J) {
        float s = 0.0f;
        for (int P = 0; P < J; ++P) {
          const acc_t u = load_cd(tile(S, P, J), g, c);
#pragma unroll
          for (int t = 0; t < 4; ++t) s = fmaf(u[t], zvec[P * 16 + 4 * g + t], s);
        }
        const float bj = bvec[J * 16 + c] - Sm::group_sum(s);
        const acc_t W = load_cd(tile(S, J, J), g, c);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float z = Sm::row_sum(W[t] * bj);
          if (c == 0) zvec[J * 16 + 4 * g + t] = z;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
      }
      for (int J = NB - 1; J >= 0; --J) {
        float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int bj = J + 1; bj < NB; ++bj) {
          const acc_t u = load_cd(tile(S, J, bj), g, c);
          const float x = xvec[bj * 16 + c];
#pragma unroll
          for (int t = 0; t < 4; ++t) part[t] = fmaf(u[t], x, part[t]);
        }
        const acc_t W = load_cd(tile(S, J, J), g, c);
        float s = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) s = fmaf(W[t], zvec[J * 16 + 4 * g + t] - Sm::row_sum(part[t]), s);
        const float x = Sm::group_sum(s);
        if (g == 0) xvec[J * 16 + c] = x;
        __builtin_amdgcn_s_waitcnt(0xc07f);
      }
      float *out = a.solved + (int64_t)sr.row * k;
      for (int i = lane; i < k; i += 64) out[i] = xvec[i];
      if (bad && lane == 0) {
        atomicAdd(&a.err->count, 1);
        a.err->firstRow = sr.row;
      }
    }
  }
};

template <int NB>
__global__ __launch_bounds__(256) void als_solve_big_kernel(StepArgs<float> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SplitRow sr = a.split[a.firstDual + blockIdx.x];
  SolveBig<NB>::run(a, sr, reinterpret_cast<float *>(smem));
}

}  // namespace ycnr
