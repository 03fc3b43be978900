// als_pair_kernels.hip.h -- the solve of the 240 < k <= 256 path by a FEW waves per row with register-resident tiles (round 3).
//
// Why.  The 8-wave workgroup solve of als_wg_kernels.hip.h takes 62 us of a whole CU per 256 x 256 system: per
// block step a panel phase, a barrier, the 16 sequential pivots of the next diagonal tile on one wave and a second
// barrier, with the whole matrix going through LDS for every tile update -- its MFMA work is 13 us.  The one-wave solves
// of the other paths get four times that throughput out of a CU because four independent rows hide each other's pivot
// chains (the 176 x 176 dual class costs 66 us of ONE SIMD, Gramian included).  A 256 x 256 system does not fit one
// wave -- 136 tiles are 544 registers, and past ~70 resident tiles hipcc spills by the kilobyte (tried: 105 tiles in
// registers + 31 parked in LDS: 2.9 KB of scratch per lane) -- but it fits two waves (68 tiles each, one wave per SIMD)
// or four (34 tiles each in 256 registers, TWO waves per SIMD: the waves of two rows share every SIMD, so a wave
// waiting for its row's pivot chain leaves the SIMD to the other row -- the shipped form).
//
// How.  The Gramian stays with the 8-wave workgroup kernel (bf16 pipe, als_wg_kernels.hip.h), which writes the row's
// image to a slab in global memory instead of solving it (als_wg_gram_rowslab_kernel); rows are processed in batches
// whose slabs fit an arena.  The solve kernel (als_slab_solve2_kernel) runs one workgroup of kPairWaves waves per row,
// two workgroups per CU:
//   * block rows are dealt back and forth over the waves (0 1 2 3 3 2 1 0 ...: 34 tiles each at NB = 16), all in
//     registers, every tile index a compile-time constant (the waves run specialisations of the same code);
//   * block step J: the OWNER of row J factors its diagonal tile (the pivot sequence of SolveMfmaF32), turns its row
//     into the panel U[J][.] = W T[J][.] (kept in its registers for the back substitution) and publishes panel and
//     z_J through LDS (double-buffered: ONE workgroup barrier per step); every wave updates the rows it owns; the owner of
//     row J + 1 updates that row first and factors it while the others are still in the trailing update (look-ahead);
//   * back substitution: the owner of row J folds the x blocks behind it into x_J and publishes it, one barrier per block.
// Same arithmetic as the one-wave solve (float32 MFMA tile products, v_rsq pivots), same C/D layout as the image.
#pragma once
#include "als_wg_kernels.hip.h"

namespace ycnr {

#ifndef YCNR_PAIR_WAVES
#define YCNR_PAIR_WAVES 4  // waves per row: 2 (one per SIMD, 512 registers each) or 4 (two per SIMD, 256 each; C5 shard 146.7 -> 141.6 ms)
#endif
constexpr int kPairWaves = YCNR_PAIR_WAVES;
constexpr int kPairThreads = 64 * kPairWaves;

template <int NB>
struct PairCfg {
  static constexpr int NW = kPairWaves;
  // owner of block row r: rows dealt back and forth over the waves (0 1 1 0 0 1 1 0 ... / 0 1 2 3 3 2 1 0 ...), which
  // balances the tile counts of a triangular matrix (NB = 16: 68 tiles each of two waves, 34 each of four)
  static __host__ __device__ constexpr int owner(int r) { return (r % (2 * NW)) < NW ? (r % (2 * NW)) : 2 * NW - 1 - (r % (2 * NW)); }
  // position of tile (bi, bj) among the tiles of wave W's rows (row-major over its rows)
  static __host__ __device__ constexpr int idx(int W, int bi, int bj) {
    int n = 0;
    for (int r = 0; r < bi; ++r)
      if (owner(r) == W) n += NB - r;
    return n + (bj - bi);
  }
  static __host__ __device__ constexpr int count(int W) { return idx(W, NB, NB); }
  static constexpr int LDW = 20;
  // LDS (floats): two panel buffers of NB tiles, z, x, one pair of 16 x 16 images per wave, a flag
  static constexpr int PANEL = NB * 256;
  static constexpr int OFF_Z = 2 * PANEL;
  static constexpr int OFF_X = OFF_Z + NB * 16;
  static constexpr int OFF_IMG = OFF_X + NB * 16;
  static constexpr int OFF_FLAG = OFF_IMG + NW * 2 * 16 * LDW;
  static constexpr int LDS_BYTES = (OFF_FLAG + 4) * 4;
};

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1.  The tile indices below must be constants
// whatever the optimizer's unrolling thresholds say (a rolled loop would index the tile registers dynamically:
// the first version of this file compiled to 1.2 KB of scratch per lane and 180 MFMAs instead of 3200).
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

template <int NB, int W>
struct PairSolve {
  using C = PairCfg<NB>;
  using Tr = MfmaTraits<float>;
  using acc_t = typename Tr::acc_t;
  using Sm = SolveMfmaF32<1>;
  static constexpr int LDW = C::LDW;
  static constexpr int NREG = C::count(W) > 0 ? C::count(W) : 1;

  // slab: the row's image (tiles in image layout) + b.  L: the workgroup's LDS.  Returns true when a real pivot of
  // this wave's diagonal tiles was not positive.  x is left in L + OFF_X (NB * 16 floats) after the last barrier.
  static __device__ __forceinline__ bool run(const float *__restrict__ slab, float *L, int kDiag, float lam, int lane) {
    const int g = lane >> 4, c = lane & 15;
    float *panel = L, *zb = L + C::OFF_Z, *xb = L + C::OFF_X;
    float *Dt = L + C::OFF_IMG + W * (2 * 16 * LDW), *Wt = Dt + 16 * LDW;
    const int off = wg_tile_lane_off(lane) >> 2;
    constexpr int NT = tile_count(NB);
    acc_t reg[NREG];
    float bpart[NB];  // own rows only
    // ---- load the rows this wave owns; + lam on the real diagonal, 1 on the padded one
    static_for<0, NB>([&](auto BI) {
      constexpr int bi = decltype(BI)::value;
      if constexpr (C::owner(bi) == W) {
        static_for<bi, NB>([&](auto BJ) {
          constexpr int bj = decltype(BJ)::value;
          acc_t v = *reinterpret_cast<const acc_t *>(slab + tile_index(bi, bj, NB) * 256 + off);
          if constexpr (bi == bj) {
            const float add = (bi * 16 + c < kDiag) ? lam : 1.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] += ((c >> 2) == g && t == (c & 3)) ? add : 0.0f;
          }
          reg[C::idx(W, bi, bj)] = v;
        });
        bpart[bi] = g == 0 ? slab[NT * 256 + bi * 16 + c] : 0.0f;
      } else {
        bpart[bi] = 0.0f;
      }
    });
    float dmin = 3.0e38f;
    // ---- factor the diagonal tile of row J (owned by this wave) and turn the row into the panel U[J][.], published
    // with z_J through LDS buffer J & 1
    auto factor_panel = [&](auto JJ) {
      constexpr int J = decltype(JJ)::value;
      float *pb = panel + (J & 1) * C::PANEL;
      float *zJ = zb + J * 16;  // written once, by the owner of row J
      // diagonal tile -> L (16 pivots), W = L^-1 through the wave's LDS images (SolveMfmaF32::solve, steps 1-2)
      {
        const acc_t d = reg[C::idx(W, J, J)];
#pragma unroll
        for (int t = 0; t < 4; ++t) Dt[(4 * g + t) * LDW + c] = d[t];
      }
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
      Sm::template pivots_dpp<0, 16>(R, dmin);  // (padded pivots are rows of the identity: exact)
      if (g == 1) {
        float4 *dst = reinterpret_cast<float4 *>(Wt + c * LDW);
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) dst[m4] = float4{R[4 * m4], R[4 * m4 + 1], R[4 * m4 + 2], R[4 * m4 + 3]};
      }
      acc_t Wd;
      {
        const float4 v = *reinterpret_cast<const float4 *>(Wt + c * LDW + 4 * g);
        Wd = acc_t{v.x, v.y, v.z, v.w};
      }
      float Aop[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) Aop[q] = Wt[(4 * g + q) * LDW + c];
      reg[C::idx(W, J, J)] = Wd;
      // z_J = W b_J, published in row form: zJ[r] = z[16 J + r]
      {
        const float bcolJ = Sm::group_sum(bpart[J]);
        float zr[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) zr[t] = Wd[t] * bcolJ;
        Sm::row_sum4(zr[0], zr[1], zr[2], zr[3]);
        if (c == 0) *reinterpret_cast<float4 *>(zJ + 4 * g) = float4{zr[0], zr[1], zr[2], zr[3]};
      }
      // panel U[J][bj] = W T[J][bj]: kept in registers (back substitution), published to LDS as [lane][4]
      static_for<J + 1, NB>([&](auto BJ) {
        constexpr int bj = decltype(BJ)::value;
        const acc_t T = reg[C::idx(W, J, bj)];
        acc_t P = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 4; ++q) P = Tr::mma(Aop[q], T[q], P);
        reg[C::idx(W, J, bj)] = P;
        *reinterpret_cast<acc_t *>(pb + bj * 256 + lane * 4) = P;
      });
    };
    // ---- row bi (owned by this wave) of the trailing update of step J: T[bi][bj] -= U[J][bi]^T U[J][bj], b_bi -= U[J][bi]^T z_J
    auto update_row = [&](auto JJ, auto BI) {
      constexpr int J = decltype(JJ)::value, bi = decltype(BI)::value;
      const float *pb = panel + (J & 1) * C::PANEL;
      const float4 z4 = *reinterpret_cast<const float4 *>(zb + J * 16 + 4 * g);
      const acc_t Pi = *reinterpret_cast<const acc_t *>(pb + bi * 256 + lane * 4);
      {
        float s = bpart[bi];
        s = fmaf(-Pi[0], z4.x, s);
        s = fmaf(-Pi[1], z4.y, s);
        s = fmaf(-Pi[2], z4.z, s);
        s = fmaf(-Pi[3], z4.w, s);
        bpart[bi] = s;
      }
      float nA[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) nA[q] = -Pi[q];
      static_for<bi, NB>([&](auto BJ) {
        constexpr int bj = decltype(BJ)::value;
        const acc_t Pj = *reinterpret_cast<const acc_t *>(pb + bj * 256 + lane * 4);
        acc_t t = reg[C::idx(W, bi, bj)];
#pragma unroll
        for (int q = 0; q < 4; ++q) t = Tr::mma(nA[q], Pj[q], t);
        reg[C::idx(W, bi, bj)] = t;
      });
    };
    // Look-ahead: in step J the owner of row J + 1 updates THAT row first, factors it and publishes panel J + 1 (into
    // the other LDS buffer) while its partner is still in the trailing update of step J; then it updates its other
    // rows.  One barrier per step: behind it panel J + 1 is complete and nobody reads panel J any more.  (Without the
    // look-ahead the partner idled through every factorisation: 37 us of a CU per row instead of ~20.)
    if constexpr (C::owner(0) == W) factor_panel(std::integral_constant<int, 0>{});
    __syncthreads();
    static_for<0, NB - 1>([&](auto JJ) {
      constexpr int J = decltype(JJ)::value;
      if constexpr (C::owner(J + 1) == W) {
        update_row(JJ, std::integral_constant<int, J + 1>{});
        factor_panel(std::integral_constant<int, J + 1>{});
        static_for<J + 2, NB>([&](auto BI) {
          if constexpr (C::owner(decltype(BI)::value) == W) update_row(JJ, BI);
        });
      } else {
        static_for<J + 1, NB>([&](auto BI) {
          if constexpr (C::owner(decltype(BI)::value) == W) update_row(JJ, BI);
        });
      }
      __syncthreads();
    });
    // ---- back substitution: x_J = W_J^T (z_J - sum_{bj > J} U[J][bj] x_bj), the owner of row J publishes x_J
    static_for<0, NB>([&](auto JR) {
      constexpr int J = NB - 1 - decltype(JR)::value;
      if constexpr (C::owner(J) == W) {
        float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        static_for<J + 1, NB>([&](auto BJ) {
          constexpr int bj = decltype(BJ)::value;
          const acc_t u = reg[C::idx(W, J, bj)];
          const float xc = xb[16 * bj + c];
#pragma unroll
          for (int t = 0; t < 4; ++t) part[t] = fmaf(u[t], xc, part[t]);
        });
        const acc_t Wd = reg[C::idx(W, J, J)];
        const float4 z4 = *reinterpret_cast<const float4 *>(zb + 16 * J + 4 * g);
        const float z[4] = {z4.x, z4.y, z4.z, z4.w};
        float sx = 0.0f;
        if (J + 1 < NB) Sm::row_sum4(part[0], part[1], part[2], part[3]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float y = (J + 1 < NB) ? z[t] - part[t] : z[t];
          sx = fmaf(Wd[t], y, sx);
        }
        sx = Sm::group_sum(sx);
        if (g == 0) xb[16 * J + c] = sx;
      }
      __syncthreads();
    });
    return !(dmin > 0.0f);
  }
};

// whole rows of a batch: Gramian -> the batch's slab number (unit - first), persistent over [first, first + count)
template <int NB>
__global__ __launch_bounds__(kWgThreads, 2) void als_wg_gram_rowslab_kernel(StepArgs<float> a, float *rowSlabs, int32_t first, int32_t count) {
  using G = WgGram<NB>;
  using C = WgCfg<NB>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l16 = tid & 15, rho = tid >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  typename G::Stage s0, s1;
  typename G::Meta m2;
  int32_t ui = blockIdx.x;
  if (ui >= count) return;
  Unit u = a.units[first + ui];
  auto prefetch = [&](const Unit &v) {  // a row's first two steps, requested while the row before it is finished
    const int64_t n = v.end - v.beg;
    const typename G::Meta m0 = G::load_meta(a, v.beg, n, 0, rho), m1 = G::load_meta(a, v.beg, n, 1, rho);
    m2 = G::load_meta(a, v.beg, n, 2, rho);
    G::load_rows(s0, a, m0, l16);
    G::load_rows(s1, a, m1, l16);
  };
  prefetch(u);
  while (true) {
    typename G::acc_t acc[G::NACC];
    float bacc[4][4];
    G::run(a, u.beg, u.end - u.beg, smem, acc, bacc, s0, s1, m2);
    float *slab = rowSlabs + (int64_t)ui * C::SLAB_FLOATS;
    ui += gridDim.x;
    const bool more = ui < count;
    if (more) {
      u = a.units[first + ui];
      prefetch(u);
    }
    G::store_tiles_w(wave, acc, slab, lane);
    G::reduce_b(bacc, smem);
    if (tid < NB * 16) slab[(int64_t)C::NT * 256 + tid] = reinterpret_cast<const float *>(smem + C::VEC_OFF)[tid];
    if (!more) break;
    __syncthreads();
  }
}

// one 2-wave workgroup per row of the batch: slab -> x -> the row of the solved matrix
template <int NB>
__global__ __launch_bounds__(kPairThreads, kPairWaves == 2 ? 1 : 2) void als_slab_solve2_kernel(StepArgs<float> a, const float *rowSlabs, int32_t first) {
  extern __shared__ __attribute__((aligned(16))) float ldsp[];
  using PC = PairCfg<NB>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Unit u = a.units[first + blockIdx.x];
  const float lam = (float)(a.lambda * (double)(u.end - u.beg));
  const int kDiag = a.kReal > 0 ? a.kReal : a.k;
  const float *slab = rowSlabs + (int64_t)blockIdx.x * wg_slab_floats(NB);
  int *flag = reinterpret_cast<int *>(ldsp + PC::OFF_FLAG);
  if (tid == 0) *flag = 0;
  bool bad;
  if constexpr (kPairWaves == 2) {
    if (wave == 0) bad = PairSolve<NB, 0>::run(slab, ldsp, kDiag, lam, lane);
    else bad = PairSolve<NB, 1>::run(slab, ldsp, kDiag, lam, lane);
  } else {
    if (wave == 0) bad = PairSolve<NB, 0>::run(slab, ldsp, kDiag, lam, lane);
    else if (wave == 1) bad = PairSolve<NB, 1>::run(slab, ldsp, kDiag, lam, lane);
    else if (wave == 2) bad = PairSolve<NB, 2 % kPairWaves>::run(slab, ldsp, kDiag, lam, lane);
    else bad = PairSolve<NB, 3 % kPairWaves>::run(slab, ldsp, kDiag, lam, lane);
  }
  // (the last barrier of the back substitution lies behind both waves: x is complete in LDS)
  float *out = a.solved + (int64_t)u.row * a.k;
  const float *xb = ldsp + PC::OFF_X;
  float chk = 0.0f;
  for (int i = tid; i < a.k; i += kPairThreads) {
    const float x = xb[i];
    out[i] = x;
    chk = fmaf(x, 0.0f, chk);
  }
  if ((bad && lane == 0) || !(chk == 0.0f)) atomicOr(flag, 1);  // a pivot that was not positive, or NaN / Inf in x
  __syncthreads();
  if (tid == 0 && *flag) {
    atomicAdd(&a.err->count, 1);
    a.err->firstRow = u.row;
  }
}

}  // namespace ycnr
