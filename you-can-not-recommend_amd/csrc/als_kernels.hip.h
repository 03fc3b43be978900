// als_kernels.hip.h -- CDNA4 (gfx950) device code of the ALS half-step.
//
// What one row of a half-step computes (reference: EmfWorker.mw_calcTrainAlsPortion,
// lib/emf/EmfWorker.js:214-248):
//     Y = fixed[indx[0..n), :]            gather        (EmfBase.js:537-555)
//     A = Y^T Y + (lambda*n) I            k x k         (EmfWorker.js:231-235)
//     b = Y^T r                           k             (EmfWorker.js:238-245)
//     x = A^-1 b -> solved[row, :]        in place      (EmfWorker.js:246-247)
//
// Mapping to the machine (DESIGN.md has the full account):
//  * One 64-lane wavefront owns one work unit = a row, or a chunk of a heavy row.
//  * The Gramian is accumulated on the matrix cores with v_mfma_f32_16x16x4_f32
//    (v_mfma_f64_16x16x4_f64 in double mode): exact IEEE fma chains, no reduced
//    precision.  k is padded to NB*16; only the NB*(NB+1)/2 upper 16x16 tiles are
//    kept, all of them in the wave's accumulator registers.
//  * The gathered factor rows are read straight from global memory in MFMA operand
//    layout: lane l holds Y[n0 + (l>>4)][16*cb + (l&15)], which is at once the
//    A operand of tile row cb and the B operand of tile column cb, so 4 ratings cost
//    NB loads and NB*(NB+1)/2 MFMAs.  No LDS staging is needed at this ratio.
//  * b is accumulated beside it on the VALU from the same operand registers.
//  * Rows that fit one unit are solved by the same wave (Cholesky in LDS); split rows
//    write partial tiles to a slab and als_reduce_solve sums them in a fixed order,
//    so results do not depend on how many GPUs or workgroups took part.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

// Register budget of the fused Gramian + solve kernel: 2 waves per SIMD = at most 256
// registers per lane (k = 100 needs 206 and does not spill; 3 would spill 220 B per lane).
#ifndef YCNR_FUSED_WAVES_PER_SIMD
#define YCNR_FUSED_WAVES_PER_SIMD 2
#endif
// Gather prefetch depth (4-rating steps) of the fused kernel's Gramian loop; 1 = shallow loop.
#ifndef YCNR_FUSED_PREFETCH
#define YCNR_FUSED_PREFETCH 2
#endif

// How many 4-rating steps ahead the Gramian-only kernel requests its gathered operands.
// 0 selects the shallow (one step ahead) loop.
// Dual-form kernels: read all multipliers of a pivot before its updates
// (SolveMfmaF32::solve<BATCH>); 6.78 -> 6.54 ms over the six classes of a MAL-scale iteration
#ifndef YCNR_DUAL_BATCH
#define YCNR_DUAL_BATCH true
#endif
#ifndef YCNR_SLAB_PREFETCH
#define YCNR_SLAB_PREFETCH 3
#endif
// Pivots of the float32 diagonal tile: 1 = multipliers by DPP row_newbcast from a copy of the pivot
// column replicated into every 16-lane row (one instruction per updated column), 0 = v_readlane + v_fma
#ifndef YCNR_DPP_PIVOTS
#define YCNR_DPP_PIVOTS 1
#endif
// Primal float32 solve of k = 16 m + 4 factors: eliminate the four edge columns first (SolveMfmaF32::solve_edge4)
#ifndef YCNR_EDGE4_SOLVE
#define YCNR_EDGE4_SOLVE 1
#endif
#ifndef YCNR_SLAB_WAVES_PER_SIMD
#define YCNR_SLAB_WAVES_PER_SIMD 2
#endif

namespace ycnr {

// A wave-level work unit: ratings [beg, end) of the local CSR belong to `row`.
// slab < 0: the unit covers the whole row and solves it; else it writes partial
// sums to slab number `slab`.
struct Unit {
  int64_t beg;
  int64_t end;
  int32_t row;
  int32_t slab;
};

// A row whose Gramian was split over nslabs units (slab0 .. slab0+nslabs-1).
struct SplitRow {
  int64_t n;  // ratings of the row (for lambda * n)
  int32_t row;
  int32_t slab0;
  int32_t nslabs;
  int32_t pad;
};

struct ErrInfo {
  int32_t count;     // rows with a non-positive pivot
  int32_t firstRow;  // one of them
};

template <typename T>
struct MfmaTraits;

template <>
struct MfmaTraits<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout: col = lane & 15, row = 4*(lane>>4) + reg
  static __device__ __forceinline__ int cd_row(int lane, int reg) { return ((lane >> 4) << 2) + reg; }
};

template <>
struct MfmaTraits<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // f64 C/D layout differs: col = lane & 15, row = (lane>>4) + 4*reg
  static __device__ __forceinline__ int cd_row(int lane, int reg) { return (lane >> 4) + (reg << 2); }
};

__host__ __device__ constexpr int tile_count(int nb) { return nb * (nb + 1) / 2; }
__host__ __device__ constexpr int tile_index(int bi, int bj, int nb) {
  return bi * nb - bi * (bi - 1) / 2 + (bj - bi);
}
// elements of T one split unit writes: NT tiles x 4 regs x 64 lanes + NB b-partials x 64 lanes
__host__ __device__ constexpr int64_t slab_elems(int nb) { return (int64_t)(tile_count(nb) * 4 + nb) * 64; }

template <typename T>
__device__ __forceinline__ T wave_shfl_xor(T v, int mask);
template <>
__device__ __forceinline__ float wave_shfl_xor<float>(float v, int mask) { return __shfl_xor(v, mask, 64); }
template <>
__device__ __forceinline__ double wave_shfl_xor<double>(double v, int mask) { return __shfl_xor(v, mask, 64); }

template <typename T, int NB>
struct Gram {
  using Tr = MfmaTraits<T>;
  using acc_t = typename Tr::acc_t;
  static constexpr int NT = tile_count(NB);
  static constexpr int KP = NB * 16;

  // Operand registers for 4 consecutive ratings: lane (g = l>>4, c = l&15) reads
  // row[16*cb + c] for every column block cb, where row is the gathered factor row of its
  // rating, or a row of zeros when the rating index is past the unit's end (branch-free).
  // Only the last block can run past k: those lanes read a zero instead.
  static __device__ __forceinline__ void load_y(T (&y)[NB], const T *__restrict__ row,
                                                const T *__restrict__ zeros, int k, int c) {
#pragma unroll
    for (int cb = 0; cb < NB - 1; ++cb) y[cb] = row[cb * 16 + c];
    const int col = (NB - 1) * 16 + c;
    const T *plast = col < k ? row + col : zeros;  // select the address, not the loaded value
    y[NB - 1] = *plast;
  }

  static __device__ __forceinline__ void mma_step(acc_t (&acc)[NT], T (&bacc)[NB], const T (&y)[NB], T r) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) bacc[cb] = fma(y[cb], r, bacc[cb]);
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
#pragma unroll
      for (int bj = bi; bj < NB; ++bj) {
        acc[tile_index(bi, bj, NB)] = Tr::mma(y[bi], y[bj], acc[tile_index(bi, bj, NB)]);
      }
    }
  }

  // acc += Y^T Y (upper tiles), bacc += per-lane-group partials of Y^T r over ratings [beg, end).
  // Two-deep software pipeline: indices two steps ahead, operands one step ahead of the MFMAs.
  static __device__ __forceinline__ void accumulate(acc_t (&acc)[NT], T (&bacc)[NB],
                                                    const int32_t *__restrict__ indx,
                                                    const T *__restrict__ vals,
                                                    const T *__restrict__ fixed,
                                                    const T *__restrict__ zeros, int k, int64_t beg,
                                                    int64_t end, int lane) {
    const int g = lane >> 4, c = lane & 15;
    const int64_t nsteps = (end - beg + 3) >> 2;
    const int64_t last = end - 1;
    // (index, rating, row pointer) of this lane group's rating in a step; past-the-end
    // ratings read the last valid entry but point at the zero row and carry rating 0
    int64_t n = beg + g;
    int64_t nc = n < end ? n : last;
    const T *row0 = n < end ? fixed + (int64_t)indx[nc] * k : zeros;
    T rv = vals[nc];
    T r0 = n < end ? rv : T(0);
    n += 4;
    nc = n < end ? n : last;
    int32_t id1 = indx[nc];
    rv = vals[nc];
    T r1 = n < end ? rv : T(0);
    bool v1 = n < end;
    T yA[NB], yB[NB];
    load_y(yA, row0, zeros, k, c);
    for (int64_t i = 0; i < nsteps; ++i) {
      const T *row1 = v1 ? fixed + (int64_t)id1 * k : zeros;
      load_y(yB, row1, zeros, k, c);  // operands of step i+1
      const T ra = r0;
      r0 = r1;
      n += 4;  // indices of step i+2
      nc = n < end ? n : last;
      v1 = n < end;
      id1 = indx[nc];
      rv = vals[nc];
      r1 = v1 ? rv : T(0);
      mma_step(acc, bacc, yA, ra);
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) yA[cb] = yB[cb];
    }
  }
  // Deep-prefetch variant for long units whose gathered rows come from HBM (the item side:
  // the user matrix does not fit the Infinity Cache).  Column ids and ratings are fetched 64
  // at a time, one per lane, and handed to the step that needs them with a bpermute; the
  // operands of step s + PF are requested while step s runs on the matrix cores, through a
  // ring of PF + 1 register sets (PF + 1 divides 16, so ring positions are static).
  template <int PF>
  static __device__ __forceinline__ void accumulate_deep(acc_t (&acc)[NT], T (&bacc)[NB],
                                                         const int32_t *__restrict__ indx,
                                                         const T *__restrict__ vals,
                                                         const T *__restrict__ fixed,
                                                         const T *__restrict__ zeros, int k, int64_t beg,
                                                         int64_t end, int lane) {
    static_assert(16 % (PF + 1) == 0, "ring size must divide the 16 steps of a 64-rating block");
    const int g = lane >> 4, c = lane & 15;
    const int64_t n = end - beg;
    const int64_t nblk = (n + 63) >> 6;
    auto fetch_blk = [&](int64_t b, int32_t &id, T &r) {
      const int64_t q = b * 64 + lane;
      const bool v = q < n;
      const int64_t qc = beg + (v ? q : n - 1);
      const int32_t i = indx[qc];
      const T t = vals[qc];
      id = v ? i : -1;  // -1: past the end of the unit -> the zero row
      r = v ? t : T(0);
    };
    auto row_of = [&](int32_t id) {  // select, not branch: the product is formed for the clamped id
      const T *base = fixed + (int64_t)(id < 0 ? 0 : id) * k;
      return id >= 0 ? base : zeros;
    };
    int32_t idc, idn;
    T rc, rn;
    fetch_blk(0, idc, rc);
    fetch_blk(1, idn, rn);
    T y[PF + 1][NB];
#pragma unroll
    for (int t = 0; t < PF; ++t) load_y(y[t], row_of(__shfl(idc, 4 * t + g, 64)), zeros, k, c);
    const int64_t ngrp = nblk * (16 / (PF + 1));  // groups of PF + 1 steps
#pragma unroll 1
    for (int64_t grp = 0; grp < ngrp; ++grp) {
      const int sbase = (int)(grp % (16 / (PF + 1))) * (PF + 1);
      if (sbase == 0 && grp != 0) {  // entering the next 64-rating block
        idc = idn;
        rc = rn;
        fetch_blk(grp / (16 / (PF + 1)) + 1, idn, rn);
      }
#pragma unroll
      for (int u = 0; u <= PF; ++u) {
        const int t = sbase + u + PF;  // step whose operands are requested now
        const int32_t src = t < 16 ? idc : idn;
        const int32_t id = __shfl(src, (4 * t + g) & 63, 64);
        load_y(y[(u + PF) % (PF + 1)], row_of(id), zeros, k, c);
        const T r = __shfl(rc, 4 * (sbase + u) + g, 64);
        mma_step(acc, bacc, y[u % (PF + 1)], r);
      }
    }
  }
};

// Uniform interface over the two Gramian forms, used by the kernels below:
//   State, init, accumulate, accumulate_deep<PF>, slab_count, store_slab, add_slab, to_tiles.
template <typename T, int NB>
struct GramPlain {
  using G = Gram<T, NB>;
  using acc_t = typename G::acc_t;
  static constexpr int NT = G::NT;
  struct State {
    acc_t acc[NT];
    T bacc[NB];
  };
  static __host__ __device__ constexpr int slab_regs() { return NT * 4 + NB; }
  static __device__ __forceinline__ void init(State &s) {
#pragma unroll
    for (int t = 0; t < NT; ++t) s.acc[t] = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) s.bacc[cb] = T(0);
  }
  static __device__ __forceinline__ void accumulate(State &s, const int32_t *indx, const T *vals, const T *fixed,
                                                    const T *zeros, int k, int64_t beg, int64_t end, int lane) {
    G::accumulate(s.acc, s.bacc, indx, vals, fixed, zeros, k, beg, end, lane);
  }
  template <int PF>
  static __device__ __forceinline__ void accumulate_deep(State &s, const int32_t *indx, const T *vals, const T *fixed,
                                                         const T *zeros, int k, int64_t beg, int64_t end, int lane) {
    G::template accumulate_deep<PF>(s.acc, s.bacc, indx, vals, fixed, zeros, k, beg, end, lane);
  }
  template <int PF>
  static __device__ __forceinline__ void accumulate_ring(State &s, const int32_t *indx, const T *vals, const T *fixed,
                                                         const T *zeros, int k, int64_t beg, int64_t end, int lane) {
    G::accumulate(s.acc, s.bacc, indx, vals, fixed, zeros, k, beg, end, lane);  // plain form: one step ahead
  }
  // Plain slab layout (also written by the bf16x6 chunk kernels): tile t as 64 lanes x 4 registers, i.e. ONE
  // 16-byte (float) access per lane and tile, then the rhs partials one register per 64-lane row.  s points at
  // the slab's element of this lane (base + lane), as before.  (Round 1 stored one register per 64-lane row
  // throughout: 119 four-byte loads per slab against a 63-deep vmcnt; the reduce kernel now has 4x the bytes in
  // flight per wave.)
  struct alignas(sizeof(T) * 4) Quad {
    T v[4];
  };
  static __device__ __forceinline__ void store_slab(const State &st, T *s) {
    Quad *q = reinterpret_cast<Quad *>(s + 3 * (int)threadIdx.x);  // base + 4 lane
#pragma unroll
    for (int t = 0; t < NT; ++t) q[t * 64] = Quad{{st.acc[t][0], st.acc[t][1], st.acc[t][2], st.acc[t][3]}};
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) s[(NT * 4 + cb) * 64] = st.bacc[cb];
  }
  static __device__ __forceinline__ void add_slab(State &st, const T *s) {
    const Quad *q = reinterpret_cast<const Quad *>(s + 3 * (int)threadIdx.x);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const Quad v = q[t * 64];
#pragma unroll
      for (int r = 0; r < 4; ++r) st.acc[t][r] += v.v[r];
    }
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) st.bacc[cb] += s[(NT * 4 + cb) * 64];
  }
  static __device__ __forceinline__ void to_tiles(State &st, acc_t (&full)[NT], T (&bfull)[NB], T *, int) {
#pragma unroll
    for (int t = 0; t < NT; ++t) full[t] = st.acc[t];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) bfull[cb] = st.bacc[cb];
  }
  // The slabs of a row with MANY chunks are added in double and rounded to T once (als_reduce_solve_kernel): a float32 chain over
  // n slabs in slab order carries an error of eps sqrt(n / 3) of the total -- 37 eps for the 4096 slabs of a 10 M-rating item --,
  // far above what the chunks' own chains contribute (each 1 / n of the total); in double the sum over the slabs is exact to
  // float32 precision.  Order still fixed: independent of how the rows are dealt to GPUs.
  // Registers: 119 double accumulators (k = 100) and the loads of a slab do not fit the 256 vector registers the vector ALU can
  // address -- the first build (all tiles at once) waited for every 16-byte load before it issued the next: 0.44 -> 1.0 ms for the
  // MAL item side's reduce.  Hence TP tiles per pass over the row's slabs, four slabs' loads of those tiles in flight.
  template <int T0, int TP>
  static __device__ __forceinline__ void sum_tiles_wide(State &st, const T *s0, int nslabs) {
    double d[TP][4];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) d[t][r] = 0.0;
    }
    const Quad *q = reinterpret_cast<const Quad *>(s0 + 3 * (int)threadIdx.x) + T0 * 64;
    constexpr int64_t STRIDE = (int64_t)slab_regs() * 16;  // slab stride in quads: slab_regs() * 64 elements
    // GS slabs' loads of these tiles are issued together, then added in slab order (left to itself hipcc waited for every
    // 16-byte load before it issued the next one: a row of 64 slabs was a chain of 1800 round trips)
    constexpr int GS = 4;
    for (int sl = 0; sl < nslabs; sl += GS) {
      Quad v[GS][TP];
#pragma unroll
      for (int u = 0; u < GS; ++u) {
        const Quad *qs = q + (int64_t)(sl + u < nslabs ? sl + u : nslabs - 1) * STRIDE;
#pragma unroll
        for (int t = 0; t < TP; ++t) v[u][t] = qs[t * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < GS; ++u) {
        if (sl + u < nslabs) {  // wave-uniform
#pragma unroll
          for (int t = 0; t < TP; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) d[t][r] += (double)v[u][t].v[r];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < TP; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) st.acc[T0 + t][r] = (T)d[t][r];
    }
  }
  template <int T0>
  static __device__ __forceinline__ void sum_passes_wide(State &st, const T *s0, int nslabs) {
    constexpr int TP = NT - T0 < 7 ? NT - T0 : 7;
    if constexpr (TP > 0) {
      sum_tiles_wide<T0, TP>(st, s0, nslabs);
      sum_passes_wide<T0 + TP>(st, s0, nslabs);
    }
  }
  // st <- sum over the nslabs slabs at s0 (this lane's element of the first slab: base + lane), every element summed in double
  static __device__ __forceinline__ void sum_slabs_wide(State &st, const T *s0, int nslabs) {
    sum_passes_wide<0>(st, s0, nslabs);
    double b[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) b[cb] = 0.0;
#pragma unroll 4
    for (int sl = 0; sl < nslabs; ++sl) {
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) b[cb] += (double)s0[(int64_t)sl * (slab_regs() * 64) + (NT * 4 + cb) * 64];
    }
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) st.bacc[cb] = (T)b[cb];
  }
};

// Gramian for k = 16 (NB-1) + 4 (k = 20, 36, ..., 100, 116), float32: the last 4 columns would
// cost a whole tile column on the matrix cores (7 of 28 tiles at k = 100 for 4 of 100 columns),
// so they are accumulated on the VALU instead, in the shadow of the MFMAs:
//   edge[cb][j]  += y[16 cb + c] * e[j]     e[j] = the rating's factors 16 (NB-1) + j, one
//   corner[j]    += e[c] * e[j]   (c < 4)   float4 load per lane group
//   be[j]        += e[j] * r
// 21 MFMAs + 34 v_fma per 4 ratings instead of 28 + 7.  to_tiles rebuilds the NB-block tile set
// the solvers expect through a 1.6 KB LDS image.
template <int NB>
struct GramEdge {
  using Tr = MfmaTraits<float>;
  using acc_t = typename Tr::acc_t;
  static constexpr int NBM = NB - 1;
  static constexpr int NTM = tile_count(NBM);
  static constexpr int NT = tile_count(NB);
  struct State {
    acc_t acc[NTM];
    float bacc[NBM];
    float edge[NBM][4];
    float corner[4];
    float be[4];
  };
  struct Ops {
    float y[NBM];
    float4 e;
  };
  static __host__ __device__ constexpr int slab_regs() { return NTM * 4 + NBM + NBM * 4 + 8; }
  static __device__ __forceinline__ void init(State &s) {
#pragma unroll
    for (int t = 0; t < NTM; ++t) s.acc[t] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) {
      s.bacc[cb] = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s.edge[cb][j] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s.corner[j] = s.be[j] = 0.0f;
  }
  static __device__ __forceinline__ void load(Ops &o, const float *__restrict__ row, int c) {
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) o.y[cb] = row[cb * 16 + c];
    o.e = *reinterpret_cast<const float4 *>(row + 16 * NBM);
  }
  static __device__ __forceinline__ void step(State &s, const Ops &o, float r, int c) {
    const float e[4] = {o.e.x, o.e.y, o.e.z, o.e.w};
    const float ec = c == 0 ? e[0] : (c == 1 ? e[1] : (c == 2 ? e[2] : e[3]));
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) {
      s.bacc[cb] = fmaf(o.y[cb], r, s.bacc[cb]);
#pragma unroll
      for (int j = 0; j < 4; ++j) s.edge[cb][j] = fmaf(o.y[cb], e[j], s.edge[cb][j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s.be[j] = fmaf(e[j], r, s.be[j]);
      s.corner[j] = fmaf(ec, e[j], s.corner[j]);
    }
#pragma unroll
    for (int bi = 0; bi < NBM; ++bi) {
#pragma unroll
      for (int bj = bi; bj < NBM; ++bj)
        s.acc[tile_index(bi, bj, NBM)] = Tr::mma(o.y[bi], o.y[bj], s.acc[tile_index(bi, bj, NBM)]);
    }
  }
  static __device__ __forceinline__ void accumulate(State &st, const int32_t *__restrict__ indx,
                                                    const float *__restrict__ vals, const float *__restrict__ fixed,
                                                    const float *__restrict__ zeros, int k, int64_t beg, int64_t end,
                                                    int lane) {
    const int g = lane >> 4, c = lane & 15;
    const int64_t nsteps = (end - beg + 3) >> 2;
    const int64_t last = end - 1;
    int64_t n = beg + g;
    int64_t nc = n < end ? n : last;
    const float *row0 = n < end ? fixed + (int64_t)indx[nc] * k : zeros;
    float rv = vals[nc];
    float r0 = n < end ? rv : 0.0f;
    n += 4;
    nc = n < end ? n : last;
    int32_t id1 = indx[nc];
    rv = vals[nc];
    float r1 = n < end ? rv : 0.0f;
    bool v1 = n < end;
    Ops A, B;
    load(A, row0, c);
    for (int64_t i = 0; i < nsteps; ++i) {
      const float *row1 = v1 ? fixed + (int64_t)id1 * k : zeros;
      load(B, row1, c);
      const float ra = r0;
      r0 = r1;
      n += 4;
      nc = n < end ? n : last;
      v1 = n < end;
      id1 = indx[nc];
      rv = vals[nc];
      r1 = v1 ? rv : 0.0f;
      step(st, A, ra, c);
      A = B;
    }
  }
  // Like accumulate, with the operands of step s + PF requested while step s runs (ring of
  // PF + 1 register sets, loop unrolled by PF + 1; at most PF zero steps are added at the end).
  // Column ids are loaded one step before their operands are requested.
  template <int PF>
  static __device__ __forceinline__ void accumulate_ring(State &st, const int32_t *__restrict__ indx,
                                                         const float *__restrict__ vals,
                                                         const float *__restrict__ fixed,
                                                         const float *__restrict__ zeros, int k, int64_t beg,
                                                         int64_t end, int lane) {
    const int g = lane >> 4, c = lane & 15;
    const int64_t nsteps = (end - beg + 3) >> 2;
    const int64_t last = end - 1;
    auto fetch = [&](int64_t step, int32_t &id, float &r) {  // id < 0: past the end -> zero row
      const int64_t n = beg + (step << 2) + g;
      const bool v = n < end;
      const int64_t nc = v ? n : last;
      const int32_t i = indx[nc];
      const float t = vals[nc];
      id = v ? i : -1;
      r = v ? t : 0.0f;
    };
    auto row_of = [&](int32_t id) {  // select, not branch: the product is formed for the clamped id
      const float *base = fixed + (int64_t)(id < 0 ? 0 : id) * k;
      return id >= 0 ? base : zeros;
    };
    Ops y[PF + 1];
    float rr[PF + 1];
    int32_t idn;
    float rn;
#pragma unroll
    for (int t = 0; t < PF; ++t) {
      int32_t id;
      fetch(t, id, rr[t]);
      load(y[t], row_of(id), c);
    }
    fetch(PF, idn, rn);
    for (int64_t i = 0; i < nsteps; i += PF + 1) {
#pragma unroll
      for (int u = 0; u <= PF; ++u) {
        const int slot = (u + PF) % (PF + 1);
        load(y[slot], row_of(idn), c);  // operands of step i + u + PF
        rr[slot] = rn;
        fetch(i + u + PF + 1, idn, rn);
        step(st, y[u], rr[u], c);
      }
    }
  }
  template <int PF>
  static __device__ __forceinline__ void accumulate_deep(State &st, const int32_t *__restrict__ indx,
                                                         const float *__restrict__ vals,
                                                         const float *__restrict__ fixed,
                                                         const float *__restrict__ zeros, int k, int64_t beg,
                                                         int64_t end, int lane) {
    static_assert(16 % (PF + 1) == 0, "ring size must divide the 16 steps of a 64-rating block");
    const int g = lane >> 4, c = lane & 15;
    const int64_t n = end - beg;
    const int64_t nblk = (n + 63) >> 6;
    auto fetch_blk = [&](int64_t b, int32_t &id, float &r) {
      const int64_t q = b * 64 + lane;
      const bool v = q < n;
      const int64_t qc = beg + (v ? q : n - 1);
      const int32_t i = indx[qc];
      const float t = vals[qc];
      id = v ? i : -1;
      r = v ? t : 0.0f;
    };
    auto row_of = [&](int32_t id) {  // select, not branch: the product is formed for the clamped id
      const float *base = fixed + (int64_t)(id < 0 ? 0 : id) * k;
      return id >= 0 ? base : zeros;
    };
    int32_t idc, idn;
    float rc, rn;
    fetch_blk(0, idc, rc);
    fetch_blk(1, idn, rn);
    Ops y[PF + 1];
#pragma unroll
    for (int t = 0; t < PF; ++t) load(y[t], row_of(__shfl(idc, 4 * t + g, 64)), c);
    const int64_t ngrp = nblk * (16 / (PF + 1));
#pragma unroll 1
    for (int64_t grp = 0; grp < ngrp; ++grp) {
      const int sbase = (int)(grp % (16 / (PF + 1))) * (PF + 1);
      if (sbase == 0 && grp != 0) {
        idc = idn;
        rc = rn;
        fetch_blk(grp / (16 / (PF + 1)) + 1, idn, rn);
      }
#pragma unroll
      for (int u = 0; u <= PF; ++u) {
        const int t = sbase + u + PF;
        const int32_t src = t < 16 ? idc : idn;
        const int32_t id = __shfl(src, (4 * t + g) & 63, 64);
        load(y[(u + PF) % (PF + 1)], row_of(id), c);
        const float r = __shfl(rc, 4 * (sbase + u) + g, 64);
        step(st, y[u % (PF + 1)], r, c);
      }
    }
  }
  // slab order: MFMA tiles, bacc, edge, corner, be -- one register per 64-lane row
  static __device__ __forceinline__ void store_slab(const State &st, float *s) {
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[(t * 4 + r) * 64] = st.acc[t][r];
    }
    float *p = s + NTM * 4 * 64;
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) p[cb * 64] = st.bacc[cb];
    p += NBM * 64;
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) p[(cb * 4 + j) * 64] = st.edge[cb][j];
    }
    p += NBM * 4 * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      p[j * 64] = st.corner[j];
      p[(4 + j) * 64] = st.be[j];
    }
  }
  static __device__ __forceinline__ void add_slab(State &st, const float *s) {
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) st.acc[t][r] += s[(t * 4 + r) * 64];
    }
    const float *p = s + NTM * 4 * 64;
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) st.bacc[cb] += p[cb * 64];
    p += NBM * 64;
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) st.edge[cb][j] += p[(cb * 4 + j) * 64];
    }
    p += NBM * 4 * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      st.corner[j] += p[j * 64];
      st.be[j] += p[(4 + j) * 64];
    }
  }
  // (see GramPlain::Wide)
  struct Wide {
    double acc[NTM][4];
    double bacc[NBM];
    double edge[NBM][4];
    double corner[4];
    double be[4];
  };
  static __device__ __forceinline__ void init_wide(Wide &w) {
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) w.acc[t][r] = 0.0;
    }
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) {
      w.bacc[cb] = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) w.edge[cb][j] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) w.corner[j] = w.be[j] = 0.0;
  }
  static __device__ __forceinline__ void add_slab_wide(Wide &w, const float *s) {
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) w.acc[t][r] += (double)s[(t * 4 + r) * 64];
    }
    const float *p = s + NTM * 4 * 64;
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) w.bacc[cb] += (double)p[cb * 64];
    p += NBM * 64;
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) w.edge[cb][j] += (double)p[(cb * 4 + j) * 64];
    }
    p += NBM * 4 * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w.corner[j] += (double)p[j * 64];
      w.be[j] += (double)p[(4 + j) * 64];
    }
  }
  static __device__ __forceinline__ void narrow(State &st, const Wide &w) {
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) st.acc[t][r] = (float)w.acc[t][r];
    }
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) {
      st.bacc[cb] = (float)w.bacc[cb];
#pragma unroll
      for (int j = 0; j < 4; ++j) st.edge[cb][j] = (float)w.edge[cb][j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      st.corner[j] = (float)w.corner[j];
      st.be[j] = (float)w.be[j];
    }
  }
  // (the entry point of GramPlain; this form -- fixed matrices of 2 GB and more at k = 16 m + 4 -- keeps the one-pass sum)
  static __device__ __forceinline__ void sum_slabs_wide(State &st, const float *s0, int nslabs) {
    Wide w;
    init_wide(w);
    for (int sl = 0; sl < nslabs; ++sl) add_slab_wide(w, s0 + (int64_t)sl * (slab_regs() * 64));
    narrow(st, w);
  }
  // Rebuild the NB-block upper tile set (C/D layout) and the rhs partials from the state.
  // S: at least (16 NBM + 4) * 16 bytes of LDS, free for reuse after the call.
  static __device__ __forceinline__ void to_tiles(State &st, acc_t (&full)[NT], float (&bfull)[NB], float *S, int lane) {
    const int g = lane >> 4, c = lane & 15;
    float *E = S, *Cm = S + 16 * NBM * 4;
    auto gsum = [](float v) {
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      return v;
    };
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) {
      const float4 v = float4{gsum(st.edge[cb][0]), gsum(st.edge[cb][1]), gsum(st.edge[cb][2]), gsum(st.edge[cb][3])};
      if (g == 0) *reinterpret_cast<float4 *>(E + (16 * cb + c) * 4) = v;
    }
    {
      const float4 v = float4{gsum(st.corner[0]), gsum(st.corner[1]), gsum(st.corner[2]), gsum(st.corner[3])};
      if (g == 0 && c < 4) *reinterpret_cast<float4 *>(Cm + c * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int bi = 0; bi < NBM; ++bi) {
#pragma unroll
      for (int bj = bi; bj < NBM; ++bj) full[tile_index(bi, bj, NB)] = st.acc[tile_index(bi, bj, NBM)];
      acc_t v;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float x = E[(16 * bi + 4 * g + t) * 4 + (c & 3)];
        v[t] = c < 4 ? x : 0.0f;
      }
      full[tile_index(bi, NBM, NB)] = v;
    }
    {
      acc_t v;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float x = Cm[t * 4 + (c & 3)];
        v[t] = (g == 0 && c < 4) ? x : 0.0f;
      }
      full[tile_index(NBM, NBM, NB)] = v;
    }
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) bfull[cb] = st.bacc[cb];
    // be is uniform over c inside a lane group and still partial over groups: the solver sums groups
    bfull[NBM] = c == 0 ? st.be[0] : (c == 1 ? st.be[1] : (c == 2 ? st.be[2] : (c == 3 ? st.be[3] : 0.0f)));
    __syncthreads();
  }
};

template <typename T, int NB, bool EDGE>
struct GramSel {
  using type = GramPlain<T, NB>;
};
template <int NB>
struct GramSel<float, NB, true> {
  using type = GramEdge<NB>;
};

// ---------------------------------------------------------------------------------------
// Solve (A + lam I) x = b for one row, A given as upper MFMA tiles in registers.
// Version 1: dump to LDS, right-looking Cholesky A = U^T U by one wave with b carried as
// column k (so the forward substitution is part of the factorisation), then a
// column-oriented back substitution.  S is [KP][LD], LD = KP + 1 (odd: conflict-free
// row and column sweeps); only the upper triangle and column k are used.
template <typename T, int NB>
struct SolveLds {
  using Tr = MfmaTraits<T>;
  using acc_t = typename Tr::acc_t;
  static constexpr int NT = tile_count(NB);
  static constexpr int KP = NB * 16;
  static constexpr int LD = KP + 1;
  static constexpr size_t lds_bytes() { return sizeof(T) * (size_t)KP * LD; }

  // kd: the first kd of the k columns are factors, the rest zero padding of a padded upload (StepArgs::kReal): their diagonal
  // entry is 1, not lambda n (a zero lambda must not leave a zero pivot there); < 0: all k
  static __device__ __forceinline__ void run(const acc_t (&acc)[NT], const T (&bacc)[NB], T *S, int k,
                                             T lam, T *__restrict__ out_row, int row, ErrInfo *err,
                                             int lane, int kd = -1) {
    const int g = lane >> 4, c = lane & 15;
    if (kd < 0) kd = k;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
#pragma unroll
      for (int bj = bi; bj < NB; ++bj) {
        const acc_t a = acc[tile_index(bi, bj, NB)];
#pragma unroll
        for (int t = 0; t < 4; ++t) S[(bi * 16 + Tr::cd_row(lane, t)) * LD + bj * 16 + c] = a[t];
      }
    }
    __syncthreads();
    // b: sum the four lane-group partials; group 0 stores b[16*cb + c] into column k
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      T v = bacc[cb];
      v += wave_shfl_xor<T>(v, 16);
      v += wave_shfl_xor<T>(v, 32);
      const int i = cb * 16 + c;
      if (g == 0 && i < k) S[i * LD + k] = v;
    }
    for (int i = lane; i < k; i += 64) S[i * LD + i] += i < kd ? lam : T(1);
    __syncthreads();

    bool bad = false;
    for (int p = 0; p < k; ++p) {
      T d = S[p * LD + p];
      if (!(d > T(0))) {
        bad = true;
        d = T(1);
      }
      const T rs = T(1) / sqrt(d);
      const int c0 = p + 1 + lane, c1 = c0 + 64;
      T u0 = T(0), u1 = T(0);
      if (c0 <= k) {
        u0 = S[p * LD + c0] * rs;
        S[p * LD + c0] = u0;
      }
      if (c1 <= k) {
        u1 = S[p * LD + c1] * rs;
        S[p * LD + c1] = u1;
      }
      if (lane == 0) S[p * LD + p] = rs;  // keep 1/U[p][p]
      __syncthreads();
      for (int r = p + 1; r < k; ++r) {
        const T ur = S[p * LD + r];
        if (c0 >= r && c0 <= k) S[r * LD + c0] -= ur * u0;
        if (c1 >= r && c1 <= k) S[r * LD + c1] -= ur * u1;
      }
      __syncthreads();
    }
    // back substitution, x overwrites column k
    for (int p = k - 1; p >= 0; --p) {
      const T x = S[p * LD + k] * S[p * LD + p];
      __syncthreads();
      if (lane == 0) S[p * LD + k] = x;
      for (int r = lane; r < p; r += 64) S[r * LD + k] -= S[r * LD + p] * x;
      __syncthreads();
    }
    for (int i = lane; i < k; i += 64) out_row[i] = S[i * LD + k];
    if (bad && lane == 0) {
      atomicAdd(&err->count, 1);
      err->firstRow = row;
    }
  }
};

// ---------------------------------------------------------------------------------------
// Version 2 of the float32 solve: block Cholesky that never leaves the register file.
//
// A (upper 16x16 tiles, MFMA C/D layout: lane (g = l>>4, c = l&15), reg t <-> row 4g+t, col c)
// is factored as A = U^T U block row by block row (right-looking, NB block steps):
//   1. the 16x16 diagonal tile D goes through a 1.25 KB LDS image to a lane-per-row form;
//      lanes 0-15 run a right-looking Cholesky of D (lane i = row i of L) while lanes 16-31
//      apply the very same instruction stream to the identity (lane 16+c = column c of
//      L^-1): per pivot one v_rsq, one scale and (15-p) v_readlane + v_fma pairs;
//   2. W = L^-1 comes back through LDS twice: in C/D layout (kept in place of D, used by
//      the triangular solves) and in MFMA A-operand layout;
//   3. panel: U[J][bj] = W * T[J][bj] -- 4 MFMAs per tile; the B operand is T's own
//      registers after a 4x4 (register <-> lane-group) transpose made of two
//      v_permlane16_swap + two v_permlane32_swap;
//   4. trailing update: T[bi][bj] -= U[J][bi]^T U[J][bj] -- 4 MFMAs per tile from the
//      transposed panel registers;
//   5. the right-hand side rides along on the VALU: z_J = W b_J (row sums by DPP row_ror),
//      b_bj -= U[J][bj]^T z_J (sums over lane groups); back substitution mirrors it.
// Only MFMA, VALU, DPP and ~20 LDS instructions per block step; no barrier other than the
// wave's own LDS ordering.  k need not be a multiple of 16: the padded diagonal is 1.
// WSYNC: the struct runs inside a workgroup of SEVERAL waves that each solve a row of their own (als_row_pair_kernel): its LDS
// images are wave-private, so its synchronisation points order the wave's own LDS traffic (program order + a compiler barrier)
// instead of being workgroup barriers, which the other waves would never reach.  With one wave per workgroup the two mean the same.
template <int NB, bool WSYNC = false>
struct SolveMfmaF32 {
  using Tr = MfmaTraits<float>;
  using acc_t = typename Tr::acc_t;
  static constexpr int NT = tile_count(NB);
  static constexpr int LDW = 20;  // floats per LDS image row: 80 B keeps b128 accesses aligned
  static constexpr size_t lds_bytes() { return 2 * 16 * LDW * sizeof(float); }
  static __device__ __forceinline__ void lds_sync() {
    if constexpr (WSYNC) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else {
      __syncthreads();
    }
  }

  template <int N>
  static __device__ __forceinline__ float row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false));
  }
  // sum over the 16 lanes of each lane group, result in every lane of the group
  static __device__ __forceinline__ float row_sum(float v) {
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    return v;
  }
  // row_sum of FOUR values at once, one v_add_f32_dpp per value and stage (round 5).  hipcc makes two instructions of `v += row_ror(v)`
  // (v_mov_b32_dpp + v_add_f32, and now and then a third that clears the mov's `old` operand): 32+ per four sums, 224 per block step of a
  // solve and 7 x 32 in the dual classes' x = Y^T w -- a tenth of the user half-step's vector instructions at MAL scale.  Same operands,
  // same adds, same order: bit for bit row_sum.  Hazards hipcc does not pad around inline asm: a DPP operand written by a vector
  // instruction needs two wait states -- `s_nop 1` in front (the compiler's producer may sit directly before the statement), and
  // between the stages the other three values' instructions lie (isa_lint.py: lint_dpp walks these too).
#ifndef YCNR_ROW_SUM4_ASM
#define YCNR_ROW_SUM4_ASM 1
#endif
  static __device__ __forceinline__ void row_sum4(float &a, float &b, float &c, float &d) {
    // (NB >= 10 -- the dual classes of 145+ ratings at k > 128 -- keeps the compiler's form: with one more asm statement in their
    // solves hipcc spilled 536 bytes per lane in the 12-block class; see the pivots' note on NB >= 10)
    if constexpr (YCNR_ROW_SUM4_ASM && NB <= 9) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    } else {
      a = row_sum(a);
      b = row_sum(b);
      c = row_sum(c);
      d = row_sum(d);
    }
  }
  // sum over the 4 lane groups (same c), result in every group
  static __device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
  }
  static __device__ __forceinline__ float readlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
  }
  // out[q] at lane (g, c) = in[g] at lane (q, c): 4x4 transpose between the register index
  // and the lane-group index
  static __device__ __forceinline__ void transpose_rg(const acc_t &in, float (&out)[4]) {
    // Written as inline asm: with ROCm 7.2's __builtin_amdgcn_permlane{16,32}_swap hipcc folded
    // the four inputs into one (devtest/panelprobe.hip shows a single load feeding all swaps).
    // v_permlane16_swap a, b: a.row1 <-> b.row0, a.row3 <-> b.row2 (row = 16 lanes)
    // v_permlane32_swap a, b: a.rows{2,3} <-> b.rows{0,1}
    // The s_nop covers the VALU-write -> permlane-swap-read wait states hipcc does not insert
    // around inline asm.
    float r0 = in[0], r1 = in[1], r2 = in[2], r3 = in[3];
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(r0), "+v"(r1));
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(r2), "+v"(r3));
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r0), "+v"(r2));
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r1), "+v"(r3));
    out[0] = r0;
    out[1] = r1;
    out[2] = r2;
    out[3] = r3;
  }

  // value of lane N of the caller's 16-lane row, in every lane of the row (DPP row_newbcast)
  template <int N>
  static __device__ __forceinline__ float row_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + N, 0xF, 0xF, true));
  }
  // Lane groups 0 / 2 hold the rows of D (lane i: row i), groups 1 / 3 the columns of the identity that become L^-1.
  // Pivot P of the diagonal tile without v_readlane for the multipliers.  The multiplier of column j is
  // L[j][P], the value lane j of group 0 has in R[P]: one v_permlane16_swap copies that register's group
  // 0 / 2 values into groups 1 / 3, after which every 16-lane row finds L[j][P] in its own lane j and the
  // update of column j is ONE v_fmac_f32 with a DPP row_newbcast operand for all four groups (v_readlane +
  // v_fma before): same products, same fma, bit for bit the results of the v_readlane form.
  // hipcc does not fold a DPP move into v_fmac, so the updates are inline asm (the compiler still
  // interleaves the next pivot's v_readlane -> v_rsq -> scale chain with them; statements grouped per
  // pivot behind scheduling barriers were slower).  hipcc pads no hazards around inline asm, so the asm
  // is arranged to need none from it:
  //   * the swap has the two wait states a v_permlane swap needs after a VALU write of its operands in
  //     front and the two a DPP read needs after a VALU write of its source behind it;
  //   * everything that consumes the v_rsq result is compiler-generated (a multiply inside asm directly
  //     behind the v_rsq read a stale scale: the transcendental-result wait state was missing);
  //   * column P + 1, which the next pivot's v_readlane reads, is updated together with column P + 2 in
  //     one statement (or with an s_nop behind it), so one instruction always separates the two;
  //   * devtest/isa_lint.py checks these three rules, and that no register copy the compiler placed in
  //     front of a v_fmac_f32_dpp renews the DPP hazard, on the device assembly of every build
  //     (tests/test_isa_lint.py).
  template <int P, int Jn>
  struct PivotDpp {
    static __device__ __forceinline__ void updates(float (&R)[16], float l) {
      if constexpr (Jn < 16) {
        asm("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
            : "+v"(R[Jn])
            : "v"(l), "v"(R[P]), "n"(Jn));
        PivotDpp<P, Jn + 1>::updates(R, l);
      }
    }
    // columns P + 1 and P + 2 in one statement
    static __device__ __forceinline__ void first_two(float (&R)[16], float l) {
      if constexpr (P + 2 < 16) {
        asm("v_fmac_f32_dpp %0, -%2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_fmac_f32_dpp %1, -%2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf bound_ctrl:1"
            : "+v"(R[P + 1]), "+v"(R[P + 2])
            : "v"(l), "v"(R[P]), "n"(P + 1), "n"(P + 2));
        PivotDpp<P, P + 3>::updates(R, l);
      } else if constexpr (P + 1 < 16) {
        asm("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 0"
            : "+v"(R[P + 1])
            : "v"(l), "v"(R[P]), "n"(P + 1));
      }
    }
  };
  // Pivots P .. N-1 (N = 4, 8, 12 or 16; pivots past the real ones are rows of the identity: scale 1,
  // multipliers 0, exact -- a count fixed at compile time keeps the sequence free of branches)
  template <int P, int N>
  static __device__ __forceinline__ void pivots_dpp(float (&R)[16], float &dmin) {
    if constexpr (P < N) {
      const float d = readlane(R[P], P);
      dmin = fminf(dmin, d);
      float rs = __builtin_amdgcn_rsqf(d);
#if defined(YCNR_PIVOT_PROBE) && YCNR_PIVOT_PROBE == 3  // devtest: sixteen wait states between the v_rsq and the first use of its result
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(rs));
#endif
      R[P] *= rs;  // L[i][P] in groups 0 / 2, Linv[P][c] in groups 1 / 3
      float l = R[P], b = R[P];
      // l.row1 <-> b.row0, l.row3 <-> b.row2: l = L[j][P] in lane j of every 16-lane row (b is scratch)
#if defined(YCNR_PIVOT_PROBE) && YCNR_PIVOT_PROBE == 1  // devtest (the 7-block dual class at two waves per SIMD): generous wait states around the swap
      asm("s_nop 7\n\ts_nop 7\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 7\n\ts_nop 7" : "+v"(l), "+v"(b));
#elif defined(YCNR_PIVOT_PROBE) && YCNR_PIVOT_PROBE == 2  // devtest: nothing of the compiler's scheduled between the statements of a pivot
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(l), "+v"(b));
      __builtin_amdgcn_sched_barrier(0);
#elif defined(YCNR_PIVOT_PROBE) && YCNR_PIVOT_PROBE == 4  // devtest: the same l without v_permlane16_swap (lane j of rows 1 / 3 from rows 0 / 2 through the LDS crossbar)
      l = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)((threadIdx.x & ~16u) << 2), __builtin_bit_cast(int, l)));
      asm volatile("s_nop 1" : "+v"(l), "+v"(b));
#else
      asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(l), "+v"(b));
#endif
      PivotDpp<P, P + 1>::first_two(R, l);
#if defined(YCNR_PIVOT_PROBE) && YCNR_PIVOT_PROBE == 1
      asm volatile("s_nop 7" ::: "memory");
#elif defined(YCNR_PIVOT_PROBE) && YCNR_PIVOT_PROBE == 2
      __builtin_amdgcn_sched_barrier(0);
#endif
      pivots_dpp<P + 1, N>(R, dmin);
    }
  }

  // Solves (A + lam I) x = b.  In: upper tiles of A in acc, per-lane-group partials of b in
  // bacc.  Out: xcol[cb] = x[16 cb + c] in every lane group; returns true when a real pivot
  // was not positive.  acc is destroyed.
  // BATCH: read all multipliers of a pivot before its updates (fewer s_nop between v_readlane and
  // the v_fma that uses its SGPR).  Faster in the row kernels (8.3 -> 8.1 ms); in the dual-form
  // kernels it was slower while their Gramian ran on the float32 pipe (7.2 -> 7.4 ms) and is
  // faster since it runs on the bf16 pipe (6.78 -> 6.54 ms), so the caller chooses.
  // kd (< 0: k): columns kd .. k - 1 are the zero padding of a padded upload (StepArgs::kReal) and take a unit diagonal like
  // the columns past k (with lambda n there, userFactReg = 0 -- which ycnr_als_create accepts -- left a zero pivot: NaN rows)
  template <bool BATCH = false>
  static __device__ __forceinline__ bool solve(acc_t (&acc)[NT], const float (&bacc)[NB], float *S, int k, float lam,
                                               float (&xcol)[NB], int lane, int kd = -1, [[maybe_unused]] unsigned *tr = nullptr) {
    const int g = lane >> 4, c = lane & 15;
    if (kd < 0) kd = k;
    float *Dt = S;             // D image, [row][col], row stride LDW
    float *Wt = S + 16 * LDW;  // W image stored transposed: Wt[col][row] = W[row][col]
    // right-hand side as per-lane-group partials: b[16 cb + c] = sum over the four groups of bpart[cb].
    // The updates b_bj -= U[J][bj]^T z_J subtract each group's share of the product; the sum over the
    // groups is taken once per block, when the block becomes the pivot block (one cross-group sum per
    // block instead of one per panel tile).
    float bpart[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) bpart[cb] = bacc[cb];
    // A += lam I on the real diagonal, 1 on the padded one (rows/cols >= k are otherwise 0)
    {
      const bool mine = (c >> 2) == g;
      const int t0 = c & 3;
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) {
        const float add = (bi * 16 + c < kd) ? lam : 1.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[tile_index(bi, bi, NB)][t] += (mine && t == t0) ? add : 0.0f;
      }
    }
    float dmin = 3.0e38f;
    float zrow[NB][4];  // z in row form: zrow[J][t] = z[16 J + 4 g + t] in every lane of group g
#pragma unroll
    for (int J = 0; J < NB; ++J) {
#ifdef YCNR_DUAL_TRACE  // devtest: when block step J began (100 MHz clock)
      if (tr && lane == 0 && J < 8) tr[8 + J] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
      // ---- 1. diagonal tile -> LDS -> lane-per-row Cholesky + inverse
      {
        const acc_t d = acc[tile_index(J, J, NB)];
#pragma unroll
        for (int t = 0; t < 4; ++t) Dt[(4 * g + t) * LDW + c] = d[t];
      }
      lds_sync();
      float R[16];
      {
        const bool xlane = (g & 1) != 0;  // groups 1 and 3 carry the identity, 0 and 2 carry D
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
      // NB >= 10 (dual rows of 145 .. 176 ratings at k > 128) keeps the v_readlane form: with the inline-asm
      // updates hipcc stopped using the accumulator half of the register file there (183 + 8 registers and
      // 1104 bytes of scratch per lane instead of 256 + 256 and none) and the class ran twice as long
      if constexpr (YCNR_DPP_PIVOTS && NB <= 9) {
        // padded pivots (index >= k; only the last tile has any) are skipped; one-tile systems run all 16
        // (see the v_readlane form below for why)
        if (NB > 1 && J == NB - 1) {
          // the last tile of a system with more than one: only the real pivots, in fours (wave-uniform)
          const int n4 = (k - J * 16 + 3) >> 2;
          if (n4 == 1) pivots_dpp<0, 4>(R, dmin);
          else if (n4 == 2) pivots_dpp<0, 8>(R, dmin);
          else if (n4 == 3) pivots_dpp<0, 12>(R, dmin);
          else pivots_dpp<0, 16>(R, dmin);
        } else {
          pivots_dpp<0, 16>(R, dmin);
        }
      } else {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          // padded pivots (index >= k) are rows of the identity: their scale is 1 and their
          // multipliers are 0, so skipping them is exact (a wave-uniform branch).  Only the last
          // tile has any: 16 (NB - 1) < k by the choice of NB (and of the dual class).
          // Not for NB = 1: there hipcc kept the pivot loop rolled because of the break (up to 120
          // trips of three s_set_gpr_idx moves each), and a guard per pivot made it copy R between
          // the guarded blocks; 16 straight-line pivots are cheaper than either.
          if constexpr (NB > 1) {
            if (J == NB - 1 && J * 16 + p >= k) break;
          }
          const float d = readlane(R[p], p);  // D[p][p] after the updates of pivots < p (lane p, group 0)
          // the smallest pivot decides whether the row is reported: the first pivot that is not
          // positive is an ordinary number (NaNs only appear after it), so a plain minimum keeps it
          dmin = fminf(dmin, d);
          // v_rsq_f32 as it is (1 ulp).  A Newton step on top of it (3 more instructions on the
          // critical path of each of the 16 pivots) changed nothing measurable: lanes 16-31 invert
          // the L that was actually computed, and the row errors against float64 had the same
          // median / p99 / max with and without it (tests/tools/errstats.py), 0.37 ms per MAL iteration.
          const float rs = __builtin_amdgcn_rsqf(d);
          R[p] *= rs;                             // L[i][p] in lanes 0-15, Linv[p][c] in lanes 16-31
          if constexpr (BATCH) {
            float mult[16];
#pragma unroll
            for (int j = p + 1; j < 16; ++j) mult[j] = readlane(R[p], j);  // L[j][p]
#pragma unroll
            for (int j = p + 1; j < 16; ++j) R[j] = fmaf(-R[p], mult[j], R[j]);
          } else {
#pragma unroll
            for (int j = p + 1; j < 16; ++j) {
              const float s = readlane(R[p], j);  // L[j][p]
              R[j] = fmaf(-R[p], s, R[j]);
            }
          }
        }
      }
      // ---- 2. W = L^-1 (column c in lanes 16-31) -> LDS -> C/D layout and A-operand layout
      if (g == 1) {
        float4 *dst = reinterpret_cast<float4 *>(Wt + c * LDW);
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) dst[m4] = float4{R[4 * m4], R[4 * m4 + 1], R[4 * m4 + 2], R[4 * m4 + 3]};
      }
      lds_sync();
      acc_t W;  // W[4g+t][c]
      {
        const float4 v = *reinterpret_cast<const float4 *>(Wt + c * LDW + 4 * g);
        W = acc_t{v.x, v.y, v.z, v.w};
      }
      // A operand of MFMA q: W[i = c][kk = 4g + q].  The contraction index of the four MFMAs of a tile
      // product is split as kk = 4g + q (lane group g, MFMA q) instead of the natural 4q + g: any split
      // works as long as both operands use it, and with this one register q of a tile in C/D layout
      // (row 4g + q, column c) already IS the B operand of MFMA q -- the panel and the trailing update
      // need no register <-> lane-group transposes (8 v_permlane swaps with their wait states per tile
      // before).  Two lane groups share a bank pair here: conflict-free in each half of the wave.
      float Aop[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) Aop[q] = Wt[(4 * g + q) * LDW + c];
      acc[tile_index(J, J, NB)] = W;
      // ---- 5a. z_J = W b_J, row form
      {
        const float bcolJ = group_sum(bpart[J]);  // b[16 J + c] after the updates of the block steps before
#ifdef YCNR_DUAL_TRACE  // devtest: the right-hand side block as step J found it
        if (tr && g == 0 && J < 7) reinterpret_cast<float *>(tr)[256 + 16 * J + c] = bcolJ;
        if (tr && g == 1 && J < 7) reinterpret_cast<float *>(tr)[384 + 16 * J + c] = bpart[J];
#endif
#pragma unroll
        for (int t = 0; t < 4; ++t) zrow[J][t] = W[t] * bcolJ;
        row_sum4(zrow[J][0], zrow[J][1], zrow[J][2], zrow[J][3]);
      }
      // ---- 3. panel tiles and 5b. rhs update
#pragma unroll
      for (int bj = J + 1; bj < NB; ++bj) {
        const acc_t T = acc[tile_index(J, bj, NB)];
        acc_t P = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 4; ++q) P = Tr::mma(Aop[q], T[q], P);
        acc[tile_index(J, bj, NB)] = P;  // U[J][bj]
        float s = bpart[bj];
        s = fmaf(-P[0], zrow[J][0], s);
        s = fmaf(-P[1], zrow[J][1], s);
        s = fmaf(-P[2], zrow[J][2], s);
        s = fmaf(-P[3], zrow[J][3], s);
#ifdef YCNR_SOLVE_NO_PK_FMA  // devtest (dual7): the updates of two blocks' right-hand sides must not pair into v_pk_fma_f32
        asm volatile("" : "+v"(s));
#endif
        bpart[bj] = s;
      }
      // ---- 4. trailing update: T[bi][bj] -= U[J][bi]^T U[J][bj], operands straight from the panel tiles
#pragma unroll
      for (int bi = J + 1; bi < NB; ++bi) {
        const acc_t Pi = acc[tile_index(J, bi, NB)];
        float nA[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) nA[q] = -Pi[q];
#pragma unroll
        for (int bj = bi; bj < NB; ++bj) {
          const acc_t Pj = acc[tile_index(J, bj, NB)];
          acc_t t = acc[tile_index(bi, bj, NB)];
#pragma unroll
          for (int q = 0; q < 4; ++q) t = Tr::mma(nA[q], Pj[q], t);
          acc[tile_index(bi, bj, NB)] = t;
        }
      }
    }
#ifdef YCNR_DUAL_TRACE  // devtest: z = L^-1 b as the forward elimination left it
    if (tr && c == 0) {
#pragma unroll
      for (int J = 0; J < NB && J < 7; ++J)
#pragma unroll
        for (int t = 0; t < 4; ++t) reinterpret_cast<float *>(tr)[16 + 16 * J + 4 * g + t] = zrow[J][t];
    }
    if (tr && lane == 0) tr[15] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
    // ---- back substitution: x_J = W_J^T (z_J - sum_{bj > J} U[J][bj] x_bj)
#pragma unroll
    for (int J = NB - 1; J >= 0; --J) {
      float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int bj = J + 1; bj < NB; ++bj) {
        const acc_t u = acc[tile_index(J, bj, NB)];
#pragma unroll
        for (int t = 0; t < 4; ++t) part[t] = fmaf(u[t], xcol[bj], part[t]);
      }
      const acc_t W = acc[tile_index(J, J, NB)];
      float s = 0.0f;
      if (J + 1 < NB) row_sum4(part[0], part[1], part[2], part[3]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float y = (J + 1 < NB) ? zrow[J][t] - part[t] : zrow[J][t];
        s = fmaf(W[t], y, s);
      }
      xcol[J] = group_sum(s);
    }
#ifdef YCNR_DUAL_TRACE  // devtest: the solution itself
    if (tr && g == 0) {
#pragma unroll
      for (int J = 0; J < NB && J < 7; ++J) reinterpret_cast<float *>(tr)[128 + 16 * J + c] = xcol[J];
    }
#endif
    // NaN / Inf in the input never shows as a small pivot (fminf drops NaNs) but ends up in x
    float chk = 0.0f;
#pragma unroll
    for (int J = 0; J < NB; ++J) chk = fmaf(xcol[J], 0.0f, chk);
    return !(dmin > 0.0f) || __any(!(chk == 0.0f));
  }

  // k = 16 (NB - 1) + 4 (k = 100, 20, 36, ...): the last block has four real columns.  Instead of carrying
  // them through every block step as a 16-wide tile column (108 of the 308 float32 MFMAs at NB = 7, one
  // more diagonal tile and six panel tiles), they are eliminated FIRST:
  //     A = [A11 A12; A21 A22],  A22 + lam I = C C^T (4 x 4),  V = C^-1 A21 (4 x 16 (NB-1)),  z2 = C^-1 b2
  //     (A11 + lam I - V^T V) x1 = b1 - V^T z2      -- a rank-4 update: ONE MFMA of K = 4 per tile
  //     x2 = C^-T (z2 - V x1)
  // The 4 x 4 factor and its inverse are computed redundantly in every lane from v_readlane values; A21
  // reaches the MFMA operand layout (lane (g, c): V[g][16 bi + c]) through a 256-byte LDS image per block.
  // The (NB - 1)-block system then goes through the ordinary solve.  Cholesky with another elimination
  // order: the same arithmetic class and error bound as the plain form.
  // kd (< 0: all four real): edge columns 16 (NB - 1) + j >= kd are zero padding (a padded upload, k = 97 .. 99 -> 100): unit diagonal
  static __device__ __forceinline__ bool solve_edge4(acc_t (&acc)[NT], const float (&bacc)[NB], float *S, float lam,
                                                     float (&xcol)[NB], int lane, int kd = -1) {
    static_assert(NB >= 2 && (NB - 1) * 64 * sizeof(float) <= lds_bytes(), "edge image must fit the solver's LDS");
    using Inner = SolveMfmaF32<NB - 1, WSYNC>;
    constexpr int L = NB - 1;  // the edge block
    const int g = lane >> 4, c = lane & 15;
    // ---- A22 + lam I and b2, wave-uniform
    const acc_t e = acc[tile_index(L, L, NB)];  // element (row 4g + t, col c) in reg t: rows 0..3 <-> lanes 0..15
    const float lam1 = (kd < 0 || 16 * L + 1 < kd) ? lam : 1.0f, lam2 = (kd < 0 || 16 * L + 2 < kd) ? lam : 1.0f,
                lam3 = (kd < 0 || 16 * L + 3 < kd) ? lam : 1.0f;  // (column 16 L is always a factor: at most three columns of padding)
    const float a00 = readlane(e[0], 0) + lam;
    const float a10 = readlane(e[1], 0), a11 = readlane(e[1], 1) + lam1;
    const float a20 = readlane(e[2], 0), a21 = readlane(e[2], 1), a22 = readlane(e[2], 2) + lam2;
    const float a30 = readlane(e[3], 0), a31 = readlane(e[3], 1), a32 = readlane(e[3], 2), a33 = readlane(e[3], 3) + lam3;
    const float bs = group_sum(bacc[L]);
    const float b0 = readlane(bs, 0), b1 = readlane(bs, 1), b2 = readlane(bs, 2), b3 = readlane(bs, 3);
    // ---- C = chol(A22) (lower), M = C^-1
    const float r0 = __builtin_amdgcn_rsqf(a00);
    const float l10 = a10 * r0, l20 = a20 * r0, l30 = a30 * r0;
    const float d1 = fmaf(-l10, l10, a11);
    const float r1 = __builtin_amdgcn_rsqf(d1);
    const float l21 = fmaf(-l20, l10, a21) * r1, l31 = fmaf(-l30, l10, a31) * r1;
    const float d2 = fmaf(-l21, l21, fmaf(-l20, l20, a22));
    const float r2 = __builtin_amdgcn_rsqf(d2);
    const float l32 = fmaf(-l31, l21, fmaf(-l30, l20, a32)) * r2;
    const float d3 = fmaf(-l32, l32, fmaf(-l31, l31, fmaf(-l30, l30, a33)));
    const float r3 = __builtin_amdgcn_rsqf(d3);
    const float dmin = fminf(fminf(a00, d1), fminf(d2, d3));
    const float m10 = -(l10 * r0) * r1;
    const float m21 = -(l21 * r1) * r2;
    const float m32 = -(l32 * r2) * r3;
    const float m20 = -fmaf(l21, m10, l20 * r0) * r2;
    const float m31 = -fmaf(l32, m21, l31 * r1) * r3;
    const float m30 = -fmaf(l32, m20, fmaf(l31, m10, l30 * r0)) * r3;
    // z2 = M b2
    const float z0 = r0 * b0;
    const float z1 = fmaf(m10, b0, r1 * b1);
    const float z2 = fmaf(m20, b0, fmaf(m21, b1, r2 * b2));
    const float z3 = fmaf(m30, b0, fmaf(m31, b1, fmaf(m32, b2, r3 * b3)));
    // row g of M and z2[g] for this lane's group
    const float Mg0 = g == 0 ? r0 : g == 1 ? m10 : g == 2 ? m20 : m30;
    const float Mg1 = g == 0 ? 0.0f : g == 1 ? r1 : g == 2 ? m21 : m31;
    const float Mg2 = g < 2 ? 0.0f : g == 2 ? r2 : m32;
    const float Mg3 = g < 3 ? 0.0f : r3;
    const float zg = g == 0 ? z0 : g == 1 ? z1 : g == 2 ? z2 : z3;
    // ---- A21 through LDS: image[bi][row][c'] (4 floats per row of the block), then V in operand layout
    if (c < 4) {
#pragma unroll
      for (int bi = 0; bi < L; ++bi) {
        const acc_t t = acc[tile_index(bi, L, NB)];
#pragma unroll
        for (int q = 0; q < 4; ++q) S[bi * 64 + (4 * g + q) * 4 + c] = t[q];
      }
    }
    lds_sync();
    float V[L];
    acc_t in[Inner::NT];
    float bin[L];
#pragma unroll
    for (int bi = 0; bi < L; ++bi) {
      const float4 a = *reinterpret_cast<const float4 *>(S + bi * 64 + c * 4);
      V[bi] = fmaf(Mg3, a.w, fmaf(Mg2, a.z, fmaf(Mg1, a.y, Mg0 * a.x)));
      bin[bi] = fmaf(-V[bi], zg, bacc[bi]);  // this group's share of b1 - V^T z2
    }
    lds_sync();  // the image is read before the inner solve reuses S
    // ---- A11 - V^T V: one MFMA (K = 4: the four rows of V) per tile
#pragma unroll
    for (int bi = 0; bi < L; ++bi) {
      const float nv = -V[bi];
#pragma unroll
      for (int bj = bi; bj < L; ++bj) in[tile_index(bi, bj, L)] = Tr::mma(nv, V[bj], acc[tile_index(bi, bj, NB)]);
    }
    float x1[L];
    const bool bad = Inner::template solve<true>(in, bin, S, 16 * L, lam, x1, lane);
    // ---- x2 = M^T (z2 - V x1)
    float s = 0.0f;
#pragma unroll
    for (int bi = 0; bi < L; ++bi) {
      xcol[bi] = x1[bi];
      s = fmaf(V[bi], x1[bi], s);
    }
    const float yg = zg - row_sum(s);  // (z2 - V x1)[g] in every lane of group g
    const float y0 = readlane(yg, 0), y1 = readlane(yg, 16), y2 = readlane(yg, 32), y3 = readlane(yg, 48);
    const float x20 = fmaf(m30, y3, fmaf(m20, y2, fmaf(m10, y1, r0 * y0)));
    const float x21 = fmaf(m31, y3, fmaf(m21, y2, r1 * y1));
    const float x22 = fmaf(m32, y3, r2 * y2);
    const float x23 = r3 * y3;
    xcol[L] = c == 0 ? x20 : c == 1 ? x21 : c == 2 ? x22 : x23;
    return bad || !(dmin > 0.0f) || !(x20 * 0.0f == 0.0f) || !(x21 * 0.0f == 0.0f) || !(x22 * 0.0f == 0.0f) || !(x23 * 0.0f == 0.0f);
  }

  // E4: the caller launches this instantiation only for k = 16 (NB - 1) + 4 (the edge columns go first,
  // solve_edge4).  A compile-time choice: with both forms behind a run-time branch the fused row kernel
  // spilled 168 bytes per lane to scratch.
  template <bool E4 = false>
  static __device__ __forceinline__ void run(acc_t (&acc)[NT], const float (&bacc)[NB], float *S, int k, float lam,
                                             float *__restrict__ out_row, int row, ErrInfo *err, int lane, int kd = -1) {
    const int g = lane >> 4, c = lane & 15;
    float xcol[NB];
    bool bad;
    if constexpr (E4 && NB >= 2 && YCNR_EDGE4_SOLVE) {
      bad = solve_edge4(acc, bacc, S, lam, xcol, lane, kd);
    } else {
      bad = solve<true>(acc, bacc, S, k, lam, xcol, lane, kd);
    }
    // lane group g stores blocks g and g + 4: two full 256-byte stores per row for k >= 64
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      if ((cb & 3) == g && cb * 16 + c < k) out_row[cb * 16 + c] = xcol[cb];
    }
    if (bad && lane == 0) {
      atomicAdd(&err->count, 1);
      err->firstRow = row;
    }
  }
};

// The same register-resident block Cholesky for float64 (v_mfma_f64_16x16x4_f64).  The f64
// C/D layout puts row g + 4t in register t of lane group g, so the register with index q of a
// tile already IS the MFMA operand of contraction step q (rows 4q + g): the panel and the
// trailing update need no lane transposes at all.  Everything else mirrors SolveMfmaF32
// (two 32-bit readlanes per broadcast, 1/sqrt in double, shuffles instead of DPP).
template <int NB>
struct SolveMfmaF64 {
  using Tr = MfmaTraits<double>;
  using acc_t = typename Tr::acc_t;
  static constexpr int NT = tile_count(NB);
  static constexpr int LDW = 18;  // doubles per LDS image row (144 B: 16-byte aligned rows)
  static constexpr size_t lds_bytes() { return 2 * 16 * LDW * sizeof(double); }

  static __device__ __forceinline__ double row_sum(double v) {
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
  }
  static __device__ __forceinline__ double group_sum(double v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
  }
  static __device__ __forceinline__ double readlane(double v, int lane) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
  }

  static __device__ __forceinline__ bool solve(acc_t (&acc)[NT], const double (&bacc)[NB], double *S, int k, double lam,
                                               double (&xcol)[NB], int lane, int kd = -1) {
    const int g = lane >> 4, c = lane & 15;
    if (kd < 0) kd = k;
    double *Dt = S, *Wt = S + 16 * LDW;
    double bcol[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) bcol[cb] = group_sum(bacc[cb]);
    {
      // element (row, col) = (g + 4t, c) is on the diagonal when c % 4 == g, in register c / 4
      const bool mine = (c & 3) == g;
      const int t0 = c >> 2;
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) {
        const double add = (bi * 16 + c < kd) ? lam : 1.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[tile_index(bi, bi, NB)][t] += (mine && t == t0) ? add : 0.0;
      }
    }
    bool bad = false;
    double zrow[NB][4];  // z in row form: zrow[J][t] = z[16 J + g + 4 t]
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      {
        const acc_t d = acc[tile_index(J, J, NB)];
#pragma unroll
        for (int t = 0; t < 4; ++t) Dt[(g + 4 * t) * LDW + c] = d[t];
      }
      __syncthreads();
      double R[16];
      {
        const bool xlane = (g & 1) != 0;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          const double v = Dt[c * LDW + m];
          R[m] = xlane ? (c == m ? 1.0 : 0.0) : v;
        }
      }
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        if constexpr (NB > 1) {  // see SolveMfmaF32: the break keeps the NB = 1 loop rolled
          if (J * 16 + p >= k) break;
        }
        double d = readlane(R[p], p);
        if (!(d > 0.0)) {
          bad = true;
          d = 1.0;
        }
        const double rs = 1.0 / sqrt(d);
        R[p] *= rs;
#pragma unroll
        for (int j = p + 1; j < 16; ++j) {
          const double s = readlane(R[p], j);
          R[j] = fma(-R[p], s, R[j]);
        }
      }
      if (g == 1) {
#pragma unroll
        for (int m = 0; m < 16; ++m) Wt[c * LDW + m] = R[m];
      }
      __syncthreads();
      acc_t W;  // W[g + 4t][c]
#pragma unroll
      for (int t = 0; t < 4; ++t) W[t] = Wt[c * LDW + g + 4 * t];
      double Aop[4];  // A operand of MFMA q: W[i = c][kk = 4q + g]
#pragma unroll
      for (int q = 0; q < 4; ++q) Aop[q] = Wt[(4 * q + g) * LDW + c];
      acc[tile_index(J, J, NB)] = W;
#pragma unroll
      for (int t = 0; t < 4; ++t) zrow[J][t] = row_sum(W[t] * bcol[J]);
      acc_t Pt[NB > 1 ? NB - 1 : 1];  // panel tiles; register q is the operand of step q
#pragma unroll
      for (int bj = J + 1; bj < NB; ++bj) {
        const acc_t T = acc[tile_index(J, bj, NB)];
        acc_t P = acc_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) P = Tr::mma(Aop[q], T[q], P);
        acc[tile_index(J, bj, NB)] = P;
        Pt[bj - J - 1] = P;
        double s = P[0] * zrow[J][0];
        s = fma(P[1], zrow[J][1], s);
        s = fma(P[2], zrow[J][2], s);
        s = fma(P[3], zrow[J][3], s);
        bcol[bj] -= group_sum(s);
      }
#pragma unroll
      for (int bi = J + 1; bi < NB; ++bi) {
#pragma unroll
        for (int bj = bi; bj < NB; ++bj) {
          acc_t t = acc[tile_index(bi, bj, NB)];
#pragma unroll
          for (int q = 0; q < 4; ++q) t = Tr::mma(-Pt[bi - J - 1][q], Pt[bj - J - 1][q], t);
          acc[tile_index(bi, bj, NB)] = t;
        }
      }
    }
#pragma unroll
    for (int J = NB - 1; J >= 0; --J) {
      double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int bj = J + 1; bj < NB; ++bj) {
        const acc_t u = acc[tile_index(J, bj, NB)];
#pragma unroll
        for (int t = 0; t < 4; ++t) part[t] = fma(u[t], xcol[bj], part[t]);
      }
      const acc_t W = acc[tile_index(J, J, NB)];
      double s = 0.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double y = (J + 1 < NB) ? zrow[J][t] - row_sum(part[t]) : zrow[J][t];
        s = fma(W[t], y, s);
      }
      xcol[J] = group_sum(s);
    }
    return bad;
  }

  static __device__ __forceinline__ void run(acc_t (&acc)[NT], const double (&bacc)[NB], double *S, int k, double lam,
                                             double *__restrict__ out_row, int row, ErrInfo *err, int lane, int kd = -1) {
    const int g = lane >> 4, c = lane & 15;
    double xcol[NB];
    const bool bad = solve(acc, bacc, S, k, lam, xcol, lane, kd);
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      if ((cb & 3) == g && cb * 16 + c < k) out_row[cb * 16 + c] = xcol[cb];
    }
    if (bad && lane == 0) {
      atomicAdd(&err->count, 1);
      err->firstRow = row;
    }
  }
};

template <typename T, int NB, bool LDS_SOLVER>
struct SolverFor {
  using type = SolveLds<T, NB>;
};
template <int NB>
struct SolverFor<float, NB, false> {
  using type = SolveMfmaF32<NB>;
};
template <int NB>
struct SolverFor<double, NB, false> {
  using type = SolveMfmaF64<NB>;
};

template <typename T>
struct StepArgs {
  const Unit *units;
  const SplitRow *split;
  const int32_t *indx;
  const T *vals;
  const T *fixed;  // opposite side's factors, [fixedRows x k]
  const T *zeros;  // >= 128 zeros: the "row" gathered for ratings past a unit's end
  T *solved;       // this side's factors, [rows x k], rows written in place
  T *slabs;
  ErrInfo *err;
  double lambda;
  int32_t k;
  int32_t firstFused;  // units[0 .. firstFused) are split chunks, the rest whole rows
  int32_t firstDual;   // first unit of the dual-form launch in flight
  uint32_t fixedBytes; // size of the fixed matrix when it is below 4 GB (buffer loads), else 0
  int32_t kReal = 0;   // != 0: only the first kReal of the k columns are factors, the rest zero padding (unit diagonal)
  const unsigned short *planes = nullptr;  // the fixed matrix split into bf16 planes (GramX6P), when the half-step made one
  uint32_t planesBytes = 0;
};

// Kernel 1a: one wave per SPLIT unit -- gather + Gramian + rhs of a chunk of a heavy row,
// written as a partial slab.  Kept apart from the fused kernel so that its register
// allocation (accumulators + operand ring) is not inflated by the solve.
template <typename T, int NB, bool EDGE>
__global__ __launch_bounds__(64, sizeof(T) == 8 ? 1 : YCNR_SLAB_WAVES_PER_SIMD) void als_gram_slab_kernel(StepArgs<T> a) {
  using G = typename GramSel<T, NB, EDGE>::type;
  const int lane = threadIdx.x;
  const Unit u = a.units[blockIdx.x];
  typename G::State st;
  G::init(st);
#if YCNR_SLAB_PREFETCH > 0
  G::template accumulate_deep<YCNR_SLAB_PREFETCH>(st, a.indx, a.vals, a.fixed, a.zeros, a.k, u.beg, u.end, lane);
#else
  G::accumulate(st, a.indx, a.vals, a.fixed, a.zeros, a.k, u.beg, u.end, lane);
#endif
  G::store_slab(st, a.slabs + (int64_t)u.slab * (G::slab_regs() * 64) + lane);
}

// Kernel 1a': the chunk Gramian on the bf16 matrix cores with float32-equivalent products.
//
// Measured on MI355X: v_mfma_f32_16x16x4_f32 does not overlap with VALU work of the same
// SIMD (step time fits 35 cycles per MFMA + 4.1 per other instruction, additively), i.e.
// the float32 "matrix" rate is the vector ALU's.  The bf16 matrix pipe is separate and 16x
// faster, so each gathered float is split EXACTLY into three bf16 terms by truncation,
//     y = b1 + b2 + b3,   b1 = top 8 mantissa bits, b2 = next 8, b3 = last 8,
// and y_i * y_j is accumulated as the six products b1b1 + b1b2 + b2b1 + b1b3 + b3b1 + b2b2
// (each exact in the float32 accumulator; the dropped b2b3 + b3b2 + b3b3 are <= 2^-23 of the
// product).  The split costs ~6.5 VALU ops per value and runs beside the MFMAs.
// v_mfma_f32_16x16x32_bf16 contracts over 32 ratings: lane (g, c) holds, for column
// 16 cb + c, the values of ratings 8g .. 8g+7 -- again the same registers are the A operand of
// tile row cb and the B operand of tile column cb, and the C/D layout equals the float32
// MFMA's, so the slab format and everything downstream are unchanged.  Gathered rows are read
// with buffer loads: out-of-range offsets (ratings past the unit's end, padded columns)
// return 0 in hardware, no zero row and no selects.  Needs the fixed matrix < 2 GB.
template <int NB>
__global__ __launch_bounds__(64, 1) void als_gram_slab_x6_kernel(StepArgs<float> a) {
  using Tr = MfmaTraits<float>;
  using acc_t = typename Tr::acc_t;
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int NT = tile_count(NB);
  constexpr unsigned OOB = 0x80000000u;  // >= num_records (matrix < 2 GB) and cannot wrap when the block offset is added
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const Unit u = a.units[blockIdx.x];
  const int k = a.k;
  const unsigned rowBytes = (unsigned)k * 4u;
  const bool lastok = (NB - 1) * 16 + c < k;
  __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)a.fixed, 0, (int)a.fixedBytes, 0x00020000);
  acc_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
  float bacc[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) bacc[cb] = 0.0f;
  const int64_t n = u.end - u.beg;
  const int64_t nsteps = (n + 31) >> 5;
  // per step this lane owns ratings 32 s + 8 g + j, j = 0..7
  // Two 16-byte loads each for the 8 column ids and the 8 ratings (the arrays carry 64 bytes of
  // slack, so the vector loads of the last lanes stay inside the allocation).  Keeping the
  // count of loads in flight below the 6-bit vmcnt range matters: with 16 scalar loads here
  // (72 in flight together with the 56 gathers) the right-hand side came out wrong now and then.
  auto fetch = [&](int64_t s, unsigned (&off)[8], float (&r)[8]) {
    const int64_t q0 = (s << 5) + 8 * g;
    const int64_t qb = q0 < n ? q0 : 0;  // lanes wholly past the end read the first ratings, masked below
    const int4 i0 = *reinterpret_cast<const int4 *>(a.indx + u.beg + qb);
    const int4 i1 = *reinterpret_cast<const int4 *>(a.indx + u.beg + qb + 4);
    const float4 v0 = *reinterpret_cast<const float4 *>(a.vals + u.beg + qb);
    const float4 v1 = *reinterpret_cast<const float4 *>(a.vals + u.beg + qb + 4);
    const int32_t ids[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
    const float vs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool v = q0 + j < n;
      off[j] = v ? (unsigned)ids[j] * rowBytes + (unsigned)c * 4u : OOB;
      r[j] = v ? vs[j] : 0.0f;
    }
  };
  auto gather = [&](float (&raw)[NB][8], const unsigned (&off)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int cb = 0; cb < NB - 1; ++cb) raw[cb][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, off[j], cb * 64, 0));
      raw[NB - 1][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, lastok ? off[j] : OOB, (NB - 1) * 64, 0));
    }
  };
  // exact 3-way bf16 split of every value, packed in k order (pairs j, j+1 per register), and
  // b += y * r on the VALU from the unsplit values
  auto split = [&](const float (&raw)[NB][8], const float (&r)[8], u32x4 (&p1)[NB], u32x4 (&p2)[NB], u32x4 (&p3)[NB]) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bacc[cb] = fmaf(raw[cb][j], r[j], bacc[cb]);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float x0 = raw[cb][2 * jj], x1 = raw[cb][2 * jj + 1];
        const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
#ifdef YCNR_ABLATE_X6_SPLIT  // timing experiments only: no split, raw bits as operands
        p1[cb][jj] = u0; p2[cb][jj] = u1; p3[cb][jj] = u0 ^ u1;
        continue;
#endif
        p1[cb][jj] = __builtin_amdgcn_perm(u1, u0, 0x07060302);
        const float s0 = x0 - __builtin_bit_cast(float, u0 & 0xFFFF0000u);
        const float s1 = x1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
        const unsigned v0 = __builtin_bit_cast(unsigned, s0), v1 = __builtin_bit_cast(unsigned, s1);
        p2[cb][jj] = __builtin_amdgcn_perm(v1, v0, 0x07060302);
        const float t0 = s0 - __builtin_bit_cast(float, v0 & 0xFFFF0000u);
        const float t1 = s1 - __builtin_bit_cast(float, v1 & 0xFFFF0000u);
        p3[cb][jj] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, t1), __builtin_bit_cast(unsigned, t0), 0x07060302);
      }
    }
  };
  auto mma6 = [&](const u32x4 (&p1)[NB], const u32x4 (&p2)[NB], const u32x4 (&p3)[NB]) {
#ifdef YCNR_ABLATE_X6_MMA  // timing experiments only: keep the split results alive, skip the products
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) asm volatile("" ::"v"(p1[cb]), "v"(p2[cb]), "v"(p3[cb]));
    return;
#endif
    // product type outermost: consecutive MFMAs update different tiles, so no MFMA waits for
    // the previous one's accumulator (smallest terms still first per tile)
#pragma unroll
    for (int term = 0; term < 6; ++term) {
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) {
#pragma unroll
        for (int bj = bi; bj < NB; ++bj) {
          const u32x4 &pa = term == 0 ? p2[bi] : (term == 1 || term == 3 || term == 5) ? p1[bi] : (term == 2 ? p3[bi] : p2[bi]);
          const u32x4 &pb = term == 0 ? p2[bj] : term == 1 ? p3[bj] : term == 2 ? p1[bj] : term == 3 ? p2[bj] : p1[bj];
          acc[tile_index(bi, bj, NB)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, pb), acc[tile_index(bi, bj, NB)], 0, 0, 0);
        }
      }
    }
  };
  // Software pipeline, two steps per trip so that no register set is copied:
  //   matrix pipe: products of step s        (from p / q, split during step s - 1)
  //   VALU:        split + rhs of step s + 1 (from the gather issued during step s - 1)
  //   memory:      gather of step s + 2, column ids and ratings of step s + 3
  // A gather is issued only after the previous one has been consumed, so at most 56 + 4 loads
  // are in flight (see fetch).
  unsigned offG[8];
  float rG[8], rX[8], rY[8];
  float rawX[NB][8], rawY[NB][8];
  u32x4 p1[NB], p2[NB], p3[NB], q1[NB], q2[NB], q3[NB];
  auto keep = [](float (&dst)[8], const float (&src)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = src[j];
  };
  fetch(0, offG, rG);
  gather(rawX, offG);
  keep(rX, rG);
  fetch(1, offG, rG);
  split(rawX, rX, p1, p2, p3);  // step 0 (waits for its gather)
  gather(rawY, offG);           // step 1
  keep(rY, rG);
  fetch(2, offG, rG);
  for (int64_t s = 0; s < nsteps; s += 2) {
    split(rawY, rY, q1, q2, q3);  // step s + 1
    gather(rawX, offG);           // step s + 2
    keep(rX, rG);
    fetch(s + 3, offG, rG);
    mma6(p1, p2, p3);             // step s
    split(rawX, rX, p1, p2, p3);  // step s + 2
    gather(rawY, offG);           // step s + 3
    keep(rY, rG);
    fetch(s + 4, offG, rG);
    mma6(q1, q2, q3);             // step s + 1 (all zero when s + 1 == nsteps)
  }
  // same slab layout as GramPlain: [tile][reg][lane], then NB rhs partials
  // plain slab layout (Gram<float, NB>::store_slab): 16 bytes per lane and tile, then the rhs partials
  float *sl = a.slabs + (int64_t)u.slab * slab_elems(NB) + lane;
  float4 *sq = reinterpret_cast<float4 *>(a.slabs + (int64_t)u.slab * slab_elems(NB)) + lane;
#pragma unroll
  for (int t = 0; t < NT; ++t) sq[t * 64] = float4{acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) sl[(NT * 4 + cb) * 64] = bacc[cb];
}

// ds_bpermute the optimizer cannot see through.  hipcc (ROCm 7.2) folds
//   v = (c & 3) == t ? __builtin_amdgcn_ds_bpermute(a, x[t]) : v,  t = 0..3
// into a single bpermute of x[0] (the select is moved across the lane exchange), the same class of
// miscompile as the permlane swaps in transpose_rg.
// v may be fresh out of an MFMA: hipcc pads no hazard for an asm operand, so the statement opens
// with the 12 wait states an MFMA result needs before a non-MFMA reader.
template <typename V4>
__device__ __forceinline__ void bpermute4_opaque(int byteAddr, const V4 &v, float (&r)[4]) {
  asm volatile("s_nop 7\n\ts_nop 3\n\tds_bpermute_b32 %0, %4, %5\n\tds_bpermute_b32 %1, %4, %6\n\t"
               "ds_bpermute_b32 %2, %4, %7\n\tds_bpermute_b32 %3, %4, %8\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
               : "v"(byteAddr), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
}
__device__ __forceinline__ float bpermute_opaque(int byteAddr, float v) {
  float r;
  asm volatile("s_nop 7\n\ts_nop 3\n\tds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(byteAddr), "v"(v));
  return r;
}

// The bf16x6 Gramian with the gather staged through LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`),
// two waves per SIMD.
//
// Why two waves: measured with devtest/mfmarate.hip, one wave cannot hide vector work behind
// v_mfma_f32_16x16x32_bf16 -- an MFMA blocks its own wave's issue for the 16 cycles it occupies
// the matrix pipe, less one slot (19.6 cycles per MFMA bare, 27.9 with 3 VALU instructions
// each, 36.9 with 6).  Two waves of a SIMD do overlap: the pair sustains an MFMA every 16
// cycles with 2 VALU instructions each, every 17 with 3.  The split needs ~2 per MFMA.
//
// Registers (<= 256 per lane for two waves): the 3 x NB operand registers are single-buffered.
// Tiles are visited row by row (bi, bj >= bi), so after row bi nothing reads block bi again and
// the NEXT step's block bi is split into the same registers while rows bi+1.. still multiply.
// The gathered floats wait in LDS, not in registers (at NB = 7 accumulators 112 + operands 84
// leave no room for a register ring): a whole step (32 ratings x 16 NB columns, NB KiB per
// wave) is in flight while the previous one is multiplied.
//
// PADRHS (k < 16 NB): the ratings ride in the first unused column of the last block, so
// b = Y^T r comes out of the MFMAs as column k of the padded Gramian: no VALU multiply-adds and
// no accumulators for it.  It is moved to the slab's b-partials at the end and the padded
// entries are zeroed, so the slab is indistinguishable from the other kernels'.
//
//   * One DMA instruction fills one 1-KiB slot (b, h): 16 ratings x the 16 columns of block b.
//     Lane l fetches 16 bytes: columns 16 b + 4 (l & 3) .. +3 of rating
//     16 h + 8 ((l >> 2) & 1) + (l >> 3).  With this order the MFMA operand lane (g, c) finds
//     rating 8 g + j of its column at dword (2 j + (g & 1)) 16 + c of slot (b, g >> 1):
//     four conflict-free ds_read2_b32 with one address register per block.  14 DMA + 28 LDS reads
//     per step replace 56 dword loads with 8 address registers.
//   * Lanes whose rating is past the unit's end, or whose column quad is past the matrix, carry
//     an out-of-range offset: the DMA writes zeros for them (devtest/dmaprobe.hip).
//   * Column ids and ratings of a step arrive by DMA too (64 + 64 dwords per step: all lanes
//     load, the upper half repeats the next step's; a ring of 4 steps), so the loop contains
//     no load the compiler counts: it would wait vmcnt(0) for such a load and drain the ring.
//     All LDS reads are inline asm for the same reason; the waits are counted by hand.  Every
//     phase issues 2 + 2 NB vector-memory operations in a fixed order, so "the two slots of
//     block b have landed" is vmcnt(2 NB) at every b.  (These two DMAs must not sit under a
//     lane mask: with `if (lane < 32)` around them one instantiation read stale ratings in
//     about half of its launches.)
//   * A slot is refilled only after its reads have returned (lgkmcnt(0)), and the wave drains
//     its DMA before it ends: LDS is handed to the next workgroup when the wave retires.
// Needs k % 4 == 0 (16-byte aligned rows) and the fixed matrix < 2 GB.  Results are bitwise
// those of als_gram_slab_x6_kernel for the tiles; with PADRHS b comes out of the MFMAs.
//
// PK3 (k = 16 (NB - 1) + 4 with PADRHS, e.g. k = 20, 100: round 3).  The last block then has five live
// columns (four of Y and the ratings), so its three bf16 planes fit ONE operand: lanes 0..4 keep the high
// terms, lanes 5..9 take the middle terms of columns 0..4 and lanes 10..14 the low ones (two v_or_b32_dpp
// row_shr per register).  A tile of the last block column is then three MFMAs -- A_l, A_m, A_h against the
// packed operand: all nine products -- instead of six, 147 MFMAs per 32 ratings instead of 168 at k = 100;
// its accumulator holds three partial columns per live column, folded (row_shl 5 / 10) once per row in
// extract_rhs.  The chip runs these kernels at its package power limit, so an eighth
// fewer MFMAs is an eighth less of what bounds them.
typedef __attribute__((address_space(3))) void *lds_void_ptr;

template <int NB, bool PADRHS, bool PK3 = false>
struct GramX6D {
  static_assert(!PK3 || PADRHS, "the packed last block needs the padded form");
  using acc_t = typename MfmaTraits<float>::acc_t;
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  static constexpr int NT = tile_count(NB);
  static constexpr int NH = 2 * NB;
  static constexpr int META = NH * 256;           // dword index of the ids / ratings ring
  static constexpr int LDS_DWORDS = META + 4 * 128;  // per step 64 ids + 64 ratings (the upper halves repeat the next step's)
  static constexpr unsigned OOB = 0x80000000u;

  // acc += Y^T Y over ratings [beg, beg + n) of indx / vals; PADRHS: column k of the padded
  // Gramian accumulates b, else bacc does (per lane-group partials, as Gram<>).  `lds` is this
  // wave's LDS_DWORDS dwords.
  static __device__ __forceinline__ void accumulate(acc_t (&acc)[NT], float (&bacc)[NB], unsigned *lds,
                                                    const int32_t *indx, const float *vals, const float *fixed,
                                                    uint32_t fixedBytes, int k, int64_t beg, int64_t n, int lane) {
    const int g = lane >> 4, c = lane & 15;
    const int kr = k - 16 * (NB - 1);
    const unsigned rowBytes = (unsigned)k * 4u;
    const unsigned n32 = (unsigned)(n < 0x3fffffff ? n : 0x3fffffff);
    const int64_t nsteps = (n + 31) >> 5;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)fixed, 0, (int)fixedBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srdI = __builtin_amdgcn_make_buffer_rsrc((void *)(indx + beg), 0, (int)(n32 * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t srdR = __builtin_amdgcn_make_buffer_rsrc((void *)(vals + beg), 0, (int)(n32 * 4u), 0x00020000);
    const unsigned ldsBase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)lds;
    // DMA side: this lane's rating inside a half (16 ratings) and its column quad
    const unsigned rhoL = 8u * ((lane >> 2) & 1) + (lane >> 3);
    const unsigned quadBytes = (unsigned)(lane & 3) * 16u;
    // operand side: byte address of (rating 8 g, column c) in slot (0, g >> 1)
    const unsigned rdBase = ldsBase + (unsigned)(g >> 1) * 1024u + (unsigned)((g & 1) * 16 + c) * 4u;

    u32x4 p1[NB], p2[NB], p3[NB];
    [[maybe_unused]] u32x4 pk = {0u, 0u, 0u, 0u};  // PK3: the three planes of the last block's five live columns in one operand
    unsigned offH[2];
    float rr[8];

    auto meta_dma = [&](int64_t t) {  // ids and ratings of step t (and of t + 1 behind them) -> ring slot t & 3
      unsigned *slot = lds + META + ((int)t & 3) * 128;
      const unsigned qb = ((unsigned)t << 7) + (unsigned)lane * 4u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdI, (lds_void_ptr)slot, 4, qb, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdR, (lds_void_ptr)(slot + 64), 4, qb, 0, 0, 0);
    };
    auto read_offsets = [&](int64_t t) {  // gather offsets of step t for this lane's two DMA ratings
      const unsigned ma = ldsBase + (unsigned)(META + ((int)t & 3) * 128) * 4u + rhoL * 4u;
      unsigned id0, id1;
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(id0), "=&v"(id1) : "v"(ma) : "memory");
      const unsigned q0 = ((unsigned)t << 5) + rhoL;
      offH[0] = q0 < n32 ? id0 * rowBytes + quadBytes : OOB;
      offH[1] = q0 + 16u < n32 ? id1 * rowBytes + quadBytes : OOB;
    };
    auto read_r = [&](int64_t t, float (&r)[8]) {  // ratings 8 g .. 8 g + 7 of step t
      const unsigned ra = ldsBase + (unsigned)(META + ((int)t & 3) * 128 + 64) * 4u + (unsigned)g * 32u;
      f32x4 r0, r1;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(r0), "=&v"(r1) : "v"(ra) : "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) { r[j] = r0[j]; r[4 + j] = r1[j]; }
    };
    auto gather_block = [&](int b) {  // both halves of block b of the step offH[] belongs to
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_void_ptr)(lds + (2 * b) * 256), 16, offH[0], b * 64, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_void_ptr)(lds + (2 * b + 1) * 256), 16, offH[1], b * 64, 0, 0);
    };
    auto mma_row = [&](int bi) {
      constexpr int NBF = PK3 ? NB - 1 : NB;  // block columns multiplied plane by plane
#pragma unroll
      for (int term = 0; term < 6; ++term) {
#pragma unroll
        for (int bj = bi; bj < NBF; ++bj) {
          const u32x4 &pa = term == 0 ? p2[bi] : (term == 1 || term == 3 || term == 5) ? p1[bi] : (term == 2 ? p3[bi] : p2[bi]);
          const u32x4 &pb = term == 0 ? p2[bj] : term == 1 ? p3[bj] : term == 2 ? p1[bj] : term == 3 ? p2[bj] : p1[bj];
          acc[tile_index(bi, bj, NB)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, pb), acc[tile_index(bi, bj, NB)], 0, 0, 0);
        }
      }
      if constexpr (PK3) {  // the last block column against the packed operand, smallest terms first
        acc_t &t = acc[tile_index(bi, NB - 1, NB)];
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, p3[bi]), __builtin_bit_cast(bf16x8, pk), t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, p2[bi]), __builtin_bit_cast(bf16x8, pk), t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, p1[bi]), __builtin_bit_cast(bf16x8, pk), t, 0, 0, 0);
      }
    };
    const float padOne = (PADRHS && c == kr) ? 1.0f : 0.0f;
    const bool lastLive = (NB - 1) * 16 + c < k;
    // One phase: products of step s beside the split of step s + 1 and the DMA of step s + 2.
    auto phase = [&](int64_t s, auto MMA_, auto SPLIT_) {
      constexpr bool MMA = decltype(MMA_)::value, SPLIT = decltype(SPLIT_)::value;
      if constexpr (SPLIT) {
        meta_dma(s + 4);       // into the slot of step s, whose last reader was phase s - 1
        read_offsets(s + 2);   // issued two phases ago: behind every vmcnt wait of phase s - 1
        if constexpr (!PADRHS) read_r(s + 1, rr);
      }
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) {
        f32x2 x01, x23, x45, x67;
        if constexpr (SPLIT) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NH) : "memory");  // block bi of step s + 1 has landed
          const unsigned ra = rdBase + (unsigned)bi * 2048u;
#ifdef YCNR_X6D_PLAIN_LDS  // experiment on the packed-rhs failure: LDS loads the compiler counts itself
          {
            typedef __attribute__((address_space(3))) const float lds_float;
            lds_float *sp = (lds_float *)(uintptr_t)ra;
            x01 = f32x2{sp[0], sp[32]};
            x23 = f32x2{sp[64], sp[96]};
            x45 = f32x2{sp[128], sp[160]};
            x67 = f32x2{sp[192], sp[224]};
          }
#else
          asm volatile("ds_read2_b32 %0, %4 offset1:32\n\tds_read2_b32 %1, %4 offset0:64 offset1:96\n\t"
                       "ds_read2_b32 %2, %4 offset0:128 offset1:160\n\tds_read2_b32 %3, %4 offset0:192 offset1:224"
                       : "=&v"(x01), "=&v"(x23), "=&v"(x45), "=&v"(x67) : "v"(ra) : "memory");
#endif
        }
        if constexpr (MMA) mma_row(bi);
        if constexpr (SPLIT) {
#ifdef YCNR_X6D_PK_NOP  // experiment on the packed-rhs failure: idle cycles between the wait and the first use
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(x01), "+v"(x23), "+v"(x45), "+v"(x67)::"memory");
#else
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x01), "+v"(x23), "+v"(x45), "+v"(x67)::"memory");
#endif
          gather_block(bi);  // step s + 2 into the slots just read
          float x[8] = {x01[0], x01[1], x23[0], x23[1], x45[0], x45[1], x67[0], x67[1]};
          if (bi == NB - 1) {
            if constexpr (PADRHS) {
              float rv[8];
              read_r(s + 1, rv);
#pragma unroll
              for (int j = 0; j < 8; ++j) x[j] = lastLive ? x[j] : (padOne != 0.0f ? rv[j] : 0.0f);  // (not rv * 0: a negative rating would leave -0,
                                                                                                     // whose sign bit the packed operand of PK3 ORs into another column's term)
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) x[j] = lastLive ? x[j] : 0.0f;
            }
          }
          if constexpr (!PADRHS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bacc[bi] = fmaf(x[j], rr[j], bacc[bi]);
            // Keeps the compiler from pairing the multiply-adds of two blocks into v_pk_fma_f32
            // (packed float32 beside MFMAs is slower anyway).  With that pairing, and only with >= 4
            // workgroups per CU, the FIRST block of each pair (whose values are copied and kept until
            // its partner arrives) got a wrong b in lanes 48..63 -- tiles right, b wrong in ~7 % of
            // rows; devtest/x6many.hip built with -DYCNR_X6D_ALLOW_PK reproduces it and counts the
            // failures per block.  Ruled out: late LDS data (a sentinel read behind the eight values
            // never arrived late in 2.5 M reads, and ~20 MFMAs lie between read and wait), the two
            // small DMAs (moving them changes nothing).  The cause is open.
#ifndef YCNR_X6D_ALLOW_PK
            asm volatile("" : "+v"(bacc[bi]));
#endif
          }
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
          p1[bi] = u32x4{h[0], h[1], h[2], h[3]};
          p2[bi] = u32x4{m[0], m[1], m[2], m[3]};
          p3[bi] = u32x4{l[0], l[1], l[2], l[3]};
          if constexpr (PK3) {
            if (bi == NB - 1) {  // columns > kr hold zeros in every plane, so the shifted planes land on empty lanes
#pragma unroll
              for (int e = 0; e < 4; ++e)
                pk[e] = h[e] | (unsigned)__builtin_amdgcn_update_dpp(0, (int)m[e], 0x115, 0xF, 0xF, true) |  // row_shr:5
                        (unsigned)__builtin_amdgcn_update_dpp(0, (int)l[e], 0x11A, 0xF, 0xF, true);           // row_shr:10
            }
          }
        }
      }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    // ids / ratings of steps 0..2, then the whole of step 0 into the ring (same issue order as
    // a phase: 2 NB gathers last, so phase(-1) can use the same counted waits)
    meta_dma(0);
    meta_dma(1);
    meta_dma(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    read_offsets(0);
#pragma unroll
    for (int b = 0; b < NB; ++b) gather_block(b);
    phase(-1, no{}, yes{});
    for (int64_t s = 0; s + 1 < nsteps; ++s) phase(s, yes{}, yes{});
    phase(nsteps - 1, yes{}, no{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this wave may land in LDS after it ends
  }

  // PADRHS: move b out of column kr of the padded Gramian into the slab's b-partials (lane
  // (0, c') carries b[16 cb + c'], the other lane groups 0) and zero the padded entries.
  static __device__ __forceinline__ void extract_rhs(acc_t (&acc)[NT], float (&bacc)[NB], int k, int lane) {
    const int g = lane >> 4, c = lane & 15;
    const int kr = k - 16 * (NB - 1);
    const int src = ((c >> 2) << 4) + kr;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      acc_t &tl = acc[tile_index(cb, NB - 1, NB)];
#ifndef YCNR_PK3_NOFOLD  // (devtest/x6many.hip prints the unfolded partial columns with it)
      if constexpr (PK3) {  // columns c, c + 5, c + 10 are the partial sums of live column c; the rest is emptied
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float e = tl[t];
          asm volatile("" : "+v"(e));  // hipcc (ROCm 7.2) otherwise feeds element 0 of the tile to all four shifts (cf. bpermute_opaque)
          const int v = __builtin_bit_cast(int, e);
          const float a5 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x105, 0xF, 0xF, true));   // row_shl:5
          const float a10 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x10A, 0xF, 0xF, true));  // row_shl:10
          tl[t] = c <= kr ? (e + a5) + a10 : 0.0f;
        }
      }
#endif
      float w[4];
      bpermute4_opaque(src << 2, tl, w);
      float v = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) v = (c & 3) == t ? w[t] : v;
      if (g != 0 || (cb == NB - 1 && c >= kr)) v = 0.0f;
      bacc[cb] = v;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (c == kr) tl[t] = 0.0f;
        if (cb == NB - 1 && 4 * g + t == kr) tl[t] = 0.0f;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// GramX6P (round 4): the same Gramian from a PRE-SPLIT copy of the fixed matrix.
//
// PMC of round 3 (profiles/r03_v6_mal_pmc_by_kernel.json): the user half-step at MAL scale issues 4.7 G vector
// instructions per half-step -- 8.9 ms of the SIMDs' issue slots in a 12.4 ms step -- and 40 % of the fused row kernel's
// are the float -> 3 x bf16 split of its gathered values (6.5 per value, 364 per 32-rating step of a wave, next to 147
// MFMAs that each hold the issue port for 8 of their 16 cycles): the kernel is bound by vector ISSUE, not by the matrix
// pipe, and (DESIGN.md "power") the split also costs 18 % of the clock.  The item matrix is split the same way by every
// one of the 10^5..10^6 waves that gather it.  Here it is split ONCE per half-step (als_split_planes_kernel: 12.7 K x 100
// floats at MAL scale) into three bf16 planes per 16-column block,
//     planes[row][block b][plane p][16 columns]      (p = 0, 1, 2: high, middle, low 8 mantissa bits; 96 NB bytes per row)
// and a wave gathers the planes of its 32 ratings straight into LDS by LDS-DMA -- one instruction per (block, plane):
// lane l fetches the 16 bytes of columns 8 (l & 1) .. + 7 of rating l >> 1, so a slot is a plain [32 ratings][16 columns]
// image with a 32-byte pitch -- and reads the MFMA operand (lane (g, c): ratings 8 g .. 8 g + 7 of column c) back with two
// ds_read_b64_tr_b16, the transposing LDS read of gfx950 (conflict-free on this image: cdna_hip_programming.md).  No
// split, no selects, no second image: per step 21 DMAs + 42 LDS reads + ~40 vector instructions at NB = 7.
// Same products in the same order as GramX6D<NB, true, false> (the planes are the same truncations): bit-identical.
//   * PADRHS only (k < 16 NB): the ratings ride in column k of the last block; their three planes are written into the
//     landed slots (3 ds_write_b16 per rating) before the operand of the last block is read.
//   * Operand registers are single-buffered as in GramX6D: after tile row bi nothing reads block bi again, so the next
//     step's block bi is read into the same registers (the reads return behind the MFMAs of row bi + 1) and the slots of
//     block bi are refilled with step s + 2 once those reads have returned (LDS reads complete in order: lgkmcnt(6)).
//   * Every wait is counted by hand (inline-asm LDS reads, DMAs through the builtin): a phase issues 2 + 3 NB vector-memory
//     operations in a fixed order; "block bi of the next step has landed" is vmcnt(3 NB - 1) for bi = 0 and
//     vmcnt(3 NB - 4) after it.
// LDS: 3 NB KB of slots + 2 KB of ids / ratings per wave (23 KB at NB = 7: six waves per CU with the solver's image).
// Needs k % 4 == 0, k <= 112, k % 16 != 0 and the plane matrix below 2 GB; used where the fixed matrix is cache-resident
// (the user half-step): the planes are 1.68 x the bytes of the float rows.
// PACK (k = 16 (NB - 1) + 4: k = 100, 20, 36 ...): the last block has four live columns and the ratings' column, so its three
// planes are stored side by side in ONE 16-column slot -- [h0..h3, hR, m0..m3, mR, l0..l3, lR, 0]; the R positions are
// zero in the matrix and receive the ratings' planes in LDS -- and a tile of the last block column is A_l, A_m, A_h of its
// block row against that packed operand (three MFMAs: all nine plane products, as GramX6D's PK3), the corner tile the packed
// operand against itself (one).  145 MFMAs, 19 DMAs and 38 LDS reads per 32 ratings at k = 100, 19 KB of slots per wave.
__host__ __device__ constexpr int planes_row_bytes(int nb, bool pack) { return pack ? (nb - 1) * 96 + 32 : nb * 96; }
__host__ __device__ constexpr bool planes_pack(int k) { return k > 16 && k % 16 == 4; }

// One thread per (row, block, column quad): the exact 3-way split by truncation of GramX6D, columns >= k are zero.
__global__ void als_split_planes_kernel(const float *__restrict__ fixed, unsigned short *__restrict__ planes, int64_t rows, int k, int nb, int pack) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = rows * nb * 4;
  if (i >= total) return;
  const int q = (int)(i & 3);
  const int b = (int)((i >> 2) % nb);
  const int64_t r = (i >> 2) / nb;
  const int c0 = 16 * b + 4 * q;
  unsigned short h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x = c0 + j < k ? fixed[r * k + c0 + j] : 0.0f;
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const float s1 = x - __builtin_bit_cast(float, u & 0xFFFF0000u);
    const unsigned v = __builtin_bit_cast(unsigned, s1);
    const float t1 = s1 - __builtin_bit_cast(float, v & 0xFFFF0000u);
    h[j] = (unsigned short)(u >> 16);
    m[j] = (unsigned short)(v >> 16);
    l[j] = (unsigned short)(__builtin_bit_cast(unsigned, t1) >> 16);
  }
  typedef unsigned short us4 __attribute__((ext_vector_type(4)));
  unsigned short *row = planes + r * (int64_t)(planes_row_bytes(nb, pack != 0) / 2) + (int64_t)b * 48;
  if (pack && b == nb - 1) {
    if (q != 0) return;  // the four live columns are this block's first quad: [h0..h3, 0, m0..m3, 0, l0..l3, 0, 0]
    *reinterpret_cast<us4 *>(row) = us4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<us4 *>(row + 4) = us4{0, m[0], m[1], m[2]};
    *reinterpret_cast<us4 *>(row + 8) = us4{m[3], 0, l[0], l[1]};
    *reinterpret_cast<us4 *>(row + 12) = us4{l[2], l[3], 0, 0};
    return;
  }
  *reinterpret_cast<us4 *>(row + 4 * q) = us4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<us4 *>(row + 16 + 4 * q) = us4{m[0], m[1], m[2], m[3]};
  *reinterpret_cast<us4 *>(row + 32 + 4 * q) = us4{l[0], l[1], l[2], l[3]};
}

template <int NB, bool PACK>
struct GramX6P {
  static_assert(!PACK || NB >= 2, "the packed last block needs a block in front of it");
  using acc_t = typename MfmaTraits<float>::acc_t;
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  static constexpr int NT = tile_count(NB);
  static constexpr int L = NB - 1;                       // the last block
  static constexpr int SL = PACK ? 1 : 3;                // its slots
  static constexpr int NS = 3 * L + SL;                  // slots of 1 KB: (block, plane)
  static constexpr int LDS_DWORDS = NS * 256;
  static constexpr unsigned OOB = 0x80000000u;
  // vector-memory operations issued behind the DMAs of block bi until the wait for them one phase later (a phase issues, behind
  // its first wait, the two loads of ids / ratings, then the blocks' DMAs in order, block bi - 1 inside iteration bi, the last
  // block at the end).  The loads are retired by a wait of their own at the end of the phase that issued them.
  static constexpr int W0 = NS - 3, WMID = 3 * L - 4 + SL, WLAST = 3 * L - 1;
  static constexpr int RL = PACK ? 2 : 6;                // LDS reads of the last block's operand

  // acc += Y^T Y over ratings [beg, beg + n); column k of the padded Gramian accumulates b = Y^T r.  `lds`: LDS_DWORDS dwords.
  static __device__ __forceinline__ void accumulate(acc_t (&acc)[NT], unsigned *lds, const int32_t *indx, const float *vals,
                                                    const unsigned short *planes, uint32_t planesBytes, int k, int64_t beg, int64_t n, int lane) {
    const int g = lane >> 4;
    const int kr = k - 16 * L;
    constexpr unsigned pitch = (unsigned)planes_row_bytes(NB, PACK);
    const unsigned n32 = (unsigned)(n < 0x3fffffff ? n : 0x3fffffff);
    const int64_t nsteps = (n + 31) >> 5;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)planes, 0, (int)planesBytes, 0x00020000);
    // ids and ratings come by plain buffer loads into registers (inline asm: a load the compiler counted would drain the DMAs)
    const uint64_t pI = (uint64_t)(uintptr_t)(indx + beg), pR = (uint64_t)(uintptr_t)(vals + beg);
    const u32x4 srdI = u32x4{(unsigned)pI, (unsigned)(pI >> 32) & 0xFFFFu, n32 * 4u, 0x00020000u};
    const u32x4 srdR = u32x4{(unsigned)pR, (unsigned)(pR >> 32) & 0xFFFFu, n32 * 4u, 0x00020000u};
    const unsigned ldsBase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)lds;
    // DMA side: this lane's rating of a step and its half of a 16-column block
    const unsigned rhoL = (unsigned)lane >> 1, halfBytes = (unsigned)(lane & 1) * 16u;
    // operand side: lane 4 q + p of 16-lane group g supplies the address of (rating 8 g + q, columns 4 p .. 4 p + 3)
    const unsigned rdBase = ldsBase + (unsigned)g * 256u + (unsigned)((lane >> 2) & 3) * 32u + (unsigned)(lane & 3) * 8u;
    // rating column: lane l < 32 writes the three planes of rating l into the landed slot(s) of the last block
    const unsigned wrBase = ldsBase + (unsigned)(3 * L) * 1024u + (unsigned)(lane & 31) * 32u + (PACK ? 8u : (unsigned)kr * 2u);

    u32x4 p1[NB], p2[NB], p3[NB];  // PACK: p1[L] is the packed operand, p2[L] / p3[L] unused
    u32x2 tr[2][6];                // operand reads in flight: block bi in tr[bi & 1]
    unsigned idNext = 0, idCur = 0, offH = OOB;
    float rvNext = 0.0f, rvCur = 0.0f;

    auto load_meta = [&](int64_t tId, int64_t tR) {  // id of this lane's rating of step tId, rating lane & 31 of step tR: IN FLIGHT on return
      const unsigned oi = (((unsigned)tId << 5) + rhoL) * 4u, orr = (((unsigned)tR << 5) + (unsigned)(lane & 31)) * 4u;
      asm volatile("buffer_load_dword %0, %2, %4, 0 offen\n\tbuffer_load_dword %1, %3, %5, 0 offen"
                   : "=&v"(idNext), "=&v"(rvNext) : "v"(oi), "v"(orr), "s"(srdI), "s"(srdR) : "memory");
    };
    auto retire_meta = [&]() { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(idNext), "+v"(rvNext) : "n"(NS) : "memory"); };
    auto take_meta = [&](int64_t tId) {  // the pair loaded one phase ago has landed (retired by that phase's counted waits)
      asm volatile("" : "+v"(idNext), "+v"(rvNext));
      idCur = idNext;
      rvCur = rvNext;
      const unsigned q0 = ((unsigned)tId << 5) + rhoL;
      offH = q0 < n32 ? idCur * pitch + halfBytes : OOB;
    };
    auto gather_block = [&](int b) {  // the planes of block b of the step offH belongs to
      if (PACK && b == L) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_void_ptr)(lds + (3 * L) * 256), 16, offH, (3 * L) * 32, 0, 0);
      } else {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_void_ptr)(lds + (3 * b + p) * 256), 16, offH, (3 * b + p) * 32, 0, 0);
      }
    };
    auto write_rating_column = [&]() {  // planes of the ratings of the step whose last block has just landed
      const unsigned u = __builtin_bit_cast(unsigned, rvCur);
      const float s1 = rvCur - __builtin_bit_cast(float, u & 0xFFFF0000u);
      const unsigned v = __builtin_bit_cast(unsigned, s1);
      const float t1 = s1 - __builtin_bit_cast(float, v & 0xFFFF0000u);
      const unsigned h = u >> 16, m = v >> 16, l = __builtin_bit_cast(unsigned, t1) >> 16;
      if (lane < 32) {
        if constexpr (PACK)  // columns 4, 9, 14 of the packed slot
          asm volatile("ds_write_b16 %0, %1\n\tds_write_b16 %0, %2 offset:10\n\tds_write_b16 %0, %3 offset:20" ::"v"(wrBase), "v"(h), "v"(m), "v"(l) : "memory");
        else
          asm volatile("ds_write_b16 %0, %1\n\tds_write_b16 %0, %2 offset:1024\n\tds_write_b16 %0, %3 offset:2048" ::"v"(wrBase), "v"(h), "v"(m), "v"(l) : "memory");
      }
    };
    // operand planes of block bi from its slots: transposing reads, IN FLIGHT on return -- nothing may touch t[] before the
    // counted wait that retires them (take_block), not even a register copy (devtest/isa_lint.py checks)
    auto read_block = [&](int bi, u32x2 (&t)[6]) {
      const unsigned ra = rdBase + (unsigned)bi * 3072u;
      if (PACK && bi == L)
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:128" : "=&v"(t[0]), "=&v"(t[1]) : "v"(ra) : "memory");
      else
        asm volatile("ds_read_b64_tr_b16 %0, %6\n\tds_read_b64_tr_b16 %1, %6 offset:128\n\t"
                     "ds_read_b64_tr_b16 %2, %6 offset:1024\n\tds_read_b64_tr_b16 %3, %6 offset:1152\n\t"
                     "ds_read_b64_tr_b16 %4, %6 offset:2048\n\tds_read_b64_tr_b16 %5, %6 offset:2176"
                     : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]) : "v"(ra) : "memory");
    };
    auto take_block = [&](int bi, u32x2 (&t)[6], auto PENDING_) {  // all LDS operations but the youngest PENDING have completed
      constexpr int PENDING = decltype(PENDING_)::value;
      if (PACK && bi == L) {
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(t[0]), "+v"(t[1]) : "n"(PENDING) : "memory");
        p1[bi] = u32x4{t[0][0], t[0][1], t[1][0], t[1][1]};
      } else {
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]) : "n"(PENDING) : "memory");
        p1[bi] = u32x4{t[0][0], t[0][1], t[1][0], t[1][1]};
        p2[bi] = u32x4{t[2][0], t[2][1], t[3][0], t[3][1]};
        p3[bi] = u32x4{t[4][0], t[4][1], t[5][0], t[5][1]};
      }
    };
    auto mma = [&](acc_t &t, const u32x4 &a, const u32x4 &b) {
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), t, 0, 0, 0);
    };
    auto mma_row = [&](int bi) {
      constexpr int NBF = PACK ? L : NB;  // block columns multiplied plane by plane
      if (PACK && bi == L) {
        mma(acc[tile_index(L, L, NB)], p1[L], p1[L]);  // the packed operand against itself: all nine plane products of the corner
        return;
      }
#pragma unroll
      for (int term = 0; term < 6; ++term) {
#pragma unroll
        for (int bj = bi; bj < NBF; ++bj) {
          const u32x4 &pa = term == 0 ? p2[bi] : (term == 1 || term == 3 || term == 5) ? p1[bi] : (term == 2 ? p3[bi] : p2[bi]);
          const u32x4 &pb = term == 0 ? p2[bj] : term == 1 ? p3[bj] : term == 2 ? p1[bj] : term == 3 ? p2[bj] : p1[bj];
          mma(acc[tile_index(bi, bj, NB)], pa, pb);
        }
      }
      if constexpr (PACK) {  // the last block column against the packed operand, smallest terms first (as GramX6D's PK3)
        acc_t &t = acc[tile_index(bi, L, NB)];
        mma(t, p3[bi], p1[L]);
        mma(t, p2[bi], p1[L]);
        mma(t, p1[bi], p1[L]);
      }
    };
    // One phase: products of step s beside the operand reads of step s + 1 and the DMA of step s + 2.
    auto phase = [&](int64_t s, auto MMA_, auto LOAD_) {
      constexpr bool MMA = decltype(MMA_)::value, LOAD = decltype(LOAD_)::value;
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) {
        if constexpr (LOAD) {  // block bi of step s + 1 has landed
          if (bi == 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W0) : "memory");
            take_meta(s + 2);  // id of step s + 2, ratings of step s + 1: loaded in phase s - 1, retired by this wait
            load_meta(s + 3, s + 2);
          } else if (bi < L) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WMID) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WLAST) : "memory");
          }
          if (bi == L) write_rating_column();
        }
        // (the MFMAs are pure operations to the compiler: without the barriers it hoists all of a phase's above the reads
        // and the reads' latency is exposed once per block)
        if constexpr (MMA && LOAD) __builtin_amdgcn_sched_barrier(0);
        if constexpr (MMA) mma_row(bi);
        if constexpr (MMA && LOAD) __builtin_amdgcn_sched_barrier(0);
        if constexpr (LOAD) {
          read_block(bi, tr[bi & 1]);  // step s + 1, behind the MFMAs of row bi + 1
          if (bi > 0) {  // the reads of block bi - 1 have returned (LDS operations complete in order): its registers take them
                         // (row bi - 1 has finished with the old ones), its slots take step s + 2
            if (bi == L) take_block(bi - 1, tr[(bi - 1) & 1], std::integral_constant<int, RL + 3>{});  // (+ the three ds_write_b16)
            else take_block(bi - 1, tr[(bi - 1) & 1], std::integral_constant<int, 6>{});
            gather_block(bi - 1);
          }
        }
      }
      if constexpr (LOAD) {
        take_block(L, tr[L & 1], std::integral_constant<int, 0>{});
        gather_block(L);
        // Everything but this phase's NS DMAs has completed -- in particular the two loads of ids / ratings issued behind the
        // phase's first wait.  They are consumed in the NEXT phase, i.e. carried around the loop: a register copy the compiler
        // places at the back edge must find them landed (a build without this wait copied them one operation too early:
        // one row in 10^5 gathered through a stale id).
#ifndef YCNR_X6P_NO_RETIRE  // (devtest: the build that copied an in-flight rating at the loop's back edge; isa_lint.py must flag it)
        retire_meta();
#endif
      }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    // ids of steps 0 and 1 and the ratings of step 0, then the whole of step 0 into the slots (same issue order as a phase: the
    // gathers last, so phase(-1) can use the same counted waits)
    {  // (one round trip for the three of them: a row's start-up is a chain of dependent loads, and a small upload is one round of waves)
      unsigned id0;
      const unsigned o0 = rhoL * 4u, o1 = (32u + rhoL) * 4u, or0 = (unsigned)(lane & 31) * 4u;
      asm volatile("buffer_load_dword %0, %3, %6, 0 offen\n\tbuffer_load_dword %1, %4, %6, 0 offen\n\tbuffer_load_dword %2, %5, %7, 0 offen\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(id0), "=&v"(idNext), "=&v"(rvNext) : "v"(o0), "v"(o1), "v"(or0), "s"(srdI), "s"(srdR) : "memory");
      offH = rhoL < n32 ? id0 * pitch + halfBytes : OOB;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) gather_block(b);
    phase(-1, no{}, yes{});
    for (int64_t s = 0; s + 1 < nsteps; ++s) phase(s, yes{}, yes{});
    phase(nsteps - 1, yes{}, no{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this wave may land in LDS after it ends
  }

  // PACK: the corner tile (packed operand x packed operand) holds the plane products of live row i in rows i, i + 5, i + 10:
  // fold them (through a 1 KB LDS image: once per row) and clear rows 5..15.  The columns are folded by
  // GramX6D<NB, true, true>::extract_rhs like every tile of the last block column.
  // (WSYNC: see SolveMfmaF32)
  template <bool WSYNC = false>
  static __device__ __forceinline__ void fold_corner_rows(acc_t &tl, float *S, int lane) {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) S[(4 * g + t) * 16 + c] = tl[t];
    SolveMfmaF32<1, WSYNC>::lds_sync();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = 4 * g + t;
      const float v = i <= 4 ? (S[i * 16 + c] + S[(i + 5) * 16 + c]) + S[(i + 10) * 16 + c] : 0.0f;
      tl[t] = i <= 4 ? v : 0.0f;
    }
    SolveMfmaF32<1, WSYNC>::lds_sync();
  }
};

template <int NB, bool PADRHS, bool PK3 = false>
__global__ __launch_bounds__(64, NB <= 7 ? 2 : 1) void als_gram_slab_x6d_kernel(StepArgs<float> a) {
  using G = GramX6D<NB, PADRHS, PK3>;
  using acc_t = typename G::acc_t;
  constexpr int NT = G::NT;
  __shared__ __attribute__((aligned(16))) unsigned lds[G::LDS_DWORDS];
  const int lane = threadIdx.x;
  const Unit u = a.units[blockIdx.x];
  acc_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
  float bacc[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) bacc[cb] = 0.0f;
  G::accumulate(acc, bacc, lds, a.indx, a.vals, a.fixed, a.fixedBytes, a.k, u.beg, u.end - u.beg, lane);
  if constexpr (PADRHS) G::extract_rhs(acc, bacc, a.k, lane);
  // plain slab layout (Gram<float, NB>::store_slab): 16 bytes per lane and tile, then the rhs partials
  float *sl = a.slabs + (int64_t)u.slab * slab_elems(NB) + lane;
  float4 *sq = reinterpret_cast<float4 *>(a.slabs + (int64_t)u.slab * slab_elems(NB)) + lane;
#pragma unroll
  for (int t = 0; t < NT; ++t) sq[t * 64] = float4{acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) sl[(NT * 4 + cb) * 64] = bacc[cb];
}

// Kernel 1b (dominant on the user side): one wave per row that fits one unit -- gather +
// Gramian + rhs, then the row's solve, all in registers.
template <typename T, int NB, bool LDS_SOLVER, bool EDGE, bool E4 = false>
__global__ __launch_bounds__(64, sizeof(T) == 8 ? 1 : YCNR_FUSED_WAVES_PER_SIMD) void als_gram_solve_kernel(StepArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using G = typename GramSel<T, NB, EDGE>::type;
  using acc_t = typename G::acc_t;
  const int lane = threadIdx.x;
  const Unit u = a.units[a.firstFused + blockIdx.x];
  typename G::State st;
  G::init(st);
#ifdef YCNR_ABLATE_GRAM  // timing experiments only: skip the Gramian (results are wrong)
  st.acc[0][0] = (T)u.beg;
#elif YCNR_FUSED_PREFETCH > 1
  G::template accumulate_ring<YCNR_FUSED_PREFETCH>(st, a.indx, a.vals, a.fixed, a.zeros, a.k, u.beg, u.end, lane);
#else
  G::accumulate(st, a.indx, a.vals, a.fixed, a.zeros, a.k, u.beg, u.end, lane);
#endif
  acc_t acc[G::NT];
  T bacc[NB];
  G::to_tiles(st, acc, bacc, reinterpret_cast<T *>(smem), lane);
  // lambda.diagonal(_lambda * _n): the product is formed in double and rounded to T once
  const T lam = (T)(a.lambda * (double)(u.end - u.beg));
#ifdef YCNR_ABLATE_SOLVE  // timing experiments only: skip the solve, keep the Gramian live
  {
    T sum = lam;
    for (int t = 0; t < G::NT; ++t) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int cb = 0; cb < NB; ++cb) sum += bacc[cb];
    if (lane < a.k) a.solved[(int64_t)u.row * a.k + lane] = sum;
  }
#else
  if constexpr (E4 && std::is_same<T, float>::value && !LDS_SOLVER) {
    SolveMfmaF32<NB>::template run<true>(acc, bacc, reinterpret_cast<float *>(smem), a.k, lam, a.solved + (int64_t)u.row * a.k, u.row, a.err, lane,
                                         a.kReal > 0 ? a.kReal : -1);
  } else {
    SolverFor<T, NB, LDS_SOLVER>::type::run(acc, bacc, reinterpret_cast<T *>(smem), a.k, lam,
                                          a.solved + (int64_t)u.row * a.k, u.row, a.err, lane, a.kReal > 0 ? a.kReal : -1);
  }
#endif
}

// Kernel 1b': the fused row kernel with the bf16x6 / LDS-DMA Gramian (GramX6D) in place of the
// float32-MFMA one; the solve is unchanged.  float32, k % 4 == 0, k <= 112, fixed matrix < 2 GB.
template <int NB, bool PADRHS, bool LDS_SOLVER, bool E4 = false, bool PK3 = false>
__global__ __launch_bounds__(64, NB <= 7 ? 2 : 1) void als_gram_solve_x6d_kernel(StepArgs<float> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using G = GramX6D<NB, PADRHS, PK3>;
  using acc_t = typename G::acc_t;
  __shared__ __attribute__((aligned(16))) unsigned ring[G::LDS_DWORDS];
  const int lane = threadIdx.x;
  const Unit u = a.units[a.firstFused + blockIdx.x];
  acc_t acc[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t) acc[t] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
  float bacc[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) bacc[cb] = 0.0f;
#ifdef YCNR_ABLATE_GRAM  // timing experiments only: skip the Gramian (results are wrong)
  acc[0][0] = (float)u.beg;
  (void)ring;
#else
  G::accumulate(acc, bacc, ring, a.indx, a.vals, a.fixed, a.fixedBytes, a.k, u.beg, u.end - u.beg, lane);
  if constexpr (PADRHS) G::extract_rhs(acc, bacc, a.k, lane);
#endif
  const float lam = (float)(a.lambda * (double)(u.end - u.beg));
#ifdef YCNR_ABLATE_SOLVE  // timing experiments only: skip the solve, keep the Gramian live
  {
    float sum = lam;
    for (int t = 0; t < G::NT; ++t) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int cb = 0; cb < NB; ++cb) sum += bacc[cb];
    if (lane < a.k) a.solved[(int64_t)u.row * a.k + lane] = sum;
  }
#else
  if constexpr (E4 && !LDS_SOLVER) {
    SolveMfmaF32<NB>::template run<true>(acc, bacc, reinterpret_cast<float *>(smem), a.k, lam, a.solved + (int64_t)u.row * a.k, u.row, a.err, lane,
                                         a.kReal > 0 ? a.kReal : -1);
  } else {
    SolverFor<float, NB, LDS_SOLVER>::type::run(acc, bacc, reinterpret_cast<float *>(smem), a.k, lam,
                                                a.solved + (int64_t)u.row * a.k, u.row, a.err, lane, a.kReal > 0 ? a.kReal : -1);
  }
#endif
}

// Kernel 1b'': the fused row kernel on the pre-split planes of the fixed matrix (GramX6P); the solve is unchanged and works
// in the slots' LDS once the Gramian has drained (19 KB per wave at k = 100: eight waves per CU).
// float32, k % 4 == 0, k <= 112, k % 16 != 0, plane matrix < 2 GB.
template <int NB, bool PACK, bool E4>
__global__ __launch_bounds__(64, NB <= 7 ? 2 : 1) void als_gram_solve_x6p_kernel(StepArgs<float> a) {
  using G = GramX6P<NB, PACK>;
  using GD = GramX6D<NB, true, PACK>;  // (extract_rhs: the padded column leaves the tiles the same way)
  using acc_t = typename G::acc_t;
  static_assert(SolveMfmaF32<NB>::lds_bytes() <= (size_t)G::LDS_DWORDS * 4, "the solver's image must fit the slots");
  __shared__ __attribute__((aligned(16))) unsigned ring[G::LDS_DWORDS];
  const int lane = threadIdx.x;
  const Unit u = a.units[a.firstFused + blockIdx.x];
  acc_t acc[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t) acc[t] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
  float bacc[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) bacc[cb] = 0.0f;
#ifdef YCNR_ABLATE_GRAM  // timing experiments only: skip the Gramian (results are wrong)
  acc[0][0] = (float)u.beg;
#else
  G::accumulate(acc, ring, a.indx, a.vals, a.planes, a.planesBytes, a.k, u.beg, u.end - u.beg, lane);
  if constexpr (PACK) G::fold_corner_rows(acc[tile_index(NB - 1, NB - 1, NB)], reinterpret_cast<float *>(ring), lane);
  GD::extract_rhs(acc, bacc, a.k, lane);
#endif
  const float lam = (float)(a.lambda * (double)(u.end - u.beg));
#ifdef YCNR_ABLATE_SOLVE  // timing experiments only: skip the solve, keep the Gramian live
  {
    float sum = lam;
    for (int t = 0; t < G::NT; ++t) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int cb = 0; cb < NB; ++cb) sum += bacc[cb];
    if (lane < a.k) a.solved[(int64_t)u.row * a.k + lane] = sum;
    return;
  }
#endif
  if constexpr (E4) {
    SolveMfmaF32<NB>::template run<true>(acc, bacc, reinterpret_cast<float *>(ring), a.k, lam, a.solved + (int64_t)u.row * a.k, u.row, a.err, lane,
                                         a.kReal > 0 ? a.kReal : -1);
  } else {
    SolveMfmaF32<NB>::run(acc, bacc, reinterpret_cast<float *>(ring), a.k, lam, a.solved + (int64_t)u.row * a.k, u.row, a.err, lane, a.kReal > 0 ? a.kReal : -1);
  }
}

// Kernel 1c: the same row solve in its DUAL form, for rows with fewer ratings than factors.
//   x = (Y^T Y + lam I)^-1 Y^T r  =  Y^T (Y Y^T + lam I)^-1 r        (push-through identity)
// so a row with n <= 16 NBN ratings needs the n x n matrix G = Y Y^T (NBN (NBN+1)/2 tiles,
// contraction over the k factors) and an n x n Cholesky instead of the k x k ones, then
// x = Y^T w.  Same arithmetic class (exact f32 MFMA chains), far fewer of them: at k = 100 a
// row with 30 ratings costs 75 + 8 MFMAs and 2 diagonal tiles instead of 224 + 308 and 7.
// Operands: lane (g, c) of block ba holds the float4 Y[idx[16 ba + c]][16 s + 4 g .. +3];
// element j feeds MFMA step (s, j), whose 4 contraction indices are {16 s + 4 g + j : g}.
// Requires k % 4 == 0 (16-byte aligned rows).
// X6: G = Y Y^T on the bf16 matrix pipe with the exact 3-way split (as GramX6D): the contraction
// runs over the k factors, which are contiguous in a gathered row, so lane (g, c) loads its eight
// operand values 32 s + 8 g .. + 7 of row 16 ba + c as two float4 and splits them in place; 6 MFMAs
// of K = 32 per tile replace 8 float32 MFMAs of K = 4 that cost 35 cycles each.
// YCNR_DUAL7_WAVES: the 7-block class (97 ... 112 ratings, k > 128) at TWO waves per SIMD (256 registers, no scratch).  Rounds 3-4
// this build solved 1 - 3 % of its rows wrong at full occupancy and was fenced off; round 5 found the instruction: hipcc's SLP
// vectoriser had paired the right-hand-side updates of two blocks (b_bj -= U[J][bj]^T z_J in SolveMfmaF32::solve) into v_pk_fma_f32
// on kept copies of the panel tiles, and one such product came out wrong when the SIMD's other wave was in its bf16 MFMA phase
// (devtest/dual7/README.md: traced to the block, 0 wrong rows of 2 x 19 200 without the pairing).  The library is built with
// -fno-slp-vectorize and isa_lint.py refuses any packed float32 multiply; one GPU's eighth of C5: 135.8 -> 134.3 ms.
#ifndef YCNR_DUAL7_WAVES
#define YCNR_DUAL7_WAVES 2
#endif
// The 5-block class (65 ... 80 ratings, the largest of the MAL shape's dual classes) at THREE waves per SIMD: with the next K-step's raw
// values double-buffered it takes 224 registers; in the row-by-row form of the classes above 6 blocks (operand planes single-
// buffered, AHEAD = false) and with the ratings loaded behind the Gramian loop (LATE_RHS) it takes 168, no scratch.
// MAL user half-step 12.23 -> 12.13 ms (interleaved A/B, same box); classes 3 / 4 one wave higher (5 / 4) spill 20 bytes.
#ifndef YCNR_DUAL5_WAVES
#define YCNR_DUAL5_WAVES 3
#endif
#ifndef YCNR_DUAL_AHEAD_MAX  // classes up to this many blocks double-buffer the raw values of a whole K-step in registers
#define YCNR_DUAL_AHEAD_MAX 6
#endif
#ifndef YCNR_DUAL_XDEPTH  // steps of x = Y^T w whose loads are in flight in the classes of 7 blocks and more (1: one step, as measured best)
#define YCNR_DUAL_XDEPTH 1
#endif
template <int NBN, bool X6>
__global__ __launch_bounds__(64, NBN == 3 ? 4 : NBN == 4 ? 3 : NBN == 5 ? YCNR_DUAL5_WAVES : NBN == 7 ? YCNR_DUAL7_WAVES : 1) void als_dual_solve_kernel(StepArgs<float> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using Sv = SolveMfmaF32<NBN>;
  using Tr = MfmaTraits<float>;
  using acc_t = typename Tr::acc_t;
  constexpr int NT = tile_count(NBN);
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const Unit u = a.units[a.firstDual + blockIdx.x];
  const int n = (int)(u.end - u.beg);
  const int k = a.k;
  const int ksteps = (k + 15) >> 4;
#ifdef YCNR_DUAL_TRACE  // devtest (tests/tools/dual_trace.py): where and when every row ran -- HW_ID, XCC_ID and the 100 MHz clock at the phase
                        // boundaries go to the rows of `solved` from row YCNR_DUAL_TRACE on, which the probe leaves without ratings
  const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long tr1 = 0, tr2 = 0;
#endif
  // this lane's rating of each 16-rating block: its factor row and its value
  const float *rowp[NBN];
  float bacc[NBN];
  // The ratings are only needed by the solve.  Classes of 7 and more blocks (k > 128) have no registers to keep them
  // across the Gramian loop -- at two waves per SIMD the 7-block class spilled exactly these (5 dwords stored in front
  // of the loop, reloaded behind it) -- so they load them after the loop (LATE_RHS): one more load latency per row,
  // behind the last K-step's MFMAs.
#ifdef YCNR_DUAL_EARLY_RHS  // devtest: the round-3 form (with YCNR_DUAL7_WAVES=2: 24 bytes of scratch per lane in the 7-block class)
  constexpr bool LATE_RHS = false;
#else
  constexpr bool LATE_RHS = NBN >= 7 || (NBN == 5 && YCNR_DUAL5_WAVES >= 3);
#endif
#pragma unroll
  for (int ba = 0; ba < NBN; ++ba) {
    const int i = ba * 16 + c;
    const int64_t q = u.beg + (i < n ? i : n - 1);
    rowp[ba] = i < n ? a.fixed + (int64_t)a.indx[q] * k : a.zeros;
    if constexpr (!LATE_RHS) {
      const float r = a.vals[q];
      bacc[ba] = (i < n && g == 0) ? r : 0.0f;  // group_sum in the solver restores r in all groups
    }
  }
  acc_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
  float4 ya[NBN], yb[NBN];
  // lanes whose 4 factors lie past k read zeros; rows past n point at zeros already
  auto load = [&](float4 (&y)[NBN], int s) {
    const int f = 16 * s + 4 * g;
#pragma unroll
    for (int ba = 0; ba < NBN; ++ba) {
      const float *p = f < k ? rowp[ba] + f : a.zeros;
      y[ba] = *reinterpret_cast<const float4 *>(p);
    }
  };
  auto mma4 = [&](const float4 (&y)[NBN]) {
#pragma unroll
    for (int ba = 0; ba < NBN; ++ba) {
#pragma unroll
      for (int bb = ba; bb < NBN; ++bb) {
        acc_t t = acc[tile_index(ba, bb, NBN)];
        t = Tr::mma(y[ba].x, y[bb].x, t);
        t = Tr::mma(y[ba].y, y[bb].y, t);
        t = Tr::mma(y[ba].z, y[bb].z, t);
        t = Tr::mma(y[ba].w, y[bb].w, t);
        acc[tile_index(ba, bb, NBN)] = t;
      }
    }
  };
#ifdef YCNR_DUAL_ABLATE_G  // timing experiments only (results are wrong)
  if constexpr (false) {
#else
  if constexpr (X6) {
#endif
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // k = 32 m + 4 or + 8 (k = 100!): the last 4 / 8 factors would cost a whole K = 32 step of six bf16 MFMAs per
    // tile and the split of eight mostly-zero values per lane; they go through ONE / TWO exact float32 MFMAs of
    // K = 4 per tile instead (lane (g, c): factor 32 m + 4 q + g of rating 16 ba + c -- one dword, no split)
    const int krem = k & 31;
    const bool remF32 = krem == 4 || krem == 8;
    const int ksteps32 = remF32 ? (k >> 5) : ((k + 31) >> 5);
    float yr[NBN][2];
    if (remF32) {
#pragma unroll
      for (int ba = 0; ba < NBN; ++ba) {
        yr[ba][0] = rowp[ba][(k & ~31) + g];
        yr[ba][1] = krem == 8 ? rowp[ba][(k & ~31) + 4 + g] : 0.0f;
      }
    }
    float4 za[NBN][2], zb[NBN][2];
    auto load8 = [&](float4 (&y)[NBN][2], int s) {
      const int f = 32 * s + 8 * g;
#pragma unroll
      for (int ba = 0; ba < NBN; ++ba) {
        y[ba][0] = *reinterpret_cast<const float4 *>(f < k ? rowp[ba] + f : a.zeros);
        y[ba][1] = *reinterpret_cast<const float4 *>(f + 4 < k ? rowp[ba] + f + 4 : a.zeros);
      }
    };
    // NBN <= 6: the next K-step's rows are loaded while this one is split and multiplied (double-buffered raw values).
    // Larger classes (k > 128 only) have no registers for that; round 3: they visit the tiles ROW BY ROW with the three
    // operand planes single-buffered -- after tile row ba nothing reads block ba again, so the next K-step's block ba is
    // split into the same registers while the rows below still multiply, and its raw values are requested one row
    // ahead (the scheme of GramX6D).  Before: every block loaded just before its split, one wave per SIMD, nothing to
    // hide the loads behind.
    constexpr bool AHEAD = NBN <= YCNR_DUAL_AHEAD_MAX && !(NBN == 5 && YCNR_DUAL5_WAVES >= 3);
    auto split8 = [&](const float4 &z0, const float4 &z1, u32x4 &o1, u32x4 &o2, u32x4 &o3) {
      const float x[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
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
      o1 = u32x4{h[0], h[1], h[2], h[3]};
      o2 = u32x4{m[0], m[1], m[2], m[3]};
      o3 = u32x4{l[0], l[1], l[2], l[3]};
    };
    if constexpr (AHEAD) {
      if (ksteps32 > 0) load8(za, 0);
      for (int s = 0; s < ksteps32; ++s) {
        if (s + 1 < ksteps32) load8(zb, s + 1);
        u32x4 p1[NBN], p2[NBN], p3[NBN];
#pragma unroll
        for (int ba = 0; ba < NBN; ++ba) split8(za[ba][0], za[ba][1], p1[ba], p2[ba], p3[ba]);
        // smallest terms first per tile, product type outermost (consecutive MFMAs hit different tiles)
#pragma unroll
        for (int term = 0; term < 6; ++term) {
#pragma unroll
          for (int ba = 0; ba < NBN; ++ba) {
#pragma unroll
            for (int bb = ba; bb < NBN; ++bb) {
              const u32x4 &pa = term == 0 ? p2[ba] : (term == 1 || term == 3 || term == 5) ? p1[ba] : (term == 2 ? p3[ba] : p2[ba]);
              const u32x4 &pb = term == 0 ? p2[bb] : term == 1 ? p3[bb] : term == 2 ? p1[bb] : term == 3 ? p2[bb] : p1[bb];
              acc[tile_index(ba, bb, NBN)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, pb), acc[tile_index(ba, bb, NBN)], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int ba = 0; ba < NBN; ++ba) {
          za[ba][0] = zb[ba][0];
          za[ba][1] = zb[ba][1];
        }
      }
    } else {
      u32x4 p1[NBN], p2[NBN], p3[NBN];
      auto load_block = [&](float4 (&z)[2], int ba, int s) {
        const int f = 32 * s + 8 * g;
        z[0] = *reinterpret_cast<const float4 *>(f < k ? rowp[ba] + f : a.zeros);
        z[1] = *reinterpret_cast<const float4 *>(f + 4 < k ? rowp[ba] + f + 4 : a.zeros);
      };
      if (ksteps32 > 0) {
#pragma unroll
        for (int ba = 0; ba < NBN; ++ba) {
          float4 z[2];
          load_block(z, ba, 0);
          split8(z[0], z[1], p1[ba], p2[ba], p3[ba]);
        }
      }
      for (int s = 0; s < ksteps32; ++s) {
        const bool more = s + 1 < ksteps32;  // wave-uniform
        float4 zn[2];
        if (more) load_block(zn, 0, s + 1);
#pragma unroll
        for (int ba = 0; ba < NBN; ++ba) {
#pragma unroll
          for (int term = 0; term < 6; ++term) {
#pragma unroll
            for (int bb = ba; bb < NBN; ++bb) {
              const u32x4 &pa = term == 0 ? p2[ba] : (term == 1 || term == 3 || term == 5) ? p1[ba] : (term == 2 ? p3[ba] : p2[ba]);
              const u32x4 &pb = term == 0 ? p2[bb] : term == 1 ? p3[bb] : term == 2 ? p1[bb] : term == 3 ? p2[bb] : p1[bb];
              acc[tile_index(ba, bb, NBN)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, pb), acc[tile_index(ba, bb, NBN)], 0, 0, 0);
            }
          }
          if (more) {  // row ba is done with block ba: the next K-step's values go into its registers
            const float4 zc0 = zn[0], zc1 = zn[1];
            if (ba + 1 < NBN) load_block(zn, ba + 1, s + 1);
            split8(zc0, zc1, p1[ba], p2[ba], p3[ba]);
          }
        }
      }
    }
    if (remF32) {
#pragma unroll
      for (int ba = 0; ba < NBN; ++ba) {
#pragma unroll
        for (int bb = ba; bb < NBN; ++bb) {
          acc_t t = Tr::mma(yr[ba][0], yr[bb][0], acc[tile_index(ba, bb, NBN)]);
          if (krem == 8) t = Tr::mma(yr[ba][1], yr[bb][1], t);
          acc[tile_index(ba, bb, NBN)] = t;
        }
      }
    }
  } else {
#ifndef YCNR_DUAL_ABLATE_G
    load(ya, 0);
    for (int s = 0; s < ksteps; ++s) {
      if (s + 1 < ksteps) load(yb, s + 1);
      mma4(ya);
#pragma unroll
      for (int ba = 0; ba < NBN; ++ba) ya[ba] = yb[ba];
    }
#endif
  }
#ifdef YCNR_DUAL_TRACE
  tr1 = __builtin_amdgcn_s_memrealtime();
#endif
  if constexpr (LATE_RHS) {
#pragma unroll
    for (int ba = 0; ba < NBN; ++ba) {
      const int i = ba * 16 + c;
      const float r = a.vals[u.beg + (i < n ? i : n - 1)];
      bacc[ba] = (i < n && g == 0) ? r : 0.0f;
    }
  }
  const float lam = (float)(a.lambda * (double)n);
  float wcol[NBN];
#ifdef YCNR_DUAL_ABLATE_SOLVE  // timing experiments only
  bool bad = false;
  for (int ba = 0; ba < NBN; ++ba) {
    wcol[ba] = bacc[ba] + lam;
    for (int bb = ba; bb < NBN; ++bb) wcol[ba] += acc[tile_index(ba, bb, NBN)][0] + acc[tile_index(ba, bb, NBN)][1] + acc[tile_index(ba, bb, NBN)][2] + acc[tile_index(ba, bb, NBN)][3];
  }
#else
#ifdef YCNR_DUAL_TRACE
  unsigned *const trRec = reinterpret_cast<unsigned *>(a.solved + (int64_t)(YCNR_DUAL_TRACE) * k) + (int64_t)u.row * 512;
  const bool bad = Sv::template solve<YCNR_DUAL_BATCH>(acc, bacc, reinterpret_cast<float *>(smem), n, lam, wcol, lane, -1, trRec);
#else
  const bool bad = Sv::template solve<YCNR_DUAL_BATCH>(acc, bacc, reinterpret_cast<float *>(smem), n, lam, wcol, lane);
#endif
#endif
#ifdef YCNR_DUAL_TRACE
  tr2 = __builtin_amdgcn_s_memrealtime();
#endif
  // x[f] = sum_a Y[a][f] w[a]: per lane the 4 factors 16 s + 4 g + j of its NBN ratings,
  // summed over the 16 lanes of the group; lane c == 0 of each group stores them
  float *out = a.solved + (int64_t)u.row * k;
  auto xstep = [&](const float4 (&y)[NBN], int s) {
    float4 x = float4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ba = 0; ba < NBN; ++ba) {
      x.x = fmaf(y[ba].x, wcol[ba], x.x);
      x.y = fmaf(y[ba].y, wcol[ba], x.y);
      x.z = fmaf(y[ba].z, wcol[ba], x.z);
      x.w = fmaf(y[ba].w, wcol[ba], x.w);
    }
    Sv::row_sum4(x.x, x.y, x.z, x.w);
    const int f = 16 * s + 4 * g;
    if (c == 0 && f < k) *reinterpret_cast<float4 *>(out + f) = x;
  };
  // YCNR_DUAL_XDEPTH > 1 (devtest): XD steps' loads in flight in the classes of 7 blocks and more (one wave per SIMD, the accumulators are
  // dead by now).  Measured at k = 256: no class faster, the 12-block class slower (4.0 -> 4.3 / 4.5 ms at 2 / 4) -- the pass waits for bytes
  // (it gathers the row's ratings a second time from the last-level cache), not for single loads (NOTES_r05.md).
  constexpr int XD = NBN >= 7 ? YCNR_DUAL_XDEPTH : 1;
#ifdef YCNR_DUAL_ABLATE_X  // timing experiments only
  if (lane < NBN) out[lane] = wcol[0];
#else
  if constexpr (XD > 1) {
    float4 yq[XD][NBN];
#pragma unroll
    for (int d = 0; d < XD; ++d)
      if (d < ksteps) load(yq[d], d);
    for (int s0 = 0; s0 < ksteps; s0 += XD) {
#pragma unroll
      for (int d = 0; d < XD; ++d) {
        const int s = s0 + d;
        if (s < ksteps) {
          xstep(yq[d], s);
          if (s + XD < ksteps) load(yq[d], s + XD);
        }
      }
    }
  } else {
    for (int s = 0; s < ksteps; ++s) {
      load(ya, s);
      xstep(ya, s);
    }
  }
#endif
#ifdef YCNR_DUAL_TRACE
  if (lane == 0) {
    unsigned *tr = trRec;
    tr[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
    tr[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    tr[2] = (unsigned)tr0;
    tr[3] = (unsigned)tr1;
    tr[4] = (unsigned)tr2;
    tr[5] = (unsigned)__builtin_amdgcn_s_memrealtime();
    tr[6] = (unsigned)blockIdx.x;
    tr[7] = 0x7ace7aceu;
  }
#endif
  if (bad && lane == 0) {
    atomicAdd(&a.err->count, 1);
    a.err->firstRow = u.row;
  }
}

// Kernel 1e: FOUR rows of at most 16 ratings per wave (float32, bf16x6 products, k % 4 == 0, k <= 128, fixed
// matrix below 4 GB).  The one-tile class is bound by vector instructions (PMC: 966 per row at MAL scale, 72 %
// of the SIMD's issue rate), and most of them do not depend on the row's length: 16 pivots with their
// replication and the L^-1 they build for block steps that a one-tile system does not have, two LDS round
// trips, seven rounds of cross-lane sums for x = Y^T w.  Here lane group s owns row s:
//   * G_s = Y_s Y_s^T: one 16 x 16 tile per row, bf16x6 MFMAs as in als_dual_solve_kernel (the same loads and
//     splits per row); a register <-> lane-group transpose of the four tiles leaves lane (s, i) with row i
//     of G_s in 16 registers;
//   * the four 16 x 16 systems are solved TOGETHER by Gaussian elimination without pivoting (exact for a
//     symmetric positive definite matrix in the sense of Cholesky: the same pivots d_p, growth factor 1), one
//     row per lane: pivot p broadcasts row p inside each 16-lane group by DPP row_newbcast, lane i > p
//     subtracts m_i = A[i][p] / d_p times it; the right-hand side rides along; back substitution the same way.
//     No L^-1, no LDS, and every instruction works for four rows;
//   * x_s = Y_s^T w_s with lane (s, c) owning float4 c and c + 16 of the row: per rating one broadcast of its
//     id and its w inside the group, two 16-byte loads, eight fma; no cross-lane sums.
// Ratings past a row's end: zero rows of G with a unit diagonal and a zero right-hand side (w = 0).
__global__ __launch_bounds__(64) void als_dual_quad_kernel(StepArgs<float> a, int32_t count) {
  using Sv = SolveMfmaF32<1>;
  using Tr = MfmaTraits<float>;
  using acc_t = typename Tr::acc_t;
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int KS = 4;  // K-steps of 32 factors: k <= 128
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int k = a.k, kq = k >> 2;
  // (k = 32 m + 4 / + 8: the remainder through exact float32 MFMAs of K = 4, as in als_dual_solve_kernel)
  const int krem = k & 31;
  const bool remF32 = krem == 4 || krem == 8;
  const int ksteps32 = remF32 ? (k >> 5) : ((k + 31) >> 5);
  const int first = 4 * (int)blockIdx.x;
  // the four rows (a short last quad repeats its last row; only valid groups store)
  Unit us[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) us[s] = a.units[a.firstDual + min(first + s, count - 1)];
  int ns[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) ns[s] = (int)(us[s].end - us[s].beg);
  const int nmax = max(max(ns[0], ns[1]), max(ns[2], ns[3]));
  // this lane group's row
  const int n_g = g == 0 ? ns[0] : g == 1 ? ns[1] : g == 2 ? ns[2] : ns[3];
  const int64_t beg_g = g == 0 ? us[0].beg : g == 1 ? us[1].beg : g == 2 ? us[2].beg : us[3].beg;
  const int row_g = g == 0 ? us[0].row : g == 1 ? us[1].row : g == 2 ? us[2].row : us[3].row;
  const bool valid_g = first + g < count;
  const int64_t q_g = beg_g + (c < n_g ? c : n_g - 1);
  const unsigned id_g = (unsigned)a.indx[q_g];            // a rating past the end repeats the last one (w = 0)
  float y = c < n_g ? a.vals[q_g] : 0.0f;                  // right-hand side of row g, lane i: r[i]
  // ---- Gramians: one tile per row
  acc_t acc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int64_t q = us[s].beg + (c < ns[s] ? c : ns[s] - 1);
    const float *rowp = c < ns[s] ? a.fixed + (int64_t)a.indx[q] * k : a.zeros;
    float4 z[KS][2];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int f = 32 * kk + 8 * g;
      z[kk][0] = *reinterpret_cast<const float4 *>(f < k ? rowp + f : a.zeros);
      z[kk][1] = *reinterpret_cast<const float4 *>(f + 4 < k ? rowp + f + 4 : a.zeros);
    }
    acc_t t = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
    if (remF32) {  // wave-uniform
      const float y0 = rowp[(k & ~31) + g];
      t = Tr::mma(y0, y0, t);
      if (krem == 8) {
        const float y1 = rowp[(k & ~31) + 4 + g];
        t = Tr::mma(y1, y1, t);
      }
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      if (kk < ksteps32) {  // wave-uniform
        const float x[8] = {z[kk][0].x, z[kk][0].y, z[kk][0].z, z[kk][0].w, z[kk][1].x, z[kk][1].y, z[kk][1].z, z[kk][1].w};
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {  // exact 3-way bf16 split by truncation
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
        const bf16x8 p1 = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
        const bf16x8 p2 = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
        const bf16x8 p3 = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
        // smallest terms first
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p2, p2, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p1, p3, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p3, p1, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p1, p2, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p2, p1, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p1, p1, t, 0, 0, 0);
      }
    }
    acc[s] = t;
  }
  // ---- lane (s, i) <- row i of G_s: acc[s'][t] at lane (q, c) is G_s'[4q + t][c] = G_s'[c][4q + t]
  float R[16];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const acc_t in = acc_t{acc[0][t], acc[1][t], acc[2][t], acc[3][t]};
    float out[4];
    Sv::transpose_rg(in, out);
#pragma unroll
    for (int q = 0; q < 4; ++q) R[4 * q + t] = out[q];
  }
  // + lambda n on the diagonal of the real rows, 1 on the padded ones
  {
    const float dv = c < n_g ? (float)(a.lambda * (double)n_g) : 1.0f;
#pragma unroll
    for (int m = 0; m < 16; ++m) R[m] += c == m ? dv : 0.0f;
  }
  // ---- forward elimination, four systems at once
  float myinv = 1.0f;
  bool bad = false;
  auto bc = [](float v, auto P) { return Sv::template row_bcast<decltype(P)::value>(v); };
  auto pivot = [&](auto P) {
    constexpr int p = decltype(P)::value;
    const float d = bc(R[p], P);
    bad = bad || !(d > 0.0f);
    const float inv = __builtin_amdgcn_rcpf(d);
    const float m = c > p ? R[p] * inv : 0.0f;
    myinv = c == p ? inv : myinv;
    auto upd = [&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (j > p) R[j] = fmaf(-m, bc(R[j], P), R[j]);
    };
    upd(std::integral_constant<int, 1>{}); upd(std::integral_constant<int, 2>{}); upd(std::integral_constant<int, 3>{});
    upd(std::integral_constant<int, 4>{}); upd(std::integral_constant<int, 5>{}); upd(std::integral_constant<int, 6>{});
    upd(std::integral_constant<int, 7>{}); upd(std::integral_constant<int, 8>{}); upd(std::integral_constant<int, 9>{});
    upd(std::integral_constant<int, 10>{}); upd(std::integral_constant<int, 11>{}); upd(std::integral_constant<int, 12>{});
    upd(std::integral_constant<int, 13>{}); upd(std::integral_constant<int, 14>{}); upd(std::integral_constant<int, 15>{});
    y = fmaf(-m, bc(y, P), y);
  };
#define YCNR_P(N) pivot(std::integral_constant<int, N>{});
  YCNR_P(0) YCNR_P(1) YCNR_P(2) YCNR_P(3) YCNR_P(4) YCNR_P(5) YCNR_P(6) YCNR_P(7)
  YCNR_P(8) YCNR_P(9) YCNR_P(10) YCNR_P(11) YCNR_P(12) YCNR_P(13) YCNR_P(14) YCNR_P(15)
#undef YCNR_P
  // ---- back substitution: w[j] = (y[j] - sum_{i > j} U[j][i] w[i]) / U[j][j], columns from the right
  float w = 0.0f;
  auto back = [&](auto J) {
    constexpr int j = decltype(J)::value;
    const float t = y * myinv;             // lane j: w[j]
    w = c == j ? t : w;
    y = fmaf(-R[j], bc(t, J), y);          // lanes < j take column j out (the others are done or do not matter)
  };
#define YCNR_B(N) back(std::integral_constant<int, N>{});
  YCNR_B(15) YCNR_B(14) YCNR_B(13) YCNR_B(12) YCNR_B(11) YCNR_B(10) YCNR_B(9) YCNR_B(8)
  YCNR_B(7) YCNR_B(6) YCNR_B(5) YCNR_B(4) YCNR_B(3) YCNR_B(2) YCNR_B(1) YCNR_B(0)
#undef YCNR_B
  // ---- x = Y^T w: lane (g, c) owns float4 c and c + 16 of row g's result
  const char *base = reinterpret_cast<const char *>(a.fixed);
  const unsigned off0 = (unsigned)min(c, kq - 1) * 16u, off1 = (unsigned)min(c + 16, kq - 1) * 16u;
  float4 x0 = float4{0.0f, 0.0f, 0.0f, 0.0f}, x1 = float4{0.0f, 0.0f, 0.0f, 0.0f};
  auto gather = [&](auto I) {
    constexpr int i = decltype(I)::value;
    if (i < nmax) {  // wave-uniform
      const unsigned idi = __builtin_bit_cast(unsigned, bc(__builtin_bit_cast(float, id_g), I));
      const float wi = bc(w, I);
      const unsigned ro = idi * (unsigned)(k * 4);
      const float4 v0 = *reinterpret_cast<const float4 *>(base + (ro + off0));
      const float4 v1 = *reinterpret_cast<const float4 *>(base + (ro + off1));
      x0.x = fmaf(v0.x, wi, x0.x); x0.y = fmaf(v0.y, wi, x0.y); x0.z = fmaf(v0.z, wi, x0.z); x0.w = fmaf(v0.w, wi, x0.w);
      x1.x = fmaf(v1.x, wi, x1.x); x1.y = fmaf(v1.y, wi, x1.y); x1.z = fmaf(v1.z, wi, x1.z); x1.w = fmaf(v1.w, wi, x1.w);
    }
  };
#define YCNR_G(N) gather(std::integral_constant<int, N>{});
  YCNR_G(0) YCNR_G(1) YCNR_G(2) YCNR_G(3) YCNR_G(4) YCNR_G(5) YCNR_G(6) YCNR_G(7)
  YCNR_G(8) YCNR_G(9) YCNR_G(10) YCNR_G(11) YCNR_G(12) YCNR_G(13) YCNR_G(14) YCNR_G(15)
#undef YCNR_G
  if (valid_g) {
    float *out = a.solved + (int64_t)row_g * k;
    if (c < kq) *reinterpret_cast<float4 *>(out + 4 * c) = x0;
    if (c + 16 < kq) *reinterpret_cast<float4 *>(out + 4 * (c + 16)) = x1;
    // a pivot that was not positive, or NaN / Inf in the result
  }
  {
    // a pivot that was not positive, or NaN / Inf in the result: one report per row
    const float chk = (x0.x + x0.y + x0.z + x0.w + x1.x + x1.y + x1.z + x1.w) * 0.0f;
    const unsigned long long m = __ballot(valid_g && (bad || !(chk == 0.0f)));
    if (c == 0 && ((m >> (16 * g)) & 0xFFFFull)) {
      atomicAdd(&a.err->count, 1);
      a.err->firstRow = row_g;
    }
  }
}

// Kernel 2: one wave per split row -- sum its slabs in slab order, then solve.
// Waves per SIMD the reduce + solve kernel's registers are bounded for.  Left to itself (one) hipcc takes 332 registers at k = 100
// -- every load of a slab gets a register of its own -- and the kernel runs one wave per SIMD.  That is what rows of many
// slabs want (the 64-slab items of the MAL shape bound the kernel by their own sum: 0.45 ms unbounded, 0.50 ms at three
// waves per SIMD).  FEW: no row of the upload has more than kFewSlabs slabs -- the small shapes, where the kernel is a
// wave of solves: bounded to 168 registers (no scratch where listed) three waves share a SIMD; ML-1M shape user half-step
// 0.193 -> 0.182 ms.
constexpr int kFewSlabs = 8;
template <typename T, int NB, bool LDS_SOLVER, bool E4, bool FEW>
constexpr int reduce_waves() {
  if (!FEW || sizeof(T) != 4 || LDS_SOLVER) return 1;
  return NB <= 6 ? 3 : NB == 7 ? (E4 ? 3 : 2) : 1;
}
template <typename T, int NB, bool LDS_SOLVER, bool EDGE, bool E4 = false, bool FEW = false>
__global__ __launch_bounds__(64, (reduce_waves<T, NB, LDS_SOLVER, E4, FEW>())) void als_reduce_solve_kernel(StepArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using G = typename GramSel<T, NB, EDGE>::type;
  using acc_t = typename G::acc_t;
  const int lane = threadIdx.x;
  const SplitRow sr = a.split[blockIdx.x];
  typename G::State st;
  G::init(st);
  // Rows of more than kFewSlabs slabs (float32): the sum over the slabs in double, rounded once (G::Wide).  Which form a row
  // takes depends on its own slab count alone -- not on the piece it was uploaded in (FEW) --, so shards and pieces still
  // cannot change a row's bits.
  bool wide = false;
  if constexpr (std::is_same<T, float>::value && !FEW) wide = sr.nslabs > kFewSlabs;  // wave-uniform
  if (wide) {
    if constexpr (std::is_same<T, float>::value && !FEW)
      G::sum_slabs_wide(st, a.slabs + (int64_t)sr.slab0 * (G::slab_regs() * 64) + lane, sr.nslabs);
  } else {
    for (int sl = 0; sl < sr.nslabs; ++sl) G::add_slab(st, a.slabs + (int64_t)(sr.slab0 + sl) * (G::slab_regs() * 64) + lane);
  }
  acc_t acc[G::NT];
  T bacc[NB];
  G::to_tiles(st, acc, bacc, reinterpret_cast<T *>(smem), lane);
  const T lam = (T)(a.lambda * (double)sr.n);
  if constexpr (E4 && std::is_same<T, float>::value && !LDS_SOLVER) {
    SolveMfmaF32<NB>::template run<true>(acc, bacc, reinterpret_cast<float *>(smem), a.k, lam, a.solved + (int64_t)sr.row * a.k, sr.row, a.err, lane,
                                         a.kReal > 0 ? a.kReal : -1);
  } else {
    SolverFor<T, NB, LDS_SOLVER>::type::run(acc, bacc, reinterpret_cast<T *>(smem), a.k, lam,
                                          a.solved + (int64_t)sr.row * a.k, sr.row, a.err, lane, a.kReal > 0 ? a.kReal : -1);
  }
}

// ---------------------------------------------------------------------------------------
// The item half-step sharded by USER BANDS (ycnr_als_set_ratings_banded).  Every rank accumulates, for EVERY row of
// the side, the Gramian over the ratings of its own bands of columns -- one slab per (row, band): the chunk kernels above, and
// als_slab_sum_kernel where a (row, band) segment was cut into several chunks --, the slabs travel to the row's owner, and the
// owner adds a row's band slabs in band order and solves: the same arithmetic whatever the number of ranks.

// dst slab <- sum of n consecutive source slabs, in order (element offsets into one arena); in double beyond kFewSlabs, as the
// reduce kernels sum the slabs of a row
struct SlabSum {
  int64_t dst, src0;
  int32_t n, pad;
};
template <typename T>
__global__ __launch_bounds__(256) void als_slab_sum_kernel(T *arena, const SlabSum *list, int64_t elems) {
  const SlabSum e = list[blockIdx.x];
  for (int64_t i = threadIdx.x; i < elems; i += 256) {
    const T *src = arena + e.src0 + i;
    if (std::is_same<T, float>::value && e.n > kFewSlabs) {
      double v = 0.0;
      for (int s = 0; s < e.n; ++s) v += (double)src[(int64_t)s * elems];
      arena[e.dst + i] = (T)v;
    } else {
      T v = T(0);
      for (int s = 0; s < e.n; ++s) v += src[(int64_t)s * elems];
      arena[e.dst + i] = v;
    }
  }
}

// One wave per owned row: its band slabs (bandSlab[sr.slab0 + b], null where the row has no rating in band b: local ones in this
// rank's arena, the others where the peers delivered them) added in band order, then the solve of als_reduce_solve_kernel.
template <typename T, int NB, bool LDS_SOLVER, bool EDGE, bool E4 = false>
__global__ __launch_bounds__(64, 1) void als_band_reduce_solve_kernel(StepArgs<T> a, const T *const *bandSlab, int32_t nBands) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using G = typename GramSel<T, NB, EDGE>::type;
  using acc_t = typename G::acc_t;
  const int lane = threadIdx.x;
  const SplitRow sr = a.split[blockIdx.x];
  typename G::State st;
  G::init(st);
  for (int b = 0; b < nBands; ++b) {
    const T *p = bandSlab[(int64_t)sr.slab0 + b];
    if (p) G::add_slab(st, p + lane);  // wave-uniform
  }
  acc_t acc[G::NT];
  T bacc[NB];
  G::to_tiles(st, acc, bacc, reinterpret_cast<T *>(smem), lane);
  const T lam = (T)(a.lambda * (double)sr.n);
  if constexpr (E4 && std::is_same<T, float>::value && !LDS_SOLVER) {
    SolveMfmaF32<NB>::template run<true>(acc, bacc, reinterpret_cast<float *>(smem), a.k, lam, a.solved + (int64_t)sr.row * a.k, sr.row, a.err, lane,
                                         a.kReal > 0 ? a.kReal : -1);
  } else {
    SolverFor<T, NB, LDS_SOLVER>::type::run(acc, bacc, reinterpret_cast<T *>(smem), a.k, lam,
                                          a.solved + (int64_t)sr.row * a.k, sr.row, a.err, lane, a.kReal > 0 ? a.kReal : -1);
  }
}

// ---------------------------------------------------------------------------------------
// RMSE partial sums (EmfWorker.mw_calcRmsePortion, lib/emf/EmfWorker.js:266-315).
// One 256-thread workgroup per portion; a 16-lane group walks one user row at a time,
// its lanes striding the k factors of U[u] and I[i]; (r - pred)^2, pred and the count are
// accumulated in double per group and reduced once per workgroup.
template <typename T>
struct RmseArgs {
  const int64_t *rowPtr;  // local rows, rebased: rowPtr[0] = 0
  const int32_t *indx;
  const T *vals;
  const T *userFactors;
  const T *itemFactors;
  const int64_t *portionRowEnd;  // local exclusive upper row per portion
  double *out;                   // 3 per portion
  double shift;
  int64_t rowBegin;  // global id of local row 0
  int32_t k;
};

// NCH: chunks of 4 values per lane (rows of at most 64 NCH values whose byte length is a multiple of 16: lane l16 owns the chunks
// l16, l16 + 16, ... of both factor rows; the user's chunks stay in registers for all of its ratings); 0: any k, value by value.
// Round 5: RB ratings of the row are in flight at once -- their column ids, values and item rows are requested before the first dot
// product is formed (before: one rating at a time, two dependent trips to memory each: 2.6 ms for the 12 M validation ratings of the
// MAL-scale run).  Same products, same sums, same order as before: the partial sums are bit for bit those of rounds 1-4.
#ifndef YCNR_RMSE_RB
#define YCNR_RMSE_RB 4  // ratings of a row in flight per 16-lane group (float32, k <= 128; half / a quarter of it for wider rows)
#endif
template <typename T, int NCH>
__global__ __launch_bounds__(256) void als_rmse_kernel(RmseArgs<T> a) {
  __shared__ double red[3][16];
  const int tid = threadIdx.x, grp = tid >> 4, l16 = tid & 15;
  const int p = blockIdx.x;
  const int64_t r0 = p == 0 ? 0 : a.portionRowEnd[p - 1];
  const int64_t r1 = a.portionRowEnd[p];
  double sd2 = 0, sp = 0, cnt = 0;
  typedef T vec4 __attribute__((ext_vector_type(4)));
  constexpr int NC = NCH > 0 ? NCH : 1;
  constexpr int RB = NCH == 0 ? 1 : (NCH * (int)sizeof(T) <= 8 ? YCNR_RMSE_RB : (NCH * (int)sizeof(T) <= 16 ? YCNR_RMSE_RB / 2 : YCNR_RMSE_RB / 4));
  const int nchunk = a.k >> 2;
  for (int64_t r = r0 + grp; r < r1; r += 16) {
    const T *uF = a.userFactors + (a.rowBegin + r) * a.k;
    vec4 u[NC];
    if constexpr (NCH > 0) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int ch = l16 + 16 * i;
        u[i] = ch < nchunk ? *reinterpret_cast<const vec4 *>(uF + 4 * ch) : vec4{T(0), T(0), T(0), T(0)};
      }
    }
    const int64_t q1 = a.rowPtr[r + 1];
    for (int64_t q = a.rowPtr[r]; q < q1; q += RB) {
      int32_t id[RB];
      T val[RB];
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int64_t qq = q + j < q1 ? q + j : q1 - 1;
        id[j] = a.indx[qq];
        val[j] = a.vals[qq];
      }
      vec4 v[RB][NC];
      if constexpr (NCH > 0) {
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const T *iF = a.itemFactors + (int64_t)id[j] * a.k;
#pragma unroll
          for (int i = 0; i < NCH; ++i) {
            const int ch = l16 + 16 * i;
            v[j][i] = ch < nchunk ? *reinterpret_cast<const vec4 *>(iF + 4 * ch) : vec4{T(0), T(0), T(0), T(0)};
          }
        }
      }
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        T dot = T(0);
        if constexpr (NCH > 0) {
#pragma unroll
          for (int i = 0; i < NCH; ++i) {
            dot = fma(u[i][0], v[j][i][0], dot);
            dot = fma(u[i][1], v[j][i][1], dot);
            dot = fma(u[i][2], v[j][i][2], dot);
            dot = fma(u[i][3], v[j][i][3], dot);
          }
        } else {
          const T *iF = a.itemFactors + (int64_t)id[j] * a.k;
          for (int f = l16; f < a.k; f += 16) dot = fma(uF[f], iF[f], dot);
        }
        dot += wave_shfl_xor<T>(dot, 8);
        dot += wave_shfl_xor<T>(dot, 4);
        dot += wave_shfl_xor<T>(dot, 2);
        dot += wave_shfl_xor<T>(dot, 1);
        if (l16 == 0 && q + j < q1) {
          const double pred = (double)dot + a.shift;
          const double d = (double)val[j] - pred;
          sd2 += d * d;
          sp += pred;
          cnt += 1.0;
        }
      }
    }
  }
  if (l16 == 0) {
    red[0][grp] = sd2;
    red[1][grp] = cnt;
    red[2][grp] = sp;
  }
  __syncthreads();
  if (tid < 3) {
    double s = 0;
    for (int i = 0; i < 16; ++i) s += red[tid][i];  // fixed order: deterministic
    a.out[3 * p + tid] = s;
  }
}

__global__ void gather_i32_kernel(const int32_t *src, const int64_t *pos, int32_t *out, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[pos[i]];
}

// max over an int32 array (index validation after upload)
__global__ void max_i32_kernel(const int32_t *x, int64_t n, int32_t *out_max, int32_t *out_min) {
  int32_t mx = INT32_MIN, mn = INT32_MAX;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = x[i];
    mx = v > mx ? v : mx;
    mn = v < mn ? v : mn;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int32_t a = __shfl_xor(mx, o, 64), b = __shfl_xor(mn, o, 64);
    mx = a > mx ? a : mx;
    mn = b < mn ? b : mn;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out_max, mx);
    atomicMin(out_min, mn);
  }
}

}  // namespace ycnr
