// als_gram32_kernels.hip.h -- the Gramian of the 240 < k <= 256 float32 path on 32 x 32 bf16 MFMAs, ONE wave per SIMD
// (round 3).
//
// Why.  WgGram (als_wg_kernels.hip.h) runs two waves per SIMD with 16 x 16 x 32 MFMAs: a 32-rating step costs 2.95 us of
// a workgroup where the matrix pipe needs 2.04 -- the split of the gathered values (vector ALU) and the products of a SIMD's
// two waves add up instead of overlapping (MI355X_MICROARCH.md, "two waves per SIMD": moving work between them is
// zero-sum), a 16 x 16 x 32 MFMA leaves the vector issue 8 of its 16 cycles, and every operand read feeds 6 of them.
// Here a row belongs to a 256-thread workgroup, one wave per SIMD with the whole register file:
//   * 32 x 32 x 16 MFMAs (v_mfma_f32_32x32x16_bf16: 32 cycles, the vector issue held for 8 of them): the upper
//     triangle of A = Y^T Y is 36 tiles of 32 x 32, wave w owns block rows w and 7 - w (9 tiles = 144 accumulator
//     registers); an operand read (16 bytes per lane and plane) feeds 6 or 12 MFMAs of twice the work;
//   * the split of the NEXT step's values sits between the MFMAs of this step in the wave's own instruction stream
//     (one chunk of 4 values per thread after every block of products): the vector ALU works in the issue slots the
//     matrix pipe leaves, no second wave needed;
//   * same exact products as everywhere else: three bf16 terms per float, six of the nine products, smallest first.
// LDS planes: [plane][32-column block][rating quad Q = rho >> 2][half = 16-column half][rho & 3][column quad p] chunks of
// 8 bytes (4 columns of one rating): a wave's plane store covers 512 contiguous bytes, and the transposing read of an
// operand (ds_read_b64_tr_b16: lanes 16 h' + 4 q + p read chunk (Q, h', q, p)) covers all 64 banks once per 32 lanes.
// The accumulators leave in the 16 x 16 image layout of als_wg_kernels.hip.h (four sub-tiles per 32 x 32 tile), so the
// slab, the reduce and the solve kernels do not change.
#pragma once
#include "../als_pair_kernels.hip.h"

namespace ycnr {

constexpr int kG32Waves = 4;
constexpr int kG32Threads = kG32Waves * 64;

typedef float g32_f32x16 __attribute__((ext_vector_type(16)));

// diagnostic builds (-DYCNR_WG_STAMPS, tests/tools/g32stamps.py): the shader clock at every step boundary of the first
// rows of workgroup 0, and the 100 MHz real-time counter beside the first and the last of them (-> the clock the
// kernel holds)
#ifdef YCNR_WG_STAMPS
#define YCNR_G32_STAMP(a, slot)                                                                                        \
  do {                                                                                                                 \
    if ((slot) < 240) YCNR_STAMP(a, slot);                                                                             \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 2 && ((slot) == 0 || (slot) == 239))        \
      reinterpret_cast<unsigned long long *>((a).err)[8 + (threadIdx.x >> 6) * 256 + 240 + ((slot) != 0)] = __builtin_amdgcn_s_memrealtime(); \
    ++(slot);                                                                                                          \
  } while (0)
#else
#define YCNR_G32_STAMP(a, slot) do { } while (0)
#endif

struct G32Cfg {
  static constexpr int NBW = 8;                // 32-column blocks of the padded matrix
  static constexpr int NB16 = 16;
  static constexpr int NT16 = tile_count(NB16);
  static constexpr int REGION = 2048;          // bytes of one (plane, block): 32 ratings x 32 columns of bf16
  static constexpr int PLANE = NBW * REGION;
  static constexpr int BUF = 3 * PLANE;
  static constexpr int BPART_OFF = 2 * BUF;
  static constexpr int BPART_BYTES = kG32Waves * 256 * 4;
  static constexpr int VEC_OFF = BPART_OFF + BPART_BYTES;  // b: 256 floats
  static constexpr int LDS_BYTES = VEC_OFF + 1024;
  static constexpr int64_t SLAB_FLOATS = (int64_t)NT16 * 256 + NB16 * 16;
};
static_assert(G32Cfg::SLAB_FLOATS == WgCfg<16>::SLAB_FLOATS, "same slab as the 16 x 16 kernels");

struct Gram32 {
  using C = G32Cfg;
  using acc_t = g32_f32x16;
  static constexpr int NACC = 9;

  struct Stage {  // one step's gathered values of this thread (rating tid >> 3, columns 32 j + 4 (tid & 7) .. +3), and its rating
    wg_f32x4 x[8];
    float r;
  };
  struct Meta {
    int32_t id;
    float r;
    bool valid;
  };
  static __device__ __forceinline__ Meta load_meta(const StepArgs<float> &a, int64_t beg, int64_t n, int64_t s, int rho) {
    const int64_t q = (s << 5) + rho;
    Meta m;
    m.valid = q < n;
    const int64_t qc = beg + (m.valid ? q : n - 1);
    m.id = a.indx[qc];
    m.r = a.vals[qc];
    return m;
  }
  static __device__ __forceinline__ void load_rows(Stage &st, const StepArgs<float> &a, const Meta &m, int l8) {
#ifdef YCNR_G32_ABLATE_GATHER  // timing experiments only: every rating reads the same row (no traffic behind the L2)
    const float *row = m.valid ? a.fixed + (int64_t)(m.id & 63) * a.k : a.zeros;
#else
    const float *row = m.valid ? a.fixed + (int64_t)m.id * a.k : a.zeros;
#endif
    st.r = m.valid ? m.r : 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = 32 * j + 4 * l8;
      const float *p = col < a.k ? row + col : a.zeros;
      st.x[j] = *reinterpret_cast<const wg_f32x4 *>(p);
    }
  }

  // chunk J of a stage: b += r x, the three bf16 planes of its 4 values -> plane buffer `buf`
  template <int J>
  static __device__ __forceinline__ void split_chunk(const Stage &st, float (&bacc)[8][4], unsigned char *buf, int wofs) {
#ifdef YCNR_G32_ABLATE_SPLIT  // timing experiments only: the loads stay alive, nothing split or stored
    asm volatile("" ::"v"(st.x[J]), "v"(st.r));
    return;
#endif
    unsigned h[2], m[2], l[2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bacc[J][e] = fmaf(st.x[J][e], st.r, bacc[J][e]);
      asm volatile("" : "+v"(bacc[J][e]));  // no v_pk_fma_f32 beside MFMAs
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const float x0 = st.x[J][2 * jj], x1 = st.x[J][2 * jj + 1];
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
    unsigned char *p = buf + wofs + J * C::REGION;
    *reinterpret_cast<wg_u32x2 *>(p) = wg_u32x2{h[0], h[1]};
    *reinterpret_cast<wg_u32x2 *>(p + C::PLANE) = wg_u32x2{m[0], m[1]};
    *reinterpret_cast<wg_u32x2 *>(p + 2 * C::PLANE) = wg_u32x2{l[0], l[1]};
  }
  static __device__ __forceinline__ void split_all(const Stage &st, float (&bacc)[8][4], unsigned char *buf, int wofs) {
    static_for<0, 8>([&](auto J) { split_chunk<decltype(J)::value>(st, bacc, buf, wofs); });
  }

  struct Op {  // the three bf16 planes of one 32-column block, 16 ratings, in MFMA operand layout
    wg_bf16x8 p[3];
  };
  static __device__ __forceinline__ Op read_op(const unsigned char *buf, int block, int kap, int rofs) {
    Op o;
    typedef __attribute__((address_space(3))) wg_s16x4 *lds_s16x4_ptr;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const unsigned char *p = buf + pl * C::PLANE + block * C::REGION + kap * 1024 + rofs;
      const wg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
      const wg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 256));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      o.p[pl] = __builtin_bit_cast(wg_bf16x8, v);
    }
    return o;
  }
  static __device__ __forceinline__ acc_t mma6(const Op &A, const Op &B, acc_t acc) {
#ifdef YCNR_G32_ABLATE_MFMA  // timing experiments only: the operands stay alive, no products
    asm volatile("" ::"v"(A.p[0]), "v"(A.p[1]), "v"(A.p[2]), "v"(B.p[0]), "v"(B.p[1]), "v"(B.p[2]));
    return acc;
#endif
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.p[1], B.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.p[0], B.p[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.p[2], B.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.p[0], B.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.p[1], B.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.p[0], B.p[0], acc, 0, 0, 0);
    return acc;
  }

  // the products of one 32-rating step for wave W -- tiles (W, W..7) in acc[0 .. 8-W), (7-W, 7-W..7) behind them -- with
  // the eight chunks of `fill` spread between its blocks of products.  One wave per SIMD means nothing covers a stall,
  // so the order is pinned (sched_barrier): [operand reads of the NEXT block] | [6 or 12 MFMAs of this block with a
  // chunk's vector instructions dealt between them] | ... -- left alone, hipcc issued reads right in front of their first
  // use (ten exposed LDS latencies per step) and the vector work in bursts.
  template <int W, typename F>
  static __device__ __forceinline__ void mma_step(acc_t (&acc)[NACC], const unsigned char *buf, int rofs, F &&fill) {
    constexpr int r0 = W, r1 = C::NBW - 1 - W, n0 = C::NBW - r0;
    constexpr int NIT = 2 * n0;
    Op A0 = read_op(buf, r0, 0, rofs);
    Op A1 = A0, B = A0;
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NIT>([&](auto IT) {
      constexpr int it = decltype(IT)::value, bj = r0 + it % n0;
      Op Bn = B;
      if constexpr (it + 1 < NIT) Bn = read_op(buf, r0 + (it + 1) % n0, (it + 1) / n0, rofs);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (bj == r1) A1 = B;
      acc[bj - r0] = mma6(A0, B, acc[bj - r0]);
      if constexpr (bj >= r1) acc[n0 + bj - r1] = mma6(A1, B, acc[n0 + bj - r1]);
      constexpr int nm = bj >= r1 ? 12 : 6;
      constexpr int c0 = (it * 8 + NIT - 1) / NIT, c1 = ((it + 1) * 8 + NIT - 1) / NIT;  // chunks ch with ch * NIT / 8 == it
      static_for<c0, c1>([&](auto CH) { fill(CH); });
      if constexpr (c1 > c0) {
        constexpr int per = ((c1 - c0) * 34 + nm - 1) / nm;
        static_for<0, nm>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, per, 0);  // the chunk's share of vector instructions
        });
        __builtin_amdgcn_sched_group_barrier(0x200, 3 * (c1 - c0), 0);  // its plane stores
      }
      __builtin_amdgcn_sched_barrier(0);
      B = Bn;
      if constexpr (bj == C::NBW - 1) {  // next: block r0 of the second half of the ratings
        A0 = Bn;
        A1 = Bn;
      }
    });
  }

  // accumulator tiles of wave W -> the 16 x 16 image layout at `img`.  Lane l of a 32 x 32 tile holds column l & 31,
  // rows 8 q + 4 (l >> 5) .. +3 in registers 4 q .. 4 q + 3: sub-tile (q >> 1, (l >> 4) & 1), rows 8 (q & 1) + 4 (l >> 5) .. +3
  template <int W, typename P>
  static __device__ __forceinline__ void store_tiles(const acc_t (&acc)[NACC], P img, int lane) {
    constexpr int r0 = W, r1 = C::NBW - 1 - W, n0 = C::NBW - r0;
    const int c = lane & 15, H = lane >> 5;
    const bool right = (lane >> 4) & 1;
    static_for<0, 2>([&](auto RS) {
      constexpr int bi = decltype(RS)::value == 0 ? r0 : r1;
      constexpr int base = decltype(RS)::value == 0 ? 0 : n0;
      static_for<bi, C::NBW>([&](auto BJ) {
        constexpr int bj = decltype(BJ)::value;
        const acc_t &v = acc[base + bj - bi];
        static_for<0, 4>([&](auto Q) {
          constexpr int q = decltype(Q)::value, sr = q >> 1;
          constexpr bool below = bi == bj && sr == 1;  // sub-tile (2 bi + 1, 2 bi) lies below the diagonal
          constexpr int T1 = tile_index(2 * bi + sr, 2 * bj + 1, C::NB16);
          constexpr int T0 = below ? T1 : tile_index(2 * bi + sr, 2 * bj, C::NB16);
          const int off = c * 16 + 4 * ((2 * (q & 1) + H) ^ ((c >> 1) & 3));
          const wg_f32x4 val = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
          if (!below || right) *reinterpret_cast<wg_f32x4 *>(&img[(right ? T1 : T0) * 256 + off]) = val;
        });
      });
    });
  }
  // Gramian of ratings [beg, beg + n): on return the accumulators hold the tiles, bacc this thread's partial sums of b,
  // and all waves have passed the last barrier (LDS is free).  s0 / s1: steps 0 and 1, already requested by the caller;
  // m2: the ids of step 2.  W: the caller's wave (the whole row loop is specialised per wave: a dispatch per step
  // would merge the four code paths' accumulators at every step, which costs hundreds of register moves).
  template <int W>
  static __device__ __forceinline__ void run(const StepArgs<float> &a, int64_t beg, int64_t n, unsigned char *smem, acc_t (&acc)[NACC],
                                             float (&bacc)[8][4], Stage &s0, Stage &s1, Meta &m2, [[maybe_unused]] int &slot) {
    const int tid = threadIdx.x, lane = tid & 63, l8 = tid & 7, rho = tid >> 3;
    const int64_t nsteps = (n + 31) >> 5;
    const int wofs = (rho >> 2) * 256 + (l8 >> 2) * 128 + (rho & 3) * 32 + (l8 & 3) * 8;
    const int rofs = (lane >> 5) * 512 + ((lane >> 4) & 1) * 128 + 8 * (lane & 15);
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) bacc[j][e] = 0.0f;
    unsigned char *buf0 = smem, *buf1 = smem + C::BUF;
    split_all(s0, bacc, buf0, wofs);
    load_rows(s0, a, m2, l8);
    Meta m3 = load_meta(a, beg, n, 3, rho);
    __syncthreads();
    // two steps per trip, ONE loop exit (an exit between the phases made the compiler copy accumulator tiles around
    // the loop: 160 register moves per trip); an odd last step follows the loop, its planes are in buf0 by then
    YCNR_G32_STAMP(a, slot);
    for (int64_t s = 0; s + 1 < nsteps; s += 2) {
      // phase s: products of step s; planes of step s + 1; rows of step s + 3
      mma_step<W>(acc, buf0, rofs, [&](auto CH) { split_chunk<decltype(CH)::value>(s1, bacc, buf1, wofs); });
      load_rows(s1, a, m3, l8);
      const Meta m4 = load_meta(a, beg, n, s + 4, rho);
      __syncthreads();
      YCNR_G32_STAMP(a, slot);
      // phase s + 1: planes of step s + 2 (always written: zeros past the end)
      mma_step<W>(acc, buf1, rofs, [&](auto CH) { split_chunk<decltype(CH)::value>(s0, bacc, buf0, wofs); });
      load_rows(s0, a, m4, l8);
      m3 = load_meta(a, beg, n, s + 5, rho);
      __syncthreads();
      YCNR_G32_STAMP(a, slot);
    }
    if (nsteps & 1) {
      mma_step<W>(acc, buf0, rofs, [](auto) {});
      __syncthreads();
      YCNR_G32_STAMP(a, slot);
    }
  }

  // this thread's b partials -> b (256 floats at VEC_OFF); ends with a barrier
  static __device__ __forceinline__ void reduce_b(float (&bacc)[8][4], unsigned char *smem) {
    const int tid = threadIdx.x, lane = tid & 63, l8 = tid & 7;
    const int wave = tid >> 6;
    float *bpart = reinterpret_cast<float *>(smem + C::BPART_OFF);
    float *bvec = reinterpret_cast<float *>(smem + C::VEC_OFF);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      wg_f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = bacc[j][e];
        t += __shfl_xor(t, 8, 64);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        v[e] = t;
      }
      if (lane < 8) *reinterpret_cast<wg_f32x4 *>(bpart + wave * 256 + 32 * j + 4 * l8) = v;
    }
    __syncthreads();
    {
      float t = 0.0f;
#pragma unroll
      for (int w = 0; w < kG32Waves; ++w) t += bpart[w * 256 + tid];
      bvec[tid] = t;
    }
    __syncthreads();
  }
};

// whole rows of a batch: Gramian -> the batch's slab number (unit - first), persistent over [first, first + count)
template <int W>
__device__ __forceinline__ void g32_rowslab_body(const StepArgs<float> &a, float *rowSlabs, int32_t first, int32_t count, unsigned char *smem) {
  using G = Gram32;
  using C = G32Cfg;
  const int tid = threadIdx.x, lane = tid & 63, l8 = tid & 7, rho = tid >> 3;
  G::Stage s0, s1;
  G::Meta m2;
  int32_t ui = blockIdx.x;
  Unit u = a.units[first + ui];
  auto prefetch = [&](const Unit &v) {  // a row's first two steps, requested while the row before it is finished
    const int64_t n = v.end - v.beg;
    const G::Meta m0 = G::load_meta(a, v.beg, n, 0, rho), m1 = G::load_meta(a, v.beg, n, 1, rho);
    m2 = G::load_meta(a, v.beg, n, 2, rho);
    G::load_rows(s0, a, m0, l8);
    G::load_rows(s1, a, m1, l8);
  };
  prefetch(u);
  int slot = 0;
  while (true) {
    G::acc_t acc[G::NACC];
    float bacc[8][4];
    G::run<W>(a, u.beg, u.end - u.beg, smem, acc, bacc, s0, s1, m2, slot);
    float *slab = rowSlabs + (int64_t)ui * C::SLAB_FLOATS;
    ui += gridDim.x;
    const bool more = ui < count;
    if (more) {
      u = a.units[first + ui];
      prefetch(u);
    }
    G::store_tiles<W>(acc, slab, lane);
    G::reduce_b(bacc, smem);
    slab[(int64_t)C::NT16 * 256 + tid] = reinterpret_cast<const float *>(smem + C::VEC_OFF)[tid];
    if (!more) break;
    __syncthreads();
  }
}
__global__ __launch_bounds__(kG32Threads, 1) void als_g32_rowslab_kernel(StepArgs<float> a, float *rowSlabs, int32_t first, int32_t count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((int32_t)blockIdx.x >= count) return;
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: g32_rowslab_body<0>(a, rowSlabs, first, count, smem); break;
    case 1: g32_rowslab_body<1>(a, rowSlabs, first, count, smem); break;
    case 2: g32_rowslab_body<2>(a, rowSlabs, first, count, smem); break;
    default: g32_rowslab_body<3>(a, rowSlabs, first, count, smem); break;
  }
}

// chunks of heavy rows: Gramian -> slab (image layout + b), persistent over units [0, count)
template <int W>
__device__ __forceinline__ void g32_slab_body(const StepArgs<float> &a, int32_t count, unsigned char *smem) {
  using G = Gram32;
  using C = G32Cfg;
  const int tid = threadIdx.x, lane = tid & 63, l8 = tid & 7, rho = tid >> 3;
  int slot = 250;  // (stamps: the row kernel's only)
  for (int32_t ui = blockIdx.x; ui < count; ui += gridDim.x) {
    const Unit u = a.units[ui];
    const int64_t n = u.end - u.beg;
    G::Stage s0, s1;
    const G::Meta m0 = G::load_meta(a, u.beg, n, 0, rho), m1 = G::load_meta(a, u.beg, n, 1, rho);
    G::Meta m2 = G::load_meta(a, u.beg, n, 2, rho);
    G::load_rows(s0, a, m0, l8);
    G::load_rows(s1, a, m1, l8);
    G::acc_t acc[G::NACC];
    float bacc[8][4];
    G::run<W>(a, u.beg, n, smem, acc, bacc, s0, s1, m2, slot);
    float *slab = a.slabs + (int64_t)u.slab * C::SLAB_FLOATS;
    G::store_tiles<W>(acc, slab, lane);
    G::reduce_b(bacc, smem);
    slab[(int64_t)C::NT16 * 256 + tid] = reinterpret_cast<const float *>(smem + C::VEC_OFF)[tid];
    __syncthreads();
  }
}
__global__ __launch_bounds__(kG32Threads, 1) void als_g32_slab_kernel(StepArgs<float> a, int32_t count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: g32_slab_body<0>(a, count, smem); break;
    case 1: g32_slab_body<1>(a, count, smem); break;
    case 2: g32_slab_body<2>(a, count, smem); break;
    default: g32_slab_body<3>(a, count, smem); break;
  }
}

}  // namespace ycnr
