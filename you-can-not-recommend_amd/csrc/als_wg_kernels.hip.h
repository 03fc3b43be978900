// als_wg_kernels.hip.h -- the float32 path for 128 < factorsCount <= 256: one 512-thread workgroup
// (8 waves, two per SIMD) per row, one workgroup per CU, persistent over the rows of a launch.
//
// At k = 256 the upper triangle of A = Y^T Y is 136 tiles of 16 x 16: 544 accumulator registers per
// lane, more than a wave owns, and a 136 KB image, most of a CU's LDS.  So a row belongs to a whole
// workgroup:
//
//  Gramian (WgGram).  Wave w owns the tiles of block rows w and NB-1-w (NB + 1 tiles: 68
//    registers).  Products run on the bf16 matrix pipe with float32-equivalent results: every
//    gathered float is split exactly into three bf16 terms and six of the nine products are
//    accumulated (the exact scheme of GramX6D, als_kernels.hip.h).  A step is 32 ratings:
//      * every thread loads 4 x 16 bytes of ONE gathered row (thread t: rating t >> 4 of the step,
//        columns 64 j + 4 (t & 15) .. +3) with plain 16-byte global loads, two steps ahead of their
//        use -- each gathered row crosses L2 -> CU once per workgroup, 16 bytes per lane;
//      * it splits its 16 values in registers and writes the three bf16 planes to LDS as
//        [rating][column] rows of 4 columns = 8 bytes (ds_write_b64), into one of two plane buffers;
//      * the MFMA operand of block b -- lane (g, c): ratings 8g .. 8g+7 of column 16 b + c -- comes
//        back with two ds_read_b64_tr_b16 per plane (the transposing LDS read of gfx950): no
//        shuffles, no second image.  Inside a block's 1 KB region the 8-byte chunk of (rating rho,
//        column quad p) sits at  p + 4 (rho & 3) + 16 ((rho >> 3) & 1) + 32 ((rho >> 2) & 1) + 64 (rho >> 4):
//        a transposed read touches 32 consecutive chunks (conflict-free) and consecutive blocks are
//        32 bytes further apart than 1 KB, which spreads the 16 writers of a rating over all banks.
//    One workgroup barrier per step; b = Y^T r is accumulated on the VALU from the unsplit values.
//
//  Solve (WgSolve).  The tiles go from the accumulators to an LDS image ([col][row] per tile with a
//    16-byte XOR swizzle: every operand and every tile is ONE conflict-free ds_read_b128 /
//    ds_write_b128 at the same per-lane offset) and a right-looking block Cholesky A = U^T U runs on
//    it with float32 MFMAs, all eight waves working:
//      for J:  panel  U[J][bj] = W_J T[J][bj]            (tiles dealt over the waves)
//              update T[bi][bj] -= U[J][bi]^T U[J][bj]   wave 0 takes tile (J+1, J+1) FIRST and
//                     factors + inverts it (16 sequential pivots) while the other seven waves do the
//                     rest of the trailing update: the next diagonal tile is ready when they are
//                     (look-ahead), two barriers per block step;
//      the right-hand side rides along (z_J = W_J b_J in the panel phase, b_bj -= U[J][bj]^T z_J in
//      the update phase, one wave per block column), and the back substitution is right-looking
//      too: after x_J every wave updates the block rows it owns, one barrier per block.
//
//  Rows longer than a launch wants in one workgroup are cut into chunks whose tiles go to a slab in
//  global memory in image layout (als_wg_gram_slab_kernel) and are summed in slab order before the
//  same solve (als_wg_reduce_solve_kernel).  Rows of <= 176 ratings never come here: they take the
//  dual form (als_dual_solve_kernel).
#pragma once
#include "als_kernels.hip.h"

namespace ycnr {

constexpr int kWgWaves = 8;
constexpr int kWgThreads = kWgWaves * 64;

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wg_u32x2 __attribute__((ext_vector_type(2)));
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));

// In-kernel stamps (diagnostic builds only, -DYCNR_WG_STAMPS): lane 0 of waves 0 and 1 of workgroup 0
// writes the shader clock at phase boundaries of the FIRST row it solves into the words behind
// ErrInfo (the host allocates 64 KB there); ycnr_als_sync dumps them when YCNR_DUMP_STAMPS names a file.
#ifdef YCNR_WG_STAMPS
#define YCNR_STAMP(a, slot)                                                                                   \
  do {                                                                                                        \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 2 && (slot) < 250)                 \
      reinterpret_cast<unsigned long long *>((a).err)[8 + (threadIdx.x >> 6) * 256 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define YCNR_STAMP(a, slot) do { } while (0)
#endif

template <int NB>
struct WgCfg {
  static constexpr int NT = tile_count(NB);
  static constexpr int REGION = 1024 + 32;  // bytes of one (plane, block): 32 ratings x 16 columns of bf16, + bank rotation
  static constexpr int PLANE = NB * REGION;
  static constexpr int BUF = 3 * PLANE;
  static constexpr int GRAM_BYTES = 2 * BUF;
  static constexpr int IMG_BYTES = NT * 1024;
  static constexpr int BPART_OFF = IMG_BYTES > GRAM_BYTES ? IMG_BYTES : GRAM_BYTES;
  static constexpr int BPART_BYTES = kWgWaves * NB * 16 * 4;
  static constexpr int VEC_OFF = BPART_OFF + BPART_BYTES;  // bvec, zvec, xvec: NB * 16 floats each
  static constexpr int DT_OFF = VEC_OFF + 3 * NB * 64;
  static constexpr int LDW = 20;
  static constexpr int FLAG_OFF = DT_OFF + 16 * LDW * 4;
  static constexpr int LDS_BYTES = FLAG_OFF + 64;
  // elements of one slab in global memory: the image + b
  static constexpr int64_t SLAB_FLOATS = (int64_t)NT * 256 + NB * 16;
};
__host__ __device__ constexpr int64_t wg_slab_floats(int nb) { return (int64_t)tile_count(nb) * 256 + nb * 16; }

// byte offset of a lane's 16 bytes in a tile of the image: element (row r, col c) of the tile lives
// at float c * 16 + 4 ((r >> 2) ^ ((c >> 1) & 3)) + (r & 3); lane (g, c) owns rows 4g .. 4g+3 of column c
__device__ __forceinline__ int wg_tile_lane_off(int lane) {
  const int g = lane >> 4, c = lane & 15;
  return (c * 16 + 4 * (g ^ ((c >> 1) & 3))) * 4;
}

// block rows of wave W: r0 = W, r1 = NB-1-W (one row when they coincide, none when W > NB-1-W)
template <int NB, int W>
struct WgRows {
  static constexpr int r0 = W, r1 = NB - 1 - W;
  static constexpr bool any = r0 <= r1, two = r0 < r1;
  static constexpr int n0 = any ? NB - r0 : 0;          // tiles of row r0: bj = r0 .. NB-1
  static constexpr int n1 = two ? NB - r1 : 0;          // tiles of row r1
};

template <int NB>
struct WgGram {
  using C = WgCfg<NB>;
  using acc_t = typename MfmaTraits<float>::acc_t;
  static constexpr int NACC = NB + 1;

  struct Stage {  // one step's gathered values of this thread, and its rating
    wg_f32x4 x[4];
    float r;
  };

  // ids and ratings are read one step ahead of the rows they address
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
  static __device__ __forceinline__ void load_rows(Stage &st, const StepArgs<float> &a, const Meta &m, int l16) {
    const float *row = m.valid ? a.fixed + (int64_t)m.id * a.k : a.zeros;
    st.r = m.valid ? m.r : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = 64 * j + 4 * l16;
      if (64 * j < NB * 16) {  // static: loads a block of this NB can need
        const float *p = col < a.k ? row + col : a.zeros;
        st.x[j] = *reinterpret_cast<const wg_f32x4 *>(p);
      }
    }
  }

  // split + plane writes of one stage into plane buffer `buf` (byte address of the buffer in LDS)
  static __device__ __forceinline__ void split_store(const Stage &st, float (&bacc)[4][4], unsigned char *buf, int wofs, int l16) {
#ifdef YCNR_WG_ABLATE_SPLIT  // timing experiments only: loads stay alive, nothing split or stored
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(st.x[j]));
    return;
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (64 * j >= NB * 16) continue;
      if (4 * j + (l16 >> 2) >= NB) continue;  // a block beyond the padded matrix (columns >= 16 NB)
      unsigned h[2], m[2], l[2];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bacc[j][e] = fmaf(st.x[j][e], st.r, bacc[j][e]);
        asm volatile("" : "+v"(bacc[j][e]));  // no v_pk_fma_f32: packed float32 beside MFMAs is slower than two v_fma
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const float x0 = st.x[j][2 * jj], x1 = st.x[j][2 * jj + 1];
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
      unsigned char *p = buf + wofs + 4 * j * C::REGION;
      *reinterpret_cast<wg_u32x2 *>(p) = wg_u32x2{h[0], h[1]};
      *reinterpret_cast<wg_u32x2 *>(p + C::PLANE) = wg_u32x2{m[0], m[1]};
      *reinterpret_cast<wg_u32x2 *>(p + 2 * C::PLANE) = wg_u32x2{l[0], l[1]};
    }
  }

  struct Op {  // the three bf16 planes of one block in MFMA operand layout
    wg_bf16x8 p[3];
  };
  static __device__ __forceinline__ Op read_op(const unsigned char *buf, int block, int rofs) {
    Op o;
    typedef __attribute__((address_space(3))) wg_s16x4 *lds_s16x4_ptr;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const unsigned char *p = buf + pl * C::PLANE + block * C::REGION + rofs;
      const wg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
      const wg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 256));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      o.p[pl] = __builtin_bit_cast(wg_bf16x8, v);
    }
    return o;
  }
  // acc += A^T B with float32-equivalent products: the six significant bf16 products, smallest first
  static __device__ __forceinline__ acc_t mma6(const Op &A, const Op &B, acc_t acc) {
#ifdef YCNR_WG_ABLATE_MFMA  // timing experiments only: operands stay alive, no products
    asm volatile("" ::"v"(A.p[0]), "v"(A.p[1]), "v"(A.p[2]), "v"(B.p[0]), "v"(B.p[1]), "v"(B.p[2]));
    return acc;
#endif
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.p[1], B.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.p[0], B.p[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.p[2], B.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.p[0], B.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.p[1], B.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.p[0], B.p[0], acc, 0, 0, 0);
    return acc;
  }

  // the products of one step for wave W: tiles (r0, r0..NB-1) in acc[0 .. n0), (r1, r1..NB-1) behind them
  template <int W>
  static __device__ __forceinline__ void mma_step(acc_t (&acc)[NACC], const unsigned char *buf, int rofs) {
    using R = WgRows<NB, W>;
    if constexpr (R::any) {
      // B operands one block ahead of their products, and no further: the compiler would otherwise
      // hoist every read of the row to the top (12 registers per block) and spill the accumulators
      const Op A0 = read_op(buf, R::r0, rofs);
      Op A1 = A0, B = A0;
#pragma unroll
      for (int bj = R::r0; bj < NB; ++bj) {
        Op Bn = B;
        if (bj + 1 < NB) Bn = read_op(buf, bj + 1, rofs);  // (two blocks ahead bought nothing: the step is bound by the matrix pipe)
        asm volatile("" ::: "memory");
        if (R::two && bj == R::r1) A1 = B;
        acc[bj - R::r0] = mma6(A0, B, acc[bj - R::r0]);
        if (R::two && bj >= R::r1) acc[R::n0 + bj - R::r1] = mma6(A1, B, acc[R::n0 + bj - R::r1]);
        B = Bn;
      }
    }
  }
  static __device__ __forceinline__ void mma_step_w(int wave, acc_t (&acc)[NACC], const unsigned char *buf, int rofs) {
#ifdef YCNR_WG_ABLATE_MMASTEP  // timing experiments only
    return;
#endif
    switch (wave) {
      case 0: mma_step<0>(acc, buf, rofs); break;
      case 1: mma_step<1>(acc, buf, rofs); break;
      case 2: mma_step<2>(acc, buf, rofs); break;
      case 3: mma_step<3>(acc, buf, rofs); break;
      case 4: mma_step<4>(acc, buf, rofs); break;
      case 5: mma_step<5>(acc, buf, rofs); break;
      case 6: mma_step<6>(acc, buf, rofs); break;
      default: mma_step<7>(acc, buf, rofs); break;
    }
  }

  // accumulator tiles of wave W -> image layout at `img` (LDS or a slab in global memory), float units
  template <int W, typename P>
  static __device__ __forceinline__ void store_tiles(const acc_t (&acc)[NACC], P img, int lane) {
    using R = WgRows<NB, W>;
    const int off = wg_tile_lane_off(lane) >> 2;
    if constexpr (R::any) {
#pragma unroll
      for (int bj = R::r0; bj < NB; ++bj) *reinterpret_cast<wg_f32x4 *>(&img[tile_index(R::r0, bj, NB) * 256 + off]) = acc[bj - R::r0];
      if constexpr (R::two) {
#pragma unroll
        for (int bj = R::r1; bj < NB; ++bj)
          *reinterpret_cast<wg_f32x4 *>(&img[tile_index(R::r1, bj, NB) * 256 + off]) = acc[R::n0 + bj - R::r1];
      }
    }
  }
  template <typename P>
  static __device__ __forceinline__ void store_tiles_w(int wave, const acc_t (&acc)[NACC], P img, int lane) {
    switch (wave) {
      case 0: store_tiles<0>(acc, img, lane); break;
      case 1: store_tiles<1>(acc, img, lane); break;
      case 2: store_tiles<2>(acc, img, lane); break;
      case 3: store_tiles<3>(acc, img, lane); break;
      case 4: store_tiles<4>(acc, img, lane); break;
      case 5: store_tiles<5>(acc, img, lane); break;
      case 6: store_tiles<6>(acc, img, lane); break;
      default: store_tiles<7>(acc, img, lane); break;
    }
  }

  // Gramian of ratings [beg, beg + n): on return the workgroup's accumulators hold the tiles, bacc
  // this thread's partial sums of b, and all waves have passed the last barrier (LDS is free).
  // pre0 / pre1 / m2: steps 0 and 1 already requested by the caller (prefetch during the previous
  // row's solve) and the ids of step 2.
  static __device__ __forceinline__ void run(const StepArgs<float> &a, int64_t beg, int64_t n, unsigned char *smem, acc_t (&acc)[NACC],
                                             float (&bacc)[4][4], Stage &s0, Stage &s1, Meta &m2) {
    const int tid = threadIdx.x, lane = tid & 63, l16 = tid & 15, rho = tid >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t nsteps = (n + 31) >> 5;
    // writer: chunk of (rho, p = l16 & 3) in the region of block (l16 >> 2) [+ 4 j]
    const int wofs = (l16 >> 2) * C::REGION +
                     8 * ((l16 & 3) + 4 * (rho & 3) + 16 * ((rho >> 3) & 1) + 32 * ((rho >> 2) & 1) + 64 * (rho >> 4));
    // reader: lane 4q + p of a 16-lane group supplies row q, column quad p; group g reads ratings 8g .. 8g+3 (+4)
    const int rofs = 8 * (lane & 31) + 512 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) bacc[j][e] = 0.0f;
    unsigned char *buf0 = smem, *buf1 = smem + C::BUF;
    // phase -1: planes of step 0; rows of step 2; ids of step 3
    split_store(s0, bacc, buf0, wofs, l16);
    load_rows(s0, a, m2, l16);
    Meta m3 = load_meta(a, beg, n, 3, rho);
    __syncthreads();
    // Within a phase a wave writes one plane buffer and reads the other, in either order.  Waves 0-3
    // split first and multiply second, waves 4-7 the other way round: the two waves of a SIMD (w and
    // w + 4) then keep its vector ALU and its matrix pipe busy at the same time instead of both
    // splitting (matrix pipe idle) and then both multiplying.
    const bool splitFirst = wave < 4;
    for (int64_t s = 0; s < nsteps; s += 2) {
      // phase s: planes of step s + 1 (always written: zeros past the end), products of step s
      if (splitFirst) {
        split_store(s1, bacc, buf1, wofs, l16);
        load_rows(s1, a, m3, l16);                      // step s + 3
      }
      Meta m4 = load_meta(a, beg, n, s + 4, rho);
      mma_step_w(wave, acc, buf0, rofs);
      if (!splitFirst) {
        split_store(s1, bacc, buf1, wofs, l16);
        load_rows(s1, a, m3, l16);
      }
      __syncthreads();
      if (s + 1 >= nsteps) break;
      // phase s + 1
      if (splitFirst) {
        split_store(s0, bacc, buf0, wofs, l16);         // step s + 2
        load_rows(s0, a, m4, l16);                      // step s + 4
      }
      m3 = load_meta(a, beg, n, s + 5, rho);
      mma_step_w(wave, acc, buf1, rofs);
      if (!splitFirst) {
        split_store(s0, bacc, buf0, wofs, l16);
        load_rows(s0, a, m4, l16);
      }
      __syncthreads();
    }
  }

  // this thread's b partials -> bvec (NB * 16 floats in LDS); ends with a barrier
  static __device__ __forceinline__ void reduce_b(float (&bacc)[4][4], unsigned char *smem) {
    const int tid = threadIdx.x, lane = tid & 63, l16 = tid & 15;
    const int wave = tid >> 6;
    float *bpart = reinterpret_cast<float *>(smem + C::BPART_OFF);
    float *bvec = reinterpret_cast<float *>(smem + C::VEC_OFF);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (64 * j >= NB * 16) continue;
      wg_f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = bacc[j][e];
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        v[e] = t;
      }
      const int col = 64 * j + 4 * l16;
      if (lane < 16 && col < NB * 16) *reinterpret_cast<wg_f32x4 *>(bpart + wave * (NB * 16) + col) = v;
    }
    __syncthreads();
    if (tid < NB * 16) {
      float t = 0.0f;
#pragma unroll
      for (int w = 0; w < kWgWaves; ++w) t += bpart[w * (NB * 16) + tid];
      bvec[tid] = t;
    }
    __syncthreads();
  }
};

// Block Cholesky + substitutions on the LDS image.  In: tiles (upper triangle, image layout) at smem,
// bvec; lam I not yet added.  Out: x written to a.solved[row].  All 512 threads call it.
template <int NB>
struct WgSolve {
  using C = WgCfg<NB>;
  using Tr = MfmaTraits<float>;
  using acc_t = typename Tr::acc_t;
  using Sm = SolveMfmaF32<1>;

  static __device__ __forceinline__ float *tile(float *S, int bi, int bj) { return S + tile_index(bi, bj, NB) * 256; }
  static __device__ __forceinline__ acc_t ld(const float *T, int off) { return *reinterpret_cast<const acc_t *>(T + off); }
  static __device__ __forceinline__ void st(float *T, int off, const acc_t &v) { *reinterpret_cast<acc_t *>(T + off) = v; }

  // y[c] = sum_r M[r][c] v[r] for the tile in registers (lane (g, c): M[4g+t][c]), v given per row
  // (vrow[t] = v[4g + t]); result in every lane group
  static __device__ __forceinline__ float matvec_t(const acc_t &m, const acc_t &vrow) {
    float s = m[0] * vrow[0];
    s = fmaf(m[1], vrow[1], s);
    s = fmaf(m[2], vrow[2], s);
    s = fmaf(m[3], vrow[3], s);
    return Sm::group_sum(s);
  }
  // y[4g+t] = sum_c M[4g+t][c] v[c], v given per column (vc = v[c]); result in every lane of group g
  static __device__ __forceinline__ acc_t matvec_n(const acc_t &m, float vc) {
    float y0 = m[0] * vc, y1 = m[1] * vc, y2 = m[2] * vc, y3 = m[3] * vc;
    Sm::row_sum4(y0, y1, y2, y3);
    return acc_t{y0, y1, y2, y3};
  }

  // value of lane N of the caller's 16-lane row, in every lane of the row (DPP row_newbcast)
  template <int N>
  static __device__ __forceinline__ float row_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + N, 0xF, 0xF, true));
  }
  template <int P, int Jn>
  struct Pivot {
    static __device__ __forceinline__ void updates(float (&R)[16], float (&X)[16]) {
      if constexpr (Jn < 16) {
        const float m = row_bcast<Jn>(R[P]);  // L[Jn][P]
        R[Jn] = fmaf(-R[P], m, R[Jn]);
        X[Jn] = fmaf(-X[P], m, X[Jn]);
        Pivot<P, Jn + 1>::updates(R, X);
      }
    }
  };
  template <int P>
  static __device__ __forceinline__ void pivot(float (&R)[16], float (&X)[16], float &dmin) {
    const float d = row_bcast<P>(R[P]);  // D[P][P] after the updates of the pivots before it
    dmin = fminf(dmin, d);
    const float rs = __builtin_amdgcn_rsqf(d);
    R[P] *= rs;  // column P of L (lane i: L[i][P])
    X[P] *= rs;
    Pivot<P, P + 1>::updates(R, X);
  }

  // wave 0: factor the diagonal tile (J, J) in place: on return it holds V = U_JJ^-1 (upper triangular,
  // image layout), i.e. image[c][r] = W[c][r] with W = L^-1.  Returns true when a real pivot was not positive.
  // Lane (g, i) holds row i of D in R (every 16-lane row carries a copy) and row i of the identity
  // in X; the column operations that turn D into L turn the identity into L^-T.  The multipliers
  // L[j][p] come from lane j of the lane's own row by DPP (row_newbcast): no v_readlane, no SGPR hops.
  static __device__ __forceinline__ bool factor_diag(float *S, float *Dt, int J, int k, int lane) {
    const int g = lane >> 4, c = lane & 15;
    float *T = tile(S, J, J);
    // Round 2: the pivot loop of the one-wave solve (SolveMfmaF32::pivots_dpp).  Lane groups 0 / 2 hold the rows
    // of D, groups 1 / 3 the columns of the identity that become L^-1; one v_permlane16_swap per pivot copies
    // the scaled pivot column into the identity groups and every updated column is ONE v_fmac with a DPP
    // row_newbcast operand (before: every lane carried a row of D and a row of the identity, two multiply-adds
    // per column -- these 16 pivots are the critical chain of the late block steps).
    float R[16];
    {
      const int sw = (c >> 1) & 3;
      const bool xlane = (g & 1) != 0;
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) {
        const acc_t v = *reinterpret_cast<const acc_t *>(T + c * 16 + 4 * (m4 ^ sw));  // row c = column c (D is symmetric)
#pragma unroll
        for (int t = 0; t < 4; ++t) R[4 * m4 + t] = xlane ? (c == 4 * m4 + t ? 1.0f : 0.0f) : v[t];
      }
    }
    float dmin = 3.0e38f;
    // padded pivots (index >= k; only the last tile has any) are rows of the identity (diagonal 1, nothing
    // else): their scale is 1 and their multipliers 0, so running them is exact
    Sm::template pivots_dpp<0, 16>(R, dmin);
    (void)k;
    // groups 1 / 3: R[j] = L^-1[j][c] = W[j][c] = element (row c, col j) of the tile; group 1 writes columns
    // 0 .. 7, group 3 columns 8 .. 15
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the tile reads above have returned before the tile is overwritten
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if ((g & 1) && (j >> 3) == (g >> 1)) T[j * 16 + 4 * ((c >> 2) ^ ((j >> 1) & 3)) + (c & 3)] = R[j];
    (void)Dt;
    return !(dmin > 0.0f);
  }

  static __device__ void run(const StepArgs<float> &a, int32_t row, int64_t nRatings, unsigned char *smem) {
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = a.k;
    float *S = reinterpret_cast<float *>(smem);
    float *bvec = reinterpret_cast<float *>(smem + C::VEC_OFF), *zvec = bvec + NB * 16, *xvec = zvec + NB * 16;
    float *Dt = reinterpret_cast<float *>(smem + C::DT_OFF);
    int *flag = reinterpret_cast<int *>(smem + C::FLAG_OFF), *rowCounter = flag + 1;
    const float lam = (float)(a.lambda * (double)nRatings);
    const int off = wg_tile_lane_off(lane) >> 2;
    // diagonal: + lam on real indices, 1 on padded ones (also the zero columns a row padded to a multiple of 4
    // carries: with lam there, a regularisation of 0 would make their pivots 0)
    const int kDiag = a.kReal > 0 ? a.kReal : k;
    for (int i = tid; i < NB * 16; i += kWgThreads) {
      const int r = i & 15;
      tile(S, i >> 4, i >> 4)[r * 16 + 4 * ((r >> 2) ^ ((r >> 1) & 3)) + (r & 3)] += (i < kDiag) ? lam : 1.0f;
    }
    if (tid == 0) *flag = 0;
    YCNR_STAMP(a, 0);
    __syncthreads();
    YCNR_STAMP(a, 1);
    bool bad = false;
#ifndef YCNR_WG_ABLATE_FACTOR  // timing experiments only
    if (wave == 0) bad = factor_diag(S, Dt, 0, k, lane);
#endif
    YCNR_STAMP(a, 2);
    __syncthreads();
    YCNR_STAMP(a, 3);
    // The right-hand side is block column NB of the matrix: "tile" (bi, NB) is the 16 x 16 matrix with
    // b_bi in column 0, kept as 16 floats (lanes c == 0 carry it, the others zeros), so that
    // z_J = W b_J is one more panel tile and b_bi -= U[J][bi]^T z_J one more trailing update --
    // MFMAs like the rest, no cross-lane sums.
    const bool c0 = c == 0;
    auto ld_rhs = [&](const float *vec, int blk) {
      const acc_t v = *reinterpret_cast<const acc_t *>(vec + blk * 16 + 4 * g);
      return c0 ? v : acc_t{0.0f, 0.0f, 0.0f, 0.0f};
    };
    for (int J = 0; J < NB; ++J) {
      // ---- panel: U[J][bj] = W T[J][bj] for bj = J+1 .. NB-1, and z_J = W b_J (bj = NB)
      if (tid == 0) *rowCounter = J + 1;  // rows of the trailing update that follows (read after the barrier)
      {
        const acc_t Wop = ld(tile(S, J, J), off);  // A operand: W[c][4g + q]
        // items bj = J+1 .. NB (NB: the right-hand side), at most two per wave (NB <= 16): both at once
        const int bjA = J + 1 + wave, bjB = bjA + kWgWaves;
        if (bjA <= NB) {
          const bool rA = bjA == NB, hasB = bjB <= NB, rB = bjB == NB;
          float *TA = tile(S, J, rA ? J : bjA), *TB = tile(S, J, (rB || !hasB) ? J : bjB);
          const acc_t BA = rA ? ld_rhs(bvec, J) : ld(TA, off);  // B operand: T[4g + q][c]
          const acc_t BB = rB ? ld_rhs(bvec, J) : ld(TB, off);
          acc_t PA = acc_t{0.0f, 0.0f, 0.0f, 0.0f}, PB = PA;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            PA = Tr::mma(Wop[q], BA[q], PA);
            PB = Tr::mma(Wop[q], BB[q], PB);
          }
          if (!rA) st(TA, off, PA);
          else if (c0) *reinterpret_cast<acc_t *>(zvec + J * 16 + 4 * g) = PA;
          if (hasB) {
            if (!rB) st(TB, off, PB);
            else if (c0) *reinterpret_cast<acc_t *>(zvec + J * 16 + 4 * g) = PB;
          }
        }
      }
      YCNR_STAMP(a, 4 + 4 * J);
      __syncthreads();
      YCNR_STAMP(a, 5 + 4 * J);
      if (J + 1 == NB) break;
      // ---- trailing update with look-ahead.  Wave 0 updates tile (J+1, J+1) FIRST and factors it; block
      // rows bi = J+1 .. NB-1 of the update (items bj = bi .. NB, bj = NB: the right-hand side) are
      // handed out by a counter in LDS, longest first, to whichever wave is free -- wave 0 joins when
      // its factorization is done.  A row's panel operand is loaded once, its items go two at a time
      // so that their LDS round trips and dependent MFMA chains overlap.
      if (wave == 0) {
        float *T = tile(S, J + 1, J + 1);
        const acc_t Pi = ld(tile(S, J, J + 1), off);
        acc_t t = ld(T, off);
#pragma unroll
        for (int q = 0; q < 4; ++q) t = Tr::mma(-Pi[q], Pi[q], t);
        st(T, off, t);
        __builtin_amdgcn_s_waitcnt(0xc07f);
#ifndef YCNR_WG_ABLATE_FACTOR
        bad = factor_diag(S, Dt, J + 1, k, lane) || bad;
#endif
      }
#ifndef YCNR_WG_ABLATE_TRAIL
      // (Tried in round 3: keeping the wave that shares wave 0's SIMD -- or any other single wave -- out of the trailing
      // update while wave 0 factors the next diagonal tile: 1.5 ms SLOWER per C5 shard whichever wave it was; the
      // pivot chain is not stretched by the partner's MFMAs.)
      {
        auto next_row = [&]() {
          int r = 0;
          if (lane == 0) r = __hip_atomic_fetch_add(rowCounter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          return __builtin_amdgcn_readfirstlane(r);
        };
        const acc_t Zop = ld_rhs(zvec, J);
        int bi = next_row();
        while (bi < NB) {
          const int nxt = next_row();  // in flight while this row is worked on
          acc_t nPi = ld(tile(S, J, bi), off);
#pragma unroll
          for (int q = 0; q < 4; ++q) nPi[q] = -nPi[q];
          // tiles (bi, b0 .. NB-1) lie 1 KB apart in the image, and so do the panel tiles (J, b0 .. NB-1):
          // four at a time off two address registers with immediate offsets.  Past the end of the row
          // the loads run into the next tiles (valid LDS; the products are dropped).
          const int b0 = bi == J + 1 ? bi + 1 : bi, nt = NB - b0;
          const float *Pp = tile(S, J, b0) + off;
          float *Tp = tile(S, bi, b0) + off;
          for (int i = 0; i < nt; i += 4) {
            acc_t Pj[4], t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              Pj[u] = *reinterpret_cast<const acc_t *>(Pp + 256 * (i + u));
              t[u] = *reinterpret_cast<const acc_t *>(Tp + 256 * (i + u));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int u = 0; u < 4; ++u) t[u] = Tr::mma(nPi[q], Pj[u][q], t[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (i + u < nt) *reinterpret_cast<acc_t *>(Tp + 256 * (i + u)) = t[u];
          }
          {  // the right-hand side: b_bi -= U[J][bi]^T z_J
            acc_t t = ld_rhs(bvec, bi);
#pragma unroll
            for (int q = 0; q < 4; ++q) t = Tr::mma(nPi[q], Zop[q], t);
            if (c0) *reinterpret_cast<acc_t *>(bvec + bi * 16 + 4 * g) = t;
          }
          bi = nxt;
        }
      }
#endif
      YCNR_STAMP(a, 6 + 4 * J);
      __syncthreads();
      YCNR_STAMP(a, 7 + 4 * J);
    }
    YCNR_STAMP(a, 80);
    // ---- back substitution, right-looking: x_J = V_J z_J; z_bi -= U[bi][J] x_J for bi < J
#ifdef YCNR_WG_ABLATE_BACK
    for (int J = -1; J >= 0; --J) {
#else
    for (int J = NB - 1; J >= 0; --J) {
#endif
      if (wave == 0) {
        const acc_t V = ld(tile(S, J, J), off);
        const acc_t x = matvec_n(V, zvec[J * 16 + c]);
        if (c == 0) *reinterpret_cast<acc_t *>(xvec + J * 16 + 4 * g) = x;
      }
      __syncthreads();
      if (J == 0) break;
      {
        const float xc = xvec[J * 16 + c];
        for (int bi = wave; bi < J; bi += kWgWaves) {
          const acc_t U = ld(tile(S, bi, J), off);
          const acc_t d = matvec_n(U, xc);
          if (c == 0) {
            acc_t z = *reinterpret_cast<acc_t *>(zvec + bi * 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < 4; ++t) z[t] -= d[t];
            *reinterpret_cast<acc_t *>(zvec + bi * 16 + 4 * g) = z;
          }
        }
      }
      __syncthreads();
    }
    YCNR_STAMP(a, 81);
    float *out = a.solved + (int64_t)row * k;
    float chk = 0.0f;
    for (int i = tid; i < k; i += kWgThreads) {
      const float x = xvec[i];
      out[i] = x;
      chk = fmaf(x, 0.0f, chk);
    }
    if (!(chk == 0.0f)) atomicOr(flag, 1);  // NaN / Inf in the input ends up in x
    if (bad && lane == 0) atomicOr(flag, 1);
    __syncthreads();
    if (tid == 0 && *flag) {
      atomicAdd(&a.err->count, 1);
      a.err->firstRow = row;
    }
  }
};

// whole rows: Gramian + solve, persistent over units [firstFused + blockIdx.x, firstFused + count) by gridDim.x
template <int NB>
__global__ __launch_bounds__(kWgThreads, 2) void als_wg_gram_solve_kernel(StepArgs<float> a, int32_t count) {
  using G = WgGram<NB>;
  using C = WgCfg<NB>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l16 = tid & 15, rho = tid >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  typename G::Stage s0, s1;
  typename G::Meta m2;
  int32_t ui = blockIdx.x;
  if (ui >= count) return;
  Unit u = a.units[a.firstFused + ui];
  auto prefetch = [&](const Unit &v) {
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
    G::store_tiles_w(wave, acc, reinterpret_cast<float *>(smem), lane);
    G::reduce_b(bacc, smem);
    const int32_t row = u.row;
    const int64_t n = u.end - u.beg;
    ui += gridDim.x;
    const bool more = ui < count;
    if (more) {
      u = a.units[a.firstFused + ui];
      prefetch(u);  // the next row's first steps land during this row's solve
    }
    WgSolve<NB>::run(a, row, n, smem);
    if (!more) break;
    __syncthreads();  // the image is dead: the next row's planes may overwrite it
  }
  (void)sizeof(C);
}

// chunks of heavy rows: Gramian -> slab (image layout + b), persistent over units [0, count)
template <int NB>
__global__ __launch_bounds__(kWgThreads, 2) void als_wg_gram_slab_kernel(StepArgs<float> a, int32_t count) {
  using G = WgGram<NB>;
  using C = WgCfg<NB>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l16 = tid & 15, rho = tid >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int32_t ui = blockIdx.x; ui < count; ui += gridDim.x) {
    const Unit u = a.units[ui];
    const int64_t n = u.end - u.beg;
    typename G::Stage s0, s1;
    const typename G::Meta m0 = G::load_meta(a, u.beg, n, 0, rho), m1 = G::load_meta(a, u.beg, n, 1, rho);
    typename G::Meta m2 = G::load_meta(a, u.beg, n, 2, rho);
    G::load_rows(s0, a, m0, l16);
    G::load_rows(s1, a, m1, l16);
    typename G::acc_t acc[G::NACC];
    float bacc[4][4];
    G::run(a, u.beg, n, smem, acc, bacc, s0, s1, m2);
    float *slab = a.slabs + (int64_t)u.slab * C::SLAB_FLOATS;
    G::store_tiles_w(wave, acc, slab, lane);
    G::reduce_b(bacc, smem);
    if (tid < NB * 16) slab[(int64_t)C::NT * 256 + tid] = reinterpret_cast<const float *>(smem + C::VEC_OFF)[tid];
    __syncthreads();
  }
}

// split rows: slabs summed in slab order -> image -> solve, persistent over split rows [0, count)
template <int NB>
__global__ __launch_bounds__(kWgThreads, 2) void als_wg_reduce_solve_kernel(StepArgs<float> a, int32_t count) {
  using C = WgCfg<NB>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  float *S = reinterpret_cast<float *>(smem);
  float *bvec = reinterpret_cast<float *>(smem + C::VEC_OFF);
  for (int32_t si = blockIdx.x; si < count; si += gridDim.x) {
    const SplitRow sr = a.split[si];
    constexpr int NQ = C::NT * 64;  // float4s of the image
    // rows of more than kFewSlabs slabs: the sum over the slabs in double, rounded once (GramPlain::Wide in als_kernels.hip.h: a
    // float32 chain over the 4096 slabs of a 10 M-rating item carries 37 eps of the total); the form follows the row's own
    // slab count, so it is the same however the rows are dealt to GPUs and pieces
    const bool wide = sr.nslabs > kFewSlabs;  // uniform over the workgroup
    for (int q0 = tid; q0 < NQ; q0 += 4 * kWgThreads) {
      wg_f32x4 v[4];
      if (wide) {
        double d[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int e = 0; e < 4; ++e) d[u][e] = 0.0;
        }
        for (int sl = 0; sl < sr.nslabs; ++sl) {
          const wg_f32x4 *src = reinterpret_cast<const wg_f32x4 *>(a.slabs + (int64_t)(sr.slab0 + sl) * C::SLAB_FLOATS);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * kWgThreads;
            if (q < NQ) {
              const wg_f32x4 x = src[q];
#pragma unroll
              for (int e = 0; e < 4; ++e) d[u][e] += (double)x[e];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = wg_f32x4{(float)d[u][0], (float)d[u][1], (float)d[u][2], (float)d[u][3]};
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = wg_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int sl = 0; sl < sr.nslabs; ++sl) {
          const wg_f32x4 *src = reinterpret_cast<const wg_f32x4 *>(a.slabs + (int64_t)(sr.slab0 + sl) * C::SLAB_FLOATS);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * kWgThreads;
            if (q < NQ) v[u] += src[q];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * kWgThreads;
        if (q < NQ) reinterpret_cast<wg_f32x4 *>(S)[q] = v[u];
      }
    }
    if (tid < NB * 16) {
      if (wide) {
        double t = 0.0;
        for (int sl = 0; sl < sr.nslabs; ++sl) t += (double)a.slabs[(int64_t)(sr.slab0 + sl) * C::SLAB_FLOATS + (int64_t)C::NT * 256 + tid];
        bvec[tid] = (float)t;
      } else {
        float t = 0.0f;
        for (int sl = 0; sl < sr.nslabs; ++sl) t += a.slabs[(int64_t)(sr.slab0 + sl) * C::SLAB_FLOATS + (int64_t)C::NT * 256 + tid];
        bvec[tid] = t;
      }
    }
    __syncthreads();
    WgSolve<NB>::run(a, sr.row, sr.n, smem);
    __syncthreads();
  }
}

}  // namespace ycnr
