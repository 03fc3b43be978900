// als_rowpair_kernels.hip.h -- the fused row kernel with SPECIALISED waves (round 5).
//
// What a whole row costs in als_gram_solve_x6p_kernel (EmfWorker.mw_calcTrainAlsPortion, lib/emf/EmfWorker.js:214-248, one
// wave per row, k = 100): a Gramian phase that is bound by the bf16 matrix pipe (145 MFMAs + ~40 vector instructions per 32
// ratings) and a solve phase that is bound by the vector ALU (pivot chains, DPP updates) and the float32 MFMAs, which run at
// the vector ALU's rate.  With two such waves per SIMD the two phases meet by chance: PMC of round 4 has the matrix pipe
// busy 57 % and the vector ALU 41 % of the kernel's 7.4 ms at MAL scale -- their SUM is the kernel's time.
//
// Here the two waves of a SIMD have fixed roles and work on consecutive rows:
//   G  gathers the planes of row j + 1 and accumulates its Gramian (GramX6P::accumulate, unchanged), folds it into the tile set
//      the solver expects and hands the tiles over through LDS;
//   S  takes the tiles of row j into its registers and solves (SolveMfmaF32, unchanged but for wave-level synchronisation),
// so that the matrix pipe (G) and the vector ALU (S) of the SIMD are busy at the same time, every row.  A workgroup is eight
// waves -- one G and one S per SIMD, 256 registers each: the whole register file of a CU -- and persistent: pair p of P takes
// the rows p, p + P, p + 2 P ... of the launch's whole rows (sorted by descending length: every pair sees the same mix).
//
// Which wave is which is decided on the machine: a wave reads the SIMD it runs on (HW_ID) and takes a ticket of that SIMD --
// the first is G, the second S -- so nothing depends on the order in which the dispatcher deals a workgroup's waves.
//
// Hand-over through ONE buffer per pair (28 tiles x 1 KB + 7 x 256 B at k = 100: the solver's registers as they are), which
// is also the Gramian's gather slots (19 KB): G starts row j + 1 once S has TAKEN row j (counter `taken`), publishes its
// tiles when its own gathers have drained (counter `full`).  Steady state: one row per max(T_G, T_S) and pair.  LDS per pair:
// 30.5 KB buffer + 2.5 KB solver image + counters = 33 KB, 132 KB per workgroup (of 160).
// Same products, same sums, same solve as als_gram_solve_x6p_kernel: bit-identical (tests/test_gpu_parity.py).
#pragma once
#include "als_kernels.hip.h"

namespace ycnr {

constexpr int kRowPairThreads = 512;

template <int NB, bool PACK>
struct RowPairCfg {
  using G = GramX6P<NB, PACK>;
  static constexpr int NT = tile_count(NB);
  static constexpr int BUF_DWORDS = (NT * 4 + NB) * 64;  // tiles as 64 lanes x 16 bytes, then the rhs partials
  static constexpr int SLOT_DWORDS = G::LDS_DWORDS;       // (alias the buffer)
  static constexpr int AREA_DWORDS = BUF_DWORDS > SLOT_DWORDS ? BUF_DWORDS : SLOT_DWORDS;
  static constexpr int IMG_DWORDS = (int)(SolveMfmaF32<NB, true>::lds_bytes() / 4);
  static constexpr int CTR_DWORDS = 16;                   // full, taken (+ padding to 64 bytes)
  static constexpr int PAIR_DWORDS = AREA_DWORDS + IMG_DWORDS + CTR_DWORDS;
  static constexpr int LDS_BYTES = (4 * PAIR_DWORDS + 16) * 4;  // + the SIMDs' ticket counters
};

// spin until the LDS counter at byte address `addr` has reached `want` (wave-uniform; the poll is one LDS read per ~100 cycles)
__device__ __forceinline__ void rowpair_wait(unsigned addr, int want) {
  for (;;) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    if (__builtin_amdgcn_readfirstlane(v) >= want) break;
    __builtin_amdgcn_s_sleep(2);
  }
}
// publish: everything this wave wrote to LDS before is visible to whoever reads the counter afterwards (LDS operations of a
// wave complete in order; the wait keeps the compiler's and the hardware's order explicit)
__device__ __forceinline__ void rowpair_post(unsigned addr, int value) {
  asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(value) : "memory");
}

template <int NB, bool PACK, bool E4>
__global__ __launch_bounds__(kRowPairThreads, 2) void als_row_pair_kernel(StepArgs<float> a, int32_t nPrimal) {
  using C = RowPairCfg<NB, PACK>;
  using G = typename C::G;
  using GD = GramX6D<NB, true, PACK>;  // (extract_rhs: the padded column leaves the tiles the same way)
  using Sv = SolveMfmaF32<NB, true>;
  using acc_t = typename G::acc_t;
  constexpr int NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_bytes[];
  unsigned *smem = reinterpret_cast<unsigned *>(smem_bytes);
  const int lane = threadIdx.x & 63;
  int *tickets = reinterpret_cast<int *>(smem + 4 * C::PAIR_DWORDS);
  if (threadIdx.x < 4) {
    tickets[threadIdx.x] = 0;
    smem[threadIdx.x * C::PAIR_DWORDS + C::AREA_DWORDS + C::IMG_DWORDS] = 0;      // full
    smem[threadIdx.x * C::PAIR_DWORDS + C::AREA_DWORDS + C::IMG_DWORDS + 1] = 0;  // taken
  }
  __syncthreads();  // (workgroup barriers: this one and the one behind the tickets, both before the waves part ways)
  // HW_ID bits 5:4: the SIMD this wave runs on
  const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;
  int role = 0;
  if (lane == 0) role = atomicAdd(&tickets[simd], 1);
  role = __builtin_amdgcn_readfirstlane(role);
  unsigned *area = smem + simd * C::PAIR_DWORDS;
  float *img = reinterpret_cast<float *>(area + C::AREA_DWORDS);
  const unsigned ctr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)(area + C::AREA_DWORDS + C::IMG_DWORDS);
  const unsigned fullAddr = ctr, takenAddr = ctr + 4;
  const int pairs = 4 * (int)gridDim.x, p0 = 4 * (int)blockIdx.x + simd;
  // Two waves per SIMD is what the register allocation enforces (the clobber below claims all 256 registers of a wave at two
  // per SIMD, whatever the instantiation needs); a SIMD with another count would leave a wave without its partner: checked by
  // the whole workgroup before anybody waits for anybody, reported as a failed row instead of a hang.
  asm volatile("" ::: "v255");
  __syncthreads();
  if (tickets[0] != 2 || tickets[1] != 2 || tickets[2] != 2 || tickets[3] != 2) {
    if (threadIdx.x == 0) {
      atomicAdd(&a.err->count, 1);
      a.err->firstRow = -1;
    }
    return;
  }
  float4 *buf4 = reinterpret_cast<float4 *>(area);
  float *bufb = reinterpret_cast<float *>(area) + NT * 256;
  if (role == 0) {
    // ---- G: gather + Gramian of rows p0, p0 + pairs, ...
    int j = 0;
    for (int ui = p0; ui < nPrimal; ui += pairs, ++j) {
      const Unit u = a.units[a.firstFused + ui];
      acc_t acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
      float bacc[NB];
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) bacc[cb] = 0.0f;
      rowpair_wait(takenAddr, j);  // S holds the tiles of the row before: the area is free for this row's gathers
      G::accumulate(acc, area, a.indx, a.vals, a.planes, a.planesBytes, a.k, u.beg, u.end - u.beg, lane);
      if constexpr (PACK) G::template fold_corner_rows<true>(acc[tile_index(NB - 1, NB - 1, NB)], reinterpret_cast<float *>(area), lane);
      GD::extract_rhs(acc, bacc, a.k, lane);
#pragma unroll
      for (int t = 0; t < NT; ++t) buf4[t * 64 + lane] = float4{acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) bufb[cb * 64 + lane] = bacc[cb];
      rowpair_post(fullAddr, j + 1);
    }
  } else {
    // ---- S: solve of the same rows, one behind
    int j = 0;
    for (int ui = p0; ui < nPrimal; ui += pairs, ++j) {
      const Unit u = a.units[a.firstFused + ui];
      const float lam = (float)(a.lambda * (double)(u.end - u.beg));
      acc_t acc[NT];
      float bacc[NB];
      rowpair_wait(fullAddr, j + 1);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 v = buf4[t * 64 + lane];
        acc[t] = acc_t{v.x, v.y, v.z, v.w};
      }
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) bacc[cb] = bufb[cb * 64 + lane];
      // (the loads above must have RETURNED before the area is handed back: G's gathers write into it)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t]));
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) asm volatile("" : "+v"(bacc[cb]));
      rowpair_post(takenAddr, j + 1);
      if constexpr (E4) {
        Sv::template run<true>(acc, bacc, img, a.k, lam, a.solved + (int64_t)u.row * a.k, u.row, a.err, lane, a.kReal > 0 ? a.kReal : -1);
      } else {
        Sv::run(acc, bacc, img, a.k, lam, a.solved + (int64_t)u.row * a.k, u.row, a.err, lane, a.kReal > 0 ? a.kReal : -1);
      }
    }
  }
}

}  // namespace ycnr
