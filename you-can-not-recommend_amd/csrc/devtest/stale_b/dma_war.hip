// Candidate cause of the stale right-hand side (README.md): in the failing assembly the copy of block A's freshly read values
// is   buffer_load_dwordx4 v82, s[4:7], s42 offen lds ; v_mov_b32 v82, v66   -- a VALU write to the OFFSET register of an
// LDS-DMA load in the issue slot directly behind it.  This probe issues exactly that pair (PAD s_nop's between the two,
// NDMA loads in front of the write, every wave of a full grid at it) and counts, per lane,
//   out[lane]       how often the register did NOT hold the moved value afterwards  (the write was lost)
//   out[64 + lane]  how often the 16 bytes the DMA put into LDS were not the lane's own 16 bytes of src  (the address was
//                   taken after the write)
// Usage: ./dma_war [iterations per wave = 2000] [workgroups = 4096]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_ptr;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAD, int NDMA>
__global__ __launch_bounds__(256) void k(const unsigned *src, unsigned *out, unsigned bytes, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned ring[4][NDMA][256];  // per wave: NDMA slots of 1 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4 srd = {(unsigned)(uintptr_t)src, (unsigned)((uintptr_t)src >> 32) & 0xFFFFu, bytes, 0x00020000u};
  unsigned lost = 0, wrongData = 0;
  const unsigned m0base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_ptr)&ring[wave][0][0]);
  for (int it = 0; it < iters; ++it) {
    // this iteration's 1 KB of src (a different one per wave and iteration) and the value moved over the offset register
    const unsigned base = ((blockIdx.x * 4u + wave) * 131u + it * 17u) % (bytes / 1024u) * 1024u;
    unsigned off = base + lane * 16u;
    const unsigned val = 0xC0DE0000u ^ (it << 8) ^ lane;
    unsigned reg = off;
    if constexpr (NDMA == 2)
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %0, %2, 0 offen lds\n\t.rept %5\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %0, %1"
                   : "+v"(reg) : "v"(val), "s"(srd), "s"(m0base), "s"(m0base + 1024u), "n"(PAD) : "memory");
    else
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds\n\t.rept %4\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %0, %1"
                   : "+v"(reg) : "v"(val), "s"(srd), "s"(m0base), "n"(PAD) : "memory");
    lost += reg != val;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int d = 0; d < NDMA; ++d)
      for (int j = 0; j < 4; ++j) wrongData += ring[wave][d][lane * 4 + j] != src[(off >> 2) + j];
    __builtin_amdgcn_s_barrier();  // (keeps the waves of a workgroup in step: four DMA issuers per CU slot at a time)
  }
  if (lost) atomicAdd(out + lane, lost);
  if (wrongData) atomicAdd(out + 64 + lane, wrongData);
}

template <int PAD, int NDMA>
static void run(const unsigned *src, unsigned *out, unsigned bytes, int iters, int wgs) {
  (void)hipMemset(out, 0, 128 * 4);
  hipLaunchKernelGGL((k<PAD, NDMA>), dim3(wgs), dim3(256), 0, 0, src, out, bytes, iters);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
  unsigned h[128];
  (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  unsigned long long lost = 0, data = 0, lostHi = 0;
  for (int l = 0; l < 64; ++l) lost += h[l], data += h[64 + l], lostHi += l >= 48 ? h[l] : 0;
  printf("pad %d, %d DMA: lost writes %llu (lanes 48..63: %llu), wrong LDS dwords %llu, of %llu pairs\n", PAD, NDMA, lost, lostHi, data,
         (unsigned long long)iters * wgs * 256);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000, wgs = argc > 2 ? atoi(argv[2]) : 4096;
  const unsigned bytes = 64u << 20;
  unsigned *src, *out;
  (void)hipMalloc(&src, bytes);
  (void)hipMalloc(&out, 128 * 4);
  std::vector<unsigned> h(bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
  (void)hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
  run<0, 1>(src, out, bytes, iters, wgs);
  run<0, 2>(src, out, bytes, iters, wgs);
  run<1, 2>(src, out, bytes, iters, wgs);
  run<2, 2>(src, out, bytes, iters, wgs);
  return 0;
}
