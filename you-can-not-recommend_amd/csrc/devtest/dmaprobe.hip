// What does an out-of-range lane of `buffer_load_dwordx4 ... lds` do to its 16 LDS bytes: write zeros or
// leave them?  And does a lane masked off by EXEC leave them?  (als_gram_slab_x6d_kernel relies on the answer.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void *lds_ptr;
__global__ void k(const float *src, float *dst, unsigned bytes) {
  __shared__ __attribute__((aligned(16))) float ring[512];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) ring[i] = -7.0f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)bytes, 0x00020000);
  // lanes 0..31 in range, 32..47 out of range (offset >= num_records), 48..63 masked off
  unsigned off = lane < 32 ? (unsigned)lane * 16u : 0x80000000u;
  if (lane < 48) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr)ring, 16, off, 0, 0, 0);
  // second piece: dword size, all lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr)(ring + 256), 4, 0x80000000u + lane * 4, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 512; i += 64) dst[i] = ring[i];
}
int main() {
  float *src, *dst, h[512];
  (void)hipMalloc(&src, 4096); (void)hipMalloc(&dst, 2048);
  float init[1024]; for (int i = 0; i < 1024; ++i) init[i] = (float)i + 1;
  (void)hipMemcpy(src, init, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, dst, 4096u);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, dst, 2048, hipMemcpyDeviceToHost);
  printf("in range  lane 0: %g %g  lane 31: %g\n", h[0], h[1], h[31 * 4]);
  printf("out of range lane 32: %g %g  lane 47: %g\n", h[32 * 4], h[32 * 4 + 1], h[47 * 4]);
  printf("EXEC-masked lane 48: %g  lane 63: %g\n", h[48 * 4], h[63 * 4 + 3]);
  printf("dword piece, all out of range: %g %g %g\n", h[256], h[257], h[256 + 63]);
  return 0;
}
