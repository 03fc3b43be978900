#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *base, unsigned bytes, float *out) {
  int l = threadIdx.x;
  __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
  out[l] = __builtin_amdgcn_raw_buffer_load_b32(srd, l * 4, 0, 0);
  out[64 + l] = __builtin_amdgcn_raw_buffer_load_b32(srd, l * 4, 64, 0);
  out[128 + l] = __builtin_amdgcn_raw_buffer_load_b32(srd, 0x80000000u, 64, 0);
  out[192 + l] = __builtin_amdgcn_raw_buffer_load_b32(srd, bytes + l * 4, 0, 0);
}
int main() {
  float h[1024], *d, *o, r[256];
  for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  (void)hipMalloc(&d, 4096); (void)hipMalloc(&o, 1024);
  (void)hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 2048u, o);
  (void)hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
  printf("plain %g %g | soffset64 %g %g | oob %g %g | past-end %g %g (num_records 2048 B)\n", r[0], r[63], r[64], r[127], r[128], r[191], r[192], r[255]);
}
