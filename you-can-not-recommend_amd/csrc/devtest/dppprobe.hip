// Which way do DPP row shifts move data?  Prints, for lanes 0..15, what row_shr:5 and row_shl:5 of the lane id return
// (bound_ctrl: zeros for lanes with no source).  hipcc --offload-arch=gfx950 -o dppprobe dppprobe.hip && ./dppprobe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int *out) {
  const int lane = threadIdx.x;
  const int v = 100 + lane;
  out[lane] = __builtin_amdgcn_update_dpp(0, v, 0x115, 0xF, 0xF, true);       // row_shr:5
  out[64 + lane] = __builtin_amdgcn_update_dpp(0, v, 0x105, 0xF, 0xF, true);  // row_shl:5
}
int main() {
  int *d, h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("row_shr:5:");
  for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
  printf("\nrow_shl:5:");
  for (int i = 0; i < 16; ++i) printf(" %d", h[64 + i]);
  printf("\n");
  return 0;
}
