// How fast does the chip start 64-thread workgroups?  The dual-form kernels launch one wave per row
// (470 K workgroups for the rows of <= 16 ratings at MAL scale): this measures an empty kernel, one that
// makes the three dependent loads every row kernel starts with (unit -> column id -> factor row), and the
// same work done by a persistent grid whose waves loop over the rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Unit { long long beg, end; int row, slab; };
__global__ __launch_bounds__(64) void k_empty(float *out) {
  if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = 1.0f;
}
__global__ __launch_bounds__(64) void k_chain(const Unit *units, const int *indx, const float *fixed, float *out, int k) {
  const Unit u = units[blockIdx.x];
  const int c = threadIdx.x & 15;
  const int n = (int)(u.end - u.beg);
  const int id = indx[u.beg + (c < n ? c : n - 1)];
  const float4 y = *reinterpret_cast<const float4 *>(fixed + (long long)id * k + 4 * (threadIdx.x >> 4));
  if (threadIdx.x < 25) *reinterpret_cast<float4 *>(out + (long long)u.row * k + 4 * threadIdx.x) = y;
}
__global__ __launch_bounds__(64) void k_loop(const Unit *units, const int *indx, const float *fixed, float *out, int k, int rows) {
  const int c = threadIdx.x & 15;
  int r = blockIdx.x;
  if (r >= rows) return;
  Unit u = units[r];
  int n = (int)(u.end - u.beg);
  int id = indx[u.beg + (c < n ? c : n - 1)];
  for (; r < rows; r += gridDim.x) {
    const int rn = r + gridDim.x;
    Unit un = u;
    int idn = id;
    if (rn < rows) {
      un = units[rn];
      const int nn = (int)(un.end - un.beg);
      idn = indx[un.beg + (c < nn ? c : nn - 1)];
    }
    const float4 y = *reinterpret_cast<const float4 *>(fixed + (long long)id * k + 4 * (threadIdx.x >> 4));
    if (threadIdx.x < 25) *reinterpret_cast<float4 *>(out + (long long)u.row * k + 4 * threadIdx.x) = y;
    u = un;
    id = idn;
  }
}
int main() {
  const int rows = 470000, k = 100, items = 12700, per = 12;
  std::vector<Unit> us(rows);
  std::vector<int> indx((size_t)rows * per);
  for (int r = 0; r < rows; ++r) {
    us[r] = Unit{(long long)r * per, (long long)r * per + per, r, -1};
    for (int j = 0; j < per; ++j) indx[(size_t)r * per + j] = (int)(((long long)r * 7919 + j * 104729) % items);
  }
  Unit *dU; int *dI; float *dF, *dO;
  hipMalloc(&dU, sizeof(Unit) * rows); hipMalloc(&dI, indx.size() * 4); hipMalloc(&dF, (size_t)items * k * 4); hipMalloc(&dO, (size_t)rows * k * 4);
  hipMemcpy(dU, us.data(), sizeof(Unit) * rows, hipMemcpyHostToDevice); hipMemcpy(dI, indx.data(), indx.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dF, 0, (size_t)items * k * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char *name, auto launch) {
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-40s %8.1f us  (%.2f G rows/s)\n", name, best * 1000, rows / best / 1e6);
  };
  time("empty, one workgroup per row", [&] { k_empty<<<rows, 64>>>(dO); });
  time("unit -> id -> row, one workgroup per row", [&] { k_chain<<<rows, 64>>>(dU, dI, dF, dO, k); });
  for (int g : {2048, 4096, 8192, 16384, 32768})
    time(("looping grid of " + std::to_string(g)).c_str(), [&] { k_loop<<<g, 64>>>(dU, dI, dF, dO, k, rows); });
  return 0;
}
