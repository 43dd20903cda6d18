// Ceiling of random 192-byte row reads from a 768 MB table (4 M rows x 48 floats), 1 M rows per launch:
// 16-lane groups, 12 lanes x 16 B per row, R rows in flight per group, no LDS, results reduced into one float per group.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
template <int R>
__global__ __launch_bounds__(256) void k(const float4* tab, const int* idx, int n, float* out, int rows_per_group) {
  const int lane = threadIdx.x & 15, g = (blockIdx.x * 256 + threadIdx.x) >> 4;
  const int q = lane < 12 ? lane : 11;
  float acc = 0.f;
  const int r0 = g * rows_per_group;
  for (int r = r0; r < r0 + rows_per_group && r < n; r += R) {
    float4 v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = tab[(size_t)idx[min(r + j, n - 1)] * 12 + q];
#pragma unroll
    for (int j = 0; j < R; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
  }
  if (acc == 123.456f) out[g] = acc;
}
int main() {
  const size_t rows = 4000000; const int n = 1000000;
  float4* tab; int* idx; float* out;
  hipMalloc(&tab, rows * 192); hipMalloc(&idx, n * 4); hipMalloc(&out, 4 << 20);
  hipMemset(tab, 0, rows * 192);
  std::vector<int> h(n); srand(1); for (auto& x : h) x = (int)(((size_t)rand() * 2147483647ull + rand()) % rows);
  hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
  auto run = [&](auto kern, int R, int rpg) {
    const int groups = (n + rpg - 1) / rpg, blocks = (groups + 15) / 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, idx, n, out, rpg);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, idx, n, out, rpg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("R=%d rows/group=%3d blocks=%6d: %7.1f us  %6.0f GB/s of 192-byte rows\n", R, rpg, blocks, ms * 1e3, n * 192.0 / (ms * 1e-3) / 1e9);
  };
  for (int rpg : {4, 16, 64}) { run(k<1>, 1, rpg); run(k<2>, 2, rpg); run(k<4>, 4, rpg); }
  run(k<8>, 8, 64); run(k<8>, 8, 16);
  return 0;
}
