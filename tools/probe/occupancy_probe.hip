// How many 256-thread workgroups with V VGPRs and 33 KB of LDS does a gfx950 CU really hold?
// Each workgroup records its start time and then idles ~30 us; workgroups that start within the first 5 us were
// resident together.  Build: hipcc --offload-arch=gfx950 -O2 occupancy_probe.hip -o occupancy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int V>
__global__ __launch_bounds__(256) void k_probe(unsigned long long* start) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) start[blockIdx.x] = wall_clock64();
  // touch the highest register so the kernel descriptor asks for V VGPRs
  if constexpr (V == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if constexpr (V == 104) asm volatile("v_mov_b32 v103, 0" ::: "v103");
  if constexpr (V == 112) asm volatile("v_mov_b32 v111, 0" ::: "v111");
  if constexpr (V == 120) asm volatile("v_mov_b32 v119, 0" ::: "v119");
  if constexpr (V == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if constexpr (V == 136) asm volatile("v_mov_b32 v135, 0" ::: "v135");
  if constexpr (V == 144) asm volatile("v_mov_b32 v143, 0" ::: "v143");
  if constexpr (V == 152) asm volatile("v_mov_b32 v151, 0" ::: "v151");
  if constexpr (V == 160) asm volatile("v_mov_b32 v159, 0" ::: "v159");
  if constexpr (V == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
  if constexpr (V == 176) asm volatile("v_mov_b32 v175, 0" ::: "v175");
  if constexpr (V == 192) asm volatile("v_mov_b32 v191, 0" ::: "v191");
  if constexpr (V == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
  lds[threadIdx.x] = (float)threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 3000) __builtin_amdgcn_s_sleep(32);          // 30 us at 100 MHz
  if (lds[(threadIdx.x + 1) & 255] < 0.f) start[0] = 0;
}

template <int V>
static void run(size_t lds, unsigned long long* d, int nb) {
  hipFuncSetAttribute((const void*)k_probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_probe<V>, dim3(nb), dim3(256), lds, 0, d);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(nb);
  hipMemcpy(h.data(), d, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  const unsigned long long t0 = *std::min_element(h.begin(), h.end());
  int early = 0;
  for (auto t : h) early += (t - t0) < 500;
  printf("VGPRs %3d  LDS %6zu B: %4d of %d workgroups resident together = %.2f per CU\n", V, lds, early, nb, early / 256.0);
}

int main() {
  const int nb = 2560;
  unsigned long long* d;
  hipMalloc(&d, nb * sizeof(unsigned long long));
  for (size_t lds : {(size_t)33280, (size_t)1024}) {
    run<96>(lds, d, nb); run<104>(lds, d, nb); run<112>(lds, d, nb); run<120>(lds, d, nb); run<128>(lds, d, nb);
    run<136>(lds, d, nb); run<144>(lds, d, nb); run<152>(lds, d, nb); run<160>(lds, d, nb); run<168>(lds, d, nb);
    run<176>(lds, d, nb); run<192>(lds, d, nb); run<256>(lds, d, nb);
  }
  return 0;
}
