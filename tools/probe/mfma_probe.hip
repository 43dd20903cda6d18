// Stand-alone probe: how much of the fp32 MFMA rate survives each ingredient of wide_gemm?
//   V0 registers only; V1 + ds_read_b128 A fragments; V2 + global 16-byte B fragments (ring of R);
//   blocks/CU = 1,2,4 (4 waves each).  Prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int LDX = 260, HID = 256;

template <int V, int R>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ W, float* out, int nch, int iters) {
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  for (int k = threadIdx.x; k < 32 * LDX; k += 256) Xs[k] = 0.001f * (k & 255);
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const float* xa = Xs + i * LDX + h * (nch * 4);
  const int lane_off = ((h * nch) * HID + 64 * w + i) * 4;
  f32x4 ring[R][2];
  f32x4 a_reg = {1.f, 2.f, 3.f, 4.f};
  f32x4 b_reg0 = {0.5f, 0.25f, 0.125f, 1.f}, b_reg1 = {0.3f, 0.2f, 0.1f, 1.f};
  for (int it = 0; it < iters; ++it) {
    if (V >= 2) {
#pragma unroll
      for (int j = 0; j < R - 1; ++j) { const float* wn = W + (size_t)j * (HID * 4); ring[j][0] = *(const f32x4*)(wn + lane_off); ring[j][1] = *(const f32x4*)(wn + lane_off + 128); }
    }
    for (int c0 = 0; c0 < nch; c0 += R) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int c = c0 + j;
        if (c < nch) {
          if (V >= 2 && c + R - 1 < nch) { const float* wn = W + (size_t)(c + R - 1) * (HID * 4); ring[(j + R - 1) % R][0] = *(const f32x4*)(wn + lane_off); ring[(j + R - 1) % R][1] = *(const f32x4*)(wn + lane_off + 128); }
          __builtin_amdgcn_sched_barrier(0);
          f32x4 av = a_reg;
          if (V >= 1) av = *(const f32x4*)(xa + 4 * c);
          f32x4 b0 = b_reg0, b1 = b_reg1;
          if (V >= 2) { b0 = ring[j][0]; b1 = ring[j][1]; }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b0[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b1[u], acc1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V, int R>
double run(const float* W, float* out, int blocks_per_cu, int nch, int iters) {
  const size_t lds = 32 * LDX * 4 + (blocks_per_cu == 1 ? 100 * 1024 : blocks_per_cu == 2 ? 40 * 1024 : 0);
  hipFuncSetAttribute((const void*)probe<V, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<V, R>), dim3(grid), dim3(256), lds, 0, W, out, nch, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<V, R>), dim3(grid), dim3(256), lds, 0, W, out, nch, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 /*waves*/ * iters * nch * 8 /*mfma*/ * 4096.0;
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  float *W, *out;
  hipMalloc(&W, 256 * 256 * 4 * 4); hipMemset(W, 0, 256 * 256 * 4 * 4);
  hipMalloc(&out, 1024 * 256 * 4 * 4);
  const int nch = 32, iters = 40;
  for (int bpc : {1, 2, 4}) {
    printf("blocks/CU %d:  V0(reg) %.1f TF | V1(+lds A) %.1f | V2(+global B, ring5) %.1f | V2 ring3 %.1f | V2 ring9 %.1f\n", bpc,
           run<0, 5>(W, out, bpc, nch, iters), run<1, 5>(W, out, bpc, nch, iters), run<2, 5>(W, out, bpc, nch, iters),
           run<2, 3>(W, out, bpc, nch, iters), run<2, 9>(W, out, bpc, nch, iters));
  }
  return 0;
}
