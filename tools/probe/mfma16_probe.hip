// Stand-alone probe for a 16-row-granular GEMM core: wave tile = (16*MT) rows x 64 columns of 16x16x4 fp32 MFMA
// tiles, A fragments by ds_read_b128 (lane (i, q) owns k = 16c + 4q + u), B fragments by 16-byte global loads
// through a ring of R chunks (chunk = 16 k).  One workgroup (4 waves, 256 columns) per CU: the question is whether a
// single wave per SIMD sustains the MFMA pipe when it owns 4*MT independent accumulators.  Prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int LDX = 260, HID = 256;

template <int MT, int R, int V>
__global__ __launch_bounds__(256, 1) void probe16(const float* __restrict__ W, float* out, int nch, int iters) {
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  for (int k = threadIdx.x; k < 16 * MT * LDX; k += 256) Xs[k] = 0.001f * (k & 255);
  __syncthreads();
  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* xa = Xs + i * LDX + 4 * q;
  // B storage [chunk][q][n = 256][u = 4]: lane's four k of a chunk are 16 contiguous bytes
  const int lane_off = (q * HID + 64 * w + i) * 4;
  f32x4 ring[R][4];
  auto ldb = [&](int c, f32x4 (&b)[4]) {
    const float* wn = W + (size_t)c * (4 * HID * 4) + lane_off;
#pragma unroll
    for (int n = 0; n < 4; ++n) b[n] = *(const f32x4*)(wn + 16 * n * 4);
  };
  for (int it = 0; it < iters; ++it) {
    if (V >= 2) {
#pragma unroll
      for (int j = 0; j < R - 1; ++j) ldb(j, ring[j]);
    }
    for (int c0 = 0; c0 < nch; c0 += R) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int c = c0 + j;
        if (c < nch) {
          if (V >= 2 && c + R - 1 < nch) ldb(c + R - 1, ring[(j + R - 1) % R]);
          __builtin_amdgcn_sched_barrier(0);
          f32x4 av[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) av[m] = V >= 1 ? *(const f32x4*)(xa + 16 * m * LDX + 16 * c) : f32x4{1.f, 2.f, 3.f, 4.f};
          f32x4 b[4];
#pragma unroll
          for (int n = 0; n < 4; ++n) b[n] = V >= 2 ? ring[j][n] : f32x4{0.5f, 0.25f, 0.125f, 1.f};
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int n = 0; n < 4; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][u], b[n][u], acc[m][n], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MT, int R, int V>
double run(const float* W, float* out, int blocks_per_cu, int nch, int iters) {
  // pad the dynamic LDS so that exactly blocks_per_cu workgroups fit a CU
  const size_t need = (size_t)16 * MT * LDX * 4;
  const size_t lds = blocks_per_cu == 1 ? 100 * 1024 : blocks_per_cu == 2 ? 60 * 1024 : need;
  hipFuncSetAttribute((const void*)probe16<MT, R, V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe16<MT, R, V>), dim3(grid), dim3(256), lds > need ? lds : need, 0, W, out, nch, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe16<MT, R, V>), dim3(grid), dim3(256), lds > need ? lds : need, 0, W, out, nch, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * nch * (16.0 * MT) * 2048.0;      // MFMAs per wave and chunk: 4u * MT * 4n
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  float *W, *out;
  hipMalloc(&W, 256 * 256 * 4 * 4); hipMemset(W, 0, 256 * 256 * 4 * 4);
  hipMalloc(&out, 1024 * 256 * 4 * 4);
  const int nch = 16, iters = 60;        // K = 256 = 16 chunks of 16
  for (int bpc : {1, 2}) {
    printf("16x16x4 core, %d workgroup(s)/CU:\n", bpc);
    printf("  MT=5 (80 rows): reg %.1f TF | +lds A %.1f | +global B ring2 %.1f | ring3 %.1f | ring4 %.1f\n",
           run<5, 3, 0>(W, out, bpc, nch, iters), run<5, 3, 1>(W, out, bpc, nch, iters), run<5, 2, 2>(W, out, bpc, nch, iters),
           run<5, 3, 2>(W, out, bpc, nch, iters), run<5, 4, 2>(W, out, bpc, nch, iters));
    printf("  MT=4 (64 rows): reg %.1f TF | +lds A %.1f | +global B ring3 %.1f | ring4 %.1f\n",
           run<4, 3, 0>(W, out, bpc, nch, iters), run<4, 3, 1>(W, out, bpc, nch, iters), run<4, 3, 2>(W, out, bpc, nch, iters),
           run<4, 4, 2>(W, out, bpc, nch, iters));
    printf("  MT=2 (32 rows): reg %.1f TF | +lds A %.1f | +global B ring3 %.1f | ring5 %.1f\n",
           run<2, 3, 0>(W, out, bpc, nch, iters), run<2, 3, 1>(W, out, bpc, nch, iters), run<2, 3, 2>(W, out, bpc, nch, iters),
           run<2, 5, 2>(W, out, bpc, nch, iters));
  }
  return 0;
}
