// Stand-alone probe for the split-precision core: fp32-equivalent TFLOP/s of a 256-deep GEMM step where every fp32 operand
// is three bf16 terms (x = x0 + x1 + x2) and a product is six v_mfma_f32_32x32x16_bf16 (x0y0 x0y1 x1y0 x0y2 x1y1 x2y0).
//   A: 3 bf16 planes in LDS [rows][256+8], one ds_read_b128 per plane / row tile / k16 step
//   B: 3 bf16 planes in global, K-interleaved by 8 ([K/8][256][8]), 16-byte loads kept R steps ahead in registers
//   workgroup = 4 waves, wave w owns columns [64w, 64w+64) of a 32*MT-row tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int LDA = 264;   // bf16 per LDS row

template <int MT, int R, int NPROD>
__global__ __launch_bounds__(256, 2) void probe(const bf16x8* __restrict__ W, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) __bf16 Xs[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  for (int k = threadIdx.x; k < 3 * 32 * MT * LDA; k += 256) Xs[k] = (__bf16)(0.001f * ((k * 37) & 255) - 0.1f);
  __syncthreads();
  f32x16 acc[MT][2];
  for (int a = 0; a < MT; ++a) for (int b = 0; b < 2; ++b) for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  const size_t plane = (size_t)32 * 256;          // bf16x8 units per plane: [K/8=32][256]
  auto ldb = [&](int s, bf16x8 (&b)[3][2]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int n = 0; n < 2; ++n) b[p][n] = W[p * plane + (size_t)(2 * s + h) * 256 + 64 * w + 32 * n + r];
  };
  bf16x8 ring[R][3][2];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < R - 1; ++j) ldb(j, ring[j]);
    for (int s0 = 0; s0 < 16; s0 += R) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int s = s0 + j;
        if (s < 16) {
          if (s + R - 1 < 16) ldb(s + R - 1, ring[(j + R - 1) % R]);
          __builtin_amdgcn_sched_barrier(0);
          bf16x8 a[3][MT];
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int m = 0; m < MT; ++m)
              a[p][m] = *reinterpret_cast<const bf16x8*>(Xs + ((size_t)p * 32 * MT + 32 * m + r) * LDA + 16 * s + 8 * h);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
              auto& c = acc[m][n];
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], ring[j][0][n], c, 0, 0, 0);
              if (NPROD >= 3) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], ring[j][1][n], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][m], ring[j][0][n], c, 0, 0, 0);
              }
              if (NPROD >= 6) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], ring[j][2][n], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][m], ring[j][1][n], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][m], ring[j][0][n], c, 0, 0, 0);
              }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  float sum = 0.f;
  for (int a = 0; a < MT; ++a) for (int b = 0; b < 2; ++b) for (int q = 0; q < 16; ++q) sum += acc[a][b][q];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int MT, int R, int NPROD>
double run(const bf16x8* W, float* out, int blocks_per_cu, int iters) {
  const size_t need = (size_t)3 * 32 * MT * LDA * 2;
  size_t lds = need;
  const size_t want = blocks_per_cu == 1 ? 120 * 1024 : blocks_per_cu == 2 ? 70 * 1024 : blocks_per_cu == 3 ? 50 * 1024 : 36 * 1024;
  if (lds < want) lds = want;           // pin residency through the LDS footprint
  hipFuncSetAttribute((const void*)probe<MT, R, NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MT, R, NPROD>), dim3(grid), dim3(256), lds, 0, W, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MT, R, NPROD>), dim3(grid), dim3(256), lds, 0, W, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * iters * (32.0 * MT) * 256 * 256 * 2;      // fp32-equivalent
  return flops / (ms * 1e-3) / 1e12;
}

// Same tile (32 rows x 64 columns per wave, six products) on v_mfma_f32_16x16x32_bf16: 2 x 4 accumulators of 16 x 16; lane
// (c = lane & 15, q = lane >> 4) holds A[row c][k = 32 s + 8 q + j] and B[k = 32 s + 8 q + j][col c].  Same bytes and the
// same number of LDS reads / global loads per K as the 32x32x16 form -- only the shape (and the clock the chip holds) differs.
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int R, int NPROD>
__global__ __launch_bounds__(256, 2) void probe16(const bf16x8* __restrict__ W, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) __bf16 Xs[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
  for (int k = threadIdx.x; k < 3 * 32 * LDA; k += 256) Xs[k] = (__bf16)(0.001f * ((k * 37) & 255) - 0.1f);
  __syncthreads();
  f32x4 acc[2][4];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.f;
  const size_t plane = (size_t)32 * 256;
  auto ldb = [&](int s, bf16x8 (&b)[3][4]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int n = 0; n < 4; ++n) b[p][n] = W[p * plane + (size_t)(4 * s + q) * 256 + 64 * w + 16 * n + c];
  };
  bf16x8 ring[R][3][4];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < R - 1; ++j) ldb(j, ring[j]);
    for (int s0 = 0; s0 < 8; s0 += R) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int s = s0 + j;
        if (s < 8) {
          if (s + R - 1 < 8) ldb(s + R - 1, ring[(j + R - 1) % R]);
          __builtin_amdgcn_sched_barrier(0);
          bf16x8 a[3][2];
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int m = 0; m < 2; ++m)
              a[p][m] = *reinterpret_cast<const bf16x8*>(Xs + ((size_t)p * 32 + 16 * m + c) * LDA + 32 * s + 8 * q);
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              auto& d = acc[m][n];
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][m], ring[j][0][n], d, 0, 0, 0);
              if (NPROD >= 3) {
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][m], ring[j][1][n], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][m], ring[j][0][n], d, 0, 0, 0);
              }
              if (NPROD >= 6) {
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][m], ring[j][2][n], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][m], ring[j][1][n], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][m], ring[j][0][n], d, 0, 0, 0);
              }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  float sum = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int e = 0; e < 4; ++e) sum += acc[a][b][e];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int R, int NPROD>
double run16(const bf16x8* W, float* out, int blocks_per_cu, int iters) {
  size_t lds = (size_t)3 * 32 * LDA * 2;
  const size_t want = blocks_per_cu == 1 ? 120 * 1024 : blocks_per_cu == 2 ? 70 * 1024 : blocks_per_cu == 3 ? 50 * 1024 : 36 * 1024;
  if (lds < want) lds = want;
  hipFuncSetAttribute((const void*)probe16<R, NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe16<R, NPROD>), dim3(grid), dim3(256), lds, 0, W, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe16<R, NPROD>), dim3(grid), dim3(256), lds, 0, W, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (double)grid * iters * 32.0 * 256 * 256 * 2 / (ms * 1e-3) / 1e12;
}

int main() {
  bf16x8* W; float* out;
  hipMalloc(&W, 3 * 32 * 256 * 16);
  {                                                     // random weights: all-zero operands let the chip hold a higher clock
    const size_t n = (size_t)3 * 32 * 256 * 8;
    unsigned short* hbuf = (unsigned short*)malloc(n * 2);
    unsigned x = 12345u;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; hbuf[i] = (unsigned short)(0x3C00u + ((x >> 16) & 0x01FFu) + ((x >> 31) << 15)); }
    hipMemcpy(W, hbuf, n * 2, hipMemcpyHostToDevice);
    free(hbuf);
  }
  hipMalloc(&out, 1024 * 256 * 4 * 4);
  const int iters = 40;
  for (int bpc : {1, 2, 3, 4}) {
    printf("blocks/CU %d (fp32-equivalent TF):  6-product MT1 R3 %.0f | MT2 R3 %.0f | MT2 R2 %.0f | MT4 R2 %.0f || 3-product MT2 R3 %.0f | 1-product MT2 R3 %.0f\n", bpc,
           run<1, 3, 6>(W, out, bpc, iters), run<2, 3, 6>(W, out, bpc, iters), run<2, 2, 6>(W, out, bpc, iters),
           bpc <= 2 ? run<4, 2, 6>(W, out, bpc, iters) : 0.0, run<2, 3, 3>(W, out, bpc, iters), run<2, 3, 1>(W, out, bpc, iters));
  }
  for (int bpc : {1, 2, 3, 4})
    printf("blocks/CU %d, 32-row tile, random data:  32x32x16 six products R3 %.0f | 16x16x32 six products R3 %.0f R2 %.0f | one product: 32x32x16 %.0f, 16x16x32 %.0f\n", bpc,
           run<1, 3, 6>(W, out, bpc, iters), run16<3, 6>(W, out, bpc, iters), run16<2, 6>(W, out, bpc, iters), run<1, 3, 1>(W, out, bpc, iters), run16<3, 1>(W, out, bpc, iters));
  return 0;
}
