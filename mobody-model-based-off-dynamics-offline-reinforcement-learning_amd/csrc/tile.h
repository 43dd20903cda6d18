// CDNA4 (gfx950) tile primitives shared by every MOBODY kernel.
//
// Geometry (fixed for the whole library):
//   * workgroup = 256 threads = 4 wave64, one row-tile of BM = 64 minibatch rows;
//   * activations of the current layer live in ONE LDS image X[64][LDX] fp32
//     (LDX = 260: 260 mod 64 == 4, so the ds_read_b128 A-fragment reads of 16
//     different rows fall on 16 different 16-byte bank slots -> conflict free);
//   * "wide" layers (N = 256 = hidden width H) use v_mfma_f32_32x32x2_f32:
//     wave w owns output columns [64w, 64w+64) of all 64 rows = 2x2 tiles of
//     32x32, 64 accumulator VGPRs; A fragments come from LDS (one b128 = four
//     k-steps), B fragments straight from global/L2 (weights are [K][N] row
//     major, so a wave-instruction reads two full 128-byte lines);
//   * "narrow" layers (N <= 128: latent heads, action/Q/reward/next-state
//     outputs, input-gradients) use v_mfma_f32_16x16x4_f32: wave w owns rows
//     [16w, 16w+16), so chains of narrow layers are wave-local (no barrier).
//   The contraction index is split between lane groups instead of interleaved:
//   lane half h (wide) / quarter q (narrow) walks its own contiguous K range, so a
//   lane's successive k-steps are contiguous in LDS.  Any k order is a valid fp32
//   sum; MFMA f32 is an exact fma chain (MI355X guide: FP32-input MFMA).
//
// Reference semantics implemented on top of these tiles are cited at each kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mobody {

constexpr int BM = 64;        // rows per workgroup tile
constexpr int HID = 256;      // hidden width (reference: hidden_dims=256 train_mobody.py:794, hidden_sizes: 256 yaml)
constexpr int LDX = 260;      // LDS leading dimension (floats)
constexpr int NTHREADS = 256;
constexpr int LATENT = 16;    // mobody_module.py:95
constexpr int NENS = 7;       // mobody_module.py:247 hard-codes 7

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SWISH = 2 };

template <int ACT>
__device__ __forceinline__ float activate(float x) {
  if (ACT == ACT_RELU) return fmaxf(x, 0.f);
  if (ACT == ACT_SWISH) return x / (1.f + __expf(-x));   // x*sigmoid(x), mobody_module.py:13-15
  return x;
}

// --------------------------------------------------------------------------------------------
// wide GEMM:  acc[mt][nt] (+)= X[64 x Kp] (LDS) * W[Kp x 256] (global, row major, ld = 256)
// Kp multiple of 8.  Columns of this wave: 64*w + 32*nt + (lane&31).
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void wide_zero(f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}

__device__ __forceinline__ void wide_gemm(const float* __restrict__ Xs, const float* __restrict__ W, int Kp,
                                          f32x16 (&acc)[2][2]) {
  const int lane = lane_id(), w = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const int kh = Kp >> 1;                       // K range of this lane half, multiple of 4
  const float* xa0 = Xs + i * LDX + h * kh;
  const float* xa1 = xa0 + 32 * LDX;
  const float* wb = W + (size_t)(h * kh) * HID + 64 * w + i;
  float b0[4], b1[4], nb0[4], nb1[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { b0[u] = wb[u * HID]; b1[u] = wb[u * HID + 32]; }
  for (int t = 0; t < kh; t += 4) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(xa0 + t);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(xa1 + t);
    const bool more = (t + 4) < kh;
    const float* wn = wb + (size_t)(more ? t + 4 : t) * HID;   // prefetch next four k-rows (re-read last on the tail)
#pragma unroll
    for (int u = 0; u < 4; ++u) { nb0[u] = wn[u * HID]; nb1[u] = wn[u * HID + 32]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b1[u], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b0[u], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { b0[u] = nb0[u]; b1[u] = nb1[u]; }
  }
}

// Visit every accumulator element of a wide result: f(row 0..63, col 0..255, value).
template <class F>
__device__ __forceinline__ void wide_foreach(f32x16 (&acc)[2][2], F&& f) {
  const int lane = lane_id(), w = wave_id();
  const int i = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;   // C/D map of 32x32 MFMA
        const int col = 64 * w + 32 * nt + i;
        f(row, col, acc[mt][nt][r]);
      }
}

// --------------------------------------------------------------------------------------------
// narrow GEMM: rows [16w,16w+16) of X (LDS, Kp cols) times W[Kp x Np] (global, ld = Np),
// NT consecutive 16-column tiles starting at column tile nt0.  Kp multiple of 8.
// acc[nt][r]: row = 16w + 4*(lane>>4) + r, col = 16*(nt0+nt) + (lane&15).
// --------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void narrow_gemm(const float* __restrict__ Xs, const float* __restrict__ W, int Kp, int Np,
                                            int nt0, f32x4 (&acc)[NT]) {
  const int lane = lane_id(), w = wave_id();
  const int i = lane & 15, q = lane >> 4;
  const int kq = Kp >> 2;                       // K range of this lane quarter, multiple of 2
  const float* xa = Xs + (16 * w + i) * LDX + q * kq;
  const float* wb = W + (size_t)(q * kq) * Np + 16 * nt0 + i;
  f32x4 acc2[NT];                               // second chain hides the 40-cycle dependent latency
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[n][r] = 0.f; acc2[n][r] = 0.f; }
  for (int t = 0; t < kq; t += 2) {
    const float2 a = *reinterpret_cast<const float2*>(xa + t);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const float bx = wb[(size_t)t * Np + 16 * n];
      const float by = wb[(size_t)(t + 1) * Np + 16 * n];
      acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bx, acc[n], 0, 0, 0);
      acc2[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, by, acc2[n], 0, 0, 0);
    }
  }
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[n][r] += acc2[n][r];
}

// Run a narrow layer over all Np/16 column tiles in groups of <= 2; f(row 0..63, col, value).
template <class F>
__device__ __forceinline__ void narrow_layer(const float* __restrict__ Xs, const float* __restrict__ W, int Kp, int Np,
                                             F&& f) {
  const int lane = lane_id(), w = wave_id();
  const int i = lane & 15, q = lane >> 4;
  const int ntiles = Np >> 4;
  int nt0 = 0;
  for (; nt0 + 2 <= ntiles; nt0 += 2) {
    f32x4 acc[2];
    narrow_gemm<2>(Xs, W, Kp, Np, nt0, acc);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) f(16 * w + 4 * q + r, 16 * (nt0 + n) + i, acc[n][r]);
  }
  if (nt0 < ntiles) {
    f32x4 acc[1];
    narrow_gemm<1>(Xs, W, Kp, Np, nt0, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) f(16 * w + 4 * q + r, 16 * nt0 + i, acc[0][r]);
  }
}

// --------------------------------------------------------------------------------------------
// LDS tile fill: X[r][col0 + c] = src[(row0+r)*ld + c] for c < n (zero for rows >= rows).
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_load(float* Xs, int col0, const float* __restrict__ src, int ld, int n, int row0,
                                          int rows) {
  for (int idx = threadIdx.x; idx < BM * n; idx += NTHREADS) {
    const int r = idx / n, c = idx - r * n;
    const int gr = row0 + r;
    Xs[r * LDX + col0 + c] = (gr < rows) ? src[(size_t)gr * ld + c] : 0.f;
  }
}
__device__ __forceinline__ void tile_zero_cols(float* Xs, int c0, int c1) {
  const int n = c1 - c0;
  if (n <= 0) return;
  for (int idx = threadIdx.x; idx < BM * n; idx += NTHREADS) {
    const int r = idx / n, c = idx - r * n;
    Xs[r * LDX + c0 + c] = 0.f;
  }
}

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace mobody
