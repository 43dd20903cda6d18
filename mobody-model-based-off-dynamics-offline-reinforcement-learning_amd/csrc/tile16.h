// 16-row-granular GEMM core (v_mfma_f32_16x16x4_f32) for the fused MLP kernels: a workgroup of 4 waves owns a tile of
// 16*MT rows (MT = 1..9), wave w owns columns [64w, 64w+64) of ALL rows = MT x 4 tiles of 16x16, so a weight fragment
// is reused by MT row tiles (the 32x32x2 core of tile.h reuses it once or twice) and ONE workgroup per CU sustains the
// MFMA pipe: tools/probe/mfma16_probe.hip measures 138-140 TFLOP/s for MT = 4, 5 at one workgroup per CU against
// 109-122 for the 32x32 core at four.  MT is chosen per launch so that the grid is ~one workgroup per CU (no
// 2.5 -> 3 rounding of workgroups per CU).
//
// Fragment maps: lane (i = lane & 15, q = lane >> 4).  A: row 16m + i, k = 16c + 4q + u (one ds_read_b128 per row tile
// and chunk of 16 k);  B: k = 16c + 4q + u, column 64w + 16n + i -- in the K-interleaved weight storage (wide_idx) the
// four u are 16 contiguous bytes;  C: acc[m][n][r] = (row 16m + 4q + r, column 64w + 16n + i).
#pragma once
#include "tile.h"

namespace mobody {

constexpr int RING16 = 3;                           // chunks of 16 k in flight (3 x 4 x 16 B per lane)
struct Ring16 { f32x4 r[RING16][4]; };

__device__ __forceinline__ void ldb16(const float* __restrict__ W, int Kp, int c, f32x4 (&b)[4]) {
  const int lane = lane_id();
  const int i = lane & 15, q = lane >> 4;
  const int kq = min(16 * c + 4 * q, Kp - 4);       // partial last chunk: clamp to valid rows (their A is zeroed)
  const float* wn = W + ((size_t)(kq >> 2) * HID + 64 * wave_id() + i) * 4;
#pragma unroll
  for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const f32x4*>(wn + 64 * n);
}

__device__ __forceinline__ void prefetch16(const float* __restrict__ W, int Kp, Ring16& ring) {
  const int nch = (Kp + 15) >> 4;
#pragma unroll
  for (int j = 0; j < RING16 - 1; ++j)
    if (j < nch) ldb16(W, Kp, j, ring.r[j]);
  __builtin_amdgcn_sched_barrier(0);
}

template <int MT>
__device__ __forceinline__ void zero16(f32x4 (&acc)[MT][4]) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// acc += X[16 MT x Kp] (LDS) * W[Kp x 256];  Kp multiple of 4;  `ring` holds prefetch16(W, Kp).
template <int MT>
__device__ __forceinline__ void gemm16(const float* __restrict__ Xs, const float* __restrict__ W, int Kp,
                                       f32x4 (&acc)[MT][4], Ring16& ring) {
  constexpr int R = RING16;
  const int lane = lane_id();
  const int i = lane & 15, q = lane >> 4;
  const int nch = (Kp + 15) >> 4;
  const float* xa = Xs + i * LDX + 4 * q;
  auto mma = [&](int c, f32x4 (&b)[4]) {
    f32x4 av[MT];
    const bool kvalid = 16 * c + 4 * q < Kp;        // only false in a partial last chunk
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      av[m] = *reinterpret_cast<const f32x4*>(xa + 16 * m * LDX + 16 * c);
      if (!kvalid) av[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][u], b[n][u], acc[m][n], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  for (int c0 = 0; c0 < nch; c0 += R) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int c = c0 + j;
      if (c < nch) {
        if (c + R - 1 < nch) ldb16(W, Kp, c + R - 1, ring.r[(j + R - 1) % R]);
        __builtin_amdgcn_sched_barrier(0);          // keep the prefetch distance (see tile.h wide_gemm)
        mma(c, ring.r[j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// f(row, col, value) for every accumulator element of the wave.
template <int MT, class F>
__device__ __forceinline__ void foreach16(f32x4 (&acc)[MT][4], F&& f) {
  const int lane = lane_id();
  const int i = lane & 15, q = lane >> 4;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) f(16 * m + 4 * q + r, 64 * wave_id() + 16 * n + i, acc[m][n][r]);
}

}  // namespace mobody
