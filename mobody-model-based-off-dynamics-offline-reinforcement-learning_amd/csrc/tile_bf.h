// Split-precision MFMA core for the 256 x 256 layers (throughput modes of BASELINE.json configs[1], "bf16 MFMA inputs /
// fp32 accumulate").  An fp32 value is carried as NPL bf16 terms x = x0 + x1 (+ x2) (8 mantissa bits each) and a product
// keeps every term pair (i, j) with i + j < NPL:
//     NPL = 1   "bf16"    1 product   plain bf16 inputs, ~3e-3 relative per product
//     NPL = 2   "bf16x2"  3 products  x0y0 + x0y1 + x1y0, ~2^-16 relative per product
//     NPL = 3   "bf16x3"  6 products  fp32-grade (2^-24)
// all accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (32 cycles per instruction against 64 for the K = 2 fp32 form:
// 16x the MACs per cycle).  Activations: NPL bf16 planes in LDS, row stride LDP = 264 (528 B = 132 dwords, 132 mod 64 = 4:
// the ds_read_b128 A-fragment reads of 16 rows fall on 16 different 16-byte bank slots).  Weights: NPL planes in the T
// blob, K-interleaved by eight ([K/8][256][8] bf16 per plane), so a lane's B fragment (eight consecutive k of one column)
// is one 16-byte load and a wave instruction reads 1 KB contiguous.  Lane maps (MI355X guide, "A/B operand lane maps"):
// lane (r = lane & 31, h = lane >> 5) holds A[row r][k = 8h + j] and B[k = 8h + j][col r], j = 0..7; C/D as the fp32 form.
#pragma once
#include "tile.h"

namespace mobody {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int LDP = 264;                       // bf16 elements per plane row
constexpr int BF_RING = 3;                     // k16 steps of weight fragments in flight
constexpr long long BF_PLANE = 32LL * HID;     // bf16x8 units per weight plane ([K/8 = 32][256])

template <int NPL>
struct BfRing { bf16x8 r[BF_RING][NPL][2]; };

// element index (in bf16 units) of weight (k, n) inside plane p of a member's plane block
__host__ __device__ inline long long bf_plane_idx(int p, int k, int n) { return (((long long)p * 32 + (k >> 3)) * HID + n) * 8 + (k & 7); }

template <int NPL>
__device__ __forceinline__ void bf_split(float y, __bf16 (&t)[NPL]) {
  t[0] = (__bf16)y;
  if constexpr (NPL >= 2) { const float r1 = y - (float)t[0]; t[1] = (__bf16)r1;
    if constexpr (NPL >= 3) t[2] = (__bf16)(r1 - (float)t[1]); }
}

template <int NPL>
__device__ __forceinline__ void bf_ldb(const bf16x8* __restrict__ Wb, int s, bf16x8 (&b)[NPL][2]) {
  const int lane = lane_id(), r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int p = 0; p < NPL; ++p)
#pragma unroll
    for (int n = 0; n < 2; ++n) b[p][n] = Wb[p * BF_PLANE + (long long)(2 * s + h) * HID + 64 * wave_col() + 32 * n + r];
}

template <int NPL>
__device__ __forceinline__ void bf_prefetch(const bf16x8* __restrict__ Wb, BfRing<NPL>& ring) {
#pragma unroll
  for (int j = 0; j < BF_RING - 1; ++j) bf_ldb<NPL>(Wb, j, ring.r[j]);
  __builtin_amdgcn_sched_barrier(0);
}

// acc[mt][nt] += X (planes in LDS, rows_total rows per plane, K = 256) * W (planes in global).  `ring` holds bf_prefetch.
template <int MT, int NPL>
__device__ __forceinline__ void bf_gemm(const __bf16* __restrict__ Ps, int rows_total, const bf16x8* __restrict__ Wb,
                                        f32x16 (&acc)[MT][2], BfRing<NPL>& ring) {
  constexpr int R = BF_RING;
  const int lane = lane_id(), r = lane & 31, h = lane >> 5;
  const __bf16* xa = Ps + (size_t)(32 * MT * wave_rg() + r) * LDP + 8 * h;
  for (int s0 = 0; s0 < 16; s0 += R) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int s = s0 + j;
      if (s < 16) {
        if (s + R - 1 < 16) bf_ldb<NPL>(Wb, s + R - 1, ring.r[(j + R - 1) % R]);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 a[NPL][MT];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            a[p][m] = *reinterpret_cast<const bf16x8*>(xa + ((size_t)p * rows_total + 32 * m) * LDP + 16 * s);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            // smallest terms first
#pragma unroll
            for (int d = NPL - 1; d >= 0; --d)
#pragma unroll
              for (int i = 0; i <= d; ++i)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][m], ring.r[j][d - i][n], acc[m][n], 0, 0, 0);
          }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

}  // namespace mobody
