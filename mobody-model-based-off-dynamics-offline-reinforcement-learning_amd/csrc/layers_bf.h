// Layer-level helpers of the split-precision modes (tile_bf.h): an fp32 layer whose epilogue writes bf16 planes, and a
// plane-fed 256 x 256 layer whose epilogue writes the fp32 image; shared by mlp_fwd_bf.hip and dynamics.hip.
#pragma once
#include <type_traits>

#include "layers.h"
#include "tile_bf.h"

namespace mobody {

template <int MT>
__device__ __forceinline__ void relu_mask_words(f32x16 (&acc)[MT][2], float bias0, float bias1, uint32_t* mask, int mask_groups) {
  const int i = lane_id() & 31, hh = lane_id() >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float bias = nt ? bias1 : bias0;
      uint32_t word = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) word |= (uint32_t)((acc[mt][nt][r] + bias) > 0.f) << ((r & 3) + 8 * (r >> 2) + 4 * hh);
      word |= (uint32_t)__shfl_xor((int)word, 32);
      const int grp = MT * wave_rg() + mt;
      if (hh == 0 && grp < mask_groups) mask[grp * HID + 64 * wave_col() + 32 * nt + i] = word;
    }
}

// fp32 layer whose output goes to NPL bf16 planes (rows_total rows per plane) instead of the fp32 image
template <int ACT, int MT, int NPL, class Extra, class Between>
__device__ __forceinline__ void wide_layer_to_planes(float* Xs, __bf16* Ps, int rows_total, const float* __restrict__ W,
                                                     const float* __restrict__ b, int Kp, WideRing& ring, Extra&& extra,
                                                     Between&& between, uint32_t* mask, bool full, int mask_groups) {
  const float bias0 = b[64 * wave_col() + (lane_id() & 31)], bias1 = b[64 * wave_col() + 32 + (lane_id() & 31)];
  f32x16 acc[MT][2];
  wide_zero<MT>(acc);
  wide_gemm<MT>(Xs, W, Kp, acc, ring);
  TR(6);
  between();
  lds_barrier();
  auto body = [&](auto guarded) {
    wide_foreach<MT>(acc, [&](int row, int col, float v) {
      const float y = activate<ACT>(v + ((col & 32) ? bias1 : bias0));
      __bf16 t[NPL];
      bf_split<NPL>(y, t);
#pragma unroll
      for (int p = 0; p < NPL; ++p) Ps[((size_t)p * rows_total + row) * LDP + col] = t[p];
      extra(guarded, row, col, y);
    });
  };
  if (full) body(std::false_type{});
  else body(std::true_type{});
  TR(7);
  if (mask != nullptr) relu_mask_words<MT>(acc, bias0, bias1, mask, mask_groups);
  lds_barrier();
}

// split-precision layer: planes -> fp32 image
template <int ACT, int MT, int NPL, class Extra, class Between>
__device__ __forceinline__ void bf_layer(float* Xs, const __bf16* Ps, int rows_total, const bf16x8* __restrict__ Wb,
                                         const float* __restrict__ b, BfRing<NPL>& ring, Extra&& extra, Between&& between,
                                         uint32_t* mask, bool full, int mask_groups) {
  const float bias0 = b[64 * wave_col() + (lane_id() & 31)], bias1 = b[64 * wave_col() + 32 + (lane_id() & 31)];
  f32x16 acc[MT][2];
  wide_zero<MT>(acc);
  bf_gemm<MT, NPL>(Ps, rows_total, Wb, acc, ring);
  TR(3);
  between();
  lds_barrier();
  auto body = [&](auto guarded) {
    wide_foreach<MT>(acc, [&](int row, int col, float v) {
      const float y = activate<ACT>(v + ((col & 32) ? bias1 : bias0));
      Xs[row * LDX + col] = y;
      extra(guarded, row, col, y);
    });
  };
  if (full) body(std::false_type{});
  else body(std::true_type{});
  if (mask != nullptr) relu_mask_words<MT>(acc, bias0, bias1, mask, mask_groups);
  lds_barrier();
}

}  // namespace mobody
