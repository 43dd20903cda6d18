// Layer-level helpers of the split-precision modes (tile_bf.h): an fp32 layer whose epilogue writes 16-bit planes, and a
// plane-fed 256 x 256 layer whose epilogue writes the fp32 image; shared by mlp_fwd_bf.hip and dynamics.hip.
// PM = precision mode (1 bf16, 2 bf16x2, 3 bf16x3, 4 f16x2), TB = rows of the workgroup's tile (32 or 64).
#pragma once
#include <type_traits>

#include "layers.h"
#include "tile_bf.h"

namespace mobody {

// sign words of a ReLU layer's OUTPUT y (y > 0 <=> pre-activation > 0); layout as wide_layer's masks
template <int MT>
__device__ __forceinline__ void relu_mask_words(f32x16 (&y)[MT][2], uint32_t* mask, int mask_groups) {
  const int i = lane_id() & 31, hh = lane_id() >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      uint32_t word = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) word |= (uint32_t)(y[mt][nt][r] > 0.f) << ((r & 3) + 8 * (r >> 2) + 4 * hh);
      word |= (uint32_t)__shfl_xor((int)word, 32);
      const int grp = MT * wave_rg() + mt;
      if (hh == 0 && grp < mask_groups) mask[grp * HID + 64 * wave_col() + 32 * nt + i] = word;
    }
}

// Optional global copy of a tile's planes for the weight-gradient GEMM (f16 mode): the SAME scaled terms that go to LDS, in
// the layout that GEMM reads as fragments -- plane[p][row / 8][256 columns][8 rows] (a lane's A/B fragment of
// v_mfma_f32_32x32x16_f16, eight consecutive rows of one column, is ONE 16-byte load; csrc/mlp_bwd.hip wgrad_tile_f16) --
// plus the tile's scale exponent.  `base` points at this tile's first 8-row block of plane 0.
struct PlaneSave {
  short* base;               // null: no copy
  long long plane_stride;    // 16-bit elements between planes (rows rounded up to 32, times 256)
  int* e_out;                // this tile's scale exponent
};

// Write the planes of a wide result held in accumulators (values y, scaled by 2^e in the f16 mode) and hand every element
// to extra(guarded, row, col, y): four consecutive rows of a lane's feature go out as one 8-byte store per plane.
template <int MT, int PM, int TB, class Extra, class Guard>
__device__ __forceinline__ void planes_from_acc(f32x16 (&acc)[MT][2], char* Ps, int e, Extra&& extra, Guard guarded,
                                                const PlaneSave& gs = PlaneSave{nullptr, 0, nullptr}) {
  const int lane = lane_id(), i = lane & 31, h = lane >> 5;
  const float sc = Split<PM>::F16 ? exp2i(e) : 1.f;
  if (gs.base != nullptr && (int)threadIdx.x < TB / 32) gs.e_out[threadIdx.x] = e;      // one exponent per 32 rows
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = 64 * wave_col() + 32 * nt + i;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float y4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y4[j] = Split<PM>::F16 ? acc[mt][nt][4 * g + j] * sc : acc[mt][nt][4 * g + j];
        planes_store4<PM, TB>(Ps, col, 8 * (MT * wave_rg() + mt) + 2 * g + h, y4, gs.base, gs.plane_stride);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          extra(guarded, 32 * (MT * wave_rg() + mt) + 8 * g + 4 * h + j, col, acc[mt][nt][4 * g + j]);
      }
    }
}

// fp32 layer whose output goes to planes instead of the fp32 image.  Returns the tile's scale exponent (planes hold
// y * 2^e; 0 outside the f16 mode).  `scr`: 8 floats of LDS for the tile maximum (f16 mode).
template <int ACT, int MT, int PM, int TB, class Extra, class Between>
__device__ __forceinline__ int wide_layer_to_planes(float* Xs, char* Ps, float* scr, const float* __restrict__ W,
                                                    const float* __restrict__ b, int Kp, WideRing& ring, Extra&& extra,
                                                    Between&& between, uint32_t* mask, bool full, int mask_groups,
                                                    int rows_here = 1 << 30, const PlaneSave& gs = PlaneSave{nullptr, 0, nullptr}) {
  const float bias0 = b[64 * wave_col() + (lane_id() & 31)], bias1 = b[64 * wave_col() + 32 + (lane_id() & 31)];
  f32x16 acc[MT][2];
  wide_zero<MT>(acc);
  wide_gemm<MT>(Xs, W, Kp, acc, ring);
  TR(6);
  between();
  float mx = 0.f;                                  // the activations in place; their largest magnitude for the f16 scale
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float y = activate<ACT>(acc[mt][nt][r] + (nt ? bias1 : bias0));
        // the global plane copy feeds a contraction over ROWS: rows past the end of the batch must be zero there
        if (gs.base != nullptr && !full && 32 * (MT * wave_rg() + mt) + (r & 3) + 8 * (r >> 2) + 4 * (lane_id() >> 5) >= rows_here) y = 0.f;
        acc[mt][nt][r] = y;
        if constexpr (Split<PM>::F16) mx = fmaxf(mx, fabsf(y));
      }
  if constexpr (Split<PM>::F16) f16_tile_max_put(mx, scr);
  lds_barrier();                                   // every wave has read the old image (and posted its maximum)
  int e = 0;
  if constexpr (Split<PM>::F16) e = f16_scale_exp(f16_tile_max_get(scr));
  if (full) planes_from_acc<MT, PM, TB>(acc, Ps, e, extra, std::false_type{}, gs);
  else planes_from_acc<MT, PM, TB>(acc, Ps, e, extra, std::true_type{}, gs);
  TR(7);
  if (mask != nullptr) relu_mask_words<MT>(acc, mask, mask_groups);
  lds_barrier();
  return e;
}

// split-precision layer: planes (scaled by 2^e_in in the f16 mode) -> fp32 image
template <int ACT, int MT, int PM, int TB, class Extra, class Between>
__device__ __forceinline__ void bf_layer(float* Xs, const char* Ps, int e_in, const s16x8* __restrict__ Wb,
                                         const float* __restrict__ b, BfRing<PM>& ring, Extra&& extra, Between&& between,
                                         uint32_t* mask, bool full, int mask_groups) {
  const float bias0 = b[64 * wave_col() + (lane_id() & 31)], bias1 = b[64 * wave_col() + 32 + (lane_id() & 31)];
  f32x16 acc[MT][2];
  wide_zero<MT>(acc);
  bf_gemm<MT, PM, TB>(Ps, Wb, acc, ring);
  TR(3);
  between();
  lds_barrier();
  const float inv = Split<PM>::F16 ? exp2i(-(e_in + F16_WSHIFT)) : 1.f;     // exact: both scales are powers of two
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bias = nt ? bias1 : bias0;
        acc[mt][nt][r] = activate<ACT>(Split<PM>::F16 ? fmaf(acc[mt][nt][r], inv, bias) : acc[mt][nt][r] + bias);
      }
  auto body = [&](auto guarded) {
    wide_foreach<MT>(acc, [&](int row, int col, float y) {
      Xs[row * LDX + col] = y;
      extra(guarded, row, col, y);
    });
  };
  if (full) body(std::false_type{});
  else body(std::true_type{});
  if (mask != nullptr) relu_mask_words<MT>(acc, mask, mask_groups);
  lds_barrier();
}

// LDS bytes of a split-precision tile kernel: the larger of the fp32 image and the planes (they alias), plus the 8-float
// scratch of the tile maximum behind them
template <int PM, int TB>
constexpr size_t split_lds_bytes() {
  constexpr size_t f32b = (size_t)TB * LDX * sizeof(float), plb = (size_t)Split<PM>::NPL * plane_bytes<TB>();
  return (f32b > plb ? f32b : plb) + 32;
}
template <int PM, int TB>
constexpr size_t split_scr_offset() { return split_lds_bytes<PM, TB>() - 32; }

}  // namespace mobody
