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
      for (int r = 0; r < 16; ++r)                      // y is a ReLU output (>= +0, never -0): y > 0 <=> its bit pattern is non-zero --
        word |= min(__float_as_uint(y[mt][nt][r]), 1u) << ((r & 3) + 8 * (r >> 2) + 4 * hh);   // v_min_u32 + v_lshl_or instead of cmp + cndmask + lshl_or
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

// Row-major fp32 copy of a wide result held in accumulators to global memory (tile base dst, leading dimension 256): 32
// coalesced dword stores per lane as ONE branch-free burst when the tile is full; the per-element row guard (v_cmp + exec
// save / restore around every store) only on the batch's last, ragged tile.
#ifndef ROWS_BUFFER_STORES
#define ROWS_BUFFER_STORES 1
#endif
template <int MT>
__device__ __forceinline__ void wide_store_rows(f32x16 (&acc)[MT][2], float* dst, bool full, int rows_here) {
#if ROWS_BUFFER_STORES
  // through a descriptor over the tile's REAL rows: a row past the end of the batch is out of range and the hardware drops
  // its store (one code path for full and ragged tiles), and an address is one per-lane offset + a scalar row offset -- no
  // 64-bit vector add per store (row offsets reach 31 KB, beyond the instruction's immediate field)
  (void)full;
  const int lane = lane_id();
  const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(dst, (unsigned)min(rows_here, 1024) * (unsigned)(HID * 4));   // (tiles hold <= 64 rows)
  const int voff = ((32 * MT * wave_rg() + 4 * (lane >> 5)) * HID + 64 * wave_col() + (lane & 31)) * 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mt][nt][r]), rs, voff,
                                              ((32 * mt + (r & 3) + 8 * (r >> 2)) * HID + 32 * nt) * 4, 0);
#else
  if (full) wide_foreach<MT>(acc, [&](int row, int col, float y) { dst[row * HID + col] = y; });
  else wide_foreach<MT>(acc, [&](int row, int col, float y) { if (row < rows_here) dst[row * HID + col] = y; });
#endif
}

// Write the planes of a wide result held in accumulators (values y, scaled by 2^e in the f16 mode): four consecutive rows
// of a lane's feature go out as one 8-byte store per plane (and, optionally, one more to the global copy gs).
// `groups`: 32-row groups of the tile that hold real rows -- the global copy is padded to whole groups of 32 rows, not to whole
// tiles, so a taller tile's groups past the end of the batch must not be written (they would land in the next plane / member).
template <int MT, int PM, int TB>
__device__ __forceinline__ void planes_from_acc(f32x16 (&acc)[MT][2], char* Ps, int e, const PlaneSave& gs, int groups = 1 << 30) {
  const int lane = lane_id(), i = lane & 31, h = lane >> 5;
  const float sc = Split<PM>::F16 ? exp2i(e) : 1.f;
  if (gs.base != nullptr && (int)threadIdx.x < min(TB / 32, groups)) gs.e_out[threadIdx.x] = e;      // one exponent per 32 rows
  auto sweep = [&](short* gbase) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = 64 * wave_col() + 32 * nt + i;
        // (whole-vector arithmetic: the compiler emits v_pk_mul_f32, two elements per instruction; element loops stay scalar)
        const f32x16 ys = Split<PM>::F16 ? acc[mt][nt] * sc : acc[mt][nt];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float y4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) y4[j] = ys[4 * g + j];
          planes_store4<PM, TB>(Ps, col, 8 * (MT * wave_rg() + mt) + 2 * g + h, y4, (MT * wave_rg() + mt) < groups ? gbase : nullptr,
                                gs.plane_stride);
        }
      }
  };
  if (gs.base != nullptr) sweep(gs.base);          // (wave uniform: two straight-line variants instead of a test per store)
  else sweep(nullptr);
}

// fp32 layer whose output goes to planes instead of the fp32 image.  Returns the tile's scale exponent (planes hold
// y * 2^e; 0 outside the f16 mode).  `scr`: 8 floats of LDS for the tile maximum (f16 mode).  Optional copies of the
// activations: gsave (fp32 rows, this tile's base) and gs (fp16 planes for the weight-gradient GEMM); optional sign words.
// DS (Swish training forward): dsave also receives d = dy/dz of every element (fp32 rows; mobody_module.py:9-15), with the
// fp32 training kernel's formulas (sig = 1 / (1 + exp(-z)), y = z sig, d = sig (1 + z (1 - sig)); layers.h wide_layer_swish_d).
template <int ACT, int MT, int PM, int TB, bool DS = false, class Ring, class Between>
__device__ __forceinline__ int wide_layer_to_planes(float* Xs, char* Ps, float* scr, const float* __restrict__ W,
                                                    const float* __restrict__ b, int Kp, Ring& ring, Between&& between,
                                                    uint32_t* mask = nullptr, bool full = true, int mask_groups = 0,
                                                    int rows_here = 1 << 30, float* gsave = nullptr,
                                                    const PlaneSave& gs = PlaneSave{nullptr, 0, nullptr}, float* dsave = nullptr) {
  const float bias0 = b[64 * wave_col() + (lane_id() & 31)], bias1 = b[64 * wave_col() + 32 + (lane_id() & 31)];
  f32x16 acc[MT][2];
  wide_zero<MT>(acc);
  wide_gemm<MT>(Xs, W, Kp, acc, ring);
  TR(6);
  between();
  float mx = 0.f;                                  // the activations in place; their largest magnitude for the f16 scale
  if constexpr (DS) {
    static_assert(ACT == ACT_SWISH, "derivative saves belong to the Swish nets");
    f32x16 dv[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float z = acc[mt][nt][r] + (nt ? bias1 : bias0);
          const float sig = fast_rcp(1.f + __expf(-z));
          const float y = z * sig;
          acc[mt][nt][r] = y;
          dv[mt][nt][r] = sig * (1.f + z * (1.f - sig));
          if constexpr (Split<PM>::F16) mx = fmaxf(mx, fabsf(y));
        }
    wide_store_rows<MT>(dv, dsave, full, rows_here);
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x16 z = acc[mt][nt] + (nt ? bias1 : bias0);          // v_pk_add_f32
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float y = activate<ACT>(z[r]);
          acc[mt][nt][r] = y;
          if constexpr (Split<PM>::F16) mx = fmaxf(mx, fabsf(y));
        }
      }
  }
  // the global plane copy feeds a contraction over ROWS: rows past the end of the batch must be zero there
  // (wave uniform and only on the ragged last tile; their activations are act(bias), harmless for the tile maximum)
  if (gs.base != nullptr && !full) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * (MT * wave_rg() + mt) + (r & 3) + 8 * (r >> 2) + 4 * (lane_id() >> 5) >= rows_here) acc[mt][nt][r] = 0.f;
  }
  if constexpr (Split<PM>::F16) f16_tile_max_put(mx, scr);
  lds_barrier();                                   // every wave has read the old image (and posted its maximum)
  int e = 0;
  if constexpr (Split<PM>::F16) e = f16_scale_exp(f16_tile_max_get(scr));
  planes_from_acc<MT, PM, TB>(acc, Ps, e, gs, (min(rows_here, TB) + 31) / 32);
  if (gsave != nullptr) wide_store_rows<MT>(acc, gsave, full, rows_here);
  TR(7);
  if (mask != nullptr) relu_mask_words<MT>(acc, mask, mask_groups);
  lds_barrier();
  return e;
}

// split-precision layer: planes (scaled by 2^e_in in the f16 mode) -> fp32 image (+ optional fp32 copy gsave, sign words)
template <int ACT, int MT, int PM, int TB, bool DS = false, class Between>
__device__ __forceinline__ void bf_layer(float* Xs, const char* Ps, int e_in, const s16x8* __restrict__ Wb,
                                         const float* __restrict__ b, BfRing<PM>& ring, Between&& between,
                                         uint32_t* mask = nullptr, bool full = true, int mask_groups = 0,
                                         int rows_here = 1 << 30, float* gsave = nullptr, float* dsave = nullptr) {
  const float bias0 = b[64 * wave_col() + (lane_id() & 31)], bias1 = b[64 * wave_col() + 32 + (lane_id() & 31)];
  f32x16 acc[MT][2];
  wide_zero<MT>(acc);
  bf_gemm<MT, PM, TB>(Ps, Wb, acc, ring);
  TR(3);
  between();
  const float inv = Split<PM>::F16 ? exp2i(-(e_in + F16_WSHIFT)) : 1.f;     // exact: both scales are powers of two
  if constexpr (DS) {
    f32x16 dv[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float bias = nt ? bias1 : bias0;
          const float z = Split<PM>::F16 ? fmaf(acc[mt][nt][r], inv, bias) : acc[mt][nt][r] + bias;
          const float sig = fast_rcp(1.f + __expf(-z));
          acc[mt][nt][r] = z * sig;
          dv[mt][nt][r] = sig * (1.f + z * (1.f - sig));
        }
    wide_store_rows<MT>(dv, dsave, full, rows_here);
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float bias = nt ? bias1 : bias0;
        const f32x16 bv = {bias, bias, bias, bias, bias, bias, bias, bias, bias, bias, bias, bias, bias, bias, bias, bias};
        const f32x16 iv = {inv, inv, inv, inv, inv, inv, inv, inv, inv, inv, inv, inv, inv, inv, inv, inv};
        const f32x16 z = Split<PM>::F16 ? __builtin_elementwise_fma(acc[mt][nt], iv, bv) : acc[mt][nt] + bias;   // v_pk_fma_f32 / v_pk_add_f32
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = activate<ACT>(z[r]);
      }
  }
  lds_barrier();                                   // every wave has read the planes
  wide_foreach<MT>(acc, [&](int row, int col, float y) { Xs[row * LDX + col] = y; });
  if (gsave != nullptr) wide_store_rows<MT>(acc, gsave, full, rows_here);
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
