// Layer-level device helpers built on tile.h, and the argument block of the generic fused
// 3-layer MLP forward kernel (in -> 256 -> 256 -> out) used for:
//   actor / twin-Q / target twin-Q / V     (ReLU;  mobody.py:35-83)
//   the ensemble reward head               (Swish; mobody_module.py:295-302)
//   the DARA classifier heads              (ReLU;  mobody.py:11-33)
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "tile.h"

namespace mobody {

// One hidden layer in place on the LDS image: X <- act(X[:, :Kp] * W + b); `extra(row, col, y)` sees every output.
// `ring` holds wide_prefetch(W, Kp); `between()` runs after the last MFMA and before the epilogue -- the place to
// request the NEXT layer's first weight fragments (the ring's registers are free again), so their L2/HBM round trip
// overlaps this layer's barrier + epilogue instead of stalling the next GEMM.
// `mask` (optional): this tile's [groups][256] words, `mask_groups` of them real; bit r of word (g, col) =
// [y(row 32 g + r, col) > 0].  The backward
// pass of a ReLU net needs only these signs, 32 B per row instead of the 1 KB activation row.
template <int ACT, int MT = 2, class Extra, class Between>
__device__ __forceinline__ void wide_layer(float* Xs, const float* __restrict__ W, const float* __restrict__ b, int Kp,
                                           WideRing& ring, Extra&& extra, Between&& between, uint32_t* mask = nullptr,
                                           bool full = false, int mask_groups = 1 << 30) {
  // the wave's two bias values (columns 64w + 32nt + lane&31): requested before the GEMM, consumed after it
  const float bias0 = b[64 * wave_col() + (lane_id() & 31)], bias1 = b[64 * wave_col() + 32 + (lane_id() & 31)];
  f32x16 acc[MT][2];
  wide_zero<MT>(acc);
  wide_gemm<MT>(Xs, W, Kp, acc, ring);
  if (Kp == HID) TR(3);
  between();
  lds_barrier();                         // every wave has finished reading the old image
  // `full` (wave uniform): every row of the tile is a real row, so `extra` may skip its row guard -- a per-element
  // row < rows_here test costs a v_cmp plus exec-mask save/restore around each of the 32 stores of a lane
  if (full) {
    wide_foreach<MT>(acc, [&](int row, int col, float v) {
      const float y = activate<ACT>(v + ((col & 32) ? bias1 : bias0));
      Xs[row * LDX + col] = y;
      extra(std::false_type{}, row, col, y);
    });
  } else {
    wide_foreach<MT>(acc, [&](int row, int col, float v) {
      const float y = activate<ACT>(v + ((col & 32) ? bias1 : bias0));
      Xs[row * LDX + col] = y;
      extra(std::true_type{}, row, col, y);
    });
  }
  if (mask != nullptr) {
    const int i = lane_id() & 31, hh = lane_id() >> 5;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float bias = nt ? bias1 : bias0;
        uint32_t word = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          word |= (uint32_t)(activate<ACT>(acc[mt][nt][r] + bias) > 0.f) << ((r & 3) + 8 * (r >> 2) + 4 * hh);
        word |= (uint32_t)__shfl_xor((int)word, 32);        // the other lane half holds the other 16 rows
        // 32-row groups past the end of the batch have no words (a taller tile's last groups would land in the next member)
        const int grp = MT * wave_rg() + mt;
        if (hh == 0 && grp < mask_groups) mask[grp * HID + 64 * wave_col() + 32 * nt + i] = word;
      }
  }
  lds_barrier();
}

template <int ACT, int MT = 2, class Extra>
__device__ __forceinline__ void wide_layer(float* Xs, const float* __restrict__ W, const float* __restrict__ b, int Kp,
                                           Extra&& extra) {
  WideRing ring;
  wide_prefetch(W, Kp, ring);
  wide_layer<ACT, MT>(Xs, W, b, Kp, ring, extra, [] {});
}

// Training forward of a Swish layer: X <- y = z*sigmoid(z), z = X*W + b, and `extra(guard, row, col, y, d)` also sees
// d = dy/dz = sig*(1 + z*(1 - sig)) (saved for the backward pass: Swish is not invertible, so the sign words of the ReLU
// nets do not carry over; mobody_module.py:9-15).
template <int MT, class Extra, class Between>
__device__ __forceinline__ void wide_layer_swish_d(float* Xs, const float* __restrict__ W, const float* __restrict__ b,
                                                   int Kp, WideRing& ring, Extra&& extra, Between&& between, bool full) {
  const float bias0 = b[64 * wave_col() + (lane_id() & 31)], bias1 = b[64 * wave_col() + 32 + (lane_id() & 31)];
  f32x16 acc[MT][2];
  wide_zero<MT>(acc);
  wide_gemm<MT>(Xs, W, Kp, acc, ring);
  between();
  lds_barrier();
  auto body = [&](auto guarded) {
    wide_foreach<MT>(acc, [&](int row, int col, float v) {
      const float z = v + ((col & 32) ? bias1 : bias0);
      const float sig = fast_rcp(1.f + __expf(-z));
      const float y = z * sig;
      Xs[row * LDX + col] = y;
      extra(guarded, row, col, y, sig * (1.f + z * (1.f - sig)));
    });
  };
  if (full) body(std::false_type{});
  else body(std::true_type{});
  lds_barrier();
}

struct NoExtra {
  template <class Guard>
  __device__ __forceinline__ void operator()(Guard, int, int, float) const {}
};

struct Mlp3FwdArgs {
  const float* src[3];      // concatenated inputs, src[k] is [rows][n[k]] with leading dim ld[k]; unused: n = 0
  int ld[3], n[3];
  long long src_ms[3];      // member stride of src[k] in floats: 0 = one input shared by all members (the hot path),
                            // rows*ld = per-member rows (dynamics pre-training / validation: [E][rows][n])
  long long x_ms;           // member stride of save_x: 0 = written once by member 0, else every member saves its own
  const float *w1, *b1, *w2, *b2, *w3, *b3;   // member 0
  long long sw1, sb1, sw2, sb2, sw3, sb3;     // member strides (floats)
  int Kp1, Np3, nout;
  long long rows;
  float* out;               // out[m*out_mstride + row*out_ld + c], c < nout
  long long out_mstride;
  int out_ld;
  float* save_x;            // [rows][Kp1]           (optional, written by member 0)
  float* save_h1;           // [members][rows][256]  (optional)
  float* save_h2;
  uint32_t* mask1;          // [members][ceil(rows/32)][256] sign bits of h1 / h2 (optional, see wide_layer)
  uint32_t* mask2;
  const unsigned short* w2_planes;   // split-precision modes: bf16 planes of W2, member 0 ([3][32][256][8] bf16 per member, tile_bf.h)
  long long planes_ms;               // member stride of w2_planes in bf16 elements
  float* save_d1;           // [members][rows][256] Swish derivative at the pre-activations of layers 1 / 2 (training
  float* save_d2;           // forward of the ensemble nets only: k_mlp3_fwd_train)
  unsigned short* save_h1p; // f16 mode, instead of save_h1: the layer-1 activations as the two fp16 planes the weight-gradient GEMM
  long long h1p_ms;         // reads ([member][2 planes][rows32 / 8][256][8], layers_bf.h PlaneSave; member stride h1p_ms and plane
  long long h1p_plane;      // stride h1p_plane in 16-bit elements, rows32 = rows rounded up to 32) and
  int* save_e1;             // [members][ceil(rows / 32)] the tiles' scale exponents
  int out_mode;             // 0 raw, 1 max_action*tanh
  const float* resid;       // optional: out[m][row][c] += resid[row*resid_ld + c] (shared by the members; mopo dynamics: s + f(s,a))
  int resid_ld;
  float max_action;
};

int launch_mlp3_fwd(const Mlp3FwdArgs& a, int members, int act, hipStream_t stream);
int launch_mlp3_fwd_pair(const Mlp3FwdArgs& a, int members_a, const Mlp3FwdArgs& b, int members_b, hipStream_t stream);
// split-precision forward (mlp_fwd_bf.hip): prec 1 bf16 / 2 bf16x2 / 3 bf16x3; needs a.w2_planes (and b.w2_planes)
int launch_mlp3_fwd_bf(const Mlp3FwdArgs& a, int members_a, const Mlp3FwdArgs& b, int members_b, int act, int prec, hipStream_t st);

// Row-tile height of the fused MLP kernels.  Measured on MI355X (bench.py, S=17/A=6): 32-row tiles (33 KB LDS,
// ~124 VGPRs -> 4 workgroups = 16 waves per CU) beat 64-row tiles (2 workgroups per CU) at every batch size from
// 2.5 k to 41 k rows (forward 84 vs 73 TFLOP/s at 41 k rows, 60 vs 51 at 10 k): occupancy hides the weight-fetch
// latency better than the 2x weight reuse of the taller tile.  MOBODY_TILE_ROWS=64 selects the tall tile (tuning aid).
inline int pick_tile_rows(long long rows, int members) {
  static const int forced = tune_int("MOBODY_TILE_ROWS", 0);
  (void)rows; (void)members;
  return forced == 64 ? 64 : 32;
}

}  // namespace mobody
