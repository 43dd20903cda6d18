// Split-precision MFMA core for the 256 x 256 layers (throughput modes of BASELINE.json configs[1], "bf16 MFMA inputs /
// fp32 accumulate").  An fp32 value is carried as NPL 16-bit terms x = x0 + x1 (+ x2) and a product keeps every term pair
// (i, j) with i + j < NPL, all accumulated in fp32 by v_mfma_f32_32x32x16_{bf16,f16} (32 cycles per instruction against 64
// for the K = 2 fp32 form: 16x the MACs per cycle).  Precision modes (MobodyHyper.precision):
//     1  "bf16"    1 plane,  1 product   plain bf16 inputs, ~3e-3 relative per product
//     2  "bf16x2"  2 planes, 3 products  x0y0 + x0y1 + x1y0 in bf16 terms (8 significand bits each), ~2^-16
//     3  "bf16x3"  3 planes, 6 products  fp32-grade (2^-24), six MFMAs per fp32 product
//     4  "f16x2"   2 planes, 3 products  TWO fp16 TERMS (11 significand bits each = 22 bits; the dropped x1y1 term is 2^-22):
//                                        fp32-grade at HALF the MFMA work of bf16x3.  fp16 has 5 exponent bits, so operands
//                                        are pre-scaled by powers of two (exact in fp32): weights by 2^F16_WSHIFT when their
//                                        planes are written, activations / gradients per row tile so that the tile's largest
//                                        magnitude lands in [2^13, 2^14) (f16_scale_exp); the accumulator is un-scaled in the
//                                        consuming epilogue by one exact multiplication.
//
// Activations: NPL planes in LDS, stored FEATURE-major -- plane[k][row], a k-row is the tile's TB rows (64 B for 32-row
// tiles) -- because the producing layer's accumulators have the feature on the lane and four consecutive ROWS in four
// consecutive registers (C/D map of the 32x32 MFMA): a lane packs those four terms into ONE 8-byte ds_write_b64 per plane
// (the row-major image this replaced took a 2-byte store per element: 96 per thread and layer in bf16x3).  The consuming GEMM
// needs, per lane, eight consecutive k of one row: two ds_read_b64_tr_b16 (the hardware transpose read of gfx950: per
// 16-lane group a block of 4 k-rows x 16 rows comes back column-major).  8-byte chunk c of k-row k sits at plane_off(k, c):
// the XOR makes the 16 lanes of a store group (16 consecutive features, same chunk) hit 16 different 8-byte bank slots; the
// transposed read of a 32-lane half covers 4 whole k-rows = 256 contiguous bytes = every bank once.
// Weights: NPL planes in the T blob, K-interleaved by eight ([K/8][256][8] 16-bit per plane), so a lane's B fragment (eight
// consecutive k of one column) is one 16-byte load and a wave instruction reads 1 KB contiguous.  Lane maps (MI355X guide,
// "A/B operand lane maps"): lane (r = lane & 31, h = lane >> 5) holds A[row r][k = 8h + j] and B[k = 8h + j][col r],
// j = 0..7; C/D as the fp32 form.
#pragma once
#include <type_traits>
#include <utility>

#include "tile.h"

namespace mobody {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;       // one MFMA A/B fragment as raw 16-bit lanes (either format)
typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr long long BF_PLANE = 32LL * HID;     // s16x8 units per weight plane ([K/8 = 32][256])
constexpr int F16_WSHIFT = 8;                  // "f16x2" weight planes hold w * 2^8 (|w| < 255; residual terms stay normal down to |w| ~ 5e-4)

#ifndef SPLIT_RING
#define SPLIT_RING 3
#endif
template <int PM>
struct Split {
  static_assert(PM >= 1 && PM <= 4, "precision mode 1..4");
  static constexpr int NPL = PM == 4 ? 2 : PM;
  static constexpr bool F16 = PM == 4;
  static constexpr int RING = SPLIT_RING;        // k16 steps of weight fragments in flight
};
constexpr int split_planes(int pm) { return pm == 4 ? 2 : pm; }

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): an unrolled loop whose index is a constant expression
template <int... J, class F>
__device__ __forceinline__ void static_seq_impl(std::integer_sequence<int, J...>, F&& f) { (f(std::integral_constant<int, J>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_seq(F&& f) { static_seq_impl(std::make_integer_sequence<int, N>{}, f); }

template <int PM>
struct BfRing { s16x8 r[Split<PM>::RING][Split<PM>::NPL][2]; };

// element index (in 16-bit units) of weight (k, n) inside plane p of a member's plane block
__host__ __device__ inline long long bf_plane_idx(int p, int k, int n) { return (((long long)p * 32 + (k >> 3)) * HID + n) * 8 + (k & 7); }

// y (already scaled in the f16 mode) -> its NPL 16-bit terms
template <int PM>
__device__ __forceinline__ void split_terms(float y, short (&t)[Split<PM>::NPL]) {
  if constexpr (Split<PM>::F16) {
    const _Float16 t0 = (_Float16)y;
    const _Float16 t1 = (_Float16)(y - (float)t0);
    t[0] = __builtin_bit_cast(short, t0); t[1] = __builtin_bit_cast(short, t1);
  } else {
    const __bf16 t0 = (__bf16)y;
    t[0] = __builtin_bit_cast(short, t0);
    if constexpr (PM >= 2) {
      const float r1 = y - (float)t0;
      const __bf16 t1 = (__bf16)r1;
      t[1] = __builtin_bit_cast(short, t1);
      if constexpr (PM >= 3) { const __bf16 t2 = (__bf16)(r1 - (float)t1); t[2] = __builtin_bit_cast(short, t2); }
    }
  }
}
// the three bf16 terms of the weight planes (modes 1-3 share them) -- kept under its old name for the plane writers
template <int NPL>
__device__ __forceinline__ void bf_split(float y, __bf16 (&t)[NPL]) {
  t[0] = (__bf16)y;
  if constexpr (NPL >= 2) { const float r1 = y - (float)t[0]; t[1] = (__bf16)r1;
    if constexpr (NPL >= 3) t[2] = (__bf16)(r1 - (float)t[1]); }
}

template <int PM>
__device__ __forceinline__ f32x16 split_mfma(s16x8 a, s16x8 b, f32x16 c) {
  if constexpr (Split<PM>::F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- power-of-two tile scales of the f16 mode ---------------------------------------------------------------------------
// exponent e such that m * 2^e lies in [2^13, 2^14) (fp16 overflows at 2^16; the residual term of anything above
// 2^-3 of the tile maximum stays a normal fp16 number, smaller values degrade gracefully to an absolute error of 2^-39 of
// the maximum).  An all-zero (or denormal) tile gets the LARGEST exponent, 100: its planes are zero whatever the scale, and
// the weight-gradient GEMM, which brings the tiles of a row slice to their smallest exponent, must not let an empty tile
// set that common scale.  Inf / NaN: 0 (a non-finite tile is garbage either way and NaNs propagate through the MFMA).
__device__ __forceinline__ int f16_scale_exp(float m) {
  const int eb = (__float_as_int(m) >> 23) & 0xff;
  if (eb == 0) return 100;
  if (eb == 255) return 0;
  const int e = 13 - (eb - 127);
  return e > 100 ? 100 : e;
}
__device__ __forceinline__ float exp2i(int e) { return __int_as_float((e + 127) << 23); }      // 2^e, -126 <= e <= 127
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// Tile maximum of per-thread maxima through four LDS words: call f16_tile_max_put BEFORE the barrier that separates the GEMM
// from its epilogue, f16_tile_max_get after it.  `scr` = 4 * waves floats of LDS no other phase touches before the next barrier.
__device__ __forceinline__ void f16_tile_max_put(float v, float* scr) {
  v = wave_max(v);
  if (lane_id() == 0) scr[threadIdx.x >> 6] = v;
}
__device__ __forceinline__ float f16_tile_max_get(const float* scr) {
  float m = fmaxf(fmaxf(scr[0], scr[1]), fmaxf(scr[2], scr[3]));
  if (blockDim.x > 256) m = fmaxf(m, fmaxf(fmaxf(scr[4], scr[5]), fmaxf(scr[6], scr[7])));
  return m;
}

// ---- activation planes in LDS ---------------------------------------------------------------------------------------------
// byte offset (inside one plane) of the 8-byte chunk c (rows 4c .. 4c+3) of feature row k; TB = rows per tile (32 or 64)
template <int TB>
__device__ __forceinline__ int plane_off(int k, int c) {
  static_assert(TB == 32 || TB == 64, "row tiles of 32 or 64");
  if constexpr (TB == 32) return k * 64 + ((c ^ ((k >> 1) & 7)) << 3);
  else return k * 128 + ((c ^ ((k & 7) | ((((k >> 3) ^ (k >> 1)) & 1) << 3))) << 3);
}
template <int TB>
constexpr int plane_bytes() { return HID * TB * 2; }

// Store the NPL terms of four consecutive rows (rows 4c .. 4c+3 of the tile, chunk c) of feature k: one ds_write_b64 per plane.
// gbase (optional): the same terms also go to global planes laid out [row / 8][256][8] (plane stride gstride elements),
// gbase pointing at the tile's first 8-row block: chunk c is the half (c & 1) of block c >> 1.
template <int PM, int TB>
__device__ __forceinline__ void planes_store4(char* Ps, int k, int c, const float (&y)[4], short* gbase = nullptr,
                                              long long gstride = 0) {
  constexpr int NPL = Split<PM>::NPL;
  short t[4][NPL];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_terms<PM>(y[j], t[j]);
  const int off = plane_off<TB>(k, c);
#pragma unroll
  for (int p = 0; p < NPL; ++p) {
    s16x4 v; v[0] = t[0][p]; v[1] = t[1][p]; v[2] = t[2][p]; v[3] = t[3][p];
    *reinterpret_cast<s16x4*>(Ps + p * plane_bytes<TB>() + off) = v;
    if (gbase != nullptr) *reinterpret_cast<s16x4*>(gbase + p * gstride + ((long long)(c >> 1) * HID + k) * 8 + 4 * (c & 1)) = v;
  }
}

// Weight fragments come through a buffer descriptor (BF_BUFFER_LOADS): the per-lane part of the address is ONE 32-bit offset
// computed once, the (plane, k-step, column tile) part a scalar / immediate offset -- no vector address arithmetic per load.
// With plain pointers every 16-byte load cost a 64-bit vector add (~100 of the ~720 non-MFMA vector instructions a wave
// executes per forward tile, in a kernel whose vector ALU is active a third of its busy time).
#ifndef BF_BUFFER_LOADS
#define BF_BUFFER_LOADS 1
#endif
template <int PM>
__device__ __forceinline__ void bf_ldb(const s16x8* __restrict__ Wb, int s, s16x8 (&b)[Split<PM>::NPL][2]) {
  const int lane = lane_id(), r = lane & 31, h = lane >> 5;
#if BF_BUFFER_LOADS
  // (the member's plane block: NPL planes of 128 KB; wave-uniform base -> scalar registers)
  const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(Wb, (unsigned)(Split<PM>::NPL * BF_PLANE * 16));
  const int voff = (h * HID + 64 * wave_col() + r) * 16;
#pragma unroll
  for (int p = 0; p < Split<PM>::NPL; ++p)
#pragma unroll
    for (int n = 0; n < 2; ++n)
      b[p][n] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)((p * BF_PLANE + (long long)2 * s * HID + 32 * n) * 16), 0));
#else
#pragma unroll
  for (int p = 0; p < Split<PM>::NPL; ++p)
#pragma unroll
    for (int n = 0; n < 2; ++n) b[p][n] = Wb[p * BF_PLANE + (long long)(2 * s + h) * HID + 64 * wave_col() + 32 * n + r];
#endif
}

template <int PM>
__device__ __forceinline__ void bf_prefetch(const s16x8* __restrict__ Wb, BfRing<PM>& ring) {
#pragma unroll
  for (int j = 0; j < Split<PM>::RING - 1; ++j) bf_ldb<PM>(Wb, j, ring.r[j]);
  __builtin_amdgcn_sched_barrier(0);
}

// acc[mt][nt] = X (planes in LDS, TB rows per tile, K = 256) * W (planes in global): acc is WRITTEN, its previous contents are
// not read (a caller's wide_zero is dead code).  `ring` holds bf_prefetch.
template <int MT, int PM, int TB>
__device__ __forceinline__ void bf_gemm(const char* __restrict__ Ps, const s16x8* __restrict__ Wb, f32x16 (&acc)[MT][2],
                                        BfRing<PM>& ring) {
  constexpr int R = Split<PM>::RING, NPL = Split<PM>::NPL;
  using lds_s16x4 = __attribute__((address_space(3))) s16x4;
  const int lane = lane_id(), g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, h = lane >> 5;
  // transposed read of k-step s, half j, tile m: lane 4q + pp of a 16-lane group supplies the address of k-row
  // 16 s + 8 h + 4 j + q, rows 4 c .. 4 c + 3 with c = 8 (MT rg + m) + 4 (g & 1) + pp; it receives its row (lane & 31 of the
  // tile) of the four k-rows: elements 4 j .. 4 j + 3 of the A fragment.  The swizzle depends on k bits 1-3 only: s moves nothing.
  int base[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 2; ++j) base[m][j] = plane_off<TB>(8 * h + 4 * j + q, 8 * (MT * wave_rg() + m) + 4 * (g & 1) + pp);
  // one k-step: weight fragments of step s + R - 1 requested into slot (SL + R - 1) % R, A fragments read, the NPL (NPL + 1) / 2
  // products of every accumulator issued smallest terms first.  FIRST (step 0): the first product of each accumulator takes a
  // literal zero as its C operand -- the accumulators are never zero-initialised (32 v_mov per GEMM and wave otherwise: the
  // loop is a runtime loop, so the compiler cannot fold the caller's zero fill into the first MFMA itself).
  auto step = [&](auto first_c, int s, auto slot_c) {
    constexpr bool FIRST = GEMM_PEEL && decltype(first_c)::value;
    constexpr int SL = decltype(slot_c)::value;
    if (s + R - 1 < 16) bf_ldb<PM>(Wb, s + R - 1, ring.r[(SL + R - 1) % R]);
    __builtin_amdgcn_sched_barrier(0);
    s16x8 a[NPL][MT];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const char* src = Ps + p * plane_bytes<TB>() + s * (16 * TB * 2);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(src + base[m][0]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(src + base[m][1]));
        a[p][m] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        bool fresh = FIRST;
        // smallest terms first
#pragma unroll
        for (int d = NPL - 1; d >= 0; --d)
#pragma unroll
          for (int i = 0; i <= d; ++i) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[m][n] = split_mfma<PM>(a[i][m], ring.r[SL][d - i][n], fresh ? zero : acc[m][n]);
            fresh = false;
          }
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  step(std::true_type{}, 0, std::integral_constant<int, 0>{});
  for (int s0 = 1; s0 < 16; s0 += R) {               // steps 1 .. 15: s0 == 1 (mod R), so the ring slot of step s0 + j is (1 + j) % R
    static_seq<R>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (s0 + j < 16) step(std::false_type{}, s0 + j, std::integral_constant<int, (1 + j) % R>{});
    });
  }
}

}  // namespace mobody
