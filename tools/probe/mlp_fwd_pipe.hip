// EXPERIMENT, NOT BUILT (round 3; measured slower than k_mlp3_fwd_bf).  To try it again: copy this file into
// csrc/, declare try_launch_mlp3_fwd_pipe in layers.h and call it at the top of launch_mlp3_fwd_bf for prec == 4;
// tools/probe/ab_pipe.sh and trace_pipe.sh then A/B and phase-trace it (tools/trace_mlp.py fwd with PIPE_MT set).
//
// Software-pipelined fused 3-layer MLP forward of the "f16x2" mode (ReLU nets of the train step: actor / twin-Q and their
// targets).  In k_mlp3_fwd_bf (mlp_fwd_bf.hip) every workgroup of a launch walks the same phases at the same time -- the
// launches are one generation of workgroups -- so the matrix cores idle while every wave splits / packs / stores an
// epilogue, and the vector ALU idles while every wave waits on the MFMA pipe and the L2 weight stream: the measured launch
// time is the SUM of the MFMA, weight-stream and epilogue floors, not their maximum.  A wave cannot borrow the idle unit from
// its SIMD neighbours either (tools/probe/coexec_probe.hip: an MFMA-dense wave starves the other waves' VALU issue), but
// ONE wave that interleaves independent VALU / LDS / store instructions between its own MFMAs hides ~70 % of them.
//
// So a workgroup here owns TWO row slots (P, Q) of 32 * MT rows each and skews them by one layer:
//     load x(P), x(Q)  |  layer 1 (P, Q together: one pass over W1)  |  epilogue 1 (P)
//     | 256x256 GEMM (P)  with  epilogue 1 (Q) issued between its MFMAs
//     | 256x256 GEMM (Q)  with  epilogue 2 (P) issued between its MFMAs
//     | epilogue 2 (Q)  |  output layer (P, Q together)
// Every k-step of a hosted GEMM carries one group (four rows of one feature per lane) of the guest epilogue; both are
// straight-line code inside one scheduling region (the GEMM is fully unrolled so the guest's accumulator indices are
// compile-time constants).  Optional saves go through buffer descriptors: a save that is off (or a row past the end of the
// batch) has no records behind it and the hardware drops the store -- no branch splits the region.
// Arithmetic is that of k_mlp3_fwd_bf<ACT, 4, MT, NT> operation for operation.
#include <stdlib.h>

#include <utility>

#include "common.h"
#include "layers_bf.h"

#ifndef FWD_PIPE_MT
#define FWD_PIPE_MT 2          // row tiles of 32 per slot: a workgroup covers 4 * FWD_PIPE_MT * 16 rows; 0 = kernel off
#endif
#ifndef FWD_PIPE_RING
#define FWD_PIPE_RING 6        // k16 steps of W2 fragments in flight (MT = 2): one wave per SIMD has nobody else to cover an L2 round trip
#endif
#ifndef FWD_PIPE_RING1
#define FWD_PIPE_RING1 4       // the same at MT = 1 (two waves per SIMD, 256 registers)
#endif
#ifndef FWD_PIPE_MIN_WGS
#define FWD_PIPE_MIN_WGS 128   // below this many pipelined workgroups the one-tile kernel fills the chip better
#endif

namespace mobody {

using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
using rsrc_t = __amdgpu_buffer_rsrc_t;

// descriptor over [p, p + bytes); a null p (save off) gets zero records: every store through it is dropped
__device__ __forceinline__ rsrc_t pipe_rsrc(const void* p, long long bytes) {
  const long long n = p != nullptr ? (bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes) : 0;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)n, 0x00020000);
}
constexpr int PIPE_OOB = 0x7ffffff0;             // a byte offset beyond any descriptor here: "this lane stores nothing"

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

template <int R>
struct PipeRing { s16x8 r[R][2][2]; };

template <int R>
__device__ __forceinline__ void pipe_prefetch(const s16x8* __restrict__ Wb, PipeRing<R>& ring) {
#pragma unroll
  for (int j = 0; j < R - 1; ++j) bf_ldb<4>(Wb, j, ring.r[j]);
  __builtin_amdgcn_sched_barrier(0);
}

template <int MT>
struct PipeSlot {
  float* Xs;                 // fp32 image / planes (aliased) of this slot
  char* Ps;
  float* scr;                // tile-maximum exchange
  long long row0;
  int rows_here, mg;         // real rows, and 32-row groups holding any
  rsrc_t r_h1p, r_h2, r_m1, r_m2;
  int h1p_pstride;           // bytes between the two planes of the global copy
  int* e_out;
  int e;                     // scale exponent of the layer-1 planes
};

// Group GI = (mt, nt, g) of epilogue 1: four consecutive rows of this lane's feature -> both planes in LDS (+ the global
// copy) and the group's sign bits; the (mt, nt) word leaves with its last group.
template <int MT, int GI>
__device__ __forceinline__ void e1_group(f32x16 (&acc)[MT][2], const PipeSlot<MT>& s, float sc, uint32_t (&word)[MT][2]) {
  constexpr int PM = 4, TB = 32 * MT;
  constexpr int mt = GI / 8, nt = (GI / 4) % 2, g = GI % 4;
  const int lane = lane_id(), i = lane & 31, h = lane >> 5;
  const int col = 64 * wave_col() + 32 * nt + i, c = 8 * mt + 2 * g + h;
  short t[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float y = acc[mt][nt][4 * g + j];
    word[mt][nt] |= (uint32_t)(y > 0.f) << (j + 8 * g + 4 * h);
    split_terms<PM>(y * sc, t[j]);
  }
  const int off = plane_off<TB>(col, c);
  // (groups of 32 rows past the batch's end have no plane rows: plane 0's would land in plane 1)
  const int goff = mt < s.mg ? ((c >> 1) * HID + col) * 16 + 8 * (c & 1) : PIPE_OOB;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    s16x4 v; v[0] = t[0][p]; v[1] = t[1][p]; v[2] = t[2][p]; v[3] = t[3][p];
    *reinterpret_cast<s16x4*>(s.Ps + p * plane_bytes<TB>() + off) = v;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), s.r_h1p, goff + p * s.h1p_pstride, 0, 0);
  }
  if constexpr (g == 3) {
    const uint32_t w = word[mt][nt] | (uint32_t)__shfl_xor((int)word[mt][nt], 32);
    __builtin_amdgcn_raw_buffer_store_b32(w, s.r_m1, h == 0 ? (mt * HID + col) * 4 : PIPE_OOB, 0, 0);
  }
}

// Group GI of epilogue 2: un-scale + bias + activation, the fp32 image for the output layer, the optional fp32 copy and signs.
template <int ACT, int MT, int GI>
__device__ __forceinline__ void e2_group(f32x16 (&acc)[MT][2], const PipeSlot<MT>& s, float inv, float bias0, float bias1,
                                         uint32_t (&word)[MT][2]) {
  constexpr int mt = GI / 8, nt = (GI / 4) % 2, g = GI % 4;
  const int lane = lane_id(), i = lane & 31, h = lane >> 5;
  const int col = 64 * wave_col() + 32 * nt + i;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 32 * mt + j + 8 * g + 4 * h;
    const float y = activate<ACT>(fmaf(acc[mt][nt][4 * g + j], inv, nt ? bias1 : bias0));
    word[mt][nt] |= (uint32_t)(y > 0.f) << (j + 8 * g + 4 * h);
    s.Xs[row * LDX + col] = y;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), s.r_h2, (row * HID + col) * 4, 0, 0);
  }
  if constexpr (g == 3) {
    const uint32_t w = word[mt][nt] | (uint32_t)__shfl_xor((int)word[mt][nt], 32);
    __builtin_amdgcn_raw_buffer_store_b32(w, s.r_m2, h == 0 ? (mt * HID + col) * 4 : PIPE_OOB, 0, 0);
  }
}

// acc += planes (LDS) * W2 planes, all 16 k-steps unrolled; guest(integral_constant<step>) is issued with each step's MFMAs.
// The ring enters holding steps 0 .. R-2 in slots (S0 + j) % R; with WRAP it leaves holding them again in the slots a second
// pass (S0 + 16) expects -- the second slot's GEMM reads the same weights.
template <int MT, int S0, bool WRAP, int R, class Guest>
__device__ __forceinline__ void bf_gemm_host(const char* __restrict__ Ps, const s16x8* __restrict__ Wb, f32x16 (&acc)[MT][2],
                                             PipeRing<R>& ring, Guest&& guest) {
  constexpr int PM = 4, TB = 32 * MT, NPL = 2;
  using lds_s16x4 = __attribute__((address_space(3))) s16x4;
  const int lane = lane_id(), g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, h = lane >> 5;
  int base[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 2; ++j) base[m][j] = plane_off<TB>(8 * h + 4 * j + q, 8 * m + 4 * (g & 1) + pp);
  // A fragments one step ahead of their MFMAs (two register sets): with one wave per SIMD nobody else covers the LDS round trip
  s16x8 a[2][NPL][MT];
  auto lda = [&](int s, s16x8 (&d)[NPL][MT]) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const char* src = Ps + p * plane_bytes<TB>() + s * (16 * TB * 2);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(src + base[m][0]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(src + base[m][1]));
        d[p][m] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
  };
  lda(0, a[0]);
  static_for<16>([&](auto S) {
    constexpr int s = decltype(S)::value, slot = (S0 + s) % R, nslot = (S0 + s + R - 1) % R;
    if constexpr (s + R - 1 < 16) bf_ldb<PM>(Wb, s + R - 1, ring.r[nslot]);
    else if constexpr (WRAP) bf_ldb<PM>(Wb, s + R - 1 - 16, ring.r[nslot]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (s + 1 < 16) lda(s + 1, a[(s + 1) & 1]);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int d = NPL - 1; d >= 0; --d)
#pragma unroll
          for (int i = 0; i <= d; ++i) acc[m][n] = split_mfma<PM>(a[s & 1][i][m], ring.r[slot][d - i][n], acc[m][n]);
      }
    guest(S);
    __builtin_amdgcn_sched_barrier(0);
  });
}

// layer 1 of NS slots in one pass over W1 (fp32 MFMA): every weight fragment feeds all slots' row tiles
template <int MT, int NS, int R>
__device__ __forceinline__ void wide_gemm_slots(float* const (&Xs)[2], const float* __restrict__ W, int Kp,
                                                f32x16 (&acc)[2][MT][2], WideRingT<R>& ring) {
  const int lane = lane_id(), i = lane & 31, h = lane >> 5;
  const int kh = Kp >> 1, nch = kh >> 2;
  const float* xa[NS];
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) xa[sl] = Xs[sl] + i * LDX + h * kh;
  f32x4 av[2][NS][MT];
  auto lda = [&](int c, f32x4 (&d)[NS][MT]) {
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
      for (int x = 0; x < MT; ++x) d[sl][x] = *reinterpret_cast<const f32x4*>(xa[sl] + 32 * x * LDX + 4 * c);
  };
  auto mma = [&](f32x4 (&a4)[NS][MT], f32x4 (&b)[2]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int sl = 0; sl < NS; ++sl)
#pragma unroll
        for (int x = 0; x < MT; ++x) {
          acc[sl][x][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[sl][x][u], b[0][u], acc[sl][x][0], 0, 0, 0);
          acc[sl][x][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[sl][x][u], b[1][u], acc[sl][x][1], 0, 0, 0);
        }
  };
  lda(0, av[0]);
  for (int c0 = 0; c0 < nch; c0 += 2 * R) {            // 2 R chunks per trip: the A double buffer's parity is a constant
#pragma unroll
    for (int j = 0; j < 2 * R; ++j) {
      const int c = c0 + j;
      wide_ldb(W, Kp, min(c + R - 1, nch - 1), ring.r[(j + R - 1) % R]);       // unconditional, clamped (tile.h wide_gemm)
      __builtin_amdgcn_sched_barrier(0);
      if (c < nch) {
        lda(min(c + 1, nch - 1), av[(j + 1) & 1]);
        mma(av[j & 1], ring.r[j % R]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int ACT, int MT, int NT, bool HASQ>
__device__ __forceinline__ void fwd_pipe_body(const Mlp3FwdArgs& a, int m, char* lds, long long row0P) {
  constexpr int PM = 4, TB = 32 * MT, NS = HASQ ? 2 : 1, NG = 8 * MT;
  constexpr int SLOT = (int)split_lds_bytes<PM, TB>(), RING = MT == 1 ? FWD_PIPE_RING1 : FWD_PIPE_RING;
  static_assert(NG == 16 || NG == 8, "8 or 16 groups against the 16 k-steps of a hosted GEMM");
  const float* w1 = a.w1 + m * a.sw1;
  const s16x8* w2b = reinterpret_cast<const s16x8*>(a.w2_planes + m * a.planes_ms);
  const float* w3 = a.w3 + m * a.sw3;
  const float* b1 = a.b1 + m * a.sb1;
  const float* b2 = a.b2 + m * a.sb2;
  const float* b3 = a.b3 + m * a.sb3;
  const int lane = lane_id(), i = lane & 31;
  TR(0);
  WideRingT<3> ring1;
  wide_prefetch(w1, a.Kp1, ring1);

  PipeSlot<MT> sl[2];
  float* Xs[2];
  const long long tiles_m = cdiv(a.rows, 32);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    PipeSlot<MT>& s = sl[k];
    s.Xs = reinterpret_cast<float*>(lds + k * SLOT); s.Ps = lds + k * SLOT; Xs[k] = s.Xs;
    s.scr = reinterpret_cast<float*>(lds + k * SLOT + split_scr_offset<PM, TB>());
    s.row0 = row0P + (long long)k * TB;
    s.rows_here = (int)max(0LL, min((long long)TB, a.rows - s.row0));
    const long long mrow = (long long)m * a.rows + s.row0;                       // first row of the slot in [members][rows] saves
    const long long mtile = (long long)m * tiles_m + s.row0 / 32;
    const int mg = s.mg = (s.rows_here + 31) / 32;
    s.r_h2 = pipe_rsrc(a.save_h2 ? a.save_h2 + mrow * HID : nullptr, (long long)s.rows_here * HID * 4);
    s.r_m1 = pipe_rsrc(a.mask1 ? a.mask1 + mtile * HID : nullptr, (long long)mg * HID * 4);
    s.r_m2 = pipe_rsrc(a.mask2 ? a.mask2 + mtile * HID : nullptr, (long long)mg * HID * 4);
    s.h1p_pstride = (int)(a.h1p_plane * 2);
    s.r_h1p = pipe_rsrc(a.save_h1p ? reinterpret_cast<short*>(a.save_h1p) + m * a.h1p_ms + (s.row0 / 8) * (HID * 8) : nullptr,
                        a.h1p_plane * 2 + (long long)mg * 32 * HID * 2);
    s.e_out = a.save_e1 ? a.save_e1 + mtile : nullptr;
    s.e = 0;
  }

  // ---- inputs of every slot, then one barrier --------------------------------------------------------------------------
  if (a.n[0] <= 32 && a.n[1] <= 32 && a.n[2] == 0) {
    // the usual case (state | action, each at most 32 columns): thread t owns column t & 31 and rows (t >> 5) + 8 u of every
    // source and slot; ALL loads are requested before the first LDS write -- one round trip for the whole prologue (a wave
    // here has no neighbours to hide a chain of them), unconditional from clamped rows / columns, invalid ones zeroed by select
    constexpr int U = TB / 8;
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    float v[NS][2][U];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int qq = (q == 1 && a.n[1] > 0) ? 1 : 0;       // an absent second source re-reads the first (result unused)
        const float* src = a.src[qq] + m * a.src_ms[qq];
        const int ld = a.ld[qq], n = max(a.n[q], 1);
#pragma unroll
        for (int u = 0; u < U; ++u)
          v[k][q][u] = src[(size_t)(sl[k].row0 + min(r0 + 8 * u, max(sl[k].rows_here - 1, 0))) * ld + min(c, n - 1)];
      }
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = r0 + 8 * u;
        const bool ok = r < sl[k].rows_here;
        if (c < a.n[0]) Xs[k][r * LDX + c] = ok ? v[k][0][u] : 0.f;
        if (c < a.n[1]) Xs[k][r * LDX + a.n[0] + c] = ok ? v[k][1][u] : 0.f;
      }
#pragma unroll
    for (int k = 0; k < NS; ++k) tile_zero_cols(Xs[k], a.n[0] + a.n[1], a.Kp1, TB);
  } else {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      int c0 = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (a.n[q] > 0) {
          tile_load(Xs[k], c0, a.src[q] + m * a.src_ms[q] + sl[k].row0 * a.ld[q], a.ld[q], a.n[q], 0, sl[k].rows_here, TB);
          c0 += a.n[q];
        }
      tile_zero_cols(Xs[k], c0, a.Kp1, TB);
    }
  }
  lds_barrier();
  TR(1);
  if (a.save_x != nullptr && (m == 0 || a.x_ms != 0)) {
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    float* sx = a.save_x + m * a.x_ms;
#pragma unroll
    for (int k = 0; k < NS; ++k)
      for (int col = c; col < a.Kp1; col += 32)
        for (int r = r0; r < sl[k].rows_here; r += NTHREADS >> 5) sx[(sl[k].row0 + r) * a.Kp1 + col] = Xs[k][r * LDX + col];
  }

  // ---- layer 1: all slots against one stream of W1; activations in place; tile maxima ---------------------------------------
  const float b1_0 = b1[64 * wave_col() + i], b1_1 = b1[64 * wave_col() + 32 + i];
  f32x16 acc1[2][MT][2];
#pragma unroll
  for (int k = 0; k < NS; ++k) wide_zero<MT>(acc1[k]);
  wide_gemm_slots<MT, NS>(Xs, w1, a.Kp1, acc1, ring1);
  PipeRing<RING> bring;
  pipe_prefetch(w2b, bring);
  NarrowRegs<NT> br;
  const float b2_0 = b2[64 * wave_col() + i], b2_1 = b2[64 * wave_col() + 32 + i];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    float mx = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float y = activate<ACT>(acc1[k][mt][nt][r] + (nt ? b1_1 : b1_0));
          acc1[k][mt][nt][r] = y;
          mx = fmaxf(mx, fabsf(y));
        }
    // the global plane copy feeds a contraction over rows: rows past the end of the batch are zero there
    if (a.save_h1p != nullptr && sl[k].rows_here < TB) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) >= sl[k].rows_here) acc1[k][mt][nt][r] = 0.f;
    }
    f16_tile_max_put(mx, sl[k].scr);
  }
  lds_barrier();                                   // every wave has read the input images (and posted its maxima)
  TR(2);
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    sl[k].e = f16_scale_exp(f16_tile_max_get(sl[k].scr));
    if (sl[k].e_out != nullptr && (int)threadIdx.x < (sl[k].rows_here + 31) / 32) sl[k].e_out[threadIdx.x] = sl[k].e;
  }

  // ---- epilogue 1 of P in the open ------------------------------------------------------------------------------------
  uint32_t word[MT][2];
  auto clear_words = [&] {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { word[mt][0] = 0; word[mt][1] = 0; }
  };
  clear_words();
  {
    const float sc = exp2i(sl[0].e);
    static_for<NG>([&](auto G) { e1_group<MT, decltype(G)::value>(acc1[0], sl[0], sc, word); });
  }
  narrow_prefetch<NT>(w3, 16 * NT, br);
  lds_barrier();                                   // planes of P complete
  TR(3);

  // ---- GEMM 2 of P, hosting epilogue 1 of Q ------------------------------------------------------------------------------
  f32x16 acc2[2][MT][2];
  wide_zero<MT>(acc2[0]);
  clear_words();
  if constexpr (HASQ) {
    const float sc = exp2i(sl[1].e);
#ifdef FWD_PIPE_NOGUEST
    bf_gemm_host<MT, 0, true, RING>(sl[0].Ps, w2b, acc2[0], bring, [](auto) {});
    TR(5);
    static_for<NG>([&](auto G) { e1_group<MT, decltype(G)::value>(acc1[1], sl[1], sc, word); });
#else
    bf_gemm_host<MT, 0, true, RING>(sl[0].Ps, w2b, acc2[0], bring, [&](auto S) {
      constexpr int s = decltype(S)::value;
      if constexpr (NG == 16) e1_group<MT, s>(acc1[1], sl[1], sc, word);
      else if constexpr (s % 2 == 0) e1_group<MT, s / 2>(acc1[1], sl[1], sc, word);
    });
#endif
  } else {
    bf_gemm_host<MT, 0, false, RING>(sl[0].Ps, w2b, acc2[0], bring, [](auto) {});
  }
  lds_barrier();                                   // planes of P consumed, planes of Q complete
  TR(4);

  // ---- GEMM 2 of Q, hosting epilogue 2 of P (its image goes where P's planes were) ---------------------------------------------
  clear_words();
  {
    const float inv = exp2i(-(sl[0].e + F16_WSHIFT));
    if constexpr (HASQ) {
      wide_zero<MT>(acc2[1]);
      bf_gemm_host<MT, 16, false, RING>(sl[1].Ps, w2b, acc2[1], bring, [&](auto S) {
        constexpr int s = decltype(S)::value;
        if constexpr (NG == 16) e2_group<ACT, MT, s>(acc2[0], sl[0], inv, b2_0, b2_1, word);
        else if constexpr (s % 2 == 0) e2_group<ACT, MT, s / 2>(acc2[0], sl[0], inv, b2_0, b2_1, word);
      });
      lds_barrier();                               // planes of Q consumed
#ifndef FWD_PIPE_NOGUEST
      TR(5);
#endif
      clear_words();
      const float invq = exp2i(-(sl[1].e + F16_WSHIFT));
      static_for<NG>([&](auto G) { e2_group<ACT, MT, decltype(G)::value>(acc2[1], sl[1], invq, b2_0, b2_1, word); });
    } else {
      static_for<NG>([&](auto G) { e2_group<ACT, MT, decltype(G)::value>(acc2[0], sl[0], inv, b2_0, b2_1, word); });
    }
  }
  lds_barrier();                                   // fp32 images complete
  TR(6);

  // ---- output layer of every slot: K split across the waves, partial sums meet in the slot's own LDS -----------------------------
  constexpr int Np = 16 * NT, MTN = 2 * MT, ROWS = TB;
  const int w = wave_id(), ii = lane & 15, q = lane >> 4;
  f32x4 acc3[NS][MTN][NT];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int mm = 0; mm < MTN; ++mm)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc3[k][mm][n][r] = 0.f;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const float* xa = Xs[k] + ii * LDX + 64 * w + 16 * q;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      f32x4 av[MTN];
#pragma unroll
      for (int mm = 0; mm < MTN; ++mm) av[mm] = *reinterpret_cast<const f32x4*>(xa + 16 * mm * LDX + 4 * s4);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int mm = 0; mm < MTN; ++mm)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc3[k][mm][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mm][u], br.b[4 * s4 + u][n], acc3[k][mm][n], 0, 0, 0);
    }
  }
  const int mycol = threadIdx.x % Np;
  const float bias3 = b3[mycol < a.nout ? mycol : 0];
  lds_barrier();                                   // every wave has read its K slice of the images
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    float* P = Xs[k] + w * (ROWS * Np);
#pragma unroll
    for (int mm = 0; mm < MTN; ++mm)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(16 * mm + 4 * q + r) * Np + 16 * n + ii] = acc3[k][mm][n][r];
  }
  lds_barrier();
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    float* out = a.out + m * a.out_mstride + sl[k].row0 * a.out_ld;
    constexpr int PER = ROWS * Np / NTHREADS;      // elements per thread: 2 .. 8
    float v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = threadIdx.x + u * NTHREADS;
      v[u] = ((Xs[k][e] + Xs[k][ROWS * Np + e]) + Xs[k][2 * ROWS * Np + e]) + Xs[k][3 * ROWS * Np + e];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = threadIdx.x + u * NTHREADS, row = e / Np, col = e % Np;
      if (row < sl[k].rows_here && col < a.nout) {
        float y = v[u] + bias3;
        if (a.out_mode == 1) y = a.max_action * tanhf(y);
        if (a.resid != nullptr) y += a.resid[(sl[k].row0 + row) * a.resid_ld + col];
        out[row * a.out_ld + col] = y;
      }
    }
  }
  TR(7);
}

template <int ACT, int MT, int NT>
__global__ __launch_bounds__(NTHREADS, MT == 1 ? 2 : 1) void k_mlp3_fwd_pipe(Mlp3FwdArgs a, Mlp3FwdArgs b, int members_a) {
  extern __shared__ __attribute__((aligned(16))) char pipe_lds[];
  const bool second = (int)blockIdx.y >= members_a;
  const Mlp3FwdArgs s = second ? b : a;
  const long long row0 = (long long)blockIdx.x * (64 * MT);
  if (row0 >= s.rows) return;
  const int m = second ? (int)blockIdx.y - members_a : (int)blockIdx.y;
  if (row0 + 32 * MT < s.rows) fwd_pipe_body<ACT, MT, NT, true>(s, m, pipe_lds, row0);
  else fwd_pipe_body<ACT, MT, NT, false>(s, m, pipe_lds, row0);
}

template <int MT, int NT>
static int launch_pipe_t(const Mlp3FwdArgs& a, int members_a, const Mlp3FwdArgs& b, int members_b, hipStream_t st) {
  constexpr size_t lds = 2 * split_lds_bytes<4, 32 * MT>();
  static bool once = false;
  if (!once) {
    int rc = allow_big_lds(k_mlp3_fwd_pipe<ACT_RELU, MT, NT>, 160 * 1024);
    if (rc) return rc;
    once = true;
  }
  const long long rows = a.rows > b.rows ? a.rows : b.rows;
  ProfScope prof(PROF_MLP_FWD, st);
  hipLaunchKernelGGL((k_mlp3_fwd_pipe<ACT_RELU, MT, NT>), dim3((unsigned)cdiv(rows, 64 * MT), (unsigned)(members_a + members_b)),
                     dim3(NTHREADS), lds, st, a, b, members_a);
  MB_LAUNCH_OK("k_mlp3_fwd_pipe");
  return 0;
}

static bool pipe_ok(const Mlp3FwdArgs& a) {
  // what the pipelined kernel does not carry: fp32 copies of layer 1, saves beyond 2 GB from a slot's base (32-bit offsets)
  return a.rows <= 0 || (a.save_h1 == nullptr && a.save_d1 == nullptr && a.save_d2 == nullptr && a.h1p_plane < (1LL << 29));
}

// 1 = launched, 0 = not applicable (the caller falls back to k_mlp3_fwd_bf), < 0 = error
int try_launch_mlp3_fwd_pipe(const Mlp3FwdArgs& a, int members_a, const Mlp3FwdArgs& b, int members_b, int act, hipStream_t st) {
#if FWD_PIPE_MT == 0
  return 0;
#else
  constexpr int MT = FWD_PIPE_MT;
  if (act != ACT_RELU || !pipe_ok(a) || !pipe_ok(b)) return 0;
  const int np3 = a.rows > 0 ? a.Np3 : b.Np3;
  if (np3 != 16 && np3 != 32) return 0;
  if (members_b > 0 && b.rows > 0 && a.rows > 0 && b.Np3 != a.Np3) return 0;
  const long long wgs = cdiv(a.rows > 0 ? a.rows : 0, 64 * MT) * members_a + cdiv(b.rows > 0 ? b.rows : 0, 64 * MT) * members_b;
  if (wgs < FWD_PIPE_MIN_WGS) return 0;
  int rc = np3 == 16 ? launch_pipe_t<MT, 1>(a, members_a, b, members_b, st) : launch_pipe_t<MT, 2>(a, members_a, b, members_b, st);
  return rc ? rc : 1;
#endif
}

}  // namespace mobody
