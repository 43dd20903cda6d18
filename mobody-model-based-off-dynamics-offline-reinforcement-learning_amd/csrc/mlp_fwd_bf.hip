// Fused 3-layer MLP forward with the 256 x 256 layer on the split-precision MFMA core (tile_bf.h): modes "bf16" / "bf16x2" /
// "bf16x3" / "f16x2" of the actor / twin-Q / reward-head forwards.  Layer 1 (K = the padded input width) and the output
// layer stay on exact fp32 MFMA; layer 1's epilogue writes its activations as 16-bit planes over the fp32 image (same LDS
// region: every layer separates its reads from its writes by a barrier), layer 2 contracts the planes with the weight
// planes kept in the T blob and writes an fp32 image for the output layer.
#include <stdlib.h>

#include "common.h"
#include "layers_bf.h"

#ifndef BF_RG_DEFAULT
#define BF_RG_DEFAULT 1
#endif
#ifndef FWD_L1_RING
#define FWD_L1_RING 3
#endif
#ifndef FWD_LOAD2
#define FWD_LOAD2 1            // 0: one tile_load per input source (A/B aid)
#endif
#ifndef FWD_BF_MT
#define FWD_BF_MT 1            // 1: 32-row forward tiles (product); 2: 64-row tiles, 0: 64-row from FWD_MT2_MIN_TILES on (experiments)
#endif
#ifndef FWD_MT2_MIN_TILES
#define FWD_MT2_MIN_TILES 1536 // 32-row tiles per launch (6 per CU) from which the 64-row tile is used
#endif
#ifndef FWD_F16_WAVES
#define FWD_F16_WAVES 4
#endif
namespace mobody {

// DS: training forward of a Swish net -- save_d1 / save_d2 receive the derivatives next to h1 (planes or rows) / h2
template <int ACT, int PM, int RG, int NT, bool DS = false, int MT = 1>
__device__ __forceinline__ void mlp3_fwd_bf_tile(const Mlp3FwdArgs& a, int m, float* Xs) {
  constexpr int TB = 32 * MT * RG;
  char* Ps = reinterpret_cast<char*>(Xs);
  float* scr = reinterpret_cast<float*>(Ps + split_scr_offset<PM, TB>());
  const long long row0 = (long long)blockIdx.x * TB;
  const int rows_here = (int)min((long long)TB, a.rows - row0);
  const bool full = rows_here == TB;
  const float* w1 = a.w1 + m * a.sw1;
  const s16x8* w2b = reinterpret_cast<const s16x8*>(a.w2_planes + m * a.planes_ms);
  const float* w3 = a.w3 + m * a.sw3;
  const float* b3 = a.b3 + m * a.sb3;
  TR(0);
  WideRingT<FWD_L1_RING> ring;                  // layer 1 is K = 24 .. 120: a short ring keeps the kernel at 128 registers
  wide_prefetch(w1, a.Kp1, ring);
  int c0 = 0;
  if (FWD_LOAD2 && RG == 1 && a.n[0] <= 32 && a.n[1] <= 32 && a.n[2] == 0) {       // state | action: both sources in one round trip
    tile_load2<TB>(Xs, a.src[0] + m * a.src_ms[0] + row0 * a.ld[0], a.ld[0], a.n[0],
                   a.n[1] > 0 ? a.src[1] + m * a.src_ms[1] + row0 * a.ld[1] : nullptr, a.ld[1], a.n[1], rows_here);
    c0 = a.n[0] + a.n[1];
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (a.n[k] > 0) {
        tile_load(Xs, c0, a.src[k] + m * a.src_ms[k] + row0 * a.ld[k], a.ld[k], a.n[k], 0, rows_here, TB);
        c0 += a.n[k];
      }
    }
  }
  tile_zero_cols(Xs, c0, a.Kp1, TB);
  lds_barrier();
  TR(1);
  if (a.save_x != nullptr && (m == 0 || a.x_ms != 0)) {
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    float* sx = a.save_x + m * a.x_ms;
    for (int col = c; col < a.Kp1; col += 32)
      for (int r = r0; r < rows_here; r += (NTHREADS * RG) >> 5) sx[(row0 + r) * a.Kp1 + col] = Xs[r * LDX + col];
  }
  float* h1 = a.save_h1 ? a.save_h1 + ((long long)m * a.rows + row0) * HID : nullptr;
  float* h2 = a.save_h2 ? a.save_h2 + ((long long)m * a.rows + row0) * HID : nullptr;
  float* d1 = DS ? a.save_d1 + ((long long)m * a.rows + row0) * HID : nullptr;
  float* d2 = DS ? a.save_d2 + ((long long)m * a.rows + row0) * HID : nullptr;
  const long long mtile = ((long long)m * cdiv(a.rows, 32) + row0 / 32) * HID;
  uint32_t* mask1 = a.mask1 ? a.mask1 + mtile : nullptr;
  uint32_t* mask2 = a.mask2 ? a.mask2 + mtile : nullptr;
  const int mg = (rows_here + 31) / 32;
  BfRing<PM> bring;
  PlaneSave gs{nullptr, 0, nullptr};
  if (a.save_h1p != nullptr) {
    gs.base = reinterpret_cast<short*>(a.save_h1p) + m * a.h1p_ms + (row0 / 8) * (HID * 8);
    gs.plane_stride = a.h1p_plane;
    gs.e_out = a.save_e1 + (long long)m * cdiv(a.rows, 32) + row0 / 32;
  }
  const int e1 = wide_layer_to_planes<ACT, MT, PM, TB, DS>(Xs, Ps, scr, w1, a.b1 + m * a.sb1, a.Kp1, ring,
                                                           [&] { bf_prefetch<PM>(w2b, bring); }, mask1, full, mg, rows_here, h1, gs, d1);
  TR(2);
  float* out = a.out + m * a.out_mstride + row0 * a.out_ld;
  auto emit = [&](int row, int col, float v, float bias) {
    if (row < rows_here && col < a.nout) {
      float y = v + bias;
      if (a.out_mode == 1) y = a.max_action * tanhf(y);
      if (a.resid != nullptr) y += a.resid[(row0 + row) * a.resid_ld + col];
      out[row * a.out_ld + col] = y;
    }
  };
  if constexpr (NT > 0) {
    NarrowRegs<NT> br;
    const int mycol = threadIdx.x % (16 * NT);
    float bias;
    bf_layer<ACT, MT, PM, TB, DS>(Xs, Ps, e1, w2b, a.b2 + m * a.sb2, bring, [&] {
      narrow_prefetch<NT>(w3, 16 * NT, br);
      bias = b3[mycol < a.nout ? mycol : 0];
    }, mask2, full, mg, rows_here, h2, d2);
    TR(4);
    narrow_run<TB / 16, NT>(Xs, br, [&](int row, int col, float v) { emit(row, col, v, bias); });
  } else {
    bf_layer<ACT, MT, PM, TB, DS>(Xs, Ps, e1, w2b, a.b2 + m * a.sb2, bring, [] {}, mask2, full, mg, rows_here, h2, d2);
    narrow_layer(Xs, w3, HID, a.Np3, [&](int row, int col, float v) { emit(row, col, v, b3[col < a.nout ? col : 0]); }, TB);
  }
  TR(5);
}

// one or two independent networks per launch (blockIdx.y < members_a -> net a), as k_mlp3_fwd2
// NT / NT2: output-layer width (16-column tiles; 0 = any) of net a / net b -- a twin-Q (one output) and an actor with more than
// 16 actions (pen: 24) still share a launch
template <int ACT, int PM, int RG, int NT, bool DS = false, int MT = 1, int NT2 = NT>
__global__ __launch_bounds__(NTHREADS * RG, (PM == 4 && RG == 1 && !DS && MT == 1) ? FWD_F16_WAVES : 2) void k_mlp3_fwd_bf(Mlp3FwdArgs a, Mlp3FwdArgs b, int members_a) {
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  const bool second = (int)blockIdx.y >= members_a;
  const Mlp3FwdArgs s = second ? b : a;
  if ((long long)blockIdx.x * (32 * MT * RG) >= s.rows) return;
  if constexpr (NT2 == NT) {
    mlp3_fwd_bf_tile<ACT, PM, RG, NT, DS, MT>(s, second ? (int)blockIdx.y - members_a : (int)blockIdx.y, Xs);
  } else {
    if (second) mlp3_fwd_bf_tile<ACT, PM, RG, NT2, DS, MT>(s, (int)blockIdx.y - members_a, Xs);
    else mlp3_fwd_bf_tile<ACT, PM, RG, NT, DS, MT>(s, (int)blockIdx.y, Xs);
  }
}

template <int ACT, int PM, int RG, int NT, bool DS = false, int MT = 1, int NT2 = NT>
static int launch_bf_t(const Mlp3FwdArgs& a, int members_a, const Mlp3FwdArgs& b, int members_b, hipStream_t st) {
  constexpr size_t lds = split_lds_bytes<PM, 32 * MT * RG>();
  static bool once = false;
  if (!once) {
    int rc = allow_big_lds(k_mlp3_fwd_bf<ACT, PM, RG, NT, DS, MT, NT2>, 160 * 1024);
    if (rc) return rc;
    once = true;
  }
  const long long rows = a.rows > b.rows ? a.rows : b.rows;
  ProfScope prof(PROF_MLP_FWD, st);
  hipLaunchKernelGGL((k_mlp3_fwd_bf<ACT, PM, RG, NT, DS, MT, NT2>), dim3((unsigned)cdiv(rows, 32 * MT * RG), (unsigned)(members_a + members_b)),
                     dim3(NTHREADS * RG), lds, st, a, b, members_a);
  MB_LAUNCH_OK("k_mlp3_fwd_bf");
  return 0;
}

template <int ACT, int PM, int RG>
static int launch_bf_nt(const Mlp3FwdArgs& a, int ma, const Mlp3FwdArgs& b, int mb, hipStream_t st) {
  if (RG == 1 && PM == 4 && mb > 0 && b.rows > 0 && a.rows > 0 && a.Np3 != b.Np3) {     // two nets, two output widths (16 | 32)
    if (a.Np3 == 16 && b.Np3 == 32) return launch_bf_t<ACT, PM, RG, 1, false, 1, 2>(a, ma, b, mb, st);
    if (a.Np3 == 32 && b.Np3 == 16) return launch_bf_t<ACT, PM, RG, 2, false, 1, 1>(a, ma, b, mb, st);
    return fail(MOBODY_E_ARG, "launch_mlp3_fwd_bf: nets of output widths %d and %d do not share a launch", a.Np3, b.Np3);
  }
  const int np3 = a.rows > 0 ? a.Np3 : b.Np3;
  return np3 == 16 ? launch_bf_t<ACT, PM, RG, 1>(a, ma, b, mb, st) : np3 == 32 ? launch_bf_t<ACT, PM, RG, 2>(a, ma, b, mb, st)
                                                                              : launch_bf_t<ACT, PM, RG, 0>(a, ma, b, mb, st);
}

// prec: 1 bf16, 2 bf16x2, 3 bf16x3, 4 f16x2.  Two ReLU nets (either may be empty: rows <= 0) or one Swish net.
int launch_mlp3_fwd_bf(const Mlp3FwdArgs& a, int members_a, const Mlp3FwdArgs& b, int members_b, int act, int prec, hipStream_t st) {
  // row groups per workgroup: 1 = 32-row tiles of 4 waves (measured best: twin-Q forward at 10 240 rows 28.8 us in bf16x3
  // against 41.5 us with two row groups sharing each weight fragment, 39.5 us in fp32); MOBODY_BF_RG=2 is a tuning aid
  static const int rg = tune_int("MOBODY_BF_RG", BF_RG_DEFAULT);
  if (a.rows <= 0 && b.rows <= 0) return 0;
  Mlp3FwdArgs x = a, y = b; int mx = members_a, my = members_b;
  if (x.rows <= 0) { x = b; mx = members_b; y.rows = 0; my = 0; }
  if (y.rows <= 0) my = 0;
  if (x.save_d1 != nullptr || x.save_d2 != nullptr) {  // training forward of a Swish net (dynamics pre-training): f16x2, one net
    if (act != ACT_SWISH || prec != 4 || my != 0 || !x.save_d1 || !x.save_d2)
      return fail(MOBODY_E_ARG, "launch_mlp3_fwd_bf: derivative saves need one Swish net in the f16x2 mode");
    const int np3 = x.Np3;
    return np3 == 16 ? launch_bf_t<ACT_SWISH, 4, 1, 1, true>(x, mx, y, 0, st) : np3 == 32 ? launch_bf_t<ACT_SWISH, 4, 1, 2, true>(x, mx, y, 0, st)
                                                                              : launch_bf_t<ACT_SWISH, 4, 1, 0, true>(x, mx, y, 0, st);
  }
  // Tile height of the f16x2 ReLU launches.  64-row tiles (MT = 2: each weight fragment feeds two row tiles, half the L2 -> CU
  // weight stream, two workgroups per CU) win only on a bare twin-Q forward of several generations (40 960 rows: 48.3 us
  // against 50.8; 10 240 rows: 19.8 against 17.4) and lose in the train step, whose forwards also save activations: c3 forward
  // 302 against 297 us per step, c4 280 against 258.  The product uses 32-row tiles; FWD_BF_MT = 2 / 0 builds the experiment.
  if (act == ACT_RELU && prec == 4) {
    const int np3 = x.rows > 0 ? x.Np3 : y.Np3;
    const long long tiles32 = cdiv(x.rows > 0 ? x.rows : 0, 32) * mx + cdiv(y.rows > 0 ? y.rows : 0, 32) * my;
    const bool tall = FWD_BF_MT == 2 || (FWD_BF_MT == 0 && tiles32 >= FWD_MT2_MIN_TILES);
    if (tall && np3 == 16) return launch_bf_t<ACT_RELU, 4, 1, 1, false, 2>(x, mx, y, my, st);
    if (tall && np3 == 32) return launch_bf_t<ACT_RELU, 4, 1, 2, false, 2>(x, mx, y, my, st);
  }
#define BF_CASE(ACT, PM) (rg == 1 ? launch_bf_nt<ACT, PM, 1>(x, mx, y, my, st) : launch_bf_nt<ACT, PM, 2>(x, mx, y, my, st))
  if (act == ACT_SWISH) return prec == 1 ? BF_CASE(ACT_SWISH, 1) : prec == 2 ? BF_CASE(ACT_SWISH, 2) : prec == 3 ? BF_CASE(ACT_SWISH, 3) : BF_CASE(ACT_SWISH, 4);
  return prec == 1 ? BF_CASE(ACT_RELU, 1) : prec == 2 ? BF_CASE(ACT_RELU, 2) : prec == 3 ? BF_CASE(ACT_RELU, 3) : BF_CASE(ACT_RELU, 4);
#undef BF_CASE
}

}  // namespace mobody
