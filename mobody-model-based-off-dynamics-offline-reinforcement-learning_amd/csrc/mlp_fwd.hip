// Fused 3-layer MLP forward: one workgroup = one 32-row tile (64 with MOBODY_TILE_ROWS=64) x one member, all three
// GEMMs on fp32 MFMA with the activations resident in LDS (written to HBM only when the backward pass asks for
// them: x / h1 / h2 for the weight gradients, 32 B/row of ReLU sign words for the masks).  k_mlp3_fwd2 runs two
// independent networks in one launch.  Roofline: MFMA f32 (2*(Kp1+256+Np3)*256 FLOP per row against (in+out)*4
// bytes per row -> AI > 1000 F/B); weights (<= 340 KB per member) stream from L2.
#include <stdlib.h>

#include "common.h"
#include "layers.h"

namespace mobody {

// Output layer of the forward (256 -> nout <= 16*NT) through the K-split narrow layer; NT = 0: generic row-split path.
template <int ACT, int MT, int RG, int NT>
__device__ __forceinline__ void mlp3_fwd_tail(const Mlp3FwdArgs& a, int m, float* Xs, WideRing& ring, float* h2,
                                              uint32_t* mask2, long long row0, int rows_here) {
  constexpr int TB = 32 * MT * RG;
  const float* w3 = a.w3 + m * a.sw3;
  const float* b3 = a.b3 + m * a.sb3;
  float* out = a.out + m * a.out_mstride + row0 * a.out_ld;
  const bool full = rows_here == TB;
  auto save_h2 = [=](auto guarded, int row, int col, float y) {
    if (h2 != nullptr && (!decltype(guarded)::value || row < rows_here)) h2[row * HID + col] = y;
  };
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
    // b3 of this thread's output columns: element e = threadIdx.x + 256 k has column e % (16 NT), the same for all k
    const int mycol = threadIdx.x % (16 * NT);
    float bias;
    wide_layer<ACT, MT>(Xs, a.w2 + m * a.sw2, a.b2 + m * a.sb2, HID, ring, save_h2, [&] {
      narrow_prefetch<NT>(w3, 16 * NT, br);
      bias = b3[mycol < a.nout ? mycol : 0];
    }, mask2, full, (rows_here + 31) / 32);
    TR(4);
    narrow_run<TB / 16, NT>(Xs, br, [&](int row, int col, float v) { emit(row, col, v, bias); });
  } else {
    wide_layer<ACT, MT>(Xs, a.w2 + m * a.sw2, a.b2 + m * a.sb2, HID, ring, save_h2, [] {}, mask2, full, (rows_here + 31) / 32);
    TR(4);
    narrow_layer(Xs, w3, HID, a.Np3, [&](int row, int col, float v) { emit(row, col, v, b3[col < a.nout ? col : 0]); }, TB);
  }
}

// NT: 16-column tiles of the output layer handled by the K-split narrow layer (Np3 == 16*NT), or 0 = any Np3.
template <int ACT, int MT, int RG, int NT>
__device__ __forceinline__ void mlp3_fwd_tile(const Mlp3FwdArgs& a, int m, float* Xs) {
  constexpr int TB = 32 * MT * RG;                // rows of this workgroup's tile
  const long long row0 = (long long)blockIdx.x * TB;
  const int rows_here = (int)min((long long)TB, a.rows - row0);
  const float* w1 = a.w1 + m * a.sw1;
  const float* w2 = a.w2 + m * a.sw2;
  TR(0);
  WideRing ring;
  wide_prefetch(w1, a.Kp1, ring);                 // W1 fragments travel while the input tile is fetched

  // ---- input tile: concat(src0, src1, src2), zero padded to Kp1 columns ----
  int c0 = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (a.n[k] > 0) {
      tile_load(Xs, c0, a.src[k] + m * a.src_ms[k] + row0 * a.ld[k], a.ld[k], a.n[k], 0, rows_here, TB);
      c0 += a.n[k];
    }
  }
  tile_zero_cols(Xs, c0, a.Kp1, TB);
  lds_barrier();
  TR(1);
  if (a.save_x != nullptr && (m == 0 || a.x_ms != 0)) {   // same thread <-> element map as tile_load (no division)
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    float* sx = a.save_x + m * a.x_ms;
    for (int col = c; col < a.Kp1; col += 32)
      for (int r = r0; r < rows_here; r += (NTHREADS * RG) >> 5) sx[(row0 + r) * a.Kp1 + col] = Xs[r * LDX + col];
  }

  float* h1 = a.save_h1 ? a.save_h1 + ((long long)m * a.rows + row0) * HID : nullptr;
  float* h2 = a.save_h2 ? a.save_h2 + ((long long)m * a.rows + row0) * HID : nullptr;
  const long long mtile = ((long long)m * cdiv(a.rows, 32) + row0 / 32) * HID;      // this tile's first mask word
  uint32_t* mask1 = a.mask1 ? a.mask1 + mtile : nullptr;
  uint32_t* mask2 = a.mask2 ? a.mask2 + mtile : nullptr;
  wide_layer<ACT, MT>(Xs, w1, a.b1 + m * a.sb1, a.Kp1, ring,
                      [=](auto guarded, int row, int col, float y) {
                        if (h1 != nullptr && (!decltype(guarded)::value || row < rows_here)) h1[row * HID + col] = y;
                      },
                      [&] { wide_prefetch(w2, HID, ring); }, mask1, rows_here == TB, (rows_here + 31) / 32);
  TR(2);
  mlp3_fwd_tail<ACT, MT, RG, NT>(a, m, Xs, ring, h2, mask2, row0, rows_here);
  TR(5);
}

template <int ACT, int MT, int RG, int NT>
__global__ __launch_bounds__(NTHREADS * RG, 2) void k_mlp3_fwd(Mlp3FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  mlp3_fwd_tile<ACT, MT, RG, NT>(a, blockIdx.y, Xs);
}

// Two independent networks in ONE launch (blockIdx.y < members_a -> net a, else net b): Q(s,a) with pi(s') in the
// critic phase, Q(s_t,a_t) with pi(s) in the actor phase.  A 1-member launch of 10 k rows is only 320 workgroups
// (1.25 per CU: half the chip idles through the second round); merged with the twin-Q launch the grid is ~3.5
// workgroups per CU and one generation shorter.  The argument block is SELECTED (scalar selects), not branched on:
// two inlined copies of the tile body in an if/else made hipcc keep both live (190 VGPRs, half the occupancy).
template <int ACT, int MT, int NT>
__global__ __launch_bounds__(NTHREADS, 2) void k_mlp3_fwd2(Mlp3FwdArgs a, Mlp3FwdArgs b, int members_a) {
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  const bool second = (int)blockIdx.y >= members_a;
  const Mlp3FwdArgs s = second ? b : a;
  if ((long long)blockIdx.x * (32 * MT) >= s.rows) return;       // the two nets may differ in rows (grid.x = max)
  mlp3_fwd_tile<ACT, MT, 1, NT>(s, second ? (int)blockIdx.y - members_a : (int)blockIdx.y, Xs);
}

template <int ACT, int MT, int RG, int NT>
static int launch_fwd_t(const Mlp3FwdArgs& a, int members, hipStream_t stream) {
  size_t lds = (size_t)32 * MT * RG * LDX * sizeof(float);
  static bool once = false;
  if (!once) {
    int rc = allow_big_lds(k_mlp3_fwd<ACT, MT, RG, NT>, 160 * 1024);
    if (rc) return rc;
    once = true;
  }
  dim3 grid((unsigned)cdiv(a.rows, 32 * MT * RG), (unsigned)members);
  ProfScope prof(PROF_MLP_FWD, stream);
  hipLaunchKernelGGL((k_mlp3_fwd<ACT, MT, RG, NT>), grid, dim3(NTHREADS * RG), lds, stream, a);
  MB_LAUNCH_OK("k_mlp3_fwd");
  return 0;
}

template <int MT, int NT>
static int launch_fwd2_t(const Mlp3FwdArgs& a, int members_a, const Mlp3FwdArgs& b, int members_b, hipStream_t stream) {
  size_t lds = (size_t)32 * MT * LDX * sizeof(float);
  static bool once = false;
  if (!once) {
    int rc = allow_big_lds(k_mlp3_fwd2<ACT_RELU, MT, NT>, 160 * 1024);
    if (rc) return rc;
    once = true;
  }
  const long long rows = a.rows > b.rows ? a.rows : b.rows;
  dim3 grid((unsigned)cdiv(rows, 32 * MT), (unsigned)(members_a + members_b));
  ProfScope prof(PROF_MLP_FWD, stream);
  hipLaunchKernelGGL((k_mlp3_fwd2<ACT_RELU, MT, NT>), grid, dim3(NTHREADS), lds, stream, a, b, members_a);
  MB_LAUNCH_OK("k_mlp3_fwd2");
  return 0;
}

static int narrow_tiles(int Np3) { return Np3 == 16 ? 1 : Np3 == 32 ? 2 : 0; }

// ReLU nets a and b in one launch (either may be empty)
int launch_mlp3_fwd_pair(const Mlp3FwdArgs& a, int members_a, const Mlp3FwdArgs& b, int members_b, hipStream_t stream) {
  if (a.rows <= 0) return launch_mlp3_fwd(b, members_b, ACT_RELU, stream);
  if (b.rows <= 0) return launch_mlp3_fwd(a, members_a, ACT_RELU, stream);
  static const bool split = tune_int("MOBODY_NO_FWD_PAIR", 0) != 0;   // tuning aid (diagnostic build)
  if (split || a.Np3 != b.Np3) {                  // the merged kernel is specialised on one output-layer width
    int rc = launch_mlp3_fwd(a, members_a, ACT_RELU, stream);
    return rc ? rc : launch_mlp3_fwd(b, members_b, ACT_RELU, stream);
  }
  const bool tall = pick_tile_rows(a.rows, members_a) == 64;
  const int nt = narrow_tiles(a.Np3);
  if (tall) return nt == 1 ? launch_fwd2_t<2, 1>(a, members_a, b, members_b, stream)
                 : nt == 2 ? launch_fwd2_t<2, 2>(a, members_a, b, members_b, stream)
                           : launch_fwd2_t<2, 0>(a, members_a, b, members_b, stream);
  return nt == 1 ? launch_fwd2_t<1, 1>(a, members_a, b, members_b, stream)
       : nt == 2 ? launch_fwd2_t<1, 2>(a, members_a, b, members_b, stream)
                 : launch_fwd2_t<1, 0>(a, members_a, b, members_b, stream);
}

template <int ACT>
static int launch_fwd_act(const Mlp3FwdArgs& a, int members, hipStream_t stream) {
  static const int shape = tune_int("MOBODY_FWD_SHAPE", 0);   // tuning aid (diagnostic build)
  if (shape == 8) return launch_fwd_t<ACT, 1, 2, 0>(a, members, stream);
  const bool tall = pick_tile_rows(a.rows, members) == 64;
  const int nt = narrow_tiles(a.Np3);
  if (tall) return nt == 1 ? launch_fwd_t<ACT, 2, 1, 1>(a, members, stream)
                 : nt == 2 ? launch_fwd_t<ACT, 2, 1, 2>(a, members, stream)
                           : launch_fwd_t<ACT, 2, 1, 0>(a, members, stream);
  return nt == 1 ? launch_fwd_t<ACT, 1, 1, 1>(a, members, stream)
       : nt == 2 ? launch_fwd_t<ACT, 1, 1, 2>(a, members, stream)
                 : launch_fwd_t<ACT, 1, 1, 0>(a, members, stream);
}

int launch_mlp3_fwd(const Mlp3FwdArgs& a, int members, int act, hipStream_t stream) {
  if (a.rows <= 0) return 0;
  return act == ACT_SWISH ? launch_fwd_act<ACT_SWISH>(a, members, stream) : launch_fwd_act<ACT_RELU>(a, members, stream);
}

}  // namespace mobody

using namespace mobody;

extern "C" int mobody_mlp3_forward(const float* blob, const float* blob_T, int precision, int in_dim, int out_dim,
                                   int members, const float* src0, int n0, const float* src1, int n1, int64_t rows,
                                   int out_mode, float max_action, float* out, float* save_x, float* save_h1,
                                   float* save_h2, void* stream) {
  MobodyMlpLayout L;
  int rc = mobody_mlp_layout(in_dim, out_dim, members, &L);
  if (rc) return rc;
  MB_REQUIRE(rows >= 0, "mobody_mlp3_forward: rows < 0");
  if (rows == 0) return 0;
  MB_REQUIRE(blob && src0 && out, "mobody_mlp3_forward: null pointer");
  MB_REQUIRE(n0 + n1 == in_dim && n0 > 0 && n1 >= 0 && (n1 == 0 || src1), "mobody_mlp3_forward: n0+n1=%d != in_dim=%d", n0 + n1, in_dim);
  Mlp3FwdArgs a{};
  a.src[0] = src0; a.ld[0] = n0; a.n[0] = n0;
  a.src[1] = src1; a.ld[1] = n1; a.n[1] = n1;
  a.src[2] = nullptr; a.ld[2] = 0; a.n[2] = 0;
  a.w1 = blob + L.w1; a.b1 = blob + L.b1; a.w2 = blob + L.w2; a.b2 = blob + L.b2; a.w3 = blob + L.w3; a.b3 = blob + L.b3;
  a.sw1 = a.sb1 = a.sw2 = a.sb2 = a.sw3 = a.sb3 = L.member_floats;
  a.Kp1 = L.Kp1; a.Np3 = L.Np3; a.nout = out_dim; a.rows = rows;
  a.out = out; a.out_mstride = rows * out_dim; a.out_ld = out_dim;
  a.save_x = save_x; a.save_h1 = save_h1; a.save_h2 = save_h2;
  a.out_mode = out_mode; a.max_action = max_action;
  MB_REQUIRE(precision >= 0 && precision <= 4 && (precision == 0 || blob_T), "mobody_mlp3_forward: precision %d needs the T blob", precision);
  if (precision == 0) return launch_mlp3_fwd(a, members, ACT_RELU, as_stream(stream));
  a.w2_planes = reinterpret_cast<const unsigned short*>(blob_T + L.w2p);
  a.planes_ms = 2 * L.t_member_floats;
  return launch_mlp3_fwd_bf(a, members, Mlp3FwdArgs{}, 0, ACT_RELU, precision, as_stream(stream));
}
