// Fused 3-layer MLP forward: one workgroup = 64 rows x one member, all three GEMMs on
// fp32 MFMA with the activations resident in LDS (never written to HBM unless the
// backward pass asks for them).  Roofline: MFMA f32 (2*(Kp1+256+Np3)*256 FLOP per row
// against (in+out)*4 bytes per row -> AI > 1000 F/B); weights (<= 340 KB per member)
// stream from L2.
#include <stdlib.h>

#include "common.h"
#include "layers.h"

namespace mobody {

template <int ACT, int MT, int RG>
__global__ __launch_bounds__(NTHREADS * RG, 2) void k_mlp3_fwd(Mlp3FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  constexpr int TB = 32 * MT * RG;                // rows of this workgroup's tile
  const int m = blockIdx.y;
  const long long row0 = (long long)blockIdx.x * TB;
  const int rows_here = (int)min((long long)TB, a.rows - row0);

  // ---- input tile: concat(src0, src1, src2), zero padded to Kp1 columns ----
  int c0 = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (a.n[k] > 0) {
      tile_load(Xs, c0, a.src[k] + row0 * a.ld[k], a.ld[k], a.n[k], 0, rows_here, TB);
      c0 += a.n[k];
    }
  }
  tile_zero_cols(Xs, c0, a.Kp1, TB);
  lds_barrier();
  if (a.save_x != nullptr && m == 0) {
    for (int idx = threadIdx.x; idx < rows_here * a.Kp1; idx += NTHREADS * RG) {
      const int r = idx / a.Kp1, c = idx - r * a.Kp1;
      a.save_x[(row0 + r) * a.Kp1 + c] = Xs[r * LDX + c];
    }
  }

  float* h1 = a.save_h1 ? a.save_h1 + ((long long)m * a.rows + row0) * HID : nullptr;
  float* h2 = a.save_h2 ? a.save_h2 + ((long long)m * a.rows + row0) * HID : nullptr;
  auto saver = [&](float* dst) {
    return [=](int row, int col, float y) {
      if (dst != nullptr && row < rows_here) dst[row * HID + col] = y;
    };
  };
  wide_layer<ACT, MT>(Xs, a.w1 + m * a.sw1, a.b1 + m * a.sb1, a.Kp1, saver(h1));
  wide_layer<ACT, MT>(Xs, a.w2 + m * a.sw2, a.b2 + m * a.sb2, HID, saver(h2));

  const float* b3 = a.b3 + m * a.sb3;
  float* out = a.out + m * a.out_mstride + row0 * a.out_ld;
  narrow_layer(Xs, a.w3 + m * a.sw3, HID, a.Np3, [&](int row, int col, float v) {
    if (row < rows_here && col < a.nout) {
      float y = v + b3[col];
      if (a.out_mode == 1) y = a.max_action * tanhf(y);
      out[row * a.out_ld + col] = y;
    }
  }, TB);
}

template <int ACT, int MT, int RG>
static int launch_fwd_t(const Mlp3FwdArgs& a, int members, hipStream_t stream) {
  size_t lds = (size_t)32 * MT * RG * LDX * sizeof(float);
  static bool once = false;
  if (!once) {
    int rc = allow_big_lds(k_mlp3_fwd<ACT, MT, RG>, 160 * 1024);
    if (rc) return rc;
    once = true;
  }
  dim3 grid((unsigned)cdiv(a.rows, 32 * MT * RG), (unsigned)members);
  ProfScope prof(PROF_MLP_FWD, stream);
  hipLaunchKernelGGL((k_mlp3_fwd<ACT, MT, RG>), grid, dim3(NTHREADS * RG), lds, stream, a);
  MB_LAUNCH_OK("k_mlp3_fwd");
  return 0;
}

int launch_mlp3_fwd(const Mlp3FwdArgs& a, int members, int act, hipStream_t stream) {
  if (a.rows <= 0) return 0;
  static const int shape = [] { const char* e = getenv("MOBODY_FWD_SHAPE"); return e ? atoi(e) : 0; }();   // tuning aid
  const bool tall = pick_tile_rows(a.rows, members) == 64;
  if (act == ACT_SWISH) {
    if (shape == 8) return launch_fwd_t<ACT_SWISH, 1, 2>(a, members, stream);
    return tall ? launch_fwd_t<ACT_SWISH, 2, 1>(a, members, stream) : launch_fwd_t<ACT_SWISH, 1, 1>(a, members, stream);
  }
  if (shape == 8) return launch_fwd_t<ACT_RELU, 1, 2>(a, members, stream);
  return tall ? launch_fwd_t<ACT_RELU, 2, 1>(a, members, stream) : launch_fwd_t<ACT_RELU, 1, 1>(a, members, stream);
}

}  // namespace mobody

using namespace mobody;

extern "C" int mobody_mlp3_forward(const float* blob, int in_dim, int out_dim, int members, const float* src0, int n0,
                                   const float* src1, int n1, int64_t rows, int out_mode, float max_action, float* out,
                                   float* save_x, float* save_h1, float* save_h2, void* stream) {
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
  return launch_mlp3_fwd(a, members, ACT_RELU, as_stream(stream));
}
