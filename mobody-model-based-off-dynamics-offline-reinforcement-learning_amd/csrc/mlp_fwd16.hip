// Fused 3-layer MLP forward on the 16-row-granular core (tile16.h): one workgroup = 16*MT rows x one member, ~one
// workgroup per CU.  Same arguments, outputs and optional saves as k_mlp3_fwd (mlp_fwd.hip); the ReLU sign words are
// per 16-row group here (mask16[member][ceil(rows/16)][256], bit r = row 16g + r), because a tile of 16*MT rows
// does not end on a 32-row boundary.
#include <stdlib.h>

#include "common.h"
#include "layers.h"
#include "tile16.h"

namespace mobody {

template <int ACT, int MT, class Between>
__device__ __forceinline__ void layer16(float* Xs, const float* __restrict__ W, const float* __restrict__ b, int Kp,
                                        Ring16& ring, float* h, uint32_t* mask, int rows_here, Between&& between) {
  const int lane = lane_id(), i = lane & 15, q = lane >> 4, w = wave_id();
  float bias[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) bias[n] = b[64 * w + 16 * n + i];
  f32x4 acc[MT][4];
  zero16<MT>(acc);
  gemm16<MT>(Xs, W, Kp, acc, ring);
  between();
  lds_barrier();
  const bool full = rows_here == 16 * MT;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    uint32_t bits[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * m + 4 * q + r, col = 64 * w + 16 * n + i;
        const float y = activate<ACT>(acc[m][n][r] + bias[n]);
        Xs[row * LDX + col] = y;
        if (h != nullptr && (full || row < rows_here)) h[row * HID + col] = y;
        bits[n] |= (uint32_t)(y > 0.f) << (4 * q + r);
      }
    if (mask != nullptr && 16 * m < rows_here) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        uint32_t word = bits[n];
        word |= (uint32_t)__shfl_xor((int)word, 16);
        word |= (uint32_t)__shfl_xor((int)word, 32);
        if (q == 0) mask[m * HID + 64 * w + 16 * n + i] = word;
      }
    }
  }
  lds_barrier();
}

template <int ACT, int MT, int NT>
__global__ __launch_bounds__(NTHREADS, 1) void k_mlp3_fwd16(Mlp3FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  constexpr int TB = 16 * MT;
  const int m = blockIdx.y;
  const long long row0 = (long long)blockIdx.x * TB;
  const int rows_here = (int)min((long long)TB, a.rows - row0);
  const float* w1 = a.w1 + m * a.sw1;
  const float* w2 = a.w2 + m * a.sw2;
  const float* w3 = a.w3 + m * a.sw3;
  const float* b3 = a.b3 + m * a.sb3;
  Ring16 ring;
  prefetch16(w1, a.Kp1, ring);

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
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    for (int col = c; col < a.Kp1; col += 32)
      for (int r = r0; r < rows_here; r += NTHREADS >> 5) a.save_x[(row0 + r) * a.Kp1 + col] = Xs[r * LDX + col];
  }
  float* h1 = a.save_h1 ? a.save_h1 + ((long long)m * a.rows + row0) * HID : nullptr;
  float* h2 = a.save_h2 ? a.save_h2 + ((long long)m * a.rows + row0) * HID : nullptr;
  const long long mtile = ((long long)m * cdiv(a.rows, 16) + row0 / 16) * HID;
  uint32_t* mask1 = a.mask1 ? a.mask1 + mtile : nullptr;
  uint32_t* mask2 = a.mask2 ? a.mask2 + mtile : nullptr;

  layer16<ACT, MT>(Xs, w1, a.b1 + m * a.sb1, a.Kp1, ring, h1, mask1, rows_here, [&] { prefetch16(w2, HID, ring); });
  NarrowRegs<NT> br;
  const int mycol = threadIdx.x % (16 * NT);
  float bias;
  layer16<ACT, MT>(Xs, w2, a.b2 + m * a.sb2, HID, ring, h2, mask2, rows_here, [&] {
    narrow_prefetch<NT>(w3, 16 * NT, br);
    bias = b3[mycol < a.nout ? mycol : 0];
  });
  float* out = a.out + m * a.out_mstride + row0 * a.out_ld;
  narrow_run<MT, NT>(Xs, br, [&](int row, int col, float v) {
    if (row < rows_here && col < a.nout) {
      float y = v + bias;
      if (a.out_mode == 1) y = a.max_action * tanhf(y);
      out[row * a.out_ld + col] = y;
    }
  });
}

template <int ACT, int MT, int NT>
static int launch16_t(const Mlp3FwdArgs& a, int members, hipStream_t stream) {
  constexpr size_t lds = (size_t)16 * MT * LDX * sizeof(float);
  static bool once = false;
  if (!once) {
    int rc = allow_big_lds(k_mlp3_fwd16<ACT, MT, NT>, lds);
    if (rc) return rc;
    once = true;
  }
  dim3 grid((unsigned)cdiv(a.rows, 16 * MT), (unsigned)members);
  ProfScope prof(PROF_MLP_FWD, stream);
  hipLaunchKernelGGL((k_mlp3_fwd16<ACT, MT, NT>), grid, dim3(NTHREADS), lds, stream, a);
  MB_LAUNCH_OK("k_mlp3_fwd16");
  return 0;
}

// Rows per workgroup (in 16-row tiles) such that the grid is about one workgroup per CU.
int pick_mt16(long long row_members) {
  static const int forced = [] { const char* e = getenv("MOBODY_CORE16_MT"); return e ? atoi(e) : 0; }();   // tuning aid
  if (forced > 0) return forced;
  const long long need = cdiv(row_members, 256LL * 16);
  for (int mt : {2, 4, 5, 6, 8})
    if (mt >= need) return mt;
  return 8;
}

template <int ACT, int NT>
static int launch16_mt(const Mlp3FwdArgs& a, int members, int mt, hipStream_t stream) {
  switch (mt) {
    case 2: return launch16_t<ACT, 2, NT>(a, members, stream);
    case 4: return launch16_t<ACT, 4, NT>(a, members, stream);
    case 5: return launch16_t<ACT, 5, NT>(a, members, stream);
    case 6: return launch16_t<ACT, 6, NT>(a, members, stream);
    default: return launch16_t<ACT, 8, NT>(a, members, stream);
  }
}

// ReLU nets with a 16- or 32-wide padded output layer only (the shapes of the actor / twin-Q / V / classifier nets).
int launch_mlp3_fwd16(const Mlp3FwdArgs& a, int members, int mt, hipStream_t stream) {
  if (a.rows <= 0) return 0;
  if (a.Np3 == 16) return launch16_mt<ACT_RELU, 1>(a, members, mt, stream);
  return launch16_mt<ACT_RELU, 2>(a, members, mt, stream);
}

}  // namespace mobody
