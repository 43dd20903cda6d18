// Backward kernels of the 3-layer ReLU MLP (actor / twin-Q), fp32 MFMA.
//
//   k_mlp3_bwd   per (32-row tile, member): dz3 (read, or formed in the prologue from the row-wise inputs: TD error,
//                -p_w dmin(Q), actor d(pre-tanh)) -> dz2 = (dz3 W3^T) * [h2>0] -> dz1 = (dz2 W2^T) * [h1>0]
//                (-> dx = dz1 W1^T for the frozen-Q pass of the actor update), plus the bias-gradient and loss
//                partial sums of the tile.  Masks come from the forward's sign words (or the saved activations);
//                W^T blobs are streamed as MFMA B operands exactly like the forward weights.
//                                                                    (autograd of mobody.py:35-48)
//   k_wgrad      dW[k][n] = sum_rows A[row][k] * dZ[row][n]: rows are the contraction index, both
//                operands are read straight from global memory in MFMA fragment order (a wave
//                instruction = two full 128-byte lines); split-K over row slices, the four waves of a
//                workgroup reduce through LDS and write one deterministic partial slab.
//   k_grad_reduce slabs + bias partials -> gradient blob (deterministic, no atomics); on one GPU it applies the
//                Adam / Polyak step to each element it has just reduced, and one extra workgroup finishes the losses.
//
// Roofline: k_mlp3_bwd and k_wgrad are MFMA-f32 bound (2*256*256 FLOP per row and layer against
// ~2 KB of activations per row); k_grad_reduce is HBM/L2 streaming.
#include <stdlib.h>

#include "common.h"
#include "layers_bf.h"
#include "train.h"

#ifndef BWD_MASK_PREFETCH
#define BWD_MASK_PREFETCH 1     // 0: fetch a layer's mask operand inside its epilogue (A/B aid)
#endif

namespace mobody {

// Masked epilogue of a backward wide layer, in two parts around the barrier that separates the GEMM's LDS reads from the
// epilogue's LDS writes.
//
// wide_mask_apply (before the barrier, registers only): acc <- dz = (acc * prescale) * [h > 0].  The mask values of a lane
// are fetched with UNCONDITIONAL loads from clamped rows -- a `cond ? load : 0` select makes hipcc branch around every
// load and drain vmcnt(0) after it (64 serialized HBM round trips per layer, measured 60 % SQ_WAIT_ANY) -- so all of them
// are in flight together and cost one round trip.  Returns the lane's largest |dz| (the f16 mode's tile scale).
// MASK: 0 = ReLU mask from the saved activations (h > 0), 1 = ReLU mask from the forward's sign words,
//       2 = Swish: multiply by the saved derivative d = dy/dz (h points at save_d of the forward, mobody_module.py:9-15).
// The mask operand of one layer for this lane: two sign words per 32-row tile (MASK == 1) or the 32 saved values per tile
// (MASK 0: activations, 2: Swish derivatives).  It is requested BEFORE the GEMM whose epilogue consumes it (mask_fetch):
// fetched inside the epilogue it put one L2 / HBM round trip (~1 us) between each GEMM and its epilogue in every workgroup of
// the launch at the same moment (phase trace of a lone workgroup: mask epilogue 2.5 us of a 10.9 us backward).  All loads are
// UNCONDITIONAL from clamped rows (see above).
template <int MT, int MASK>
struct MaskPre {
  uint32_t w[MASK == 1 ? MT : 1][2];
  float v[MASK == 1 ? 1 : MT][2][MASK == 1 ? 1 : 16];
};
template <int MT, int MASK>
__device__ __forceinline__ void mask_fetch(MaskPre<MT, MASK>& p, const float* __restrict__ h, const uint32_t* __restrict__ bits,
                                           int rows_here) {
  const int lane = lane_id(), w = wave_id(), i = lane & 31, hh = lane >> 5;
  if constexpr (MASK == 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int g = min(mt, (rows_here + 31) / 32 - 1);      // groups past the end of the batch have no words (their rows are masked)
      p.w[mt][0] = bits[g * HID + 64 * w + i]; p.w[mt][1] = bits[g * HID + 64 * w + 32 + i];
    }
  } else {
    const float* hp = h + 64 * w + i;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh, rows_here - 1);
        p.v[mt][0][r] = hp[row * HID];
        p.v[mt][1][r] = hp[row * HID + 32];
      }
  }
}

template <int MT, int MASK>
__device__ __forceinline__ float wide_mask_apply(f32x16 (&acc)[MT][2], const MaskPre<MT, MASK>& pre, int rows_here, float prescale) {
  const int hh = lane_id() >> 5;
  constexpr bool BITS = MASK == 1;
  const auto& hv = pre.v;
  const auto& mw = pre.w;
  float mx = 0.f;
  f32x16 pa[MT][2];                                  // acc * prescale as whole vectors (v_pk_mul_f32)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { pa[mt][0] = acc[mt][0] * prescale; pa[mt][1] = acc[mt][1] * prescale; }
  // full tile (wave uniform): no per-element row guard
  auto sweep = [&](auto guarded) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rb = (r & 3) + 8 * (r >> 2) + 4 * hh;
          const bool valid = !decltype(guarded)::value || 32 * mt + rb < rows_here;
          const float a = pa[mt][nt][r];
          float dz;
          if constexpr (MASK == 2) {
            dz = valid ? a * hv[mt][nt][r] : 0.f;
          } else {
            bool on;
            if constexpr (BITS) on = (mw[mt][nt] >> rb) & 1u;
            else on = hv[mt][nt][r] > 0.f;
            dz = (on && valid) ? a : 0.f;
          }
          acc[mt][nt][r] = dz;
          mx = fmaxf(mx, fabsf(dz));
        }
  };
  if (rows_here == 32 * MT) sweep(std::false_type{});
  else sweep(std::true_type{});
  return mx;
}

// wide_store_colsum (after the barrier): dz -> LDS (the fp32 image, or -- PM > 0 -- the 16-bit planes the next GEMM
// contracts on the split-precision core, scaled by 2^e in the f16 mode), optional global copy (the weight-gradient operand,
// always fp32) and the per-column sums of the tile (bias gradient).  Lanes < 32 end up with the sums of columns
// 64w + 32nt + (lane&31), nt = 0,1.
template <int MT, int PM>
__device__ __forceinline__ void wide_store_colsum(f32x16 (&acc)[MT][2], float* Xs, float* gdst, int rows_here, int e,
                                                  float (&cs)[2], const PlaneSave& gs = PlaneSave{nullptr, 0, nullptr}) {
  const int lane = lane_id(), w = wave_id();
  const int i = lane & 31, hh = lane >> 5;
  cs[0] = cs[1] = 0.f;
  if constexpr (PM > 0) {
    const float sc = Split<PM>::F16 ? exp2i(e) : 1.f;
    if (gs.base != nullptr && (int)threadIdx.x < MT) gs.e_out[threadIdx.x] = e;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x16 ys = Split<PM>::F16 ? acc[mt][nt] * sc : acc[mt][nt];               // v_pk_mul_f32
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float y4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) y4[j] = ys[4 * g + j];
          planes_store4<PM, 32 * MT>(reinterpret_cast<char*>(Xs), 64 * w + 32 * nt + i, 8 * mt + 2 * g + hh, y4, gs.base, gs.plane_stride);
        }
      }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dz = acc[mt][nt][r];
        if constexpr (PM == 0) Xs[(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh) * LDX + 64 * w + 32 * nt + i] = dz;
        cs[nt] += dz;
      }
  if (gdst != nullptr) wide_store_rows<MT>(acc, gdst, rows_here == 32 * MT, rows_here);      // one branch-free burst on full tiles
  cs[0] += __shfl_xor(cs[0], 32);
  cs[1] += __shfl_xor(cs[1], 32);
}

// Seed prologue (BwdSeed modes 1-3): fills Xs[r][0..Np3) for the TB rows of this tile.  All global loads of a pass
// are independent (one round trip); the loss partials are reduced through `red` (static LDS, 8 floats).
__device__ __forceinline__ void bwd_seed(const Mlp3BwdArgs& a, float* Xs, float* red, int m, long long row0, int rows_here,
                                         int TB) {
  const BwdSeed& sd = a.seed;
  const int Np3 = a.Np3;
  const int t = threadIdx.x;
  float l0 = 0.f, l1 = 0.f;
  if (sd.mode == 1 || sd.mode == 2) {
    if (t < TB) {
      const bool ok = t < rows_here;
      const long long row = row0 + (ok ? t : 0);
      float v = 0.f;
      if (sd.mode == 1) {
        const float qn = sd.qnext ? sd.qnext[row] : fminf(sd.qt[row], sd.qt[a.rows + row]);
        const float y = sd.r[row] + sd.nd[row] * sd.gamma * qn;
        const float d = sd.q[(long long)m * a.rows + row] - y;
        v = ok ? 2.f * d * sd.inv_ng : 0.f;
        l0 = ok ? d * d : 0.f;
      } else {
        const ActorRowArgs& r = sd.ar;
        const float q0 = r.qp[row], q1 = r.qp[r.N + row];
        const float c = -policy_weight(r) / (float)r.Ng;
        const float g0 = q0 < q1 ? 1.f : (q0 == q1 ? 0.5f : 0.f);
        v = ok ? c * (m == 0 ? g0 : 1.f - g0) : 0.f;
        if (m == 0 && ok && row < r.Nt) r.bcw[row] = bc_weight(r, row);
      }
      float* o = Xs + t * LDX;
      o[0] = v;
      for (int c = 1; c < Np3; ++c) o[c] = 0.f;
      if (sd.dz3_out != nullptr && ok) {
        float* g = sd.dz3_out + ((long long)m * a.rows + row) * Np3;
        g[0] = v;
        for (int c = 1; c < Np3; ++c) g[c] = 0.f;
      }
    }
  } else {                                          // mode 3: one thread per (row, column)
    const ActorRowArgs& r = sd.ar;
    const float wscale = r.h.bc_coef * 2.f / ((float)r.Ntg * (float)r.A);
    for (int e = t; e < TB * Np3; e += NTHREADS) {
      const int rr = e / Np3, j = e - rr * Np3;
      const bool ok = rr < rows_here && j < r.A;
      const long long row = row0 + (rr < rows_here ? rr : 0);
      const int jc = j < r.A ? j : 0;
      const bool bc = row < r.Nt;
      const float p = r.pi[row * r.A + jc];
      const float d0 = r.dxa[row * r.A + jc], d1 = r.dxa[(r.N + row) * r.A + jc];
      const float act = r.act[row * r.A + jc];
      const float w = r.bcw[bc ? row : 0];
      float v = 0.f;
      if (ok) {
        float d = d0 + d1;
        if (bc) { const float df = p - act; d += wscale * w * df; l1 += w * (df * df); }
        const float th = p / r.h.max_action;
        v = d * r.h.max_action * (1.f - th * th);                   // d tanh
        if (j == 0) l0 = l0 - fminf(r.qp[row], r.qp[r.N + row]);
      }
      Xs[rr * LDX + j] = v;
      if (rr < rows_here) sd.dz3_out[(row0 + rr) * Np3 + j] = v;
    }
  }
  if (sd.mode == 2) return;
  // loss partials of this tile
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { l0 += __shfl_xor(l0, o); l1 += __shfl_xor(l1, o); }
  if ((t & 63) == 0) { red[t >> 6] = l0; red[4 + (t >> 6)] = l1; }
  lds_barrier();                                     // (LDS only: __syncthreads would also drain the dz3_out stores, ~1 us per tile)
  if (t == 0) {
    l0 = red[0] + red[1] + red[2] + red[3];
    l1 = red[4] + red[5] + red[6] + red[7];
    if (sd.mode == 1) sd.lossp[blockIdx.x * 2 + m] = l0;
    else { sd.lossp[2 * blockIdx.x] = l0; sd.lossp[2 * blockIdx.x + 1] = l1; }
  }
}

// NT (DX only): 16-column tiles of the input-gradient layer handled by the K-split narrow layer (Np1t == 16*NT),
// or 0 = any Np1t through the row-split path.
// MASK: see wide_mask_apply (1: sign words m1, m2; 0: saved activations h1, h2; 2: Swish derivatives in h1, h2).
// PM: 0 = exact fp32 MFMA; 1..4 = the 256 x 256 GEMM (dz2 W2^T) on the split-precision core, streaming W2^T's planes.
template <bool DX, int MT, int NT, int MASK, int PM = 0>
// (three workgroups per CU only for the sign-word variants: the variants that hold 32 mask / derivative values per lane next to
//  the accumulators spilled ~26 VGPRs at the 168-register budget; they serve the small generic launches -- V function, DARA
//  classifier, dynamics pre-training -- where a third resident workgroup buys nothing)
__global__ __launch_bounds__(NTHREADS, (MT == 1 && MASK == 1) ? 3 : 2) void k_mlp3_bwd(Mlp3BwdArgs a) {
  constexpr bool BITS = MASK == 1;
  __shared__ float red[8];
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  constexpr int TB = 32 * MT;
  constexpr int PMX = PM > 0 ? PM : 1;
  const int m = blockIdx.y;
  const long long row0 = (long long)blockIdx.x * TB;
  const int rows_here = (int)min((long long)TB, a.rows - row0);
  const int lane = lane_id(), w = wave_id();
  const float* w3t = a.wt + m * a.t_mstride + a.w3t;
  const float* w2t = a.wt + m * a.t_mstride + a.w2t;
  const float* w1t = a.wt + m * a.t_mstride + a.w1t;
  const float* h1 = BITS ? nullptr : a.h1 + ((long long)m * a.rows + row0) * HID;
  const float* h2 = BITS ? nullptr : a.h2 + ((long long)m * a.rows + row0) * HID;
  const long long mtile = ((long long)m * cdiv(a.rows, 32) + row0 / 32) * HID;
  const uint32_t* m1 = BITS ? a.m1 + mtile : nullptr;
  const uint32_t* m2 = BITS ? a.m2 + mtile : nullptr;
  float* dz2 = a.dz2 ? a.dz2 + ((long long)m * a.rows + row0) * HID : nullptr;
  float* dz1 = a.dz1 ? a.dz1 + ((long long)m * a.rows + row0) * HID : nullptr;
  float* dbp = a.dbp + ((long long)blockIdx.x * gridDim.y + m) * (2 * HID + a.Np3);
  float* scr = reinterpret_cast<float*>(reinterpret_cast<char*>(Xs) + split_scr_offset<PMX, TB>());   // tile maximum (f16 mode)

  TR(0);
  MaskPre<MT, MASK> mk1, mk2;                       // layer 2's operand now; layer 1's too when it is two words, else before its GEMM
  if (BWD_MASK_PREFETCH) {
    mask_fetch<MT, MASK>(mk2, h2, m2, rows_here);
    if constexpr (BITS) mask_fetch<MT, MASK>(mk1, h1, m1, rows_here);
  }
  // (split modes: the ring only serves the K = Np3 GEMM, two to four chunks -- three stages keep the kernel at 128 registers)
  WideRingT<(PM > 0 ? 3 : WIDE_RING)> ring;
  wide_prefetch(w3t, a.Np3, ring);                // weight fragments travel while the seed rows are fetched
  if (a.seed.mode == 0) tile_load(Xs, 0, a.dz3 + ((long long)m * a.rows + row0) * a.Np3, a.Np3, a.Np3, 0, rows_here, TB);
  else bwd_seed(a, Xs, red, m, row0, rows_here, TB);
  lds_barrier();
  TR(1);
  if ((int)threadIdx.x < a.Np3) {                 // db3 partial of this tile
    float s = 0.f;
    for (int r = 0; r < TB; ++r) s += Xs[r * LDX + threadIdx.x];
    dbp[2 * HID + threadIdx.x] = s;
  }

  f32x16 acc[MT][2];
  float cs[2];
  // dh2 = dz3 * W3^T ; dz2 = dh2 * [h2 > 0]
  wide_zero<MT>(acc);
  wide_gemm<MT>(Xs, w3t, a.Np3, acc, ring);
  TR(2);
  BfRing<PMX> bring;
  const s16x8* w2tp = PM > 0 ? reinterpret_cast<const s16x8*>(a.w2t_planes + m * a.planes_ms) : nullptr;
  if constexpr (PM > 0) bf_prefetch<PMX>(w2tp, bring);
  else wide_prefetch(w2t, HID, ring);             // next layer's first fragments overlap the mask epilogue
  int e2 = 0;
  {
    if (!BWD_MASK_PREFETCH) mask_fetch<MT, MASK>(mk2, h2, m2, rows_here);
    const float mx = wide_mask_apply<MT, MASK>(acc, mk2, rows_here, 1.f);
    if constexpr (PM == 4) f16_tile_max_put(mx, scr);
  }
  lds_barrier();
  if constexpr (PM == 4) e2 = f16_scale_exp(f16_tile_max_get(scr));
  PlaneSave gs{nullptr, 0, nullptr};
  if (PM == 4 && a.dz2p != nullptr) {              // the weight-gradient GEMM reads dz2 as the planes formed here
    gs.base = reinterpret_cast<short*>(a.dz2p) + m * a.dz2p_ms + (row0 / 8) * (HID * 8);
    gs.plane_stride = a.dz2p_plane;
    gs.e_out = a.e2_out + (long long)m * cdiv(a.rows, 32) + row0 / 32;
  }
  wide_store_colsum<MT, PM>(acc, Xs, dz2, rows_here, e2, cs, gs);
  if (lane < 32) { dbp[HID + 64 * w + lane] = cs[0]; dbp[HID + 64 * w + 32 + lane] = cs[1]; }
  lds_barrier();
  TR(3);
  // dh1 = dz2 * W2^T ; dz1 = dh1 * [h1 > 0]
  if constexpr (!BITS) { if (BWD_MASK_PREFETCH) mask_fetch<MT, MASK>(mk1, h1, m1, rows_here); }
  wide_zero<MT>(acc);
  if constexpr (PM > 0) bf_gemm<MT, PMX, TB>(reinterpret_cast<const char*>(Xs), w2tp, acc, bring);
  else wide_gemm<MT>(Xs, w2t, HID, acc, ring);
  TR(4);
  NarrowRegs<(NT > 0 ? NT : 1)> br;
  if constexpr (DX && NT > 0) narrow_prefetch<NT>(w1t, 16 * NT, br);
  if (!BWD_MASK_PREFETCH) mask_fetch<MT, MASK>(mk1, h1, m1, rows_here);
  wide_mask_apply<MT, MASK>(acc, mk1, rows_here, PM == 4 ? exp2i(-(e2 + F16_WSHIFT)) : 1.f);
  lds_barrier();
  wide_store_colsum<MT, 0>(acc, Xs, dz1, rows_here, 0, cs);
  if (lane < 32) { dbp[64 * w + lane] = cs[0]; dbp[64 * w + 32 + lane] = cs[1]; }
  TR(5);
  if constexpr (DX) {
    lds_barrier();
    float* dx = a.dx + ((long long)m * a.rows + row0) * a.dx_n;
    auto emit = [&](int row, int col, float v) {
      const int c = col - a.dx_c0;
      if (row < rows_here && c >= 0 && c < a.dx_n) dx[row * a.dx_n + c] = v;
    };
    if constexpr (NT > 0) narrow_run<TB / 16, NT>(Xs, br, emit);
    else narrow_layer(Xs, w1t, HID, a.Np1t, emit, TB);
  }
  TR(6);
}

template <bool DX, int MT, int NT, int BITS, int NPL = 0>
static int launch_bwd_t(const Mlp3BwdArgs& a, int members, hipStream_t st) {
  constexpr size_t lds = split_lds_bytes<(NPL > 0 ? NPL : 1), 32 * MT>();
  static bool once = false;
  if (!once) {
    int rc = allow_big_lds(k_mlp3_bwd<DX, MT, NT, BITS, NPL>, lds);
    if (rc) return rc;
    once = true;
  }
  dim3 grid((unsigned)cdiv(a.rows, 32 * MT), (unsigned)members);
  ProfScope prof(PROF_MLP_BWD, st);
  hipLaunchKernelGGL((k_mlp3_bwd<DX, MT, NT, BITS, NPL>), grid, dim3(NTHREADS), lds, st, a);
  MB_LAUNCH_OK("k_mlp3_bwd");
  return 0;
}

// split-precision backward: sign-word masks, 32-row tiles (the train step's three backward launches)
template <int NPL>
static int launch_bwd_bf(const Mlp3BwdArgs& a, int members, bool with_dx, hipStream_t st) {
  const int nt = a.Np1t == 16 ? 1 : a.Np1t == 32 ? 2 : 0;
  if (!with_dx) return launch_bwd_t<false, 1, 0, 1, NPL>(a, members, st);
  return nt == 1 ? launch_bwd_t<true, 1, 1, 1, NPL>(a, members, st) : nt == 2 ? launch_bwd_t<true, 1, 2, 1, NPL>(a, members, st)
                                                                             : launch_bwd_t<true, 1, 0, 1, NPL>(a, members, st);
}

// tile_rows (32 or 64) must be the value the caller sized `dbp` / the bias reduction with
template <int BITS>
static int launch_bwd_masks(const Mlp3BwdArgs& a, int members, bool with_dx, int tile_rows, hipStream_t st) {
  const int nt = a.Np1t == 16 ? 1 : a.Np1t == 32 ? 2 : 0;
  if (tile_rows == 32) {
    if (!with_dx) return launch_bwd_t<false, 1, 0, BITS>(a, members, st);
    return nt == 1 ? launch_bwd_t<true, 1, 1, BITS>(a, members, st) : nt == 2 ? launch_bwd_t<true, 1, 2, BITS>(a, members, st)
                                                                               : launch_bwd_t<true, 1, 0, BITS>(a, members, st);
  }
  if (!with_dx) return launch_bwd_t<false, 2, 0, BITS>(a, members, st);
  return nt == 1 ? launch_bwd_t<true, 2, 1, BITS>(a, members, st) : nt == 2 ? launch_bwd_t<true, 2, 2, BITS>(a, members, st)
                                                                             : launch_bwd_t<true, 2, 0, BITS>(a, members, st);
}

// Swish nets (the ensemble dynamics, pre-training): 32-row tiles only, derivative multipliers in h1 / h2
static int launch_bwd_swish(const Mlp3BwdArgs& a, int members, bool with_dx, hipStream_t st) {
  const int nt = a.Np1t == 16 ? 1 : a.Np1t == 32 ? 2 : 0;
  if (a.prec == 4 && a.w2t_planes != nullptr) {     // f16x2: the 256 x 256 GEMM on the split core, dz2 as planes for the weight gradients
    if (!with_dx) return launch_bwd_t<false, 1, 0, 2, 4>(a, members, st);
    return nt == 1 ? launch_bwd_t<true, 1, 1, 2, 4>(a, members, st) : nt == 2 ? launch_bwd_t<true, 1, 2, 2, 4>(a, members, st)
                                                                             : launch_bwd_t<true, 1, 0, 2, 4>(a, members, st);
  }
  if (!with_dx) return launch_bwd_t<false, 1, 0, 2>(a, members, st);
  return nt == 1 ? launch_bwd_t<true, 1, 1, 2>(a, members, st) : nt == 2 ? launch_bwd_t<true, 1, 2, 2>(a, members, st)
                                                                         : launch_bwd_t<true, 1, 0, 2>(a, members, st);
}

int launch_mlp3_bwd(const Mlp3BwdArgs& a, int members, bool with_dx, int tile_rows, hipStream_t st) {
  if (a.rows <= 0) return 0;
  if (a.swish) {
    if (tile_rows != 32) return fail(MOBODY_E_ARG, "launch_mlp3_bwd: the Swish backward uses 32-row tiles");
    return launch_bwd_swish(a, members, with_dx, st);
  }
  if (a.prec != 0 && a.w2t_planes != nullptr && a.m1 != nullptr && a.m2 != nullptr && tile_rows == 32)
    return a.prec == 1 ? launch_bwd_bf<1>(a, members, with_dx, st) : a.prec == 2 ? launch_bwd_bf<2>(a, members, with_dx, st)
         : a.prec == 3 ? launch_bwd_bf<3>(a, members, with_dx, st) : launch_bwd_bf<4>(a, members, with_dx, st);
  return a.m1 != nullptr && a.m2 != nullptr ? launch_bwd_masks<1>(a, members, with_dx, tile_rows, st)
                                            : launch_bwd_masks<0>(a, members, with_dx, tile_rows, st);
}

// ------------------------------------------------------------------------------------------------
// weight gradient GEMM (all three layers of one packed MLP in ONE launch)
//
//   job 0: dW2  = h1^T dz2      256 x 256      wave tile 64 x 64  (MT = 2)
//   job 1: dW1  = x^T  dz1      Kp1 x 256      wave tile 32 x 64  (MT = 1)
//   job 2: dW3^T = dz3^T h2     Np3 x 256      wave tile 32 x 64  (MT = 1), stored transposed into W3's [256][Np3]
//
// Rows are the contraction index: lane (i = lane&31, h = lane>>5) reads A[row+h][k0+i] and B[row+h][n0+i], i.e.
// every wave instruction is two full 128-byte lines, no LDS staging.  Split-K: the 4 waves of a workgroup take
// 4 consecutive row slices of the same output tile and reduce through LDS; workgroups along the row dimension
// write separate slabs (deterministic, summed by k_grad_reduce).  Operands of the next 8 rows are prefetched
// into a second register set while the current 8 rows feed the MFMAs (see wgrad_tile for what makes that overlap real).
// Block -> work mapping is XCD aware (blocks b and b+8 share an XCD and its L2): all output tiles of one
// (row slice, member) run on the same XCD back to back, so the slice's activations are fetched from
// HBM/Infinity Cache once and re-read 4x from that XCD's L2.
// ------------------------------------------------------------------------------------------------
// Buffer descriptor over `nrows` rows of a row-major fp32 matrix (pitch ld floats) starting at the wave-uniform pointer p, and a
// dword load through it: the per-lane byte offset rides in voffset, the wave-uniform row offset in soffset (a scalar register).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slice_rsrc(const float* p, int nrows, int ld) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)((unsigned)nrows * (unsigned)ld * 4u), 0x00020000);
}
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

// Reduce the four row slices of a workgroup through LDS and write one slab tile.  Two tile buffers (32 KB), not four:
// waves 0,1 store, waves 2,3 add onto them (same lane <-> element map, so no conflicts).  With four buffers (64 KB)
// only two workgroups fit a CU and a 768-workgroup launch needs two rounds.
template <int MT, int NT>
__device__ __forceinline__ void wgrad_store(const WgradJob& jb, const WgradArgs& a, f32x16 (&acc)[MT][NT], int k0, int n0,
                                            int slice, int m, float* red) {
  constexpr int TK = 32 * MT, TN = 32 * NT;
  const int lane = lane_id(), w = wave_id();
  const int i = lane & 31, h = lane >> 5;
  float* mine = red + (w & 1) * (TK * TN);
  auto sweep = [&](bool add) {
#pragma unroll
    for (int x = 0; x < MT; ++x)
#pragma unroll
      for (int y = 0; y < NT; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kk = 32 * x + (r & 3) + 8 * (r >> 2) + 4 * h;
          float* p = mine + kk * TN + 32 * y + i;
          *p = add ? *p + acc[x][y][r] : acc[x][y][r];
        }
  };
  if (w < 2) sweep(false);
  lds_barrier();
  if (w >= 2) sweep(true);
  lds_barrier();
  float* slab = a.slabs + (long long)slice * a.slab_stride + jb.out_off + m * a.out_mstride;
  for (int idx = threadIdx.x; idx < TK * TN; idx += NTHREADS) {
    const int kk = idx / TN, nn = idx - kk * TN;
    const float s = red[idx] + red[TK * TN + idx];
    const int gk = k0 + kk, gn = n0 + nn;
    if (gk < jb.out_k && gn < jb.out_n) {
      if (jb.transposed) slab[(long long)gn * jb.out_ld + gk] = s;
      else if (jb.wide) slab[wide_idx(gk, gn)] = s;
      else slab[(long long)gk * jb.out_ld + gn] = s;
    }
  }
}

// The row loop runs on wave-uniform row-block pointers (scalar registers) plus one per-lane 32-bit offset per operand
// column block, so a load costs no vector arithmetic, and whole row blocks carry no masks: the loaded registers feed the
// MFMAs directly and the loads of block n+1 stay in flight under the MFMAs of block n.  (Multiplying every loaded
// value by a 0/1 mask, as the first version did, made the compiler wait for each block's loads BEFORE the previous
// block's MFMAs: nothing overlapped inside a wave.)  Columns past ka / nb read column 0 instead: their products land in
// output elements wgrad_store never writes.  Only the < RB rows left at the end of a wave's slice take masked loads.
template <int MT>
__device__ __forceinline__ void wgrad_tile(const WgradJob& jb, const WgradArgs& a, int tile, int slice, int m, float* red) {
  constexpr int NT = 2, TK = 32 * MT, TN = 32 * NT, U = 4, RB = 2 * U;
  const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(wave_id());
  const int i = lane & 31, h = lane >> 5;
  const int tk = tile / jb.tiles_n, tn = tile - tk * jb.tiles_n;
  const int k0 = tk * TK, n0 = tn * TN;
  const long long r_begin = ((long long)slice * 4 + w) * a.rows_per_wave;
  const long long r_end = min(a.rows, r_begin + a.rows_per_wave);
  const int lda = jb.lda, ldb = jb.ldb;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

  unsigned oa[MT], ob[NT];                            // byte offsets; lane half h takes the odd row of each pair
#pragma unroll
  for (int x = 0; x < MT; ++x) { const int c = k0 + 32 * x + i; oa[x] = 4u * (unsigned)((c < jb.ka ? c : 0) + h * lda); }
#pragma unroll
  for (int y = 0; y < NT; ++y) { const int c = n0 + 32 * y + i; ob[y] = 4u * (unsigned)((c < jb.nb ? c : 0) + h * ldb); }

  auto mma = [&](float (&av)[U][MT], float (&bv)[U][NT]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int x = 0; x < MT; ++x)
#pragma unroll
        for (int y = 0; y < NT; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][x], bv[u][y], acc[x][y], 0, 0, 0);
  };
  if (r_begin < r_end) {
    const int nrows = (int)(r_end - r_begin), nblk = nrows / RB, tail = nrows - nblk * RB;
    const auto ra = slice_rsrc(jb.A + m * jb.a_mstride + r_begin * lda, nrows, lda);
    const auto rb = slice_rsrc(jb.B + m * jb.b_mstride + r_begin * ldb, nrows, ldb);
    const unsigned sa = 4u * lda, sb = 4u * ldb;      // row pitch in bytes
    auto load = [&](unsigned row, float (&av)[U][MT], float (&bv)[U][NT]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int x = 0; x < MT; ++x) av[u][x] = buf_ld(ra, oa[x], (row + 2 * u) * sa);
#pragma unroll
        for (int y = 0; y < NT; ++y) bv[u][y] = buf_ld(rb, ob[y], (row + 2 * u) * sb);
      }
    };
    float a0[U][MT], b0[U][NT], a1[U][MT], b1[U][NT];
    // Straight-line body (no branch between a block's loads and the previous block's MFMAs, or the compiler's vmcnt
    // bookkeeping merges the two paths and waits for the NEW loads): the last pass re-loads block nblk - 1, unused if nblk is even.
    if (nblk > 0) load(0, a0, b0);
    for (int blk = 0; blk + 1 < nblk; blk += 2) {
      load((blk + 1) * RB, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load(min(blk + 2, nblk - 1) * RB, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (nblk & 1) mma(a0, b0);
    if (tail > 0) {                                   // rows [nblk * RB, nrows) of the slice: masked lanes re-read its first row
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool ok = 2 * u + h < tail;
#pragma unroll
        for (int x = 0; x < MT; ++x) { const float v = buf_ld(ra, ok ? oa[x] + 2 * u * sa : oa[x] - h * sa, nblk * RB * sa); a0[u][x] = ok ? v : 0.f; }
#pragma unroll
        for (int y = 0; y < NT; ++y) { const float v = buf_ld(rb, ok ? ob[y] + 2 * u * sb : ob[y] - h * sb, nblk * RB * sb); b0[u][y] = ok ? v : 0.f; }
      }
      mma(a0, b0);
    }
  }
  wgrad_store<MT, NT>(jb, a, acc, k0, n0, slice, m, red);
}

// Split-precision form of the 256 x 256 job (dW2 = h1^T dz2, 86 % of the weight-gradient FLOPs): the contraction runs over
// batch ROWS, so a lane's A / B fragment of v_mfma_f32_32x32x16_bf16 is eight consecutive rows of one column -- eight
// coalesced scalar loads (a wave instruction = 2 rows x 128 bytes), split into NPL bf16 terms in registers, then the
// (i + j < NPL) products.  Same work split, addressing, LDS reduction and slab output as wgrad_tile<2>; row blocks of 16.
template <int NPL>
__device__ __forceinline__ void wgrad_tile_bf(const WgradJob& jb, const WgradArgs& a, int tile, int slice, int m, float* red) {
  constexpr int MT = 2, NT = 2, TK = 64, TN = 64, RB = 16;
  const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(wave_id());
  const int i = lane & 31, h = lane >> 5;
  const int tk = tile / jb.tiles_n, tn = tile - tk * jb.tiles_n;
  const int k0 = tk * TK, n0 = tn * TN;
  const long long r_begin = ((long long)slice * 4 + w) * a.rows_per_wave;
  const long long r_end = min(a.rows, r_begin + a.rows_per_wave);
  const int lda = jb.lda, ldb = jb.ldb;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  unsigned oa[MT], ob[NT];                                       // byte offsets; job 0: every column is real (ka = nb = 256)
#pragma unroll
  for (int x = 0; x < MT; ++x) oa[x] = 4u * (unsigned)(k0 + 32 * x + i + 8 * h * lda);
#pragma unroll
  for (int y = 0; y < NT; ++y) ob[y] = 4u * (unsigned)(n0 + 32 * y + i + 8 * h * ldb);
  auto pack = [&](const float (&v)[8], bf16x8 (&f)[NPL]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 t[NPL];
      bf_split<NPL>(v[j], t);
#pragma unroll
      for (int p = 0; p < NPL; ++p) f[p][j] = t[p];
    }
  };
  auto mma = [&](const float (&av)[MT][8], const float (&bv)[NT][8]) {
    bf16x8 af[MT][NPL], bfr[NT][NPL];
#pragma unroll
    for (int x = 0; x < MT; ++x) pack(av[x], af[x]);
#pragma unroll
    for (int y = 0; y < NT; ++y) pack(bv[y], bfr[y]);
#pragma unroll
    for (int x = 0; x < MT; ++x)
#pragma unroll
      for (int y = 0; y < NT; ++y)
#pragma unroll
        for (int d = NPL - 1; d >= 0; --d)
#pragma unroll
          for (int q = 0; q <= d; ++q)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[x][q], bfr[y][d - q], acc[x][y], 0, 0, 0);
  };
  if (r_begin < r_end) {
    const int nrows = (int)(r_end - r_begin), nblk = nrows / RB, tail = nrows - nblk * RB;
    const auto ra = slice_rsrc(jb.A + m * jb.a_mstride + r_begin * lda, nrows, lda);
    const auto rb = slice_rsrc(jb.B + m * jb.b_mstride + r_begin * ldb, nrows, ldb);
    const unsigned sa = 4u * lda, sb = 4u * ldb;
    auto load = [&](unsigned row, float (&av)[MT][8], float (&bv)[NT][8]) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int x = 0; x < MT; ++x) av[x][j] = buf_ld(ra, oa[x], (row + j) * sa);
#pragma unroll
        for (int y = 0; y < NT; ++y) bv[y][j] = buf_ld(rb, ob[y], (row + j) * sb);
      }
    };
    float a0[MT][8], b0[NT][8], a1[MT][8], b1[NT][8];
    // Straight-line body (no branch between a block's loads and the previous block's MFMAs, or the compiler's vmcnt
    // bookkeeping merges the two paths and waits for the NEW loads): the last pass re-loads block nblk - 1, unused if nblk is even.
    if (nblk > 0) load(0, a0, b0);
    for (int blk = 0; blk + 1 < nblk; blk += 2) {
      load((blk + 1) * RB, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load(min(blk + 2, nblk - 1) * RB, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (nblk & 1) mma(a0, b0);
    if (tail > 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = 8 * h + j < tail;
#pragma unroll
        for (int x = 0; x < MT; ++x) { const float v = buf_ld(ra, ok ? oa[x] + j * sa : oa[x] - 8 * h * sa, nblk * RB * sa); a0[x][j] = ok ? v : 0.f; }
#pragma unroll
        for (int y = 0; y < NT; ++y) { const float v = buf_ld(rb, ok ? ob[y] + j * sb : ob[y] - 8 * h * sb, nblk * RB * sb); b0[y][j] = ok ? v : 0.f; }
      }
      mma(a0, b0);
    }
  }
  wgrad_store<MT, NT>(jb, a, acc, k0, n0, slice, m, red);
}

// "f16x2" form of the 256 x 256 job: both operands arrive PRE-SPLIT -- the two fp16 planes the forward (h1) and backward
// (dz2) epilogues stored in fragment order, [row / 8][256][8] per plane -- so a lane's A / B fragment is ONE 16-byte load and
// the row loop has no conversion work at all: per 16-row block 8 loads and 12 MFMAs (three products on four 32 x 32 tiles).
// Every 32-row tile carries its own power-of-two scales (planes hold h1 * 2^eA[t], dz2 * 2^eB[t]); a wave brings its
// blocks to one common scale U = min_t (eA[t] + eB[t]) over its rows by multiplying the B fragments with the exact factor
// 2^(U - eA[t] - eB[t]) <= 1 (v_pk_mul_f16; a tile far below the wave's largest one underflows gracefully, its contribution
// to the sum is below fp32 resolution anyway) and un-scales its accumulators once at the end.
__device__ __forceinline__ void wgrad_tile_f16(const WgradJob& jb, const WgradArgs& a, int tile, int slice, int m, float* red) {
  constexpr int MT = 2, NT = 2, TK = 64, TN = 64, RB = 16;
  const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(wave_id());
  const int i = lane & 31, h = lane >> 5;
  const int tk = tile / jb.tiles_n, tn = tile - tk * jb.tiles_n;
  const int k0 = tk * TK, n0 = tn * TN;
  const long long r_begin = ((long long)slice * 4 + w) * a.rows_per_wave;      // multiple of 16
  const long long r_end = min(a.rows, r_begin + a.rows_per_wave);
  f32x16 acc[MT][NT];
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  int U = 0;
  if (r_begin < r_end) {
    const int nblk = (int)((r_end - r_begin + RB - 1) / RB);                   // the planes are zero beyond the batch (rows32)
    const int* eA = a.eA + m * a.e_mstride;
    const int* eB = a.eB + m * a.e_mstride;
    // Lane l keeps the summed exponent of the slice's tile t0 + l (a slice has at most 64 tiles: launch_wgrad), so the row
    // loop takes a block's scale with v_readlane instead of a memory round trip in front of its MFMAs.
    const int t0 = (int)(r_begin >> 5), t1 = (int)((r_end - 1) >> 5);
    const int Ev = (t0 + lane <= t1) ? eA[t0 + lane] + eB[t0 + lane] : 0x7fffffff;
    U = Ev;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) U = min(U, __shfl_xor(U, o));
    U = __builtin_amdgcn_readfirstlane(U);
    // 16-byte units: plane p of a member starts at p * plane_stride / 8; fragment of block b, column c: ((r / 8 + h) * 256 + c)
    const s16x8* pa = reinterpret_cast<const s16x8*>(jb.A) + (m * jb.a_mstride) / 8 + ((r_begin >> 3) + h) * HID + k0 + i;
    const s16x8* pb = reinterpret_cast<const s16x8*>(jb.B) + (m * jb.b_mstride) / 8 + ((r_begin >> 3) + h) * HID + n0 + i;
    const long long ps = a.plane_stride / 8;
    struct Blk { s16x8 a[2][MT], b[2][NT]; int sh; };
    auto load = [&](int b, Blk& k) {
      const long long o = (long long)b * 2 * HID;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int x = 0; x < MT; ++x) k.a[p][x] = pa[p * ps + o + 32 * x];
#pragma unroll
        for (int y = 0; y < NT; ++y) k.b[p][y] = pb[p * ps + o + 32 * y];
      }
      const int t = (int)((r_begin + (long long)b * RB) >> 5);
      k.sh = U - __builtin_amdgcn_readlane(Ev, t - t0);                         // <= 0
    };
    auto mma = [&](Blk& k) {
      // 2^sh as a packed fp16 pair (sh >= -24 stays exact down to the smallest subnormal; below that the tile is noise)
      const int shc = max(k.sh, -24);
      const _Float16 f = (_Float16)__int_as_float((shc + 127) << 23);
      s16x8 bs[2][NT];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int y = 0; y < NT; ++y) bs[p][y] = __builtin_bit_cast(s16x8, __builtin_bit_cast(f16x8, k.b[p][y]) * f);
#pragma unroll
      for (int x = 0; x < MT; ++x)
#pragma unroll
        for (int y = 0; y < NT; ++y) {
          acc[x][y] = split_mfma<4>(k.a[0][x], bs[1][y], acc[x][y]);          // smallest terms first
          acc[x][y] = split_mfma<4>(k.a[1][x], bs[0][y], acc[x][y]);
          acc[x][y] = split_mfma<4>(k.a[0][x], bs[0][y], acc[x][y]);
        }
    };
    Blk b0, b1;
    // straight-line body as in wgrad_tile: the last pass re-loads block nblk - 1, unused if nblk is even
    load(0, b0);
    for (int blk = 0; blk + 1 < nblk; blk += 2) {
      load(blk + 1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mma(b0);
      __builtin_amdgcn_sched_barrier(0);
      load(min(blk + 2, nblk - 1), b0);
      __builtin_amdgcn_sched_barrier(0);
      mma(b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (nblk & 1) mma(b0);
  }
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = ldexpf(acc[x][y][r], -U);
  wgrad_store<MT, NT>(jb, a, acc, k0, n0, slice, m, red);
}

__global__ __launch_bounds__(NTHREADS, 2) void k_wgrad_f16(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];     // [2][64][64]
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  const int sm = xcd + 8 * (j / a.tiles_total);
  const int t = j % a.tiles_total;
  if (sm >= a.nsplit * a.members) return;
  const int slice = sm / a.members, m = sm - slice * a.members;
  if (t < a.job[0].ntiles) wgrad_tile_f16(a.job[0], a, t, slice, m, red);
  else if (t < a.job[0].ntiles + a.job[1].ntiles) wgrad_tile<1>(a.job[1], a, t - a.job[0].ntiles, slice, m, red);
  else wgrad_tile<1>(a.job[2], a, t - a.job[0].ntiles - a.job[1].ntiles, slice, m, red);
}

template <int NPL>
__global__ __launch_bounds__(NTHREADS, 2) void k_wgrad_bf(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];     // [2][64][64]
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  const int sm = xcd + 8 * (j / a.tiles_total);
  const int t = j % a.tiles_total;
  if (sm >= a.nsplit * a.members) return;
  const int slice = sm / a.members, m = sm - slice * a.members;
  if (t < a.job[0].ntiles) wgrad_tile_bf<NPL>(a.job[0], a, t, slice, m, red);
  else if (t < a.job[0].ntiles + a.job[1].ntiles) wgrad_tile<1>(a.job[1], a, t - a.job[0].ntiles, slice, m, red);
  else wgrad_tile<1>(a.job[2], a, t - a.job[0].ntiles - a.job[1].ntiles, slice, m, red);
}

__global__ __launch_bounds__(NTHREADS, 4) void k_wgrad(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];     // [2][64][64]
  // XCD-aware decode: blocks with equal (id % 8) share an XCD; consecutive ones walk the tiles of one (slice, member)
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  const int sm = xcd + 8 * (j / a.tiles_total);
  const int t = j % a.tiles_total;
  if (sm >= a.nsplit * a.members) return;
  const int slice = sm / a.members, m = sm - slice * a.members;
  if (t < a.job[0].ntiles) wgrad_tile<2>(a.job[0], a, t, slice, m, red);
  else if (t < a.job[0].ntiles + a.job[1].ntiles) wgrad_tile<1>(a.job[1], a, t - a.job[0].ntiles, slice, m, red);
  else wgrad_tile<1>(a.job[2], a, t - a.job[0].ntiles - a.job[1].ntiles, slice, m, red);
}

int launch_wgrad(WgradArgs a, hipStream_t st) {
  if (a.rows <= 0) return 0;
  constexpr size_t lds = (size_t)2 * 64 * 64 * sizeof(float);
  static bool once = false;
  if (!once) {
    int rc = allow_big_lds(k_wgrad, lds);
    if (rc) return rc;
    once = true;
  }
  long long rpw = cdiv(a.rows, (long long)4 * a.nsplit);
  a.rows_per_wave = (rpw + 15) & ~15LL;               // whole 8- / 16-row blocks for every wave but the last one with work
  for (int k = (a.prec == 4 && a.eA != nullptr) ? 1 : 0; k < 3; ++k)     // a wave addresses its row slice through 32-bit buffer offsets
    if (a.rows_per_wave * (long long)std::max(a.job[k].lda, a.job[k].ldb) * 4 >= (1LL << 31))
      return fail(MOBODY_E_ARG, "launch_wgrad: row slice too large for 32-bit offsets (raise nsplit)");
  a.job[0].tiles_n = (a.job[0].nb + 63) / 64; a.job[0].ntiles = ((a.job[0].ka + 63) / 64) * a.job[0].tiles_n;
  for (int k = 1; k < 3; ++k) { a.job[k].tiles_n = (a.job[k].nb + 63) / 64; a.job[k].ntiles = ((a.job[k].ka + 31) / 32) * a.job[k].tiles_n; }
  a.tiles_total = a.job[0].ntiles + a.job[1].ntiles + a.job[2].ntiles;
  const int sm = a.nsplit * a.members;
  const int blocks = 8 * ((sm + 7) / 8) * a.tiles_total;
  ProfScope prof(PROF_WGRAD, st);
  // Split-precision job 0 in every bf16 mode (the operand split costs ~6 VALU instructions per value and term; with the
  // unmasked scalar-addressed row loop that still leaves a gain: per step at c2 0.058 ms fp32 job -> 0.052 bf16x3,
  // 0.044 bf16x2); MOBODY_WGRAD_BF=0 keeps the job in fp32 (tuning aid).
  if (a.prec == 4 && a.eA != nullptr) {               // "f16x2": job 0 on the pre-split fp16 planes
    if (a.rows_per_wave > 2048) return fail(MOBODY_E_ARG, "launch_wgrad: more than 64 row tiles per wave slice (raise nsplit)");
    static bool once_h = false;
    if (!once_h) {
      int rc = allow_big_lds(k_wgrad_f16, lds);
      if (rc) return rc;
      once_h = true;
    }
    hipLaunchKernelGGL(k_wgrad_f16, dim3(blocks), dim3(NTHREADS), lds, st, a);
    MB_LAUNCH_OK("k_wgrad_f16");
    return 0;
  }
  static const int bf_force = tune_int("MOBODY_WGRAD_BF", -1);
  const bool use_bf = bf_force == 0 ? false : a.prec != 0;
  if (use_bf && a.job[0].ka == HID && a.job[0].nb == HID && a.job[0].wide) {
    static bool once_bf = false;
    if (!once_bf) {
      int rc = allow_big_lds(k_wgrad_bf<1>, lds);
      if (!rc) rc = allow_big_lds(k_wgrad_bf<2>, lds);
      if (!rc) rc = allow_big_lds(k_wgrad_bf<3>, lds);
      if (rc) return rc;
      once_bf = true;
    }
    if (a.prec == 1) hipLaunchKernelGGL(k_wgrad_bf<1>, dim3(blocks), dim3(NTHREADS), lds, st, a);
    else if (a.prec == 2) hipLaunchKernelGGL(k_wgrad_bf<2>, dim3(blocks), dim3(NTHREADS), lds, st, a);
    else hipLaunchKernelGGL(k_wgrad_bf<3>, dim3(blocks), dim3(NTHREADS), lds, st, a);
    MB_LAUNCH_OK("k_wgrad_bf");
    return 0;
  }
  hipLaunchKernelGGL(k_wgrad, dim3(blocks), dim3(NTHREADS), lds, st, a);
  MB_LAUNCH_OK("k_wgrad");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// slabs + bias partials -> gradient blob
// ------------------------------------------------------------------------------------------------
// Weight entries: one thread per entry sums the split-K slabs (coalesced across threads).  Bias entries: one
// WAVE per entry strides over the row-tile partials and shuffle-reduces (a serial loop over ~160 dependent
// L2 reads per thread made the first version of this kernel latency bound: 41 us instead of ~5).
__global__ __launch_bounds__(256) void k_grad_reduce(GradReduceArgs a) {
  __shared__ float adam_sm[2];
  adam_block_consts(a.adam, adam_sm);
  const long long nW = a.L.total_floats;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid == 0 && a.adam.bump != nullptr) a.adam.bump[0] += 1;      // graph replay: advance a counter no block of this launch reads
  if (gid < nW) {
    const long long j = gid;
    const long long o = j % a.L.member_floats;
    const bool is_bias = (o >= a.L.b1 && o < a.L.b1 + HID) || (o >= a.L.b2 && o < a.L.b2 + HID) || (o >= a.L.b3);
    if (is_bias) return;
    // (requesting all 16 slab entries and the Adam state in one round trip instead of groups of four + one measured neutral at
    //  c2 and -1 % on the pre-training step: not kept)
    float s = 0.f;
    int k = 0;
    for (; k + 4 <= a.nsplit; k += 4) {            // 4 independent loads in flight, summed in slab order
      const float* p = a.slabs + (long long)k * a.slab_stride + j;
      const float v0 = p[0], v1 = p[a.slab_stride], v2 = p[2 * a.slab_stride], v3 = p[3 * a.slab_stride];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; k < a.nsplit; ++k) s += a.slabs[(long long)k * a.slab_stride + j];
    if (a.grad != nullptr) a.grad[j] = s;
    if (a.adam.on) adam_element(a.adam, a.L, j, s, adam_sm);
    return;
  }
  // ---- bias part: wave index -> (member, bias element) ----
  const int per = 2 * HID + a.L.Np3;
  const long long wv = (gid - ((nW + 255) / 256) * 256) >> 6;
  if (blockIdx.x == gridDim.x - 1 && a.loss.kind != 0) {      // the extra workgroup: loss partials -> scalars
    __shared__ float sm[8];
    const LossFinal& f = a.loss;
    const int stride = f.kind == 2 ? 2 : 1;
    float s0 = 0.f, s1 = 0.f;
    for (int k = threadIdx.x; k < f.nparts; k += 256) { s0 += f.parts[stride * k]; if (f.kind == 2) s1 += f.parts[2 * k + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = s0; sm[4 + (threadIdx.x >> 6)] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      s0 = sm[0] + sm[1] + sm[2] + sm[3];
      s1 = sm[4] + sm[5] + sm[6] + sm[7];
      if (f.kind == 1) {
        f.out[0] = s0 * f.scale;
      } else {
        const float pw = f.scale_q ? f.weight / (f.stats[0] / f.ng) : 1.f;
        const float bc = s1 / f.ntg_a;
        f.out[0] = pw * s0 / f.ng + f.bc_coef * bc;
        f.out[1] = bc;
      }
    }
    return;
  }
  if (wv < 0 || wv >= (long long)a.L.members * per) return;
  const int lane = threadIdx.x & 63;
  const int m = (int)(wv / per), off = (int)(wv % per);
  float s = 0.f;
  for (int t = lane; t < a.ntiles; t += 64) s += a.dbp[((long long)t * a.L.members + m) * per + off];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    const long long dst = (long long)m * a.L.member_floats + (off < HID ? a.L.b1 + off : off < 2 * HID ? a.L.b2 + (off - HID) : a.L.b3 + (off - 2 * HID));
    if (a.grad != nullptr) a.grad[dst] = s;
    if (a.adam.on) adam_element(a.adam, a.L, dst, s, adam_sm);
  }
}

// dW1, dW2, dW3 (one merged split-K launch) + the deterministic slab / bias-partial reduction (optionally with the
// optimizer step fused).  x_mstride: 0 = the input rows are shared by the members, rows*Kp1 = per-member inputs.
int mlp3_weight_grads(const MobodyMlpLayout& L, const float* x, long long x_mstride, const float* h1, const float* h2,
                      const float* dz3, const float* dz2, const float* dz1, long long rows, int nsplit, float* slabs,
                      const float* dbp, int ntiles, float* grad, const LossFinal& loss, const AdamTarget& adam,
                      hipStream_t st, int prec, const int* e_h1, const int* e_dz2) {
  WgradArgs g{};
  g.prec = prec;
  const long long rows32 = (rows + 31) & ~31LL;
  g.eA = e_h1; g.eB = e_dz2; g.e_mstride = rows32 / 32; g.plane_stride = rows32 * HID;
  const long long slab_stride = (L.total_floats + 3) & ~3LL;
  g.rows = rows; g.slabs = slabs; g.slab_stride = slab_stride; g.out_mstride = L.member_floats;
  g.nsplit = nsplit; g.members = L.members;
  const long long hs = rows * HID;
  // dW2 = h1^T dz2
  g.job[0] = WgradJob{h1, hs, HID, HID, dz2, hs, HID, HID, L.w2, HID, HID, HID, 0, 1, 0, 0};
  if (prec == 4 && e_h1 != nullptr) g.job[0].a_mstride = g.job[0].b_mstride = 2 * rows32 * HID;   // planes: 16-bit elements per member
  // dW1 = x^T dz1
  g.job[1] = WgradJob{x, x_mstride, L.Kp1, L.Kp1, dz1, hs, HID, HID, L.w1, HID, L.Kp1, HID, 0, 1, 0, 0};
  // dW3^T = dz3^T h2, stored transposed into W3[256][Np3]
  g.job[2] = WgradJob{dz3, rows * L.Np3, L.Np3, L.Np3, h2, hs, HID, HID, L.w3, L.Np3, L.Np3, HID, 1, 0, 0, 0};
  int rc = launch_wgrad(g, st);
  if (rc) return rc;
  GradReduceArgs r{L, slabs, slab_stride, nsplit, dbp, ntiles, grad, loss, adam};
  return launch_grad_reduce(r, st);
}

int launch_grad_reduce(const GradReduceArgs& a, hipStream_t st) {
  const long long wblocks = cdiv(a.L.total_floats, 256);
  const long long bblocks = cdiv((long long)a.L.members * (2 * HID + a.L.Np3) * 64, 256);
  const long long lblocks = a.loss.kind != 0 ? 1 : 0;
  hipLaunchKernelGGL(k_grad_reduce, dim3((unsigned)(wblocks + bblocks + lblocks)), dim3(256), 0, st, a);
  MB_LAUNCH_OK("k_grad_reduce");
  return 0;
}

}  // namespace mobody
