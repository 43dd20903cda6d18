// CDNA4 (gfx950) tile primitives shared by every MOBODY kernel.
//
// Geometry (fixed for the whole library):
//   * workgroup = 256 threads = 4 wave64, one row-tile of BM = 64 minibatch rows;
//   * activations of the current layer live in ONE LDS image X[64][LDX] fp32
//     (LDX = 260: 260 mod 64 == 4, so the ds_read_b128 A-fragment reads of 16
//     different rows fall on 16 different 16-byte bank slots -> conflict free);
//   * "wide" layers (N = 256 = hidden width H) use v_mfma_f32_32x32x2_f32:
//     wave w owns output columns [64w, 64w+64) of all 64 rows = 2x2 tiles of
//     32x32, 64 accumulator VGPRs; A fragments come from LDS (one b128 = four
//     k-steps), B fragments straight from global/L2 (weights are [K][N] row
//     major, so a wave-instruction reads two full 128-byte lines);
//   * "narrow" layers (N <= 128: latent heads, action/Q/reward/next-state
//     outputs, input-gradients) use v_mfma_f32_16x16x4_f32: wave w owns rows
//     [16w, 16w+16), so chains of narrow layers are wave-local (no barrier).
//   The contraction index is split between lane groups instead of interleaved:
//   lane half h (wide) / quarter q (narrow) walks its own contiguous K range, so a
//   lane's successive k-steps are contiguous in LDS.  Any k order is a valid fp32
//   sum; MFMA f32 is an exact fma chain (MI355X guide: FP32-input MFMA).
//
// Reference semantics implemented on top of these tiles are cited at each kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace mobody {

constexpr int BM = 64;        // rows per workgroup tile
constexpr int HID = 256;      // hidden width (reference: hidden_dims=256 train_mobody.py:794, hidden_sizes: 256 yaml)
constexpr int LDX = 260;      // LDS leading dimension (floats)
constexpr int NTHREADS = 256;
constexpr int LATENT = 16;    // mobody_module.py:95
constexpr int NENS = 7;       // mobody_module.py:247 hard-codes 7

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
// Workgroups have 4*RG waves: wave w owns output columns [64*(w&3), +64) of the row group w>>2 (rows
// [32*MT*(w>>2), +32*MT) of the tile).  RG = 2 puts two waves on every weight-column slice: their B-fragment
// requests are identical and merge in the CU's L1, so a 64-row tile streams the weights once instead of twice.
__device__ __forceinline__ int wave_col() { return (threadIdx.x >> 6) & 3; }
__device__ __forceinline__ int wave_rg() { return threadIdx.x >> 8; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also emits s_waitcnt vmcnt(0), which parks
// every wave until its outstanding GLOBAL loads/stores (activation saves, prefetched masks) have retired;
// nothing in these kernels hands global data between waves, so only lgkmcnt has to drain.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SWISH = 2 };

// 1 / x as ONE v_rcp_f32 (1 ulp) instead of the IEEE division sequence (v_div_scale x 2, v_rcp, four fma, v_div_fmas, v_div_fixup):
// a Swish is exp + reciprocal + two multiplies, and the ensemble step evaluates ~500 of them per wave and 64-row tile in a kernel
// whose vector ALU is as busy as its matrix pipe.  The results move by ~1e-7 relative, two orders inside the parity tolerance.
#ifndef SWISH_FAST_RCP
#define SWISH_FAST_RCP 1
#endif
__device__ __forceinline__ float fast_rcp(float x) {
#if SWISH_FAST_RCP
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.f / x;
#endif
}

template <int ACT>
__device__ __forceinline__ float activate(float x) {
  if (ACT == ACT_RELU) return fmaxf(x, 0.f);
  if (ACT == ACT_SWISH) return x * fast_rcp(1.f + __expf(-x));   // x*sigmoid(x), mobody_module.py:13-15
  return x;
}

// --------------------------------------------------------------------------------------------
// wide GEMM:  acc[mt][nt] = X[64 x Kp] (LDS) * W[Kp x 256] (global, row major, ld = 256)   (acc is written, not accumulated onto)
// Kp multiple of 8.  Columns of this wave: 64*w + 32*nt + (lane&31).
// --------------------------------------------------------------------------------------------
template <int MT>
__device__ __forceinline__ void wide_zero(f32x16 (&acc)[MT][2]) {
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}

// MT = 2: 64-row workgroup tile (2x2 MFMA tiles per wave); MT = 1: 32-row tile (1x2), used when a launch has too
// few 64-row tiles to fill 256 CUs (twice the workgroups, half the LDS, ~100 VGPRs -> 4 workgroups per CU).
// Storage index of element (k, n) of a 256-column ("wide") weight matrix.  Wide matrices are stored K-interleaved
// by four, [K/4][256][4]: the four k-steps one MFMA chunk needs for column n are 16 contiguous bytes, so a lane
// fetches its B fragments for a whole chunk with ONE 16-byte load and a wave instruction reads 1 KB contiguous.
__host__ __device__ inline long long wide_idx(int k, int n) { return ((long long)(k >> 2) * HID + n) * 4 + (k & 3); }

// Register ring of weight fragments: chunk c of four k-steps is requested WIDE_RING-1 chunks before its MFMAs.
// One chunk is 8*MT MFMAs = 512*MT cycles of this wave's pipe time, while an L2 round trip under load is
// ~0.9 us (~2000 cycles): throughput per wave = bytes in flight / latency, so 4 chunks (2 KB per wave) are
// kept in flight.  (With 2 chunks of 4-byte loads the 640-workgroup twin-Q forward ran at 35 % MFMA busy.)
constexpr int WIDE_RING = 5;
#ifndef GEMM_PEEL
#define GEMM_PEEL 1              // 0: accumulate onto the caller's zero-filled registers (A/B aid)
#endif
template <int R>
struct WideRingT { f32x4 r[R][2]; };
using WideRing = WideRingT<WIDE_RING>;

// A wave-uniform pointer as a buffer descriptor (records = bytes): loads through it take ONE per-lane 32-bit offset plus a scalar
// offset, no 64-bit vector address arithmetic (the pointer is forced into scalar registers: kernels pass member bases that are
// uniform by construction).
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  const void* base = (const void*)(((unsigned long long)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                   (unsigned int)__builtin_amdgcn_readfirstlane((int)a));
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// (fp32 weight fragments and the narrow layers' fragments keep plain pointer loads: through the descriptor c2 was neutral and the
//  small launches -- c1, pre-training -- 2 % slower; the 64 16-byte loads per tile of the split core are where it pays, tile_bf.h)
__device__ __forceinline__ void wide_ldb(const float* __restrict__ W, int Kp, int c, f32x4 (&b)[2]) {
  const int lane = lane_id();
  const int i = lane & 31, h = lane >> 5;
  const int nch = Kp >> 3;                      // chunks of four k-steps per lane half
  // per-lane part of the weight address as a 32-bit element offset; the chunk advance is wave-uniform
  const int lane_off = ((h * nch) * HID + 64 * wave_col() + i) * 4;
  const float* wn = W + (size_t)c * (HID * 4);
  b[0] = *reinterpret_cast<const f32x4*>(wn + lane_off);
  b[1] = *reinterpret_cast<const f32x4*>(wn + lane_off + 128);
}

// Request the first WIDE_RING-1 chunks of W.  Called as early as the data dependences allow (before the input
// tile is in LDS, before the previous layer's epilogue): a layer that starts with a cold ring puts one L2/HBM round
// trip (~1 us) in front of its first MFMA, and in a single-generation launch every workgroup does so at once.
template <int R>
__device__ __forceinline__ void wide_prefetch(const float* __restrict__ W, int Kp, WideRingT<R>& ring) {
  const int nch = Kp >> 3;
#pragma unroll
  for (int j = 0; j < R - 1; ++j) wide_ldb(W, Kp, min(j, nch - 1), ring.r[j]);   // unconditional (see wide_gemm)
  __builtin_amdgcn_sched_barrier(0);
}

// `ring` must hold wide_prefetch(W, Kp).
template <int MT, int R>
__device__ __forceinline__ void wide_gemm(const float* __restrict__ Xs, const float* __restrict__ W, int Kp,
                                          f32x16 (&acc)[MT][2], WideRingT<R>& ring) {
  const int lane = lane_id();
  const int i = lane & 31, h = lane >> 5;
  const int kh = Kp >> 1;                       // K range of this lane half, multiple of 4
  const int nch = kh >> 2;                      // chunks of four k-steps
  const float* xa = Xs + (32 * MT * wave_rg() + i) * LDX + h * kh;
  // FIRST (chunk 0): the first MFMA of each accumulator takes a literal zero C operand -- acc is WRITTEN by this GEMM, a
  // caller's wide_zero is dead code (the chunk loop is a runtime loop: the compiler cannot fold the zero fill itself, and it
  // costs 32 * MT v_mov per GEMM and wave in kernels whose vector ALU is a third of their busy time)
  auto mma = [&](auto first_c, int c, f32x4 (&b)[2]) {
    constexpr bool FIRST = GEMM_PEEL && decltype(first_c)::value;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x4 av[MT];
#pragma unroll
    for (int x = 0; x < MT; ++x) av[x] = *reinterpret_cast<const f32x4*>(xa + 32 * x * LDX + 4 * c);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int x = 0; x < MT; ++x) {
        acc[x][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][u], b[0][u], (FIRST && u == 0) ? zero : acc[x][0], 0, 0, 0);
        acc[x][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][u], b[1][u], (FIRST && u == 0) ? zero : acc[x][1], 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
  };
  // A K known at compile time (the 256-wide layers) unrolls completely and drops the loads past the last chunk.  A runtime
  // K (layer 1) keeps every load UNCONDITIONAL, from a clamped chunk index: with a load inside a branch the compiler's
  // vmcnt bookkeeping merges the two paths and waits for the loads it has just issued (seen in the ISA of the K = 24
  // layer: load, load, vmcnt(1), mfma -- one L2 round trip per chunk, 7 us for a layer with 0.6 us of MFMA work).
  const bool known = __builtin_constant_p(Kp);
  // chunk 0 (every K has one), then chunks 1 .. nch - 1: c0 == 1 (mod R), so the ring slot of chunk c0 + j is (1 + j) % R
  if (!known || R - 1 < nch) wide_ldb(W, Kp, min(R - 1, nch - 1), ring.r[(R - 1) % R]);
  __builtin_amdgcn_sched_barrier(0);
  mma(std::true_type{}, 0, ring.r[0]);
  __builtin_amdgcn_sched_barrier(0);
  for (int c0 = 1; c0 < nch; c0 += R) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int c = c0 + j;
      if (!known || c + R - 1 < nch) wide_ldb(W, Kp, min(c + R - 1, nch - 1), ring.r[(1 + j + R - 1) % R]);
      // Pin the issue order: without this fence hipcc sinks each prefetch load down to its first use (it trades
      // the ring's registers for occupancy), which collapses the 4-chunk prefetch distance to ~1 chunk and puts
      // an L2 round trip in front of every chunk's MFMAs (seen in the ISA: load ... vmcnt(1) ... mfma of it).
      __builtin_amdgcn_sched_barrier(0);
      if (c < nch) mma(std::false_type{}, c, ring.r[(1 + j) % R]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int MT>
__device__ __forceinline__ void wide_gemm(const float* __restrict__ Xs, const float* __restrict__ W, int Kp,
                                          f32x16 (&acc)[MT][2]) {
  WideRing ring;
  wide_prefetch(W, Kp, ring);
  wide_gemm<MT, WIDE_RING>(Xs, W, Kp, acc, ring);
}

// Visit every accumulator element of a wide result: f(row 0..32*MT-1, col 0..255, value).
template <int MT, class F>
__device__ __forceinline__ void wide_foreach(f32x16 (&acc)[MT][2], F&& f) {
  const int lane = lane_id(), w = wave_col();
  const int i = lane & 31, h = lane >> 5;
  const int rbase = 32 * MT * wave_rg();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;   // C/D map of 32x32 MFMA
        const int col = 64 * w + 32 * nt + i;
        f(row, col, acc[mt][nt][r]);
      }
}

// --------------------------------------------------------------------------------------------
// narrow GEMM: rows [16w,16w+16) of X (LDS, Kp cols) times W[Kp x Np] (global, ld = Np),
// NT consecutive 16-column tiles starting at column tile nt0.  Kp multiple of 8.
// acc[nt][r]: row = 16w + 4*(lane>>4) + r, col = 16*(nt0+nt) + (lane&15).
// --------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void narrow_gemm(const float* __restrict__ Xs, const float* __restrict__ W, int Kp, int Np,
                                            int nt0, f32x4 (&acc)[NT], int bm = BM) {
  constexpr int C = 8 / NT;                      // k-pairs per prefetch chunk (8 MFMAs per chunk and accumulator pair)
  const int lane = lane_id(), w = wave_id();
  if (16 * w >= bm) {                            // 32-row tiles: waves 2,3 own no rows in a narrow layer (wave uniform)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;
    return;
  }
  const int i = lane & 15, q = lane >> 4;
  const int kq = Kp >> 2;                       // K range of this lane quarter, multiple of 2
  const int steps = kq >> 1;                    // k-pairs
  const float* xa = Xs + (16 * w + i) * LDX + q * kq;
  const float* wb = W + (size_t)(q * kq) * Np + 16 * nt0 + i;
  f32x4 acc2[NT];                               // second chain hides the 40-cycle dependent latency
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[n][r] = 0.f; acc2[n][r] = 0.f; }
  // Weight fragments of a whole chunk are requested before the chunk's MFMAs and one chunk ahead (two register
  // sets): loading each fragment right before its MFMA exposed one L2 round trip per k-pair (64 per K=256 layer).
  auto loadB = [&](int p, float (&bv)[C][2][NT]) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int t = 2 * min(p + c, steps - 1);   // clamped: tail steps re-read the last pair (their A is zeroed)
#pragma unroll
      for (int n = 0; n < NT; ++n) { bv[c][0][n] = wb[(size_t)t * Np + 16 * n]; bv[c][1][n] = wb[(size_t)(t + 1) * Np + 16 * n]; }
    }
  };
  auto mma = [&](int p, float (&bv)[C][2][NT]) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const bool ok = (p + c) < steps;
      float2 av = *reinterpret_cast<const float2*>(xa + 2 * min(p + c, steps - 1));
      av.x = ok ? av.x : 0.f; av.y = ok ? av.y : 0.f;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[c][0][n], acc[n], 0, 0, 0);
        acc2[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[c][1][n], acc2[n], 0, 0, 0);
      }
    }
  };
  // The loads are unconditional (clamped inside loadB): a load behind a branch makes the compiler wait for the loads it
  // has just issued at the merge point (see wide_gemm).  Only the MFMAs of a chunk past the end are skipped.
  float b0[C][2][NT], b1[C][2][NT];
  loadB(0, b0);
  for (int p = 0; p < steps; p += 2 * C) {
    loadB(p + C, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma(p, b0);
    __builtin_amdgcn_sched_barrier(0);
    loadB(p + 2 * C, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (p + C < steps) mma(p + C, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[n][r] += acc2[n][r];
}

// Run a narrow layer over all Np/16 column tiles in groups of <= 2; f(row 0..63, col, value).
template <class F>
__device__ __forceinline__ void narrow_layer(const float* __restrict__ Xs, const float* __restrict__ W, int Kp, int Np,
                                             F&& f, int bm = BM) {
  const int lane = lane_id(), w = wave_id();
  const int i = lane & 15, q = lane >> 4;
  const int ntiles = Np >> 4;
  if (16 * w >= bm) return;
  int nt0 = 0;
  for (; nt0 + 2 <= ntiles; nt0 += 2) {
    f32x4 acc[2];
    narrow_gemm<2>(Xs, W, Kp, Np, nt0, acc, bm);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) f(16 * w + 4 * q + r, 16 * (nt0 + n) + i, acc[n][r]);
  }
  if (nt0 < ntiles) {
    f32x4 acc[1];
    narrow_gemm<1>(Xs, W, Kp, Np, nt0, acc, bm);
#pragma unroll
    for (int r = 0; r < 4; ++r) f(16 * w + 4 * q + r, 16 * nt0 + i, acc[0][r]);
  }
}

// --------------------------------------------------------------------------------------------
// K-split narrow layer (K = 256, Np = 16*NT <= 32, 4 waves, rows = 16*MTN): wave w contracts k in [64w, 64w+64)
// for ALL rows, so its weight fragments are just 16*NT registers that can be requested long before the layer
// starts (narrow_prefetch); the four partial results meet in LDS.  The row-split narrow_gemm above keeps two of
// four waves idle on a 32-row tile and walks K = 256 in four dependent load rounds (3.7 us per tile measured,
// against 0.2 us of MFMA work).  Lane (i = lane&15, q = lane>>4) supplies k = 64w + 16q + s at step s.
// --------------------------------------------------------------------------------------------
template <int NT>
struct NarrowRegs { float b[16][NT]; };

template <int NT>
__device__ __forceinline__ void narrow_prefetch(const float* __restrict__ W, int Np, NarrowRegs<NT>& br) {
  const int lane = lane_id();
  const float* wb = W + (size_t)(64 * wave_id() + 16 * (lane >> 4)) * Np + (lane & 15);
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int n = 0; n < NT; ++n) br.b[s][n] = wb[(size_t)s * Np + 16 * n];
  __builtin_amdgcn_sched_barrier(0);
}

// f(row, col, value) is called once per output element (row < 16*MTN, col < 16*NT).  Overwrites Xs.
template <int MTN, int NT, class F>
__device__ __forceinline__ void narrow_run(float* Xs, const NarrowRegs<NT>& br, F&& f) {
  constexpr int Np = 16 * NT, ROWS = 16 * MTN;
  const int lane = lane_id(), w = wave_id();
  const int i = lane & 15, q = lane >> 4;
  f32x4 acc[MTN][NT];
#pragma unroll
  for (int m = 0; m < MTN; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
  const float* xa = Xs + i * LDX + 64 * w + 16 * q;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    f32x4 av[MTN];
#pragma unroll
    for (int m = 0; m < MTN; ++m) av[m] = *reinterpret_cast<const f32x4*>(xa + 16 * m * LDX + 4 * s4);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int m = 0; m < MTN; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][u], br.b[4 * s4 + u][n], acc[m][n], 0, 0, 0);
  }
  lds_barrier();                                 // every wave has read its slice of the image
  float* P = Xs + w * (ROWS * Np);
#pragma unroll
  for (int m = 0; m < MTN; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(16 * m + 4 * q + r) * Np + 16 * n + i] = acc[m][n][r];
  lds_barrier();
  for (int e = threadIdx.x; e < ROWS * Np; e += NTHREADS) {
    const float v = ((Xs[e] + Xs[ROWS * Np + e]) + Xs[2 * ROWS * Np + e]) + Xs[3 * ROWS * Np + e];
    f(e / Np, e % Np, v);
  }
}

// The same K-split layer for wider heads (Np = 16*NT up to 128: pen 48, ant 112), 32 rows per pass so the accumulators stay at
// 8*NT registers: the four partial results of a pass meet in the LDS region of the rows it just consumed, through TWO
// buffers (waves 2,3 store, waves 0,1 add their own onto them, then everyone sums the two) because four buffers of
// 32 x 112 floats do not fit under a 32-row image.  f(row, col, value) once per output element; overwrites Xs.
template <int MTN, int NT, class F>
__device__ __forceinline__ void narrow_run_wide(float* Xs, const NarrowRegs<NT>& br, F&& f) {
  static_assert(MTN % 2 == 0 && 2 * 32 * 16 * NT <= 32 * LDX, "32-row passes, two partial buffers inside the pass's rows");
  constexpr int Np = 16 * NT, PB = 32 * Np;
  const int lane = lane_id(), w = wave_id();
  const int i = lane & 15, q = lane >> 4;
#pragma unroll 1
  for (int p = 0; p < MTN / 2; ++p) {
    f32x4 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
    const float* xa = Xs + (32 * p + i) * LDX + 64 * w + 16 * q;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      f32x4 av[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) av[m] = *reinterpret_cast<const f32x4*>(xa + 16 * m * LDX + 4 * s4);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][u], br.b[4 * s4 + u][n], acc[m][n], 0, 0, 0);
    }
    lds_barrier();                               // every wave has read this pass's 32 rows
    float* P = Xs + 32 * p * LDX + (w & 1) * PB;
    auto sweep = [&](bool add) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = P + (16 * m + 4 * q + r) * Np + 16 * n + i;
            *d = add ? *d + acc[m][n][r] : acc[m][n][r];
          }
    };
    if (w >= 2) sweep(false);
    lds_barrier();
    if (w < 2) sweep(true);
    lds_barrier();
    const float* S0 = Xs + 32 * p * LDX;
    for (int e = threadIdx.x; e < PB; e += NTHREADS) f(32 * p + e / Np, e % Np, S0[e] + S0[PB + e]);
  }
}

// --------------------------------------------------------------------------------------------
// LDS tile fill: X[r][col0 + c] = src[(row0+r)*ld + c] for c < n (zero for rows >= rows).
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_load(float* Xs, int col0, const float* __restrict__ src, int ld, int n, int row0,
                                          int rows, int bm = BM) {
  // Thread t owns column (t & 31) of every 32-column chunk and rows (t >> 5) + k * (threads/32): no integer division
  // (an idx / n decode by a runtime n cost ~20 VALU instructions per element, 700+ per wave in the forward prologue).
  // The four loads of a pass are issued before the first LDS write (a load followed directly by its dependent
  // ds_write costs one HBM round trip per element); they are unconditional from a clamped row and column (a
  // conditional load would branch and drain vmcnt per element), invalid rows are zeroed by select.
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5, rstep = (int)blockDim.x >> 5;
  for (int cb = 0; cb < n; cb += 32) {
    const int col = cb + c;
    const int colc = min(col, n - 1);
    for (int rb = r0; rb < bm; rb += 4 * rstep) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[(size_t)min(row0 + rb + u * rstep, rows - 1) * ld + colc];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = rb + u * rstep;
        if (col < n && r < bm) Xs[r * LDX + col0 + col] = (row0 + r < rows) ? v[u] : 0.f;
      }
    }
  }
}
// Two sources of at most 32 columns each (state | action, the train step's inputs) in ONE round trip: thread t owns column
// t & 31 of both and rows (t >> 5) + 8 u; every load of both sources is requested before the first LDS write (tile_load per
// source is two dependent round trips -- the second source's loads sit behind the first one's stores -- ~1 us per forward
// launch whose workgroups all wait at the same moment).  n1 may be 0 (one source: the second is not read).  TB = 32 or 64.
template <int TB>
__device__ __forceinline__ void tile_load2(float* Xs, const float* __restrict__ src0, int ld0, int n0,
                                           const float* __restrict__ src1, int ld1, int n1, int rows) {
  constexpr int U = TB / 8;
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  const float* s1 = n1 > 0 ? src1 : src0;            // an absent second source re-reads the first (result unused)
  const int l1 = n1 > 0 ? ld1 : ld0, m1 = n1 > 0 ? n1 : n0;
  float v0[U], v1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t r = (size_t)min(r0 + 8 * u, rows - 1);
    v0[u] = src0[r * ld0 + min(c, n0 - 1)];
    v1[u] = s1[r * l1 + min(c, m1 - 1)];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int r = r0 + 8 * u;
    const bool ok = r < rows;
    if (c < n0) Xs[r * LDX + c] = ok ? v0[u] : 0.f;
    if (c < n1) Xs[r * LDX + n0 + c] = ok ? v1[u] : 0.f;
  }
}

__device__ __forceinline__ void tile_zero_cols(float* Xs, int c0, int c1, int bm = BM) {
  for (int r = threadIdx.x; r < bm; r += (int)blockDim.x)
    for (int c = c0; c < c1; ++c) Xs[r * LDX + c] = 0.f;
}

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace mobody
