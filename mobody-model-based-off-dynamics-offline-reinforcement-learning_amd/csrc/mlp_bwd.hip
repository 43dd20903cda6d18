// Backward kernels of the 3-layer ReLU MLP (actor / twin-Q), fp32 MFMA.
//
//   k_mlp3_bwd   per (64-row tile, member): dz3 -> dz2 = (dz3 W3^T) * [h2>0] -> dz1 = (dz2 W2^T) * [h1>0]
//                (-> dx = dz1 W1^T for the frozen-Q pass of the actor update), plus the bias-gradient
//                partial sums of the tile.  W^T blobs are streamed as MFMA B operands exactly like the
//                forward weights.                                   (autograd of mobody.py:35-48)
//   k_wgrad      dW[k][n] = sum_rows A[row][k] * dZ[row][n]: rows are the contraction index, both
//                operands are read straight from global memory in MFMA fragment order (a wave
//                instruction = two full 128-byte lines); split-K over row slices, the four waves of a
//                workgroup reduce through LDS and write one deterministic partial slab.
//   k_grad_reduce slabs + bias partials -> gradient blob (deterministic, no atomics).
//
// Roofline: k_mlp3_bwd and k_wgrad are MFMA-f32 bound (2*256*256 FLOP per row and layer against
// ~2 KB of activations per row); k_grad_reduce is HBM/L2 streaming.
#include "common.h"
#include "layers.h"
#include "train.h"

namespace mobody {

// per-lane column sums of a wide accumulator after masking; lanes < 32 end up with the full 64-row sums
// for columns 64w + 32nt + (lane&31), nt = 0,1
template <class Mask>
__device__ __forceinline__ void wide_mask_store_colsum(f32x16 (&acc)[2][2], float* Xs, float* gdst, int rows_here,
                                                       Mask&& mask, float (&cs)[2]) {
  const int lane = lane_id(), w = wave_id();
  const int i = lane & 31, h = lane >> 5;
  cs[0] = cs[1] = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = 64 * w + 32 * nt + i;
        const bool valid = row < rows_here;
        const float dz = (valid && mask(row, col)) ? acc[mt][nt][r] : 0.f;
        Xs[row * LDX + col] = dz;
        if (gdst != nullptr && valid) gdst[row * HID + col] = dz;
        cs[nt] += dz;
      }
  cs[0] += __shfl_xor(cs[0], 32);
  cs[1] += __shfl_xor(cs[1], 32);
}

template <bool DX>
__global__ __launch_bounds__(NTHREADS, 2) void k_mlp3_bwd(Mlp3BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  const int m = blockIdx.y;
  const long long row0 = (long long)blockIdx.x * BM;
  const int rows_here = (int)min((long long)BM, a.rows - row0);
  const int lane = lane_id(), w = wave_id();
  const float* w3t = a.wt + m * a.t_mstride + a.w3t;
  const float* w2t = a.wt + m * a.t_mstride + a.w2t;
  const float* w1t = a.wt + m * a.t_mstride + a.w1t;
  const float* h1 = a.h1 + ((long long)m * a.rows + row0) * HID;
  const float* h2 = a.h2 + ((long long)m * a.rows + row0) * HID;
  float* dz2 = a.dz2 ? a.dz2 + ((long long)m * a.rows + row0) * HID : nullptr;
  float* dz1 = a.dz1 ? a.dz1 + ((long long)m * a.rows + row0) * HID : nullptr;
  float* dbp = a.dbp + ((long long)blockIdx.x * gridDim.y + m) * (2 * HID + a.Np3);

  tile_load(Xs, 0, a.dz3 + ((long long)m * a.rows + row0) * a.Np3, a.Np3, a.Np3, 0, rows_here);
  __syncthreads();
  if ((int)threadIdx.x < a.Np3) {                 // db3 partial of this tile
    float s = 0.f;
    for (int r = 0; r < BM; ++r) s += Xs[r * LDX + threadIdx.x];
    dbp[2 * HID + threadIdx.x] = s;
  }

  f32x16 acc[2][2];
  float cs[2];
  // dh2 = dz3 * W3^T ; dz2 = dh2 * [h2 > 0]
  wide_zero(acc);
  wide_gemm(Xs, w3t, a.Np3, acc);
  __syncthreads();
  wide_mask_store_colsum(acc, Xs, dz2, rows_here, [&](int row, int col) { return h2[row * HID + col] > 0.f; }, cs);
  if (lane < 32) { dbp[HID + 64 * w + lane] = cs[0]; dbp[HID + 64 * w + 32 + lane] = cs[1]; }
  __syncthreads();
  // dh1 = dz2 * W2^T ; dz1 = dh1 * [h1 > 0]
  wide_zero(acc);
  wide_gemm(Xs, w2t, HID, acc);
  __syncthreads();
  wide_mask_store_colsum(acc, Xs, dz1, rows_here, [&](int row, int col) { return h1[row * HID + col] > 0.f; }, cs);
  if (lane < 32) { dbp[64 * w + lane] = cs[0]; dbp[64 * w + 32 + lane] = cs[1]; }
  if (DX) {
    __syncthreads();
    float* dx = a.dx + ((long long)m * a.rows + row0) * a.dx_n;
    narrow_layer(Xs, w1t, HID, a.Np1t, [&](int row, int col, float v) {
      const int c = col - a.dx_c0;
      if (row < rows_here && c >= 0 && c < a.dx_n) dx[row * a.dx_n + c] = v;
    });
  }
}

int launch_mlp3_bwd(const Mlp3BwdArgs& a, int members, bool with_dx, hipStream_t st) {
  if (a.rows <= 0) return 0;
  static bool once = false;
  if (!once) {
    int rc = allow_big_lds(k_mlp3_bwd<false>, TILE_LDS_BYTES);
    if (rc) return rc;
    rc = allow_big_lds(k_mlp3_bwd<true>, TILE_LDS_BYTES);
    if (rc) return rc;
    once = true;
  }
  dim3 grid((unsigned)cdiv(a.rows, BM), (unsigned)members);
  ProfScope prof(PROF_MLP_BWD, st);
  if (with_dx) hipLaunchKernelGGL(k_mlp3_bwd<true>, grid, dim3(NTHREADS), TILE_LDS_BYTES, st, a);
  else hipLaunchKernelGGL(k_mlp3_bwd<false>, grid, dim3(NTHREADS), TILE_LDS_BYTES, st, a);
  MB_LAUNCH_OK("k_mlp3_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// weight gradient GEMM
// ------------------------------------------------------------------------------------------------
template <int MT, int NT>
__global__ __launch_bounds__(NTHREADS, 2) void k_wgrad(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];     // [4 waves][32MT][32NT]
  constexpr int TK = 32 * MT, TN = 32 * NT;
  const int lane = lane_id(), w = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const int tiles_n = (a.nb + TN - 1) / TN;
  const int tk = blockIdx.x / tiles_n, tn = blockIdx.x - tk * tiles_n;
  const int k0 = tk * TK, n0 = tn * TN;
  const int m = blockIdx.z;
  const float* A = a.A + m * a.a_mstride;
  const float* B = a.B + m * a.b_mstride;
  const long long r_begin = ((long long)blockIdx.y * 4 + w) * a.rows_per_wave;
  const long long r_end = min(a.rows, r_begin + a.rows_per_wave);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

  bool okA[MT], okB[NT];
#pragma unroll
  for (int x = 0; x < MT; ++x) okA[x] = (k0 + 32 * x + i) < a.ka;
#pragma unroll
  for (int y = 0; y < NT; ++y) okB[y] = (n0 + 32 * y + i) < a.nb;
  const float* pa = A + k0 + i;
  const float* pb = B + n0 + i;

  constexpr int U = 4;     // row pairs per unrolled chunk
  for (long long rb = r_begin; rb < r_end; rb += 2 * U) {
    float av[U][MT], bv[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = rb + 2 * u + h;          // lane half h takes the odd row of the pair
      const bool rv = row < r_end;
#pragma unroll
      for (int x = 0; x < MT; ++x) av[u][x] = (rv && okA[x]) ? pa[row * a.lda + 32 * x] : 0.f;
#pragma unroll
      for (int y = 0; y < NT; ++y) bv[u][y] = (rv && okB[y]) ? pb[row * a.ldb + 32 * y] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int x = 0; x < MT; ++x)
#pragma unroll
        for (int y = 0; y < NT; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][x], bv[u][y], acc[x][y], 0, 0, 0);
  }

  // ---- reduce the four row slices of this workgroup through LDS ----
  float* mine = red + w * (TK * TN);
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = 32 * x + (r & 3) + 8 * (r >> 2) + 4 * h;
        mine[kk * TN + 32 * y + i] = acc[x][y][r];
      }
  __syncthreads();
  float* slab = a.slabs + (long long)blockIdx.y * a.slab_stride + a.out_off + m * a.out_mstride;
  for (int idx = threadIdx.x; idx < TK * TN; idx += NTHREADS) {
    const int kk = idx / TN, nn = idx - kk * TN;
    const float s = (red[idx] + red[TK * TN + idx]) + (red[2 * TK * TN + idx] + red[3 * TK * TN + idx]);
    if (k0 + kk < a.out_k && n0 + nn < a.out_n) slab[(long long)(k0 + kk) * a.out_ld + n0 + nn] = s;
  }
}

template <int MT, int NT>
static int launch_wgrad_t(const WgradArgs& a, int members, int nsplit, hipStream_t st) {
  constexpr size_t lds = (size_t)4 * 32 * MT * 32 * NT * sizeof(float);
  static bool once = false;
  if (!once) {
    int rc = allow_big_lds(k_wgrad<MT, NT>, lds);
    if (rc) return rc;
    once = true;
  }
  const int tiles = (int)(cdiv(a.ka, 32 * MT) * cdiv(a.nb, 32 * NT));
  ProfScope prof(PROF_WGRAD, st);
  hipLaunchKernelGGL((k_wgrad<MT, NT>), dim3(tiles, nsplit, members), dim3(NTHREADS), lds, st, a);
  MB_LAUNCH_OK("k_wgrad");
  return 0;
}

// Pick the wave tile from the operand widths; `a.rows_per_wave` is derived from nsplit here.
int launch_wgrad(WgradArgs a, int members, int nsplit, hipStream_t st) {
  if (a.rows <= 0) return 0;
  long long rpw = cdiv(a.rows, (long long)4 * nsplit);
  rpw = (rpw + 1) & ~1LL;
  a.rows_per_wave = rpw;
  if (a.ka <= 32) return launch_wgrad_t<1, 2>(a, members, nsplit, st);
  if (a.nb <= 32) return launch_wgrad_t<2, 1>(a, members, nsplit, st);
  return launch_wgrad_t<2, 2>(a, members, nsplit, st);
}

// ------------------------------------------------------------------------------------------------
// slabs + bias partials -> gradient blob
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grad_reduce(GradReduceArgs a) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= a.L.total_floats) return;
  const int m = (int)(j / a.L.member_floats);
  const long long o = j - (long long)m * a.L.member_floats;
  int bsel = -1, bidx = 0;     // bias segment: 0 -> db1, 1 -> db2, 2 -> db3
  if (o >= a.L.b1 && o < a.L.b1 + HID) { bsel = 0; bidx = (int)(o - a.L.b1); }
  else if (o >= a.L.b2 && o < a.L.b2 + HID) { bsel = 1; bidx = (int)(o - a.L.b2); }
  else if (o >= a.L.b3) { bsel = 2; bidx = (int)(o - a.L.b3); }
  float s = 0.f;
  if (bsel < 0) {
    for (int k = 0; k < a.nsplit; ++k) s += a.slabs[(long long)k * a.slab_stride + j];
  } else {
    const int per = 2 * HID + a.L.Np3;
    const int off = bsel == 0 ? bidx : bsel == 1 ? HID + bidx : 2 * HID + bidx;
    for (int t = 0; t < a.ntiles; ++t) s += a.dbp[((long long)t * a.L.members + m) * per + off];
  }
  a.grad[j] = s;
}

int launch_grad_reduce(const GradReduceArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_grad_reduce, dim3((unsigned)cdiv(a.L.total_floats, 256)), dim3(256), 0, st, a);
  MB_LAUNCH_OK("k_grad_reduce");
  return 0;
}

}  // namespace mobody
