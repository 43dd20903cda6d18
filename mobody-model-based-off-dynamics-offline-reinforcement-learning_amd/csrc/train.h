// Argument blocks shared by the training translation units (mlp_bwd.hip, train.hip).
#pragma once
#include "common.h"

namespace mobody {

struct Mlp3BwdArgs {
  const float* dz3;        // [members][rows][Np3] (zero in padded columns)
  const float* h1;         // [members][rows][256] post-ReLU hidden activations saved by the forward
  const float* h2;
  const float* wt;         // transposed blob (member 0)
  long long t_mstride, w3t, w2t, w1t;
  int Np3, Np1t;
  long long rows;
  float* dz2;              // [members][rows][256] or null (not needed when only dx is wanted)
  float* dz1;
  float* dbp;              // [tiles][members][512 + Np3] bias-gradient partials: db1 | db2 | db3
  float* dx;               // [members][rows][dx_n] input gradient columns [dx_c0, dx_c0 + dx_n)  (DX only)
  int dx_c0, dx_n;
};
int launch_mlp3_bwd(const Mlp3BwdArgs& a, int members, bool with_dx, int tile_rows, hipStream_t st);

struct WgradJob {
  const float* A; long long a_mstride; int lda, ka;    // A[rows][lda], columns < ka contribute (output rows)
  const float* B; long long b_mstride; int ldb, nb;    // B[rows][ldb], columns < nb contribute (output cols)
  long long out_off; int out_ld, out_k, out_n;         // slab + out_off + m*out_mstride + k*out_ld + n  (k<out_k, n<out_n)
  int transposed;                                      // store [n][k] instead (used for dW3, computed as dz3^T h2)
  int wide;                                            // destination is a 256-wide matrix in wide_idx storage
  int tiles_n, ntiles;                                 // filled by launch_wgrad
};
struct WgradArgs {
  WgradJob job[3];                                     // job 0: 64x64 wave tiles, jobs 1-2: 32x64
  long long rows, rows_per_wave;
  float* slabs; long long slab_stride, out_mstride;    // partial slab s = slabs + s*slab_stride (gradient-blob layout)
  int nsplit, members, tiles_total;
};
int launch_wgrad(WgradArgs a, hipStream_t st);

// Final reduction of per-workgroup loss partials, done by one extra workgroup of k_grad_reduce (saves a launch).
//   kind 1 (critic): out[0] = scale * sum parts[k]
//   kind 2 (actor):  parts = pairs (sum -min q, sum w*(pi-a)^2);  bc = s1/ntg_a;
//                    out[0] = p_w * s0 / ng + bc_coef * bc, out[1] = bc,  p_w = scale_q ? weight/(stats[0]/ng) : 1
struct LossFinal {
  int kind, nparts, scale_q;
  float scale, weight, bc_coef, ng, ntg_a;
  const float* parts; const float* stats; float* out;
};

struct GradReduceArgs {
  MobodyMlpLayout L;
  const float* slabs; long long slab_stride; int nsplit;
  const float* dbp; int ntiles;
  float* grad;
  LossFinal loss;       // kind 0: none
};
int launch_grad_reduce(const GradReduceArgs& a, hipStream_t st);

// split-K factor (workgroups along the row dimension) used for a batch of `rows`
inline int wgrad_nsplit(long long rows) {
  long long s = rows / 256;
  if (s < 1) s = 1;
  if (s > 16) s = 16;
  return (int)s;
}

}  // namespace mobody
