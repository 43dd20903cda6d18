// Argument blocks shared by the training translation units (mlp_bwd.hip, train.hip).
#pragma once
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "tile.h"
#include "tile_bf.h"

namespace mobody {

// Row-wise quantities of the actor update (mobody.py:246-276, 314-345) shared by the seed prologues of k_mlp3_bwd.
struct ActorRowArgs {
  const float *qp, *qb, *stats, *pi, *act, *dxa;
  const float* v_true;       // [Nt] V(s_true) when config['advantage'] (else null)
  float* bcw;                // [Nt] BC weights (written by the frozen-Q backward, read by the actor backward)
  long long N, Nt, Ng, Ntg;
  int A;
  MobodyHyper h;
};
__device__ __forceinline__ float policy_weight(const ActorRowArgs& a) {       // p_w, mobody.py:318 / :283
  return a.h.scale_q ? a.h.weight / (a.stats[0] / (float)a.Ng) : 1.f;
}
__device__ __forceinline__ float bc_weight(const ActorRowArgs& a, long long row) {   // exp_adv, :257-267
  if (!a.h.q_weighted) return 1.f;
  const float qb = fminf(a.qb[row], a.qb[a.Nt + row]);
  const float adv = a.v_true ? qb - a.v_true[row]                      // advantage variant, mobody.py:255-256
                             : qb / (a.stats[1] / (float)a.Ntg);
  return fminf(expf(3.f * adv), 100.f);
}

// Where the output-layer gradient dz3 of a backward launch comes from.  Modes 1-3 compute it in the kernel's
// prologue from the row-wise inputs (no separate row-wise launch, no dz3 round trip through HBM before the first GEMM):
//   0  dz3 read from memory
//   1  critic: y = r + nd*gamma*min(Qt1,Qt2) (or q_next); dz3[m][row][0] = 2 (q_m - y) / N_global   (mobody.py:190-207)
//      lossp[tile*2 + m] = sum_rows (q_m - y)^2
//   2  frozen twin-Q of the actor update: dz3[m][row][0] = -p_w/N_global * d min(q0,q1)/dq_m (ties split 1/2, as
//      torch.min's backward); member 0 also writes the BC weights bcw[row < Nt]
//   3  actor: d(pre-tanh) = (dxa[0]+dxa[1] + bc_coef*2*w*(pi-a)/(Ntg*A)) * max_action*(1-tanh^2);
//      lossp[2*tile] = sum -min q, lossp[2*tile+1] = sum w*(pi-a)^2
// Modes 1 and 3 also store dz3 to `dz3_out` (the weight-gradient GEMM reads it).
struct BwdSeed {
  int mode;
  const float *q, *qt, *qnext, *r, *nd;      // mode 1 ([2][rows] q and qt, [rows] the rest)
  float gamma, inv_ng;
  ActorRowArgs ar;                           // modes 2, 3
  float* dz3_out;
  float* lossp;
};

struct Mlp3BwdArgs {
  BwdSeed seed;
  const float* dz3;        // [members][rows][Np3] (zero in padded columns); seed.mode 0 only
  const float* h1;         // [members][rows][256] post-ReLU hidden activations saved by the forward
  const float* h2;         // (swish != 0: the Swish derivatives save_d1 / save_d2 of k_mlp3_fwd_train instead)
  int swish;
  const uint32_t* m1;      // [members][ceil(rows/32)][256] sign bits of h1 / h2 written by the forward; when both are
  const uint32_t* m2;      // given they replace h1 / h2 (which may then be null)
  const float* wt;         // transposed blob (member 0)
  const unsigned short* w2t_planes;   // bf16 planes of W2^T (member 0; wt + L.w2tp) and the precision (0 = exact fp32 MFMA)
  long long planes_ms;
  int prec;
  long long t_mstride, w3t, w2t, w1t;
  int Np3, Np1t;
  long long rows;
  float* dz2;              // [members][rows][256] or null (not needed when only dx is wanted)
  float* dz1;
  unsigned short* dz2p;    // f16 mode, instead of dz2: its two fp16 planes in the weight-gradient GEMM's fragment layout
  long long dz2p_ms;       // ([member][2][rows32 / 8][256][8]; member / plane strides in 16-bit elements, layers_bf.h PlaneSave)
  long long dz2p_plane;
  int* e2_out;             // [members][ceil(rows / 32)] the tiles' scale exponents of dz2p
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
  // prec 4 ("f16x2"): job 0's operands are NOT fp32 rows but the planes the forward / backward epilogues saved
  // (job[0].A = h1 planes, job[0].B = dz2 planes: [member][2][rows32 / 8][256][8] fp16, a_mstride / b_mstride in 16-bit
  // elements) together with the 32-row tiles' scale exponents eA / eB ([member][tiles])
  const int *eA, *eB;
  long long e_mstride, plane_stride;                   // tiles per member; 16-bit elements between the two planes
  WgradJob job[3];                                     // job 0: 64x64 wave tiles, jobs 1-2: 32x64
  long long rows, rows_per_wave;
  float* slabs; long long slab_stride, out_mstride;    // partial slab s = slabs + s*slab_stride (gradient-blob layout)
  int nsplit, members, tiles_total;
  int prec;                                            // 0 exact fp32; 1..3: the 256 x 256 job on the split-precision bf16 core
};
int launch_wgrad(WgradArgs a, hipStream_t st);

// Same op forms as torch's single-tensor Adam: exp_avg.lerp_(g, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2);
// denom = sqrt(v)/sqrt(bc2) + eps; p.addcdiv_(m, denom, -lr/bc1).  The scalar constants are formed in double
// on the host and rounded to fp32 once, as torch does when it multiplies a fp32 tensor by a Python float.
struct AdamConsts { float w1, b2, w2, step_size, bc2_sqrt, eps, tau, one_minus_tau, gscale; };

// W2[k][n] = w -> its terms in the planes of W2 (as B[k][n]) and of W2^T (as B[n][k]) of a member's T blob: the three bf16
// terms (precision modes 0-3 share them), or -- precision 4, "f16x2" -- the two fp16 terms of w * 2^F16_WSHIFT in planes 0, 1
__device__ __forceinline__ void write_w2_planes(float* t_member, const MobodyMlpLayout& L, int k, int n, float w, int precision) {
  short* p2 = reinterpret_cast<short*>(t_member + L.w2p);
  short* p2t = reinterpret_cast<short*>(t_member + L.w2tp);
  if (precision == 4) {
    short t[2];
    split_terms<4>(w * exp2i(F16_WSHIFT), t);
#pragma unroll
    for (int p = 0; p < 2; ++p) { p2[bf_plane_idx(p, k, n)] = t[p]; p2t[bf_plane_idx(p, n, k)] = t[p]; }
  } else {
    short t[3];
    split_terms<3>(w, t);
#pragma unroll
    for (int p = 0; p < 3; ++p) { p2[bf_plane_idx(p, k, n)] = t[p]; p2t[bf_plane_idx(p, n, k)] = t[p]; }
  }
}

// Destination of parameter entry (member-local offset o) inside the member's T blob, or -1 (biases, padding rows).
__device__ __forceinline__ long long t_blob_index(const MobodyMlpLayout& L, long long o) {
  if (o < L.b1) {                                   // W1 (wide storage): W1T[n][k] row major, ld = Np1t
    const long long g = o >> 2;
    const int k = (int)(g / HID) * 4 + (int)(o & 3), n = (int)(g % HID);
    return L.w1t + (long long)n * L.Np1t + k;
  }
  if (o >= L.w2 && o < L.b2) {                      // W2 (wide): W2T[n][k] wide
    const long long oo = o - L.w2, g = oo >> 2;
    const int k = (int)(g / HID) * 4 + (int)(oo & 3), n = (int)(g % HID);
    return L.w2t + wide_idx(n, k);
  }
  if (o >= L.w3 && o < L.b3) {                      // W3 (narrow [256][Np3]): W3T[n3][k] wide
    const long long oo = o - L.w3;
    const int k = (int)(oo / L.Np3), n3 = (int)(oo % L.Np3);
    return L.w3t + wide_idx(n3, k);
  }
  return -1;
}

// Parameter blobs an optimizer step works on.  Used by k_adam (gradient read from a blob) and by k_grad_reduce (the
// single-GPU fused form: the element's gradient is the slab / partial sum it has just formed -- one launch and one
// gradient round trip through HBM fewer per network and step).
struct AdamTarget {
  float *p, *m, *v, *target, *blob_T;       // target / blob_T may be null
  float* target_T;                          // T blob of the target net (its W2 planes follow the Polyak update); may be null
  AdamConsts c;
  const long long* t_dev;                   // device step count (graph replay) or null
  float lr;
  int on;                                   // k_grad_reduce only: 0 = just write the gradient
  long long* bump;                          // k_grad_reduce only: device word incremented by one thread (not t_dev), or null
  int precision;                            // format of the W2 planes kept in blob_T / target_T (write_w2_planes); < 0: no planes
                                            // (dynamics pre-training runs exact fp32: six scattered 2-byte stores per W2 element saved)
};

// Bias corrections of a device-side step count (graph replay), formed ONCE per workgroup in double: thread 0 computes,
// everybody reads after the barrier.  (Every thread evaluating two double pow() per element made the fused
// reduce+Adam kernel 3x slower: 5.5 -> 17 us on the 0.5 M-parameter ensemble nets.)  Call before any early return.
__device__ __forceinline__ void adam_block_consts(const AdamTarget& a, float* sm2) {
  if (a.on && a.t_dev != nullptr) {
    if (threadIdx.x == 0) {
      const double t = (double)a.t_dev[0];
      sm2[0] = (float)((double)a.lr / (1.0 - pow(0.9, t)));
      sm2[1] = (float)sqrt(1.0 - pow(0.999, t));
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void adam_element(const AdamTarget& a, const MobodyMlpLayout& L, long long j, float g,
                                             const float* sm2 = nullptr) {
  AdamConsts c = a.c;
  if (a.t_dev != nullptr) {                         // graph replay: the step count lives in device memory
    if (sm2 != nullptr) {
      c.step_size = sm2[0]; c.bc2_sqrt = sm2[1];
    } else {
      const double t = (double)a.t_dev[0];
      c.step_size = (float)((double)a.lr / (1.0 - pow(0.9, t)));
      c.bc2_sqrt = (float)sqrt(1.0 - pow(0.999, t));
    }
  }
  const float gj = g * c.gscale;
  const float m0 = a.m[j];
  const float mj = m0 + c.w1 * (gj - m0);
  const float vj = c.b2 * a.v[j] + c.w2 * (gj * gj);
  a.m[j] = mj; a.v[j] = vj;
  const float pj = a.p[j] - c.step_size * (mj / (sqrtf(vj) / c.bc2_sqrt + c.eps));
  a.p[j] = pj;
  float tj = 0.f;
  if (a.target != nullptr) { tj = c.tau * pj + c.one_minus_tau * a.target[j]; a.target[j] = tj; }      // update_target :183-187
  if (a.blob_T != nullptr || a.target_T != nullptr) {   // keep the transposes / bf16 planes the kernels stream in sync
    const int mem = (int)(j / L.member_floats);
    const long long o = j - (long long)mem * L.member_floats;
    const long long ti = t_blob_index(L, o);
    if (ti >= 0 && a.blob_T != nullptr) a.blob_T[(long long)mem * L.t_member_floats + ti] = pj;
    if (o >= L.w2 && o < L.b2 && a.precision >= 0) {  // a W2 element (wide storage): its planes (precision < 0: nobody streams them)
      const long long oo = o - L.w2, g = oo >> 2;
      const int k = (int)(g / HID) * 4 + (int)(oo & 3), n = (int)(g % HID);
      if (a.blob_T != nullptr) write_w2_planes(a.blob_T + (long long)mem * L.t_member_floats, L, k, n, pj, a.precision);
      if (a.target_T != nullptr && a.target != nullptr) write_w2_planes(a.target_T + (long long)mem * L.t_member_floats, L, k, n, tj, a.precision);
    }
  }
}

int launch_adam(const AdamTarget& a, const float* g, const MobodyMlpLayout& L, hipStream_t st);      // k_adam over one packed MLP

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
  float* grad;          // may be null when `adam.on` (nobody reads the gradient blob on one GPU)
  LossFinal loss;       // kind 0: none
  AdamTarget adam;      // on = 0: none
};
int launch_grad_reduce(const GradReduceArgs& a, hipStream_t st);
// prec 4: h1 / dz2 point at the saved fp16 planes and e_h1 / e_dz2 at their tile exponents (else null)
int mlp3_weight_grads(const MobodyMlpLayout& L, const float* x, long long x_mstride, const float* h1, const float* h2,
                      const float* dz3, const float* dz2, const float* dz1, long long rows, int nsplit, float* slabs,
                      const float* dbp, int ntiles, float* grad, const LossFinal& loss, const AdamTarget& adam,
                      hipStream_t st, int prec = 0, const int* e_h1 = nullptr, const int* e_dz2 = nullptr);

// split-K factor (workgroups along the row dimension) used for a batch of `rows`: 24 output tiles x nsplit x members
// workgroups should reach ~3 per CU (768), so a one-member net splits twice as fine as a twin net
inline int wgrad_nsplit(long long rows, int members) {
  const int cap = members == 1 ? 32 : members == 2 ? 16 : (32 / members > 1 ? 32 / members : 1);   // 7 members: 4
  long long s = rows / (members == 1 ? 128 : 256);
  if (s < 1) s = 1;
  if (s > cap) s = cap;
  while (rows > 2048LL * 4 * s) s *= 2;             // a wave's row slice spans at most 64 tiles of 32 rows (wgrad_tile_f16)
  return (int)s;
}

}  // namespace mobody
