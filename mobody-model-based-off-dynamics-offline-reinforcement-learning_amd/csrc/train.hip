// MOBODY gradient step: critic (twin-Q TD regression) and actor (Q-scaled policy gradient +
// Q-weighted behaviour cloning), assembled from the fused MLP forward/backward kernels (which also form the row-wise
// quantities in their prologues, mlp_bwd.hip) plus the few row-wise kernels below.  Reference: algo/offline_offline/mobody.py:189-208 (critic),
// :246-276 (bc_loss), :314-345 (update_policy), :540-578 (order of updates), :183-187 (Polyak).
//
// Row-wise kernels are HBM-streaming (a few floats per row); the scalar reductions are
// single-workgroup and deterministic (no atomics), so N-GPU == 1-GPU to rounding.
#include <math.h>

#include "common.h"
#include "layers.h"
#include "train.h"

namespace mobody {

// ------------------------------------------------------------------------------------------------
// workspace carving
// ------------------------------------------------------------------------------------------------
struct TrainWs {
  uint32_t *mq1, *mq2, *ma1, *ma2;      // ReLU sign words of the twin-Q / actor hidden layers
  int *eh1q, *eh1a, *edz2;              // f16 mode: scale exponents of the 32-row tiles of the h1 / dz2 planes
  float *pin;            // pi(s') of the critic phase (pi holds pi(s) for the actor phase)
  float *pi, *qt, *q, *qb, *xq, *h1q, *h2q, *xa, *h1a, *h2a, *dz3q, *dz2, *dz1, *dz3a, *dxa, *bcw, *dbp, *slabs, *lossp;
  long long total;
  int nsplit_q, nsplit_a, ntiles, tile_rows;
  MobodyMlpLayout Lq, La;
};

static int carve(const MobodyTrainDims& d, float* base, TrainWs& w) {
  int rc = mobody_mlp_layout(d.S + d.A, 1, 2, &w.Lq);
  if (rc) return rc;
  rc = mobody_mlp_layout(d.S, d.A, 1, &w.La);
  if (rc) return rc;
  const long long N = d.N, Nt = d.Nt;
  const long long N32 = (N + 31) & ~31LL;             // h1 / dz2 hold fp16 planes of whole 32-row tiles in the f16 mode (same bytes)
  long long off = 0;
  auto take = [&](long long n) { float* p = base ? base + off : nullptr; off += (n + 3) & ~3LL; return p; };
  w.pi = take(N * d.A);
  w.pin = take(N * d.A);
  w.qt = take(2 * N);
  w.q = take(2 * N);
  w.qb = take(2 * Nt);
  w.xq = take(N * w.Lq.Kp1);
  w.h1q = take(2 * N32 * HID);
  w.h2q = take(2 * N * HID);
  w.xa = take(N * w.La.Kp1);
  w.h1a = take(N32 * HID);
  w.h2a = take(N * HID);
  const long long mw = cdiv(N, 32) * HID;
  w.mq1 = (uint32_t*)take(2 * mw); w.mq2 = (uint32_t*)take(2 * mw);
  w.ma1 = (uint32_t*)take(mw); w.ma2 = (uint32_t*)take(mw);
  w.eh1q = (int*)take(2 * (N32 / 32)); w.eh1a = (int*)take(N32 / 32); w.edz2 = (int*)take(2 * (N32 / 32));
  w.dz3q = take(2 * N * w.Lq.Np3);
  w.dz2 = take(2 * N32 * HID);
  w.dz1 = take(2 * N * HID);
  w.dz3a = take(N * w.La.Np3);
  w.dxa = take(2 * N * d.A);
  w.bcw = take(Nt > 0 ? Nt : 1);
  w.lossp = take(2 * cdiv(N, 32));                // per (row tile, member) / per-tile pairs, tiles of >= 32 rows
  w.tile_rows = pick_tile_rows(N, 1);             // one value for both nets: the bias partials are per row tile
  w.ntiles = (int)cdiv(N, w.tile_rows);
  w.nsplit_q = wgrad_nsplit(N, 2);
  w.nsplit_a = wgrad_nsplit(N, 1);
  const long long per_q = 2 * HID + w.Lq.Np3, per_a = 2 * HID + w.La.Np3;
  w.dbp = take((long long)w.ntiles * (2 * per_q > per_a ? 2 * per_q : per_a));
  {                                               // one slab area, used by the critic's and then the actor's gradients
    const long long sq = ((w.Lq.total_floats + 3) & ~3LL) * w.nsplit_q, sa = ((w.La.total_floats + 3) & ~3LL) * w.nsplit_a;
    w.slabs = take(sq > sa ? sq : sa);
  }
  w.total = off;
  return 0;
}

static int check_dims(const MobodyTrainDims* d, const char* who) {
  MB_REQUIRE(d != nullptr, "%s: dims is null", who);
  MB_REQUIRE(d->N >= 1 && d->Nt >= 0 && d->Nt <= d->N, "%s: need 1 <= N and 0 <= Nt <= N (N=%lld Nt=%lld)", who,
             (long long)d->N, (long long)d->Nt);
  MB_REQUIRE(d->N_global >= d->N && d->Nt_global >= d->Nt, "%s: global row counts smaller than local", who);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// row-wise kernels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sm) {      // blockDim.x multiple of 64, <= 1024
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int k = 0; k < nw; ++k) s += sm[k];       // same order in every thread: deterministic
  return s;
}

// stats[0] = sum_rows |min(Q1,Q2)(s,pi(s))|, stats[1] = sum_{rows<Nt} |min(Q1,Q2)(s_t,a_t)|   (:318, :259)
__global__ __launch_bounds__(1024) void k_actor_stats(const float* qp, const float* qb, long long N, long long Nt,
                                                      float* stats) {
  __shared__ float sm[16];
  constexpr int U = 4;                            // 2U independent loads in flight; each thread still adds its rows in
  float s0 = 0.f, s1 = 0.f;                       // increasing order, so the sums do not depend on U
  for (long long base = threadIdx.x; base < N; base += U * 1024) {
    float a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long r = base + u * 1024; const long long rc = r < N ? r : 0; a[u] = qp[rc]; b[u] = qp[N + rc]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 1024 < N) s0 += fabsf(fminf(a[u], b[u]));
  }
  for (long long base = threadIdx.x; base < Nt; base += U * 1024) {
    float a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long r = base + u * 1024; const long long rc = r < Nt ? r : 0; a[u] = qb[rc]; b[u] = qb[Nt + rc]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 1024 < Nt) s1 += fabsf(fminf(a[u], b[u]));
  }
  s0 = block_sum(s0, sm);
  s1 = block_sum(s1, sm);
  if (threadIdx.x == 0) { stats[0] = s0; stats[1] = s1; }
}

// expectile regression of V towards min target-Q (update_v_function, mobody.py:231-242; asymmetric_l2_loss :85-86):
// adv = min(Qt1,Qt2)(s,a) - V(s); L_V = mean(|0.7 - 1[adv<0]| * adv^2); dz3[row][0] = dL/dV
__global__ __launch_bounds__(256) void k_v_loss(const float* qt, const float* v, long long N, float invNg, int Np3,
                                                float* dz3, float* lossp) {
  __shared__ float sm[4];
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float l = 0.f;
  if (row < N) {
    const float adv = fminf(qt[row], qt[N + row]) - v[row];
    const float w = fabsf(0.7f - (adv < 0.f ? 1.f : 0.f));
    l = w * adv * adv;
    float* o = dz3 + row * Np3;
    o[0] = -2.f * w * adv * invNg;
    for (int c = 1; c < Np3; ++c) o[c] = 0.f;
  }
  l = block_sum(l, sm);
  if (threadIdx.x == 0) lossp[blockIdx.x] = l;
}
__global__ __launch_bounds__(256) void k_sum_scale(const float* parts, int n, float scale, float* out) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int k = threadIdx.x; k < n; k += blockDim.x) s += parts[k];
  s = block_sum(s, sm);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// ------------------------------------------------------------------------------------------------
// Adam + Polyak, transposes
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam(AdamTarget a, const float* g, long long n, MobodyMlpLayout L) {
  __shared__ float adam_sm[2];
  adam_block_consts(a, adam_sm);
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  adam_element(a, L, j, g[j], adam_sm);
}

int launch_adam(const AdamTarget& a, const float* g, const MobodyMlpLayout& L, hipStream_t st) {
  hipLaunchKernelGGL(k_adam, dim3((unsigned)cdiv(L.total_floats, 256)), dim3(256), 0, st, a, g, (long long)L.total_floats, L);
  MB_LAUNCH_OK("k_adam");
  return 0;
}

// W1 and W2 (and W3T, W2T of the T blob) are 256 columns wide and stored K-interleaved (tile.h wide_idx);
// W3 and W1T are narrow and row major.
__global__ __launch_bounds__(256) void k_mlp_transpose(MobodyMlpLayout L, const float* blob, float* bt, int precision) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= L.t_total_floats) return;
  const int m = (int)(j / L.t_member_floats);
  const long long o = j - (long long)m * L.t_member_floats;
  const float* src = blob + (long long)m * L.member_floats;
  if (o >= L.w2p) {                                // bf16 planes of W2 / W2^T: one thread per float slot writes nothing here;
    if (o >= L.w2p + HID * HID) return;            // the first 65536 threads of the region each split one weight
    const int e = (int)(o - L.w2p), k = e / HID, n = e % HID;
    write_w2_planes(bt + (long long)m * L.t_member_floats, L, k, n, src[L.w2 + wide_idx(k, n)], precision);
    if (precision == 4) {                          // the unused third plane slot: defined contents (a rebuilt T blob compares equal)
      short* tm = reinterpret_cast<short*>(bt + (long long)m * L.t_member_floats);
      tm[2 * L.w2p + bf_plane_idx(2, k, n)] = 0; tm[2 * L.w2tp + bf_plane_idx(2, n, k)] = 0;
    }
    return;
  }
  float val;
  if (o < L.w1t) {                               // wide regions of the T blob: decode (row kk, column c) of storage slot o
    const bool is3 = o < L.w2t;
    const long long oo = is3 ? o : o - L.w2t;
    const long long g = oo >> 2;
    const int kk = (int)(g / HID) * 4 + (int)(oo & 3), c = (int)(g % HID);
    // W3T[n3 = kk][k = c] = W3[k][n3] (narrow, row major);  W2T[n = kk][k = c] = W2[k][n] (wide)
    val = is3 ? src[L.w3 + (long long)c * L.Np3 + kk] : src[L.w2 + wide_idx(c, kk)];
  } else {                                       // W1T[n][k] = W1[k][n] (narrow [256][Np1t]), zero for k >= Kp1
    const long long oo = o - L.w1t;
    const int n = (int)(oo / L.Np1t), k = (int)(oo % L.Np1t);
    val = k < L.Kp1 ? src[L.w1 + wide_idx(k, n)] : 0.f;
  }
  bt[j] = val;
}

// ------------------------------------------------------------------------------------------------
// helpers to launch the fused MLP pieces on a packed blob
// ------------------------------------------------------------------------------------------------
// e1 != null (f16 mode): sh1 receives the layer-1 activations as fp16 planes + tile exponents instead of fp32 rows
static Mlp3FwdArgs fwd_args(const float* blob, const MobodyMlpLayout& L, const float* s0, int n0, const float* s1, int n1,
                            long long rows, float* out, int out_mode, float max_action, float* sx, float* sh1,
                            float* sh2, uint32_t* m1 = nullptr, uint32_t* m2 = nullptr, const float* blob_T = nullptr,
                            int* e1 = nullptr) {
  Mlp3FwdArgs a{};
  if (e1 != nullptr && sh1 != nullptr) {
    const long long r32 = (rows + 31) & ~31LL;
    a.save_h1p = reinterpret_cast<unsigned short*>(sh1); a.h1p_plane = r32 * HID; a.h1p_ms = 2 * r32 * HID; a.save_e1 = e1;
    sh1 = nullptr;
  }
  if (blob_T != nullptr) {                          // split-precision modes stream W2's bf16 planes from the T blob
    a.w2_planes = reinterpret_cast<const unsigned short*>(blob_T + L.w2p);
    a.planes_ms = 2 * L.t_member_floats;
  }
  a.src[0] = s0; a.ld[0] = n0; a.n[0] = n0;
  a.src[1] = s1; a.ld[1] = n1; a.n[1] = s1 ? n1 : 0;
  a.w1 = blob + L.w1; a.b1 = blob + L.b1; a.w2 = blob + L.w2; a.b2 = blob + L.b2; a.w3 = blob + L.w3; a.b3 = blob + L.b3;
  a.sw1 = a.sb1 = a.sw2 = a.sb2 = a.sw3 = a.sb3 = L.member_floats;
  a.Kp1 = L.Kp1; a.Np3 = L.Np3; a.nout = L.out_dim; a.rows = rows;
  a.out = out; a.out_mstride = rows * L.out_dim; a.out_ld = L.out_dim;
  a.save_x = sx; a.save_h1 = sh1; a.save_h2 = sh2; a.mask1 = m1; a.mask2 = m2;
  a.out_mode = out_mode; a.max_action = max_action;
  return a;
}

// two ReLU nets in one launch at the requested precision (0 = exact fp32 MFMA)
static int fwd_pair(const Mlp3FwdArgs& a, int ma, const Mlp3FwdArgs& b, int mb, int prec, hipStream_t st) {
  if (prec == 0) return launch_mlp3_fwd_pair(a, ma, b, mb, st);
  // (one launch also when the two output layers differ in width, 16 | 32 columns, in the f16x2 mode; else one launch per net)
  const bool mixed_ok = prec == 4 && ((a.Np3 == 16 && b.Np3 == 32) || (a.Np3 == 32 && b.Np3 == 16));
  if (a.rows > 0 && b.rows > 0 && a.Np3 != b.Np3 && !mixed_ok) {
    int rc = launch_mlp3_fwd_bf(a, ma, Mlp3FwdArgs{}, 0, ACT_RELU, prec, st);
    return rc ? rc : launch_mlp3_fwd_bf(b, mb, Mlp3FwdArgs{}, 0, ACT_RELU, prec, st);
  }
  return launch_mlp3_fwd_bf(a, ma, b, mb, ACT_RELU, prec, st);
}
static int fwd_one(const Mlp3FwdArgs& a, int ma, int prec, hipStream_t st) {
  return prec == 0 ? launch_mlp3_fwd(a, ma, ACT_RELU, st) : launch_mlp3_fwd_bf(a, ma, Mlp3FwdArgs{}, 0, ACT_RELU, prec, st);
}
static int check_prec(const MobodyHyper* h, const char* who, bool have_planes) {
  MB_REQUIRE(h->precision >= 0 && h->precision <= 4, "%s: precision must be 0 (f32), 1 (bf16), 2 (bf16x2), 3 (bf16x3) or 4 (f16x2)", who);
  MB_REQUIRE(h->precision == 0 || have_planes, "%s: the split-precision modes need the T blobs (bf16 planes) of every net", who);
  return 0;
}

// weight gradients of one packed MLP: one merged split-K launch + the deterministic reduction
static int weight_grads(const MobodyMlpLayout& L, const float* x, const float* h1, const float* h2, const float* dz3,
                        const float* dz2, const float* dz1, long long rows, const TrainWs& w, float* grad,
                        const LossFinal& loss, const AdamTarget& adam, hipStream_t st, int prec = 0, const int* e_h1 = nullptr) {
  return mlp3_weight_grads(L, x, 0, h1, h2, dz3, dz2, dz1, rows, L.members == 1 ? w.nsplit_a : w.nsplit_q, w.slabs, w.dbp,
                           w.ntiles, grad, loss, adam, st, prec, prec == 4 ? e_h1 : nullptr, prec == 4 ? w.edz2 : nullptr);
}

// e2 != null (f16 mode): dz2 receives fp16 planes + tile exponents instead of fp32 rows
static Mlp3BwdArgs bwd_args(const MobodyMlpLayout& L, const float* blob_T, const float* dz3, const float* h1,
                            const float* h2, long long rows, float* dz2, float* dz1, float* dbp,
                            const uint32_t* m1 = nullptr, const uint32_t* m2 = nullptr, int prec = 0, int* e2 = nullptr) {
  Mlp3BwdArgs b{};
  if (prec == 4 && e2 != nullptr && dz2 != nullptr) {
    const long long r32 = (rows + 31) & ~31LL;
    b.dz2p = reinterpret_cast<unsigned short*>(dz2); b.dz2p_plane = r32 * HID; b.dz2p_ms = 2 * r32 * HID; b.e2_out = e2;
    dz2 = nullptr;
  }
  b.prec = prec; b.w2t_planes = reinterpret_cast<const unsigned short*>(blob_T + L.w2tp); b.planes_ms = 2 * L.t_member_floats;
  b.dz3 = dz3; b.h1 = h1; b.h2 = h2; b.m1 = m1; b.m2 = m2; b.wt = blob_T; b.t_mstride = L.t_member_floats;
  b.w3t = L.w3t; b.w2t = L.w2t; b.w1t = L.w1t; b.Np3 = L.Np3; b.Np1t = L.Np1t; b.rows = rows;
  b.dz2 = dz2; b.dz1 = dz1; b.dbp = dbp;
  return b;
}

}  // namespace mobody

using namespace mobody;

extern "C" int64_t mobody_train_workspace(const MobodyTrainDims* d) {
  if (check_dims(d, "mobody_train_workspace")) return -1;
  TrainWs w;
  if (carve(*d, nullptr, w)) return -1;
  return w.total;
}

static AdamTarget adam_target(float* blob, float* blob_T, float* m, float* v, float* target, int64_t t, const int64_t* t_dev,
                              float lr, float tau, float grad_scale, int precision);

static int critic_impl(const MobodyTrainDims* d, const MobodyHyper* h, const float* actor_blob, const float* actor_blob_T,
                       const float* q_blob, const float* q_blob_T, const float* qtarg_blob, const float* qtarg_blob_T,
                       const float* state, const float* action,
                       const float* next_state, const float* reward, const float* not_done, const float* q_next,
                       float* grad_q, const AdamTarget& adam, float* loss_out, float* workspace, int policy_forward,
                       void* stream, int phase = 0) {
  // phase: 0 the whole step; 1 only its forwards (nothing of them reads `reward`); 2 only the backward, weight gradients and
  // reduction / optimizer step -- the caller may let another stream finish rewriting `reward` (penalty_type 'par': an ensemble
  // step on the source rows) between the two
  int rc = check_dims(d, "mobody_critic_step");
  if (rc) return rc;
  MB_REQUIRE(h && q_blob && q_blob_T && state && action && reward && not_done && (grad_q || adam.on) && loss_out && workspace,
             "mobody_critic_step: null pointer");
  MB_REQUIRE(q_next || (actor_blob && qtarg_blob && next_state), "mobody_critic_step: need q_next or actor/target/next_state");
  rc = check_prec(h, "mobody_critic_step", q_next != nullptr || (actor_blob_T && qtarg_blob_T));
  if (rc) return rc;
  const int prec = h->precision;
  const float *aT = prec ? actor_blob_T : nullptr, *qT = prec ? q_blob_T : nullptr, *tT = prec ? qtarg_blob_T : nullptr;
  TrainWs w;
  rc = carve(*d, workspace, w);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const long long N = d->N;
  const int S = d->S, A = d->A;
  // online twin-Q(s, a), activations kept for the backward (:196), together with a' = pi(s') (:191) in one launch
  const Mlp3FwdArgs fq = fwd_args(q_blob, w.Lq, state, S, action, A, N, w.q, 0, 1.f, w.xq, w.h1q, w.h2q, w.mq1, w.mq2, qT,
                                  prec == 4 ? w.eh1q : nullptr);
  if (phase == 2) {
    // forwards already enqueued by the phase-1 call
  } else if (q_next == nullptr) {
    rc = fwd_pair(fq, 2, fwd_args(actor_blob, w.La, next_state, S, nullptr, 0, N, w.pin, 1, h->max_action, nullptr, nullptr, nullptr, nullptr, nullptr, aT), 1, prec, st);
    // target twin-Q(s', a') (:192) -- and, when the caller asks for it, pi(s) of the coming actor phase in the same
    // launch: the actor is not updated in between, and a twin-Q launch alone is 2.5 workgroups per CU where the
    // merged one is 3.75 (the actor phase then opens with Q(s_t,a_t) alone: exactly 2 per CU)
    const Mlp3FwdArgs ft = fwd_args(qtarg_blob, w.Lq, next_state, S, w.pin, A, N, w.qt, 0, 1.f, nullptr, nullptr, nullptr, nullptr, nullptr, tT);
    if (!rc && policy_forward)
      rc = fwd_pair(ft, 2, fwd_args(actor_blob, w.La, state, S, nullptr, 0, N, w.pi, 1, h->max_action, w.xa, w.h1a, w.h2a, w.ma1, w.ma2, aT,
                                    prec == 4 ? w.eh1a : nullptr), 1, prec, st);
    else if (!rc)
      rc = fwd_one(ft, 2, prec, st);
  } else {
    rc = fwd_one(fq, 2, prec, st);                  // q_next = V(s') supplied by the caller (update_q_functions_1, :210-229)
  }
  if (rc || phase == 1) return rc;
  const float invNg = 1.f / (float)d->N_global;
  // TD error -> dz3 in the backward's prologue (mobody.py:190-207), then dz2, dz1 and the bias partials
  Mlp3BwdArgs bq = bwd_args(w.Lq, q_blob_T, w.dz3q, w.h1q, w.h2q, N, w.dz2, w.dz1, w.dbp, w.mq1, w.mq2, prec, w.edz2);
  bq.seed.mode = 1; bq.seed.q = w.q; bq.seed.qt = w.qt; bq.seed.qnext = q_next; bq.seed.r = reward; bq.seed.nd = not_done;
  bq.seed.gamma = h->gamma; bq.seed.inv_ng = invNg; bq.seed.dz3_out = w.dz3q; bq.seed.lossp = w.lossp;
  rc = launch_mlp3_bwd(bq, 2, false, w.tile_rows, st);
  if (rc) return rc;
  LossFinal lf{};                                  // q_loss = mse(q1,y)+mse(q2,y), local share of the global mean
  lf.kind = 1; lf.nparts = 2 * w.ntiles; lf.scale = invNg; lf.parts = w.lossp; lf.out = loss_out;
  return weight_grads(w.Lq, w.xq, w.h1q, w.h2q, w.dz3q, w.dz2, w.dz1, N, w, grad_q, lf, adam, st, prec, w.eh1q);
}

extern "C" int mobody_critic_step(const MobodyTrainDims* d, const MobodyHyper* h, const float* actor_blob,
                                  const float* actor_blob_T, const float* q_blob, const float* q_blob_T,
                                  const float* qtarg_blob, const float* qtarg_blob_T, const float* state,
                                  const float* action, const float* next_state, const float* reward, const float* not_done,
                                  const float* q_next, float* grad_q, float* loss_out, float* workspace, int policy_forward,
                                  void* stream) {
  MB_REQUIRE(grad_q, "mobody_critic_step: grad_q is null");
  return critic_impl(d, h, actor_blob, actor_blob_T, q_blob, q_blob_T, qtarg_blob, qtarg_blob_T, state, action, next_state, reward,
                     not_done, q_next, grad_q, AdamTarget{}, loss_out, workspace, policy_forward, stream);
}

extern "C" int mobody_critic_update(const MobodyTrainDims* d, const MobodyHyper* h, const float* actor_blob,
                                    const float* actor_blob_T, float* q_blob, float* q_blob_T, float* qtarg_blob,
                                    float* qtarg_blob_T, const float* state, const float* action, const float* next_state,
                                    const float* reward, const float* not_done, const float* q_next, float* m, float* v,
                                    int64_t t, const int64_t* t_dev, float lr, float* loss_out, float* workspace,
                                    int policy_forward, int64_t* bump, void* stream) {
  MB_REQUIRE(h && q_blob && q_blob_T && qtarg_blob && m && v, "mobody_critic_update: null pointer");
  MB_REQUIRE(t_dev != nullptr || t >= 1, "mobody_critic_update: step t must be >= 1");
  MB_REQUIRE(bump == nullptr || bump != t_dev, "mobody_critic_update: bump must not be the step word the launch reads");
  AdamTarget at = adam_target(q_blob, q_blob_T, m, v, qtarg_blob, t, t_dev, lr, h->tau, 1.f, h->precision);
  at.target_T = qtarg_blob_T;
  at.bump = (long long*)bump;
  return critic_impl(d, h, actor_blob, actor_blob_T, q_blob, q_blob_T, qtarg_blob, qtarg_blob_T, state, action, next_state, reward,
                     not_done, q_next, nullptr, at, loss_out, workspace, policy_forward, stream);
}

extern "C" int mobody_critic_update_phase(const MobodyTrainDims* d, const MobodyHyper* h, const float* actor_blob,
                                          const float* actor_blob_T, float* q_blob, float* q_blob_T, float* qtarg_blob,
                                          float* qtarg_blob_T, const float* state, const float* action, const float* next_state,
                                          const float* reward, const float* not_done, const float* q_next, float* m, float* v,
                                          int64_t t, const int64_t* t_dev, float lr, float* loss_out, float* workspace,
                                          int policy_forward, int64_t* bump, int phase, void* stream) {
  MB_REQUIRE(h && q_blob && q_blob_T && qtarg_blob && m && v, "mobody_critic_update_phase: null pointer");
  MB_REQUIRE(phase == 1 || phase == 2, "mobody_critic_update_phase: phase is 1 (forwards) or 2 (backward + update)");
  MB_REQUIRE(t_dev != nullptr || t >= 1, "mobody_critic_update_phase: step t must be >= 1");
  MB_REQUIRE(bump == nullptr || bump != t_dev, "mobody_critic_update_phase: bump must not be the step word the launch reads");
  AdamTarget at = adam_target(q_blob, q_blob_T, m, v, qtarg_blob, t, t_dev, lr, h->tau, 1.f, h->precision);
  at.target_T = qtarg_blob_T;
  at.bump = (long long*)bump;
  return critic_impl(d, h, actor_blob, actor_blob_T, q_blob, q_blob_T, qtarg_blob, qtarg_blob_T, state, action, next_state, reward,
                     not_done, q_next, nullptr, at, loss_out, workspace, policy_forward, stream, phase);
}

extern "C" int mobody_actor_forward(const MobodyTrainDims* d, const MobodyHyper* h, const float* actor_blob,
                                    const float* actor_blob_T, const float* q_blob, const float* q_blob_T,
                                    const float* state, const float* action, float* stats, float* workspace, int policy_ready,
                                    void* stream) {
  int rc = check_dims(d, "mobody_actor_forward");
  if (rc) return rc;
  MB_REQUIRE(h && actor_blob && q_blob && state && action && stats && workspace, "mobody_actor_forward: null pointer");
  rc = check_prec(h, "mobody_actor_forward", actor_blob_T && q_blob_T);
  if (rc) return rc;
  const int prec = h->precision;
  const float *aT = prec ? actor_blob_T : nullptr, *qT = prec ? q_blob_T : nullptr;
  TrainWs w;
  rc = carve(*d, workspace, w);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const long long N = d->N, Nt = d->Nt;
  const int S = d->S, A = d->A;
  // Q(s_true, a_true) for the BC weights (:251) and pi(s) on the whole mixed batch (its first Nt rows are
  // pi(s_true), mobody.py:249,315) in one launch -- unless the critic call already left pi(s) in the workspace
  const Mlp3FwdArgs fb = fwd_args(q_blob, w.Lq, state, S, action, A, Nt, w.qb, 0, 1.f, nullptr, nullptr, nullptr, nullptr, nullptr, qT);
  // Q(s, pi(s)) with the freshly updated critic (:316); dQ/da through the frozen net needs only the ReLU signs
  const Mlp3FwdArgs fp = fwd_args(q_blob, w.Lq, state, S, w.pi, A, N, w.q, 0, 1.f, nullptr, nullptr, nullptr, w.mq1, w.mq2, qT);
  static const bool split_q = tune_int("MOBODY_MERGE_ACTOR_Q", 1) == 0;   // tuning aid (diagnostic build)
  if (policy_ready && !split_q) {
    rc = fwd_pair(fb, 2, fp, 2, prec, st);                 // both on the same critic: one launch of N + Nt rows (0.384 -> 0.380 ms/step)
  } else {
    if (policy_ready)
      rc = fwd_one(fb, 2, prec, st);
    else
      rc = fwd_pair(fb, 2, fwd_args(actor_blob, w.La, state, S, nullptr, 0, N, w.pi, 1, h->max_action, w.xa, w.h1a, w.h2a, w.ma1, w.ma2, aT,
                                    prec == 4 ? w.eh1a : nullptr), 1, prec, st);
    if (!rc) rc = fwd_one(fp, 2, prec, st);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(k_actor_stats, dim3(1), dim3(1024), 0, st, w.q, w.qb, N, Nt, stats);
  MB_LAUNCH_OK("k_actor_stats");
  return 0;
}

static int actor_backward_impl(const MobodyTrainDims* d, const MobodyHyper* h, const float* actor_blob,
                               const float* actor_blob_T, const float* q_blob, const float* q_blob_T, const float* state,
                               const float* action, const float* stats, const float* v_true, float* grad_actor,
                               const AdamTarget& adam, float* loss_out, float* workspace, void* stream) {
  int rc = check_dims(d, "mobody_actor_backward");
  if (rc) return rc;
  MB_REQUIRE(h && actor_blob && actor_blob_T && q_blob && q_blob_T && state && action && stats && (grad_actor || adam.on) &&
                 loss_out && workspace, "mobody_actor_backward: null pointer");
  rc = check_prec(h, "mobody_actor_backward", true);
  if (rc) return rc;
  TrainWs w;
  rc = carve(*d, workspace, w);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const long long N = d->N;
  ActorRowArgs ra{};
  ra.qp = w.q; ra.qb = w.qb; ra.stats = stats; ra.pi = w.pi; ra.act = action; ra.dxa = w.dxa; ra.v_true = v_true;
  ra.bcw = w.bcw; ra.N = N; ra.Nt = d->Nt; ra.Ng = d->N_global; ra.Ntg = d->Nt_global > 0 ? d->Nt_global : 1;
  ra.A = d->A; ra.h = *h;
  // dq -> d(action) through the frozen twin-Q (parameters get no gradient, mobody.py:555-556); the prologue forms
  // -p_w/N d min(q1,q2) and the BC weights
  Mlp3BwdArgs bq = bwd_args(w.Lq, q_blob_T, nullptr, nullptr, nullptr, N, nullptr, nullptr, w.dbp, w.mq1, w.mq2, h->precision);
  bq.seed.mode = 2; bq.seed.ar = ra;
  bq.dx = w.dxa; bq.dx_c0 = d->S; bq.dx_n = d->A;
  rc = launch_mlp3_bwd(bq, 2, true, w.tile_rows, st);
  if (rc) return rc;
  // actor: d(pre-tanh) from both members' dx and the BC term in the prologue, then the actor's own backward
  Mlp3BwdArgs ba = bwd_args(w.La, actor_blob_T, w.dz3a, w.h1a, w.h2a, N, w.dz2, w.dz1, w.dbp, w.ma1, w.ma2, h->precision, w.edz2);
  ba.seed.mode = 3; ba.seed.ar = ra; ba.seed.dz3_out = w.dz3a; ba.seed.lossp = w.lossp;
  rc = launch_mlp3_bwd(ba, 1, false, w.tile_rows, st);
  if (rc) return rc;
  LossFinal lf{};                                  // loss_out[0] = p_w*mean(-q) + bc_coef*L_BC, [1] = L_BC (local shares)
  lf.kind = 2; lf.nparts = w.ntiles; lf.scale_q = h->scale_q; lf.weight = h->weight; lf.bc_coef = h->bc_coef;
  lf.ng = (float)ra.Ng; lf.ntg_a = (float)ra.Ntg * (float)ra.A; lf.parts = w.lossp; lf.stats = stats; lf.out = loss_out;
  return weight_grads(w.La, w.xa, w.h1a, w.h2a, w.dz3a, w.dz2, w.dz1, N, w, grad_actor, lf, adam, st, h->precision, w.eh1a);
}

extern "C" int mobody_actor_backward(const MobodyTrainDims* d, const MobodyHyper* h, const float* actor_blob,
                                     const float* actor_blob_T, const float* q_blob, const float* q_blob_T,
                                     const float* state, const float* action, const float* stats, const float* v_true,
                                     float* grad_actor, float* loss_out, float* workspace, void* stream) {
  MB_REQUIRE(grad_actor, "mobody_actor_backward: grad_actor is null");
  return actor_backward_impl(d, h, actor_blob, actor_blob_T, q_blob, q_blob_T, state, action, stats, v_true, grad_actor,
                             AdamTarget{}, loss_out, workspace, stream);
}

extern "C" int mobody_actor_update(const MobodyTrainDims* d, const MobodyHyper* h, float* actor_blob, float* actor_blob_T,
                                   const float* q_blob, const float* q_blob_T, const float* state, const float* action,
                                   const float* stats, const float* v_true, float* m, float* v, int64_t t,
                                   const int64_t* t_dev, float lr, float* loss_out, float* workspace, void* stream) {
  MB_REQUIRE(actor_blob && actor_blob_T && m && v, "mobody_actor_update: null pointer");
  MB_REQUIRE(t_dev != nullptr || t >= 1, "mobody_actor_update: step t must be >= 1");
  return actor_backward_impl(d, h, actor_blob, actor_blob_T, q_blob, q_blob_T, state, action, stats, v_true, nullptr,
                             adam_target(actor_blob, actor_blob_T, m, v, nullptr, t, t_dev, lr, -1.f, 1.f, h ? h->precision : 0), loss_out,
                             workspace, stream);
}

extern "C" int mobody_mlp_transpose(int in_dim, int out_dim, int members, const float* blob, float* blob_T, int precision,
                                    void* stream) {
  MobodyMlpLayout L;
  int rc = mobody_mlp_layout(in_dim, out_dim, members, &L);
  if (rc) return rc;
  MB_REQUIRE(blob && blob_T, "mobody_mlp_transpose: null pointer");
  MB_REQUIRE(precision >= 0 && precision <= 4, "mobody_mlp_transpose: precision must be 0..4");
  hipLaunchKernelGGL(k_mlp_transpose, dim3((unsigned)cdiv(L.t_total_floats, 256)), dim3(256), 0, as_stream(stream), L, blob, blob_T,
                     precision);
  MB_LAUNCH_OK("k_mlp_transpose");
  return 0;
}

// torch.optim.Adam scalar bookkeeping in double, as the reference's host code does
static AdamTarget adam_target(float* blob, float* blob_T, float* m, float* v, float* target, int64_t t, const int64_t* t_dev,
                              float lr, float tau, float grad_scale, int precision) {
  const double tt = t_dev ? 1.0 : (double)t;
  const double bc1 = 1.0 - pow(0.9, tt), bc2 = 1.0 - pow(0.999, tt);
  AdamTarget a{};
  a.p = blob; a.m = m; a.v = v; a.blob_T = blob_T;
  a.target = (target != nullptr && tau >= 0.f) ? target : nullptr;
  a.c.w1 = (float)(1.0 - 0.9); a.c.b2 = (float)0.999; a.c.w2 = (float)(1.0 - 0.999);
  a.c.step_size = (float)((double)lr / bc1); a.c.bc2_sqrt = (float)sqrt(bc2); a.c.eps = 1e-8f;
  a.c.tau = tau; a.c.one_minus_tau = (float)(1.0 - (double)tau); a.c.gscale = grad_scale;
  a.t_dev = (const long long*)t_dev; a.lr = lr; a.on = 1; a.precision = precision;
  return a;
}

static int adam_impl(int in_dim, int out_dim, int members, float* blob, float* blob_T, const float* grad, float* m,
                     float* v, float* target, float* target_T, int64_t t, const int64_t* t_dev, float lr, float tau,
                     float grad_scale, int precision, void* stream) {
  MobodyMlpLayout L;
  int rc = mobody_mlp_layout(in_dim, out_dim, members, &L);
  if (rc) return rc;
  MB_REQUIRE(blob && grad && m && v, "mobody_adam_polyak: null pointer");
  MB_REQUIRE(t_dev != nullptr || t >= 1, "mobody_adam_polyak: step t must be >= 1");
  MB_REQUIRE(precision >= 0 && precision <= 4, "mobody_adam_polyak: precision must be 0..4");
  AdamTarget a = adam_target(blob, blob_T, m, v, target, t, t_dev, lr, tau, grad_scale, precision);
  a.target_T = a.target ? target_T : nullptr;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)cdiv(L.total_floats, 256)), dim3(256), 0, st, a, grad, (long long)L.total_floats, L);
  MB_LAUNCH_OK("k_adam");
  return 0;       // (W1T's zero padding columns k >= Kp1 are written once by mobody_mlp_transpose and never change)
}

extern "C" int mobody_adam_polyak(int in_dim, int out_dim, int members, float* blob, float* blob_T, const float* grad,
                                  float* m, float* v, float* target, float* target_T, int64_t t, float lr, float tau,
                                  float grad_scale, int precision, void* stream) {
  return adam_impl(in_dim, out_dim, members, blob, blob_T, grad, m, v, target, target_T, t, nullptr, lr, tau, grad_scale, precision, stream);
}

extern "C" int mobody_adam_polyak_dev(int in_dim, int out_dim, int members, float* blob, float* blob_T,
                                      const float* grad, float* m, float* v, float* target, float* target_T,
                                      const int64_t* t_dev, float lr, float tau, float grad_scale, int precision, void* stream) {
  MB_REQUIRE(t_dev != nullptr, "mobody_adam_polyak_dev: t_dev is null");
  return adam_impl(in_dim, out_dim, members, blob, blob_T, grad, m, v, target, target_T, 0, t_dev, lr, tau, grad_scale, precision, stream);
}

// ---- PAR reward penalty: r -= coef * mean_d (s'_true - s'_model)^2   (mobody.py:428-434) ----
namespace mobody {
__global__ __launch_bounds__(256) void k_par_penalty(const float* ns_true, const float* ns_model, float* reward, float coef,
                                                     long long n, int S) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  float s = 0.f;
  for (int d = 0; d < S; ++d) { const float e = ns_true[row * S + d] - ns_model[row * S + d]; s += e * e; }
  reward[row] -= coef * (s / (float)S);
}
}  // namespace mobody

extern "C" int mobody_par_penalty(const float* next_state_true, const float* next_state_model, float* reward, float coef,
                                  int64_t n, int S, void* stream) {
  MB_REQUIRE(n >= 0 && S >= 1, "mobody_par_penalty: bad sizes");
  if (n == 0) return 0;
  MB_REQUIRE(next_state_true && next_state_model && reward, "mobody_par_penalty: null pointer");
  hipLaunchKernelGGL(k_par_penalty, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), next_state_true,
                     next_state_model, reward, coef, (long long)n, S);
  MB_LAUNCH_OK("k_par_penalty");
  return 0;
}


// ---- V-function expectile loss (advantage variant): dz3[N][16] and loss_out[0] = local share of L_V ----
extern "C" int mobody_value_loss_grad(const float* qt, const float* v, int64_t N, int64_t N_global, float* dz3,
                                      float* loss_out, float* lossp_ws, void* stream) {
  MB_REQUIRE(N >= 1 && N_global >= N, "mobody_value_loss_grad: bad sizes");
  MB_REQUIRE(qt && v && dz3 && loss_out && lossp_ws, "mobody_value_loss_grad: null pointer");
  const int nb = (int)cdiv(N, 256);
  const float invNg = 1.f / (float)N_global;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_v_loss, dim3(nb), dim3(256), 0, st, qt, v, (long long)N, invNg, 16, dz3, lossp_ws);
  MB_LAUNCH_OK("k_v_loss");
  hipLaunchKernelGGL(k_sum_scale, dim3(1), dim3(256), 0, st, lossp_ws, nb, invNg, loss_out);
  MB_LAUNCH_OK("k_sum_scale");
  return 0;
}
