// Dynamics pre-training: one optimizer step of MOBODYEnsembleDynamics.learn (mobody_dynamics.py:594-653) for the
// 7-member latent ensemble, default configuration (no_vae=0, latent_reward=0, inverse_sep_reward_loss=0, mopo=0):
//
//   loss = transition_loss + (5 if trg else 1) * encoder_loss_coef * encoder_loss + reward_loss * (1 if trg else 0.01)
//   encoder_loss    = 100 * recon + KL(s) + KL(s') + latent consistency      (:300-335)
//   transition_loss = sum_e mean (forward(s, a) - s')^2                          (:337-347)
//   reward_loss     = sum_e mean (r(s, a, fake) - r)^2 + sum_e mean (r(s, a, s') - r)^2,   fake = mean + eps * std_e(mean)
//                     with the gradient flowing through the mean AND the ensemble std   (:349-384)
//
// The reference evaluates the state encoder six times and the transition decoder four times per batch, each time on a
// fresh reparameterisation sample z_k = mu + eps_k * exp(logvar / 2) (mobody_module.py:237-243).  mu / logvar do not
// depend on the sample, so here the three big sub-networks are each ONE batched pass of the fused 3-layer MFMA kernels
// (state encoder on [s; s'] = 2b rows per member, decoder on [z1; z2; z5+za5; z6+za6] = 4b rows, reward head on
// [s,a,fake; s,a,s'] = 2b rows), their backward passes likewise (k_mlp3_bwd with the saved Swish derivatives, k_wgrad,
// k_grad_reduce), and everything at the 16-wide latent level -- the samples, the tiny action encoder (16+A -> 32 -> 16)
// forward and backward, KL, latent consistency, the chain rule through the ensemble std -- is row-wise work in the
// k_pre_* kernels below.  Parameters live in ONE blob (MobodyPretrainLayout: three MobodyMlpLayout regions with 7
// members + the two action encoders), gradients in a blob of the same layout, so data-parallel ranks exchange one
// all-reduce per step.
//
// Roofline: the three MLP passes are MFMA-f32 bound (25 MFLOP per sample, all members, forward + backward);
// the k_pre_* kernels move a few hundred bytes per row.
#include <math.h>
#include <string.h>

#include "common.h"
#include "layers.h"
#include "rng.h"
#include "train.h"

namespace mobody {

constexpr int ZH = 32;                   // hidden width of the action encoder (mobody_module.py:106-107)
constexpr uint32_t STREAM_PRE = 16;      // Philox stream ids 16..22: the seven noise draws of one batch

// ------------------------------------------------------------------------------------------------
// fused 3-layer Swish forward with everything the backward needs (x, h1, h2 and the Swish derivatives d1, d2)
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NTHREADS, 2) void k_mlp3_fwd_train(Mlp3FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  constexpr int TB = 32;
  const int m = blockIdx.y;
  const long long row0 = (long long)blockIdx.x * TB;
  const int rows_here = (int)min((long long)TB, a.rows - row0);
  const bool full = rows_here == TB;
  const float* w1 = a.w1 + m * a.sw1;
  const float* w2 = a.w2 + m * a.sw2;
  const float* w3 = a.w3 + m * a.sw3;
  const float* b3 = a.b3 + m * a.sb3;
  WideRing ring;
  wide_prefetch(w1, a.Kp1, ring);
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
  if (a.save_x != nullptr) {
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    float* sx = a.save_x + m * a.x_ms;
    for (int col = c; col < a.Kp1; col += 32)
      for (int r = r0; r < rows_here; r += NTHREADS >> 5) sx[(row0 + r) * a.Kp1 + col] = Xs[r * LDX + col];
  }
  const long long hoff = ((long long)m * a.rows + row0) * HID;
  float *h1 = a.save_h1 + hoff, *d1 = a.save_d1 + hoff, *h2 = a.save_h2 + hoff, *d2 = a.save_d2 + hoff;
  wide_layer_swish_d<1>(Xs, w1, a.b1 + m * a.sb1, a.Kp1, ring,
                        [=](auto guarded, int row, int col, float y, float d) {
                          if (!decltype(guarded)::value || row < rows_here) { h1[row * HID + col] = y; d1[row * HID + col] = d; }
                        },
                        [&] { wide_prefetch(w2, HID, ring); }, full);
  auto save2 = [=](auto guarded, int row, int col, float y, float d) {
    if (!decltype(guarded)::value || row < rows_here) { h2[row * HID + col] = y; d2[row * HID + col] = d; }
  };
  float* out = a.out + m * a.out_mstride + row0 * a.out_ld;
  auto emit = [&](int row, int col, float v, float bias) {
    if (row < rows_here && col < a.nout) out[row * a.out_ld + col] = v + bias;
  };
  if constexpr (NT > 0) {
    NarrowRegs<NT> br;
    const int mycol = threadIdx.x % (16 * NT);
    float bias;
    wide_layer_swish_d<1>(Xs, w2, a.b2 + m * a.sb2, HID, ring, save2, [&] {
      narrow_prefetch<NT>(w3, 16 * NT, br);
      bias = b3[mycol < a.nout ? mycol : 0];
    }, full);
    narrow_run<TB / 16, NT>(Xs, br, [&](int row, int col, float v) { emit(row, col, v, bias); });
  } else {
    wide_layer_swish_d<1>(Xs, w2, a.b2 + m * a.sb2, HID, ring, save2, [] {}, full);
    narrow_layer(Xs, w3, HID, a.Np3, [&](int row, int col, float v) { emit(row, col, v, b3[col < a.nout ? col : 0]); }, TB);
  }
}

template <int NT>
static int launch_fwd_train_t(const Mlp3FwdArgs& a, int members, hipStream_t st) {
  constexpr size_t lds = (size_t)32 * LDX * sizeof(float);
  ProfScope prof(PROF_MLP_FWD, st);
  hipLaunchKernelGGL((k_mlp3_fwd_train<NT>), dim3((unsigned)cdiv(a.rows, 32), (unsigned)members), dim3(NTHREADS), lds, st, a);
  MB_LAUNCH_OK("k_mlp3_fwd_train");
  return 0;
}
static int launch_fwd_train(const Mlp3FwdArgs& a, int members, hipStream_t st) {
  if (a.rows <= 0) return 0;
  return a.Np3 == 16 ? launch_fwd_train_t<1>(a, members, st) : a.Np3 == 32 ? launch_fwd_train_t<2>(a, members, st)
                                                                            : launch_fwd_train_t<0>(a, members, st);
}

// ------------------------------------------------------------------------------------------------
// row-wise pieces
// ------------------------------------------------------------------------------------------------
struct PreRow {
  int S, A, use_trg, Np3tr;
  long long b;               // rows per member on this rank
  float inv_bg;              // 1 / (rows per member summed over data-parallel ranks)
  float ce, cr;              // (5 if trg else 1) * encoder_loss_coef ; reward-loss factor 1 (trg) / 0.01 (src), times reward_coef
  float ct;                  // weight of transition_loss in the step's loss (1; 0 in a reward-only step, learn_sep_reward :482-519)
  const float* xenc;         // [E][2b][S]   s rows, then s' rows
  const float* act;          // [E][b][A]
  const float* rew;          // [E][b]
  const float* noise6;       // [6][E][b][16] z1(s) z2(s') z3(s) z4(s') z5(s) z6(s), or null -> Philox
  const float* noise7;       // [E][b][S], or null -> Philox
  uint32_t seed, call;
  const long long* call_dev; // optional device word added to `call` (graph replay: the step counter lives on the device)
  const float* za;           // the action encoder in use: [E][za_member_floats]
  long long za_mf, za_w1, za_b1, za_w2, za_b2;
  const float* enc_out;      // [E][2b][32]  mu | logvar
  float* zt;                 // [E][4b][16]  decoder inputs z1 | z2 | z5+za5 | z6+za6
  const float* tr_out;       // [E][4b][S]
  float* dz3tr;              // [E][4b][Np3tr]
  float* xrw;                // [E][2b][2S+A]
  const float* rw_out;       // [E][2b][2]
  float* dz3rw;              // [E][2b][16]
  const float* dfake;        // [E][2b][S]   d loss / d (third input block of the reward head)
  const float* dzt;          // [E][4b][16]
  float* dz3enc;             // [E][2b][32]
  float* zap;                // [chunks][E][za_mf] action-encoder gradient partials
  float* lossp;              // loss partial sums (see PreLossOff)
  float* fnz;                // [E][b][S] the fake-next-state noise draws of this step (k_pre_trans_loss -> k_pre_fake_bwd)
};
// lossp regions (floats): latent chunks [E*nch][2] (latent, kl) | row-tile chunks [nrt][2] (recon, trans) | reward [nrw]
struct PreLossOff { long long lat, rt, rw; int n_lat, n_rt, n_rw; };

__device__ __forceinline__ float pre_noise(const PreRow& a, int k, int e, long long row, int j, int width) {
  const long long i = ((long long)e * a.b + row) * width + j;
  if (k < 6 && a.noise6) return a.noise6[(long long)k * NENS * a.b * 16 + i];
  if (k == 6 && a.noise7) return a.noise7[i];
  const uint32_t call = a.call + (a.call_dev ? (uint32_t)a.call_dev[0] : 0u);
  return rng_normal_at(a.seed, STREAM_PRE + k, call, (uint64_t)i);
}
__device__ __forceinline__ float swishf(float z) { return z / (1.f + expf(-z)); }
__device__ __forceinline__ float dswishf(float z) { const float s = 1.f / (1.f + expf(-z)); return s * (1.f + z * (1.f - s)); }

// ---- latent level: cooperative kernels, one workgroup = LROWS rows of one member ------------------------------------
// Everything between the state encoder's (mu, logvar) and the decoder's inputs is 16..32-wide work per row: six
// reparameterisation samples, three passes through the action encoder (16+A -> 32 -> 16), KL and latent-consistency terms.
// A thread-per-row formulation runs ~10 k dependent instructions on a few thousand threads (measured 46 + 89 us per step
// at b = 256, the two longest kernels of the step); here each phase is spread over the 256 threads of the workgroup
// through LDS and the grid is (ceil(b / 8), 7) workgroups.
constexpr int LROWS = 8, LP = 3 * LROWS;     // rows per workgroup; (pass, row) pairs: pass 0,1,2 = the samples z3, z5, z6

struct LatentLds {                           // float offsets inside the dynamic LDS block
  int sw, es, es2, act, z, eps, sdv, pre, hh, za, dout, dpre, du, total;
};
__host__ __device__ inline LatentLds latent_lds(long long za_mf, int A) {
  LatentLds o; int t = 0;
  auto take = [&](int n) { const int r = t; t += (n + 3) & ~3; return r; };
  o.sw = take((int)za_mf); o.es = take(LROWS * 32); o.es2 = take(LROWS * 32); o.act = take(LROWS * A);
  o.z = take(LP * LATENT); o.eps = take(LP * LATENT); o.sdv = take(LP * LATENT); o.pre = take(LP * ZH); o.hh = take(LP * ZH);
  o.za = take(LP * LATENT); o.dout = take(LP * LATENT); o.dpre = take(LP * ZH); o.du = take(LP * LATENT);
  o.total = t;
  return o;
}

// phases 0..C: weights, (mu, logvar) rows, actions -> samples Z, hidden pre-activations PRE / activations HH, outputs ZA
__device__ __forceinline__ void latent_forward(const PreRow& a, const LatentLds& o, float* sh, int e, long long row0) {
  const int tid = threadIdx.x;
  const float* g = a.za + (long long)e * a.za_mf;
  for (int i = tid; i < (int)a.za_mf; i += 256) sh[o.sw + i] = g[i];
  {
    const int r = tid >> 5, c = tid & 31;                                   // 8 rows x 32 columns = 256 threads
    const long long rc = min(row0 + r, a.b - 1);
    sh[o.es + tid] = a.enc_out[((long long)e * 2 * a.b + rc) * 32 + c];
    sh[o.es2 + tid] = a.enc_out[((long long)e * 2 * a.b + a.b + rc) * 32 + c];
  }
  for (int i = tid; i < LROWS * a.A; i += 256) {
    const int r = i / a.A, c = i - r * a.A;
    sh[o.act + i] = a.act[((long long)e * a.b + min(row0 + r, a.b - 1)) * a.A + c];
  }
  __syncthreads();
  for (int i = tid; i < LP * LATENT; i += 256) {                            // A: z_k = mu + eps_k * exp(logvar / 2)
    const int p = i >> 4, j = i & 15, pass = p / LROWS, r = p - pass * LROWS;
    const int k = pass == 0 ? 2 : pass == 1 ? 4 : 5;
    const float eps = pre_noise(a, k, e, min(row0 + r, a.b - 1), j, 16);
    const float sd = expf(0.5f * sh[o.es + r * 32 + LATENT + j]);
    sh[o.eps + i] = eps; sh[o.sdv + i] = sd; sh[o.z + i] = sh[o.es + r * 32 + j] + eps * sd;
  }
  __syncthreads();
  const float *W1 = sh + o.sw + a.za_w1, *b1 = sh + o.sw + a.za_b1, *W2 = sh + o.sw + a.za_w2, *b2 = sh + o.sw + a.za_b2;
  for (int i = tid; i < LP * ZH; i += 256) {                                // B: hidden layer on [z, a]
    const int p = i >> 5, n = i & 31, r = p % LROWS;
    float acc = b1[n];
#pragma unroll
    for (int j = 0; j < LATENT; ++j) acc += sh[o.z + p * LATENT + j] * W1[j * ZH + n];
    for (int j = 0; j < a.A; ++j) acc += sh[o.act + r * a.A + j] * W1[(LATENT + j) * ZH + n];
    sh[o.pre + i] = acc; sh[o.hh + i] = swishf(acc);
  }
  __syncthreads();
  for (int i = tid; i < LP * LATENT; i += 256) {                            // C: output layer, mu half
    const int p = i >> 4, j = i & 15;
    float acc = b2[j];
#pragma unroll
    for (int n = 0; n < ZH; ++n) acc += sh[o.hh + p * ZH + n] * W2[n * LATENT + j];
    sh[o.za + i] = acc;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_pre_latent_fwd(PreRow a, PreLossOff lo, LatentLds o) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  __shared__ float red[8];
  const int e = blockIdx.y, tid = threadIdx.x;
  const long long row0 = (long long)blockIdx.x * LROWS, b = a.b;
  latent_forward(a, o, sh, e, row0);
  float lat = 0.f, kl = 0.f;
  {                                                      // passes 1, 2 -> decoder inputs (quarters 2, 3): 2 x 8 x 16 = 256 elements
    const int pass = 1 + (tid >> 7), r = (tid >> 4) & 7, j = tid & 15, p = pass * LROWS + r;
    if (row0 + r < b) a.zt[((long long)e * 4 * b + (pass + 1) * b + row0 + r) * LATENT + j] = sh[o.z + p * LATENT + j] + sh[o.za + p * LATENT + j];
  }
  if (tid < 128) {                                       // pass 0 against z4, the no-grad sample of s' (:323-326)
    const int r = tid >> 4, j = tid & 15;
    const long long rc = min(row0 + r, b - 1);
    const float z4 = sh[o.es2 + r * 32 + j] + pre_noise(a, 3, e, rc, j, 16) * expf(0.5f * sh[o.es2 + r * 32 + LATENT + j]);
    const float d = (sh[o.z + r * LATENT + j] + sh[o.za + r * LATENT + j]) - z4;
    lat = row0 + r < b ? d * d : 0.f;
  } else {                                               // z1, z2 -> quarters 0, 1 and the two KL sums (:332-335)
    const int r = (tid - 128) >> 4, j = tid & 15;
    const long long rc = min(row0 + r, b - 1);
    const float mu = sh[o.es + r * 32 + j], lv = sh[o.es + r * 32 + LATENT + j];
    const float mu2 = sh[o.es2 + r * 32 + j], lv2 = sh[o.es2 + r * 32 + LATENT + j];
    const float z1 = mu + pre_noise(a, 0, e, rc, j, 16) * expf(0.5f * lv);
    const float z2 = mu2 + pre_noise(a, 1, e, rc, j, 16) * expf(0.5f * lv2);
    if (row0 + r < b) {
      a.zt[((long long)e * 4 * b + row0 + r) * LATENT + j] = z1;
      a.zt[((long long)e * 4 * b + b + row0 + r) * LATENT + j] = z2;
      kl = -0.5f * (1.f + lv - mu * mu - expf(lv)) + -0.5f * (1.f + lv2 - mu2 * mu2 - expf(lv2));
    }
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) { lat += __shfl_xor(lat, s); kl += __shfl_xor(kl, s); }
  if ((tid & 63) == 0) { red[tid >> 6] = lat; red[4 + (tid >> 6)] = kl; }
  __syncthreads();
  if (tid == 0) {
    float* q = a.lossp + lo.lat + ((long long)e * gridDim.x + blockIdx.x) * 2;
    q[0] = red[0] + red[1]; q[1] = red[6] + red[7];
  }
}

// one thread per (member, row, state dim): reconstruction / transition residuals -> decoder output gradients (quarters 0..2),
// the sampled fake next state (its noise draw is kept in fnz for k_pre_fake_bwd) and the reward head's input rows.  Every load
// of a thread is independent of its stores (one round trip), and a thread draws ONE normal: with a thread per (row, dim)
// looping over the members the seven Philox + Box-Muller evaluations and seven dependent load / store rounds made this
// 17-workgroup kernel 15 us of the step's critical path.
__global__ __launch_bounds__(256) void k_pre_trans_loss(PreRow a, PreLossOff lo) {
  __shared__ float sm[8];
  const int S = a.S, A = a.A, W = 2 * S + A, Np3 = a.Np3tr;
  const long long b = a.b;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = i < NENS * b * S;
  const long long ic = ok ? i : 0;
  const int e = (int)(ic / (b * S));
  const long long row = (ic - (long long)e * b * S) / S;
  const int d = (int)(ic - ((long long)e * b + row) * S);
  const float c_rec = a.ce * 100.f * 2.f * a.inv_bg / (float)S, c_tr = a.ct * 2.f * a.inv_bg / (float)S;
  float m6[NENS], avg = 0.f;
#pragma unroll
  for (int k = 0; k < NENS; ++k) { m6[k] = a.tr_out[((long long)k * 4 * b + 3 * b + row) * S + d]; avg += m6[k]; }
  const float* to = a.tr_out + (long long)e * 4 * b * S;
  const float s = a.xenc[((long long)e * 2 * b + row) * S + d], s2 = a.xenc[((long long)e * 2 * b + b + row) * S + d];
  const float o0 = to[row * S + d], o1 = to[(b + row) * S + d], o2 = to[(2 * b + row) * S + d];
  const float aj0 = a.act[((long long)e * b + row) * A + (d < A ? d : 0)];
  const float nz = pre_noise(a, 6, e, row, d, S);
  avg *= (1.f / NENS);
  float var = 0.f, mine = 0.f;
#pragma unroll
  for (int k = 0; k < NENS; ++k) { const float t = m6[k] - avg; var += t * t; mine = k == e ? m6[k] : mine; }
  const float sd = sqrtf(var * (1.f / (NENS - 1)));                         // torch.std over the ensemble axis, unbiased (:353)
  float rec = 0.f, tr = 0.f;
  if (ok) {
    const float r0 = o0 - s, r1 = o1 - s2, t5 = o2 - s2;
    float* g = a.dz3tr + (long long)e * 4 * b * Np3;
    g[row * Np3 + d] = c_rec * r0; g[(b + row) * Np3 + d] = c_rec * r1; g[(2 * b + row) * Np3 + d] = c_tr * t5;
    for (int c = S + d; c < Np3; c += S) { g[row * Np3 + c] = 0.f; g[(b + row) * Np3 + c] = 0.f; g[(2 * b + row) * Np3 + c] = 0.f; }
    rec = r0 * r0 + r1 * r1; tr = t5 * t5;
    float* x = a.xrw + (long long)e * 2 * b * W;
    x[row * W + d] = s; x[(b + row) * W + d] = s;
    x[row * W + S + A + d] = mine + nz * sd;                                // fake next state (:353)
    x[(b + row) * W + S + A + d] = s2;
    a.fnz[ic] = nz;
    if (d < A) { x[row * W + S + d] = aj0; x[(b + row) * W + S + d] = aj0; }
    for (int j = d + S; j < A; j += S) { const float aj = a.act[((long long)e * b + row) * A + j]; x[row * W + S + j] = aj; x[(b + row) * W + S + j] = aj; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { rec += __shfl_xor(rec, o); tr += __shfl_xor(tr, o); }
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = rec; sm[4 + (threadIdx.x >> 6)] = tr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* p = a.lossp + lo.rt + (long long)blockIdx.x * 2;
    p[0] = (sm[0] + sm[1]) + (sm[2] + sm[3]); p[1] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
  }
}

// one thread per (member, reward-head row): d loss / d r_mu and the squared residuals   (:366-378)
__global__ __launch_bounds__(256) void k_pre_reward_seed(PreRow a, PreLossOff lo) {
  __shared__ float sm[4];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)NENS * 2 * a.b;
  float l = 0.f;
  if (i < n) {
    const long long e = i / (2 * a.b), row = i - e * 2 * a.b;
    const float d = a.rw_out[i * 2] - a.rew[e * a.b + (row < a.b ? row : row - a.b)];
    float* g = a.dz3rw + i * 16;
    g[0] = a.cr * 2.f * a.inv_bg * d;
#pragma unroll
    for (int c = 1; c < 16; ++c) g[c] = 0.f;
    l = d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) a.lossp[lo.rw + blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// chain rule through fake = mean + eps * std_e(mean):  d mean_e = g_e + (sum_j g_j eps_j) (mean_e - avg) / (6 std)
__global__ __launch_bounds__(256) void k_pre_fake_bwd(PreRow a) {
  const int S = a.S, Np3 = a.Np3tr;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.b * S) return;
  const long long row = i / S, b = a.b;
  const int d = (int)(i - row * S);
  float m6[NENS], g[NENS], avg = 0.f, G = 0.f;
#pragma unroll
  for (int e = 0; e < NENS; ++e) {
    m6[e] = a.tr_out[((long long)e * 4 * b + 3 * b + row) * S + d];
    g[e] = a.dfake[((long long)e * 2 * b + row) * S + d];
    avg += m6[e];
    G += g[e] * a.fnz[((long long)e * b + row) * S + d];      // the draw k_pre_trans_loss made (and kept) for this element
  }
  avg *= (1.f / NENS);
  float var = 0.f;
#pragma unroll
  for (int e = 0; e < NENS; ++e) { const float t = m6[e] - avg; var += t * t; }
  const float k = G / ((NENS - 1) * sqrtf(var * (1.f / (NENS - 1))));
#pragma unroll
  for (int e = 0; e < NENS; ++e) {
    float* o = a.dz3tr + ((long long)e * 4 * b + 3 * b + row) * Np3;
    o[d] = g[e] + k * (m6[e] - avg);
    for (int c = S + d; c < Np3; c += S) o[c] = 0.f;
  }
}

// Backward of the latent level (same workgroup shape as k_pre_latent_fwd): recomputes the action-encoder forward, runs its
// backward, reduces the d mu / d logvar contributions of the four uses of the s row (z1, z3, z5, z6) and of z2 for the
// s' row into the state encoder's output gradient, and accumulates the action encoder's weight-gradient partials.
__global__ __launch_bounds__(256) void k_pre_latent_bwd(PreRow a, LatentLds o, int za_in) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int e = blockIdx.y, tid = threadIdx.x;
  const long long row0 = (long long)blockIdx.x * LROWS, b = a.b;
  latent_forward(a, o, sh, e, row0);
  const float *W1 = sh + o.sw + a.za_w1, *W2 = sh + o.sw + a.za_w2;
  for (int i = tid; i < LP * LATENT; i += 256) {                            // D: d loss / d (z_k + za_k)
    const int p = i >> 4, j = i & 15, pass = p / LROWS, r = p - pass * LROWS;
    const bool ok = row0 + r < b;
    const long long rc = min(row0 + r, b - 1);
    float d;
    if (pass == 0) {
      const float z4 = sh[o.es2 + r * 32 + j] + pre_noise(a, 3, e, rc, j, 16) * expf(0.5f * sh[o.es2 + r * 32 + LATENT + j]);
      d = a.ce * 2.f * a.inv_bg / (float)LATENT * ((sh[o.z + i] + sh[o.za + i]) - z4);
    } else {
      d = a.dzt[((long long)e * 4 * b + (pass + 1) * b + rc) * LATENT + j];
    }
    sh[o.dout + i] = ok ? d : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < LP * ZH; i += 256) {                                // E: through the output layer and Swish
    const int p = i >> 5, n = i & 31;
    float dh = 0.f;
#pragma unroll
    for (int j = 0; j < LATENT; ++j) dh += W2[n * LATENT + j] * sh[o.dout + p * LATENT + j];
    sh[o.dpre + i] = dh * dswishf(sh[o.pre + i]);
  }
  __syncthreads();
  for (int i = tid; i < LP * LATENT; i += 256) {                            // F: d z_k = d out + (W1 d pre)[:16]
    const int p = i >> 4, j = i & 15;
    float du = sh[o.dout + i];
#pragma unroll
    for (int n = 0; n < ZH; ++n) du += W1[j * ZH + n] * sh[o.dpre + p * ZH + n];
    sh[o.du + i] = du;
  }
  __syncthreads();
  {                                                                         // G: encoder output gradients, 8 rows x 32 columns
    const int r = tid >> 5, c = tid & 31, j = c & 15;
    const long long rc = min(row0 + r, b - 1);
    const float ckl = a.ce * 0.05f * a.inv_bg / (float)LATENT;
    const float mu = sh[o.es + r * 32 + j], lv = sh[o.es + r * 32 + LATENT + j];
    const float mu2 = sh[o.es2 + r * 32 + j], lv2 = sh[o.es2 + r * 32 + LATENT + j];
    const float dz1 = a.dzt[((long long)e * 4 * b + rc) * LATENT + j], dz2 = a.dzt[((long long)e * 4 * b + b + rc) * LATENT + j];
    float v, v2;
    if (c < LATENT) {
      v = ((sh[o.du + r * LATENT + j] + sh[o.du + (LROWS + r) * LATENT + j]) + sh[o.du + (2 * LROWS + r) * LATENT + j]) + dz1 + ckl * mu;
      v2 = dz2 + ckl * mu2;
    } else {
      float t = 0.f;
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {
        const int q = (pass * LROWS + r) * LATENT + j;
        t += sh[o.du + q] * 0.5f * sh[o.eps + q] * sh[o.sdv + q];
      }
      v = t + dz1 * 0.5f * pre_noise(a, 0, e, rc, j, 16) * expf(0.5f * lv) + ckl * (-0.5f) * (1.f - expf(lv));
      v2 = dz2 * 0.5f * pre_noise(a, 1, e, rc, j, 16) * expf(0.5f * lv2) + ckl * (-0.5f) * (1.f - expf(lv2));
    }
    if (row0 + r < b) {
      a.dz3enc[((long long)e * 2 * b + row0 + r) * 32 + c] = v;
      a.dz3enc[((long long)e * 2 * b + b + row0 + r) * 32 + c] = v2;         // s' row: only z2 and its KL term carry a gradient
    }
  }
  // H: weight-gradient partials of this workgroup (rows past the batch carry zero gradients)
  float* zp = a.zap + ((long long)blockIdx.x * NENS + e) * a.za_mf;
  for (int q = tid; q < (int)a.za_mf; q += 256) {
    float s = 0.f;
    if (q >= a.za_w1 && q < a.za_w1 + za_in * ZH) {
      const int k = (q - (int)a.za_w1) / ZH, n = (q - (int)a.za_w1) % ZH;
      if (k < LATENT) { for (int p = 0; p < LP; ++p) s += sh[o.z + p * LATENT + k] * sh[o.dpre + p * ZH + n]; }
      else { for (int p = 0; p < LP; ++p) s += sh[o.act + (p % LROWS) * a.A + (k - LATENT)] * sh[o.dpre + p * ZH + n]; }
    } else if (q >= a.za_b1 && q < a.za_b1 + ZH) {
      for (int p = 0; p < LP; ++p) s += sh[o.dpre + p * ZH + (q - (int)a.za_b1)];
    } else if (q >= a.za_w2 && q < a.za_w2 + ZH * LATENT) {
      const int n = (q - (int)a.za_w2) / LATENT, j = (q - (int)a.za_w2) % LATENT;
      for (int p = 0; p < LP; ++p) s += sh[o.hh + p * ZH + n] * sh[o.dout + p * LATENT + j];
    } else if (q >= a.za_b2 && q < a.za_b2 + LATENT) {
      for (int p = 0; p < LP; ++p) s += sh[o.dout + p * LATENT + (q - (int)a.za_b2)];
    }
    zp[q] = s;
  }
}

// action-encoder gradient: sum of the workgroup partials in chunk order (deterministic)
__device__ __forceinline__ void za_adam_element(const AdamTarget& a, long long j, float g, const float* sm2) {
  AdamConsts c = a.c;
  if (a.t_dev != nullptr) { c.step_size = sm2[0]; c.bc2_sqrt = sm2[1]; }      // adam_block_consts (train.h)
  const float gj = g * c.gscale;
  const float m0 = a.m[j];
  const float mj = m0 + c.w1 * (gj - m0);
  const float vj = c.b2 * a.v[j] + c.w2 * (gj * gj);
  a.m[j] = mj; a.v[j] = vj;
  a.p[j] = a.p[j] - c.step_size * (mj / (sqrtf(vj) / c.bc2_sqrt + c.eps));
}

__global__ __launch_bounds__(256) void k_pre_za_reduce(const float* zap, int nch, long long n, float* grad, AdamTarget adam) {
  __shared__ float adam_sm[2];
  adam_block_consts(adam, adam_sm);
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  int c = 0;
  for (; c + 4 <= nch; c += 4) {                    // four independent loads in flight, summed in chunk order
    const float* p = zap + (long long)c * n + i;
    const float v0 = p[0], v1 = p[n], v2 = p[2 * n], v3 = p[3 * n];
    s += v0; s += v1; s += v2; s += v3;
  }
  for (; c < nch; ++c) s += zap[(long long)c * n + i];
  if (grad != nullptr) grad[i] = s;
  if (adam.on) za_adam_element(adam, i, s, adam_sm);       // single-GPU form: the reduction applies the optimizer step itself
}

// out[5] = (loss, transition_loss, encoder_loss, recon_loss, kl_loss) as learn() reports them (:630-650); local shares
// of the global means when data parallel.
__global__ __launch_bounds__(256) void k_pre_loss_final(const float* lossp, PreLossOff lo, float inv_bg, int S, float ce,
                                                        float cr, float ct, float* out, float* acc) {
  __shared__ float sm[5][4];
  float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};                // latent, kl, recon, trans, reward
  for (int k = threadIdx.x; k < lo.n_lat; k += 256) { v[0] += lossp[lo.lat + 2 * k]; v[1] += lossp[lo.lat + 2 * k + 1]; }
  for (int k = threadIdx.x; k < lo.n_rt; k += 256) { v[2] += lossp[lo.rt + 2 * k]; v[3] += lossp[lo.rt + 2 * k + 1]; }
  for (int k = threadIdx.x; k < lo.n_rw; k += 256) v[4] += lossp[lo.rw + k];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o);
    if ((threadIdx.x & 63) == 0) sm[q][threadIdx.x >> 6] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[5];
    for (int q = 0; q < 5; ++q) t[q] = (sm[q][0] + sm[q][1]) + (sm[q][2] + sm[q][3]);
    const float lat = t[0] * inv_bg / LATENT, kl = 0.05f * t[1] * inv_bg / LATENT;
    const float recon = t[2] * inv_bg / S, trans = t[3] * inv_bg / S, rl = cr * t[4] * inv_bg;
    const float enc = 100.f * recon + kl + lat;
    out[0] = ct * trans + ce * enc + rl; out[1] = trans; out[2] = enc; out[3] = recon; out[4] = kl;
    if (acc != nullptr) { acc[0] += out[0]; acc[1] += trans; acc[2] += enc; acc[3] += recon; acc[4] += kl; }   // learn()'s running sums (:630-650)
  }
}

// Adam on the two action encoders' region of the blob (the net that had no gradient this step is skipped, as
// torch.optim.Adam skips parameters whose .grad is None)
__global__ __launch_bounds__(256) void k_pre_za_adam(AdamTarget a, const float* g, long long n) {
  __shared__ float adam_sm[2];
  adam_block_consts(a, adam_sm);
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  za_adam_element(a, j, g[j], adam_sm);
}

// bootstrap gather of one batch: member e takes dataset rows idx[e][start + r]   (mobody_dynamics.py:604-612, the
// reference slices pre-gathered [7, n, .] CPU arrays and copies them to the device every batch)
__global__ __launch_bounds__(256) void k_pre_gather(const float* state, const float* action, const float* next_state,
                                                    const float* reward, const int32_t* idx, long long n_idx, long long start,
                                                    const long long* start_dev, long long b, int S, int A, float* xenc, float* act,
                                                    float* rew) {
  if (start_dev != nullptr) start += start_dev[0] * b;      // the device word counts batches
  const int W = 2 * S + A + 1;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)NENS * b * W) return;
  const long long er = gid / W;
  const int c = (int)(gid - er * W);
  const long long e = er / b, r = er - e * b;
  const long long src = idx[e * n_idx + min(start + r, n_idx - 1)];     // (clamped: a replayed graph can never read past the matrix)
  if (c < S) xenc[(e * 2 * b + r) * S + c] = state[src * S + c];
  else if (c < 2 * S) xenc[(e * 2 * b + b + r) * S + (c - S)] = next_state[src * S + (c - S)];
  else if (c < 2 * S + A) act[er * A + (c - 2 * S)] = action[src * A + (c - 2 * S)];
  else rew[er] = reward[src];
}

// per-member holdout losses of validate() (:1113-1140): out[e] = mean_{b,d}(mean_e - s')^2, out[7+e] = mean_b (r_e - r)^2
__global__ __launch_bounds__(256) void k_pre_validate(const float* mean, const float* next_obs, const float* r_mu,
                                                      const float* rew, long long B, int S, float* out) {
  __shared__ float sm[2][4];
  const int e = blockIdx.x;
  float t = 0.f, q = 0.f;
  for (long long i = threadIdx.x; i < B * S; i += 256) { const float d = mean[(long long)e * B * S + i] - next_obs[i]; t += d * d; }
  for (long long i = threadIdx.x; i < B; i += 256) { const float d = r_mu[(long long)e * B + i] - rew[i]; q += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { t += __shfl_xor(t, o); q += __shfl_xor(q, o); }
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = t; sm[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[e] = ((sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3])) / ((float)B * (float)S);
    out[NENS + e] = ((sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3])) / (float)B;
  }
}

// ------------------------------------------------------------------------------------------------
// workspace carving
// ------------------------------------------------------------------------------------------------
struct PreWs {
  float *enc_out, *sx_enc, *h1e, *h2e, *d1e, *d2e;
  float *zt, *h1t, *h2t, *d1t, *d2t, *tr_out;
  float *xrw, *sx_rw, *h1r, *h2r, *d1r, *d2r, *rw_out;
  float *dz3rw, *dz3tr, *dz3enc, *dfake, *dzt, *zap, *lossp, *fnz;
  // backward operands of the three nets (0 reward head, 1 decoder, 2 state encoder): one set EACH, because the weight-gradient
  // GEMM + reduction of nets 0 and 1 run on the side stream while the main stream's backward chain moves on to the next net
  float *dz2[3], *dz1[3], *dbp[3], *slabs[3];
  int *eh1e, *eh1t, *eh1r, *edz2[3];   // f16x2: scale exponents of the 32-row tiles of the h1 / dz2 planes (h1*, dz2 then hold planes)
  PreLossOff lo;
  int nch, nsplit2, nsplit4, ntiles2, ntiles4;
  long long total;
};

static int pre_carve(const MobodyPretrainLayout& L, long long b, float* base, PreWs& w) {
  const long long E = NENS, R2 = 2 * b, R4 = 4 * b;
  const long long R2p = (R2 + 31) & ~31LL, R4p = (R4 + 31) & ~31LL;     // the fp16 planes of h1 / dz2 are padded to whole 32-row tiles
  const int S = L.S, A = L.A;
  long long off = 0;
  auto take = [&](long long n) { float* p = base ? base + off : nullptr; off += (n + 3) & ~3LL; return p; };
  w.enc_out = take(E * R2 * 32); w.sx_enc = take(E * R2 * L.enc.Kp1);
  w.h1e = take(E * R2p * HID); w.h2e = take(E * R2 * HID); w.d1e = take(E * R2 * HID); w.d2e = take(E * R2 * HID);
  w.zt = take(E * R4 * LATENT);
  w.h1t = take(E * R4p * HID); w.h2t = take(E * R4 * HID); w.d1t = take(E * R4 * HID); w.d2t = take(E * R4 * HID);
  w.tr_out = take(E * R4 * S);
  w.xrw = take(E * R2 * (2 * S + A)); w.sx_rw = take(E * R2 * L.rw.Kp1);
  w.h1r = take(E * R2p * HID); w.h2r = take(E * R2 * HID); w.d1r = take(E * R2 * HID); w.d2r = take(E * R2 * HID);
  w.rw_out = take(E * R2 * 2);
  w.dz3rw = take(E * R2 * 16); w.dz3tr = take(E * R4 * L.tr.Np3); w.dz3enc = take(E * R2 * 32);
  const long long rows_of[3] = {R2, R4, R2}, rowsp_of[3] = {R2p, R4p, R2p};
  for (int k = 0; k < 3; ++k) { w.dz2[k] = take(E * rowsp_of[k] * HID); w.dz1[k] = take(E * rows_of[k] * HID); }
  w.dfake = take(E * R2 * S); w.dzt = take(E * R4 * LATENT); w.fnz = take(E * b * S);
  w.ntiles2 = (int)cdiv(R2, 32); w.ntiles4 = (int)cdiv(R4, 32);
  w.eh1e = reinterpret_cast<int*>(take(E * w.ntiles2)); w.eh1r = reinterpret_cast<int*>(take(E * w.ntiles2));
  w.eh1t = reinterpret_cast<int*>(take(E * w.ntiles4));
  for (int k = 0; k < 3; ++k) w.edz2[k] = reinterpret_cast<int*>(take(E * (k == 1 ? w.ntiles4 : w.ntiles2)));
  const long long per = 2 * HID + (L.tr.Np3 > 32 ? L.tr.Np3 : 32);
  for (int k = 0; k < 3; ++k) w.dbp[k] = take((long long)(k == 1 ? w.ntiles4 : w.ntiles2) * E * per);
  w.nsplit2 = wgrad_nsplit(R2, NENS); w.nsplit4 = wgrad_nsplit(R4, NENS);
  w.slabs[0] = take(((L.rw.total_floats + 3) & ~3LL) * w.nsplit2);
  w.slabs[1] = take(((L.tr.total_floats + 3) & ~3LL) * w.nsplit4);
  w.slabs[2] = take(((L.enc.total_floats + 3) & ~3LL) * w.nsplit2);
  w.nch = (int)cdiv(b, LROWS);
  w.zap = take((long long)w.nch * E * L.za_member_floats);
  w.lo.n_lat = (int)(E * w.nch); w.lo.n_rt = (int)cdiv(E * b * S, 256); w.lo.n_rw = (int)cdiv(E * R2, 256);
  w.lo.lat = 0; w.lo.rt = 2LL * w.lo.n_lat; w.lo.rw = w.lo.rt + 2LL * w.lo.n_rt;
  w.lossp = take(w.lo.rw + w.lo.n_rw);
  w.total = off;
  return 0;
}

// blob_T / e1 (f16x2): W2's fp16 planes from the T blob; h1 receives planes + tile exponents instead of fp32 rows
static Mlp3FwdArgs pre_fwd_args(const float* blob, const MobodyMlpLayout& L, const float* src, int n, long long rows, float* out,
                                float* sx, float* h1, float* h2, float* d1, float* d2, const float* blob_T = nullptr,
                                int* e1 = nullptr) {
  Mlp3FwdArgs a{};
  if (e1 != nullptr) {
    const long long r32 = (rows + 31) & ~31LL;
    a.save_h1p = reinterpret_cast<unsigned short*>(h1); a.h1p_plane = r32 * HID; a.h1p_ms = 2 * r32 * HID; a.save_e1 = e1;
    a.w2_planes = reinterpret_cast<const unsigned short*>(blob_T + L.w2p); a.planes_ms = 2 * L.t_member_floats;
    h1 = nullptr;
  }
  a.src[0] = src; a.ld[0] = n; a.n[0] = n; a.src_ms[0] = rows * n;
  a.w1 = blob + L.w1; a.b1 = blob + L.b1; a.w2 = blob + L.w2; a.b2 = blob + L.b2; a.w3 = blob + L.w3; a.b3 = blob + L.b3;
  a.sw1 = a.sb1 = a.sw2 = a.sb2 = a.sw3 = a.sb3 = L.member_floats;
  a.Kp1 = L.Kp1; a.Np3 = L.Np3; a.nout = L.out_dim; a.rows = rows;
  a.out = out; a.out_mstride = rows * L.out_dim; a.out_ld = L.out_dim;
  a.save_x = sx; a.x_ms = rows * L.Kp1; a.save_h1 = h1; a.save_h2 = h2; a.save_d1 = d1; a.save_d2 = d2;
  return a;
}

static Mlp3BwdArgs pre_bwd_args(const MobodyMlpLayout& L, const float* blob_T, const float* dz3, const float* d1, const float* d2,
                                long long rows, float* dz2, float* dz1, float* dbp, int* e2 = nullptr) {
  Mlp3BwdArgs b{};
  if (e2 != nullptr) {                              // f16x2: dz2 leaves as planes + tile exponents, W2^T's planes from the T blob
    const long long r32 = (rows + 31) & ~31LL;
    b.dz2p = reinterpret_cast<unsigned short*>(dz2); b.dz2p_plane = r32 * HID; b.dz2p_ms = 2 * r32 * HID; b.e2_out = e2;
    b.prec = 4; b.w2t_planes = reinterpret_cast<const unsigned short*>(blob_T + L.w2tp); b.planes_ms = 2 * L.t_member_floats;
    dz2 = nullptr;
  }
  b.dz3 = dz3; b.h1 = d1; b.h2 = d2; b.swish = 1; b.wt = blob_T; b.t_mstride = L.t_member_floats;
  b.w3t = L.w3t; b.w2t = L.w2t; b.w1t = L.w1t; b.Np3 = L.Np3; b.Np1t = L.Np1t; b.rows = rows;
  b.dz2 = dz2; b.dz1 = dz1; b.dbp = dbp;
  return b;
}

}  // namespace mobody

using namespace mobody;

extern "C" int mobody_pretrain_layout(int S, int A, MobodyPretrainLayout* out) {
  MB_REQUIRE(out != nullptr, "mobody_pretrain_layout: out is null");
  MB_REQUIRE(S >= 2 && S <= 128 && A >= 1 && A <= 64, "mobody_pretrain_layout: unsupported S=%d A=%d (S in [2,128], A in [1,64])", S, A);
  memset(out, 0, sizeof(*out));
  out->S = S; out->A = A; out->za_in = LATENT + A;
  int rc = mobody_mlp_layout(S, 2 * LATENT, NENS, &out->enc);                       // zs1-3: S -> 256 -> 256 -> 32 (mu | logvar)
  if (!rc) rc = mobody_mlp_layout(LATENT, S, NENS, &out->tr);                       // transition1-3: 16 -> 256 -> 256 -> S
  if (!rc) rc = mobody_mlp_layout(2 * S + A, 2, NENS, &out->rw);                    // reward_model1-3: 2S+A -> 256 -> 256 -> 2
  if (rc) return rc;
  int64_t o = 0;
  out->za_w1 = o; o += (int64_t)out->za_in * ZH;
  out->za_b1 = o; o += ZH;
  out->za_w2 = o; o += ZH * LATENT;                                                 // mu half of za_*2 (the logvar half is unused, :255,270)
  out->za_b2 = o; o += LATENT;
  out->za_member_floats = (o + 3) & ~(int64_t)3;
  int64_t p = 0;
  out->off_enc = p; p += (out->enc.total_floats + 3) & ~(int64_t)3;
  out->off_tr = p; p += (out->tr.total_floats + 3) & ~(int64_t)3;
  out->off_rw = p; p += (out->rw.total_floats + 3) & ~(int64_t)3;
  out->off_za_src = p; p += NENS * out->za_member_floats;
  out->off_za_trg = p; p += NENS * out->za_member_floats;
  out->total_floats = p;
  int64_t t = 0;
  out->t_off_enc = t; t += (out->enc.t_total_floats + 3) & ~(int64_t)3;
  out->t_off_tr = t; t += (out->tr.t_total_floats + 3) & ~(int64_t)3;
  out->t_off_rw = t; t += (out->rw.t_total_floats + 3) & ~(int64_t)3;
  out->t_total_floats = t;
  return 0;
}

static int pre_check_prec(int precision, const char* who) {
  MB_REQUIRE(precision == 0 || precision == 4, "%s: pre-training runs in precision 0 (f32) or 4 (f16x2)", who);
  return 0;
}

extern "C" int mobody_pretrain_transpose(int S, int A, const float* blob, float* blob_T, int precision, void* stream) {
  MobodyPretrainLayout L;
  int rc = mobody_pretrain_layout(S, A, &L);
  if (rc) return rc;
  MB_REQUIRE(blob && blob_T, "mobody_pretrain_transpose: null pointer");
  rc = pre_check_prec(precision, "mobody_pretrain_transpose");
  if (rc) return rc;
  rc = mobody_mlp_transpose(S, 2 * LATENT, NENS, blob + L.off_enc, blob_T + L.t_off_enc, precision, stream);
  if (!rc) rc = mobody_mlp_transpose(LATENT, S, NENS, blob + L.off_tr, blob_T + L.t_off_tr, precision, stream);
  if (!rc) rc = mobody_mlp_transpose(2 * S + A, 2, NENS, blob + L.off_rw, blob_T + L.t_off_rw, precision, stream);
  return rc;
}

extern "C" int64_t mobody_pretrain_workspace(int S, int A, int64_t b) {
  MobodyPretrainLayout L;
  if (mobody_pretrain_layout(S, A, &L) || b < 1) return -1;
  PreWs w;
  pre_carve(L, b, nullptr, w);
  return w.total;
}

extern "C" int mobody_pretrain_gather(const float* state, const float* action, const float* next_state, const float* reward,
                                      const int32_t* idx, int64_t n_idx, int64_t start, const int64_t* start_dev, int64_t b,
                                      int S, int A, float* xenc, float* act, float* rew, void* stream) {
  MB_REQUIRE(b >= 1 && n_idx >= 1 && start >= 0 && (start_dev != nullptr || start + b <= n_idx),
             "mobody_pretrain_gather: rows [%lld, %lld) outside the %lld indices per member", (long long)start,
             (long long)(start + b), (long long)n_idx);
  MB_REQUIRE(state && action && next_state && reward && idx && xenc && act && rew, "mobody_pretrain_gather: null pointer");
  const long long n = (long long)NENS * b * (2 * S + A + 1);
  hipLaunchKernelGGL(k_pre_gather, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), state, action, next_state, reward,
                     idx, (long long)n_idx, (long long)start, (const long long*)start_dev, (long long)b, S, A, xenc, act, rew);
  MB_LAUNCH_OK("k_pre_gather");
  return 0;
}

namespace mobody {
// torch.optim.Adam scalar bookkeeping in double (same forms as train.hip's adam_target); t_dev: device step count
static AdamTarget pre_adam_target(float* p, float* pT, float* m, float* v, int64_t t, const int64_t* t_dev, float lr,
                                  float grad_scale, int precision = 0) {
  const double tt = t_dev ? 1.0 : (double)t;
  const double bc1 = 1.0 - pow(0.9, tt), bc2 = 1.0 - pow(0.999, tt);
  AdamTarget a{};
  a.p = p; a.m = m; a.v = v; a.blob_T = pT; a.target = nullptr;
  a.c.w1 = (float)(1.0 - 0.9); a.c.b2 = (float)0.999; a.c.w2 = (float)(1.0 - 0.999);
  a.c.step_size = (float)((double)lr / bc1); a.c.bc2_sqrt = (float)sqrt(bc2); a.c.eps = 1e-8f;
  a.c.tau = -1.f; a.c.one_minus_tau = 0.f; a.c.gscale = grad_scale;
  a.t_dev = (const long long*)t_dev; a.lr = lr; a.on = 1;
  a.precision = precision == 4 ? 4 : -1;          // exact fp32 never reads the W2 planes of its T blob; f16x2 keeps them current
  return a;
}
// Side stream of the step.  After a net's backward kernel its weight-gradient GEMM and the gradient reduction (with the fused
// Adam step) feed nothing later in the step, while the main chain -- 13 dependent launches of 17 .. 224 workgroups that leave
// most of the 256 CUs idle -- still has the next net's backward to run: the reward head's and the decoder's (~26 us each) go to
// a library-owned second stream between two events and rejoin before the loss is finalised.  Under stream capture the event
// pair pulls the side stream into the caller's graph as a parallel branch.  PRE_SIDE_STREAM=0 builds the single-stream order.
#ifndef PRE_SIDE_STREAM
#define PRE_SIDE_STREAM 1
#endif
struct PreSide { hipStream_t s; hipEvent_t fork[3], join; };
static int pre_side(PreSide** out) {
  static PreSide side[16];
  static bool have[16] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return fail(MOBODY_E_LAUNCH, "pre-training: hipGetDevice failed");
  if (!have[dev]) {
    PreSide& p = side[dev];
    if (hipStreamCreateWithFlags(&p.s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&p.fork[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p.fork[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p.fork[2], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p.join, hipEventDisableTiming) != hipSuccess)
      return fail(MOBODY_E_LAUNCH, "pre-training: could not create the side stream");
    have[dev] = true;
  }
  *out = &side[dev];
  return 0;
}

struct PreOpt {             // fused optimizer step (single GPU): Adam state and step counts; on = 0 -> gradients only
  int on;
  float *blob, *blob_T, *m, *v;
  int64_t t_main, t_za;
  const int64_t* t_dev;     // device {t_main, t_za} (graph replay) or null
  float lr;
};
}  // namespace mobody

static int pretrain_impl(int S, int A, int64_t b, int64_t b_global, int use_trg, float encoder_loss_coef,
                         const float* blob, const float* blob_T, const float* xenc, const float* act,
                         const float* rew, const float* noise6, const float* noise7, uint32_t seed, uint32_t call,
                         const int64_t* call_dev, float* grad, const PreOpt& opt, float* loss_out, float* loss_acc,
                         float* workspace, int precision, void* stream, float transition_coef = 1.f, float reward_coef = 1.f) {
  MobodyPretrainLayout L;
  int rc = mobody_pretrain_layout(S, A, &L);
  if (rc) return rc;
  rc = pre_check_prec(precision, "mobody_pretrain");
  if (rc) return rc;
  const bool f16 = precision == 4;
  MB_REQUIRE(b >= 1 && b_global >= b, "mobody_pretrain: need 1 <= b <= b_global");
  MB_REQUIRE(blob && blob_T && xenc && act && rew && (grad || opt.on) && loss_out && workspace, "mobody_pretrain: null pointer");
  auto region_adam = [&](int64_t off, int64_t toff) {
    if (!opt.on) return AdamTarget{};
    return pre_adam_target(opt.blob + off, opt.blob_T + toff, opt.m + off, opt.v + off, opt.t_main, opt.t_dev, opt.lr, 1.f, precision);
  };
  auto gptr = [&](int64_t off) { return grad ? grad + off : nullptr; };
  PreWs w;
  pre_carve(L, b, workspace, w);
  hipStream_t st = as_stream(stream);
  const long long R2 = 2 * b, R4 = 4 * b;
  static bool once = false;
  if (!once) {
    rc = allow_big_lds(k_mlp3_fwd_train<0>, 160 * 1024);
    if (!rc) rc = allow_big_lds(k_mlp3_fwd_train<1>, 160 * 1024);
    if (!rc) rc = allow_big_lds(k_mlp3_fwd_train<2>, 160 * 1024);
    if (rc) return rc;
    once = true;
  }
  PreRow r{};
  r.S = S; r.A = A; r.use_trg = use_trg; r.Np3tr = L.tr.Np3; r.b = b; r.inv_bg = 1.f / (float)b_global;
  r.ce = (use_trg ? 5.f : 1.f) * encoder_loss_coef; r.cr = (use_trg ? 1.f : 0.01f) * reward_coef; r.ct = transition_coef;
  r.xenc = xenc; r.act = act; r.rew = rew; r.noise6 = noise6; r.noise7 = noise7; r.seed = seed; r.call = call;
  r.call_dev = (const long long*)call_dev;
  r.za = blob + (use_trg ? L.off_za_trg : L.off_za_src);
  r.za_mf = L.za_member_floats; r.za_w1 = L.za_w1; r.za_b1 = L.za_b1; r.za_w2 = L.za_w2; r.za_b2 = L.za_b2;
  r.enc_out = w.enc_out; r.zt = w.zt; r.tr_out = w.tr_out; r.dz3tr = w.dz3tr; r.xrw = w.xrw; r.rw_out = w.rw_out;
  r.dz3rw = w.dz3rw; r.dfake = w.dfake; r.dzt = w.dzt; r.dz3enc = w.dz3enc; r.zap = w.zap; r.lossp = w.lossp; r.fnz = w.fnz;
  const float *Penc = blob + L.off_enc, *Ptr = blob + L.off_tr, *Prw = blob + L.off_rw;
  const float *Tenc = blob_T + L.t_off_enc, *Ttr = blob_T + L.t_off_tr, *Trw = blob_T + L.t_off_rw;

  // f16x2: the 256 x 256 layers of the three nets on the split core (mlp_fwd_bf.hip with derivative saves)
  auto forward = [&](const Mlp3FwdArgs& fa) {
    return f16 ? launch_mlp3_fwd_bf(fa, NENS, Mlp3FwdArgs{}, 0, ACT_SWISH, 4, st) : launch_fwd_train(fa, NENS, st);
  };
  // ---- forward ----
  rc = forward(pre_fwd_args(Penc, L.enc, xenc, S, R2, w.enc_out, w.sx_enc, w.h1e, w.h2e, w.d1e, w.d2e, Tenc, f16 ? w.eh1e : nullptr));
  if (rc) return rc;
  const LatentLds ll = latent_lds(L.za_member_floats, A);
  hipLaunchKernelGGL(k_pre_latent_fwd, dim3((unsigned)w.nch, NENS), dim3(256), sizeof(float) * (size_t)ll.total, st, r, w.lo, ll);
  MB_LAUNCH_OK("k_pre_latent_fwd");
  rc = forward(pre_fwd_args(Ptr, L.tr, w.zt, LATENT, R4, w.tr_out, nullptr, w.h1t, w.h2t, w.d1t, w.d2t, Ttr, f16 ? w.eh1t : nullptr));
  if (rc) return rc;
  hipLaunchKernelGGL(k_pre_trans_loss, dim3((unsigned)w.lo.n_rt), dim3(256), 0, st, r, w.lo);
  MB_LAUNCH_OK("k_pre_trans_loss");
  rc = forward(pre_fwd_args(Prw, L.rw, w.xrw, 2 * S + A, R2, w.rw_out, w.sx_rw, w.h1r, w.h2r, w.d1r, w.d2r, Trw, f16 ? w.eh1r : nullptr));
  if (rc) return rc;
  PreSide* side = nullptr;
  hipStream_t st2 = st;                              // where the off-chain weight-gradient work goes
  if (PRE_SIDE_STREAM) {
    rc = pre_side(&side);
    if (rc) return rc;
    st2 = side->s;
  }
  auto fork = [&](int k) {                           // side stream: wait for everything enqueued on the main stream so far
    if (!PRE_SIDE_STREAM) return 0;
    if (hipEventRecord(side->fork[k], st) != hipSuccess || hipStreamWaitEvent(st2, side->fork[k], 0) != hipSuccess)
      return fail(MOBODY_E_LAUNCH, "pre-training: fork onto the side stream failed");
    return 0;
  };
  // ---- backward: reward head (its input gradient feeds the decoder's fourth quarter) ----
  hipLaunchKernelGGL(k_pre_reward_seed, dim3((unsigned)w.lo.n_rw), dim3(256), 0, st, r, w.lo);
  MB_LAUNCH_OK("k_pre_reward_seed");
  {
    Mlp3BwdArgs bw = pre_bwd_args(L.rw, Trw, w.dz3rw, w.d1r, w.d2r, R2, w.dz2[0], w.dz1[0], w.dbp[0], f16 ? w.edz2[0] : nullptr);
    bw.dx = w.dfake; bw.dx_c0 = S + A; bw.dx_n = S;
    rc = launch_mlp3_bwd(bw, NENS, true, 32, st);
    if (!rc) rc = fork(0);
    if (rc) return rc;
    rc = mlp3_weight_grads(L.rw, w.sx_rw, R2 * L.rw.Kp1, w.h1r, w.h2r, w.dz3rw, w.dz2[0], w.dz1[0], R2, w.nsplit2, w.slabs[0], w.dbp[0],
                           w.ntiles2, gptr(L.off_rw), LossFinal{}, region_adam(L.off_rw, L.t_off_rw), st2, precision,
                           f16 ? w.eh1r : nullptr, f16 ? w.edz2[0] : nullptr);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_pre_fake_bwd, dim3((unsigned)cdiv(b * S, 256)), dim3(256), 0, st, r);
  MB_LAUNCH_OK("k_pre_fake_bwd");
  // ---- decoder ----
  {
    Mlp3BwdArgs bw = pre_bwd_args(L.tr, Ttr, w.dz3tr, w.d1t, w.d2t, R4, w.dz2[1], w.dz1[1], w.dbp[1], f16 ? w.edz2[1] : nullptr);
    bw.dx = w.dzt; bw.dx_c0 = 0; bw.dx_n = LATENT;
    rc = launch_mlp3_bwd(bw, NENS, true, 32, st);
    if (!rc) rc = fork(1);
    if (rc) return rc;
    rc = mlp3_weight_grads(L.tr, w.zt, R4 * LATENT, w.h1t, w.h2t, w.dz3tr, w.dz2[1], w.dz1[1], R4, w.nsplit4, w.slabs[1], w.dbp[1],
                           w.ntiles4, gptr(L.off_tr), LossFinal{}, region_adam(L.off_tr, L.t_off_tr), st2, precision,
                           f16 ? w.eh1t : nullptr, f16 ? w.edz2[1] : nullptr);
    if (rc) return rc;
  }
  // ---- latent level + action encoder ----
  hipLaunchKernelGGL(k_pre_latent_bwd, dim3((unsigned)w.nch, NENS), dim3(256), sizeof(float) * (size_t)ll.total, st, r, ll, (int)L.za_in);
  MB_LAUNCH_OK("k_pre_latent_bwd");
  {
    const long long n = (long long)NENS * L.za_member_floats;
    const int64_t oz = use_trg ? L.off_za_trg : L.off_za_src;
    AdamTarget za{};
    if (opt.on) za = pre_adam_target(opt.blob + oz, nullptr, opt.m + oz, opt.v + oz, opt.t_za, opt.t_dev ? opt.t_dev + 1 : nullptr, opt.lr, 1.f);
    rc = fork(2);                                    // (the action encoder's reduction + Adam feed nothing later either)
    if (rc) return rc;
    hipLaunchKernelGGL(k_pre_za_reduce, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st2, w.zap, w.nch, n, gptr(oz), za);
    MB_LAUNCH_OK("k_pre_za_reduce");
  }
  // ---- state encoder ----
  {
    Mlp3BwdArgs bw = pre_bwd_args(L.enc, Tenc, w.dz3enc, w.d1e, w.d2e, R2, w.dz2[2], w.dz1[2], w.dbp[2], f16 ? w.edz2[2] : nullptr);
    rc = launch_mlp3_bwd(bw, NENS, false, 32, st);
    if (rc) return rc;
    rc = mlp3_weight_grads(L.enc, w.sx_enc, R2 * L.enc.Kp1, w.h1e, w.h2e, w.dz3enc, w.dz2[2], w.dz1[2], R2, w.nsplit2, w.slabs[2], w.dbp[2],
                           w.ntiles2, gptr(L.off_enc), LossFinal{}, region_adam(L.off_enc, L.t_off_enc), st, precision,
                           f16 ? w.eh1e : nullptr, f16 ? w.edz2[2] : nullptr);
    if (rc) return rc;
  }
  if (PRE_SIDE_STREAM) {                             // the step ends when both streams have: the main one waits for the side one
    if (hipEventRecord(side->join, st2) != hipSuccess || hipStreamWaitEvent(st, side->join, 0) != hipSuccess)
      return fail(MOBODY_E_LAUNCH, "pre-training: join of the side stream failed");
  }
  hipLaunchKernelGGL(k_pre_loss_final, dim3(1), dim3(256), 0, st, w.lossp, w.lo, r.inv_bg, S, r.ce, r.cr, r.ct, loss_out, loss_acc);
  MB_LAUNCH_OK("k_pre_loss_final");
  return 0;
}

extern "C" int mobody_pretrain_grads(int S, int A, int64_t b, int64_t b_global, int use_trg, float encoder_loss_coef,
                                     const float* blob, const float* blob_T, const float* xenc, const float* act,
                                     const float* rew, const float* noise6, const float* noise7, uint32_t seed, uint32_t call,
                                     float* grad, float* loss_out, float* workspace, int precision, float transition_coef,
                                     float reward_coef, void* stream) {
  MB_REQUIRE(grad, "mobody_pretrain_grads: grad is null");
  return pretrain_impl(S, A, b, b_global, use_trg, encoder_loss_coef, blob, blob_T, xenc, act, rew, noise6, noise7, seed, call,
                       nullptr, grad, PreOpt{}, loss_out, nullptr, workspace, precision, stream, transition_coef, reward_coef);
}

extern "C" int mobody_pretrain_update(int S, int A, int64_t b, int use_trg, float encoder_loss_coef, float* blob, float* blob_T,
                                      const float* xenc, const float* act, const float* rew, const float* noise6,
                                      const float* noise7, uint32_t seed, uint32_t call, const int64_t* call_dev, float* m,
                                      float* v, int64_t t_main, int64_t t_za, const int64_t* t_dev, float lr, float* loss_out,
                                      float* loss_acc, float* workspace, int precision, void* stream) {
  MB_REQUIRE(m && v, "mobody_pretrain_update: null pointer");
  MB_REQUIRE(t_dev != nullptr || (t_main >= 1 && t_za >= 1), "mobody_pretrain_update: step counts are 1-based");
  PreOpt o{1, blob, blob_T, m, v, t_main, t_za, t_dev, lr};
  return pretrain_impl(S, A, b, b, use_trg, encoder_loss_coef, blob, blob_T, xenc, act, rew, noise6, noise7, seed, call, call_dev,
                       nullptr, o, loss_out, loss_acc, workspace, precision, stream);
}

extern "C" int mobody_pretrain_adam(int S, int A, int use_trg, float* blob, float* blob_T, const float* grad, float* m,
                                    float* v, int64_t t_main, int64_t t_za, float lr, float grad_scale, int precision,
                                    int net_mask, int64_t t_rw, void* stream) {
  MobodyPretrainLayout L;
  int rc = mobody_pretrain_layout(S, A, &L);
  if (rc) return rc;
  rc = pre_check_prec(precision, "mobody_pretrain_adam");
  if (rc) return rc;
  MB_REQUIRE(blob && blob_T && grad && m && v, "mobody_pretrain_adam: null pointer");
  MB_REQUIRE(t_main >= 1 && t_za >= 1 && (!(net_mask & 4) || t_rw >= 1), "mobody_pretrain_adam: step counts are 1-based");
  hipStream_t st = as_stream(stream);
  const MobodyMlpLayout* nets[3] = {&L.enc, &L.tr, &L.rw};
  const int64_t offs[3] = {L.off_enc, L.off_tr, L.off_rw}, toffs[3] = {L.t_off_enc, L.t_off_tr, L.t_off_rw};
  for (int k = 0; k < 3; ++k) {
    if (!((net_mask >> k) & 1)) continue;              // a net whose .grad is None in the reference: Adam skips it, its count stays
    const AdamTarget a = pre_adam_target(blob + offs[k], blob_T + toffs[k], m + offs[k], v + offs[k], k == 2 ? t_rw : t_main, nullptr, lr,
                                         grad_scale, precision);
    rc = launch_adam(a, grad + offs[k], *nets[k], st);
    if (rc) return rc;
  }
  const int64_t oz = use_trg ? L.off_za_trg : L.off_za_src;
  const long long nz = (long long)NENS * L.za_member_floats;
  const AdamTarget a = pre_adam_target(blob + oz, nullptr, m + oz, v + oz, t_za, nullptr, lr, grad_scale);
  hipLaunchKernelGGL(k_pre_za_adam, dim3((unsigned)cdiv(nz, 256)), dim3(256), 0, st, a, grad + oz, nz);
  MB_LAUNCH_OK("k_pre_za_adam");
  return 0;
}

extern "C" int mobody_pretrain_za_adam(int S, int A, int use_trg, float* blob, const float* grad, float* m, float* v,
                                       int64_t t_za, float lr, float grad_scale, void* stream) {
  MobodyPretrainLayout L;
  int rc = mobody_pretrain_layout(S, A, &L);
  if (rc) return rc;
  MB_REQUIRE(blob && grad && m && v && t_za >= 1, "mobody_pretrain_za_adam: null pointer or step count < 1");
  const int64_t oz = use_trg ? L.off_za_trg : L.off_za_src;
  const long long nz = (long long)NENS * L.za_member_floats;
  const AdamTarget a = pre_adam_target(blob + oz, nullptr, m + oz, v + oz, t_za, nullptr, lr, grad_scale);
  hipLaunchKernelGGL(k_pre_za_adam, dim3((unsigned)cdiv(nz, 256)), dim3(256), 0, as_stream(stream), a, grad + oz, nz);
  MB_LAUNCH_OK("k_pre_za_adam");
  return 0;
}

extern "C" int64_t mobody_dyn_validate_workspace(int S, int A, int64_t B) {
  (void)A;
  return (int64_t)NENS * B * S + (int64_t)NENS * B;
}

extern "C" int mobody_dyn_validate(const float* dyn_blob, int S, int A, const float* obs, const float* act,
                                   const float* next_obs, const float* rew, int64_t B, int use_trg, float* out,
                                   float* workspace, void* stream) {
  MobodyDynLayout L;
  int rc = mobody_dyn_layout(S, A, &L);
  if (rc) return rc;
  MB_REQUIRE(B >= 1, "mobody_dyn_validate: B < 1");
  MB_REQUIRE(dyn_blob && obs && act && next_obs && rew && out && workspace, "mobody_dyn_validate: null pointer");
  float* mean = workspace;
  float* r_mu = workspace + (int64_t)NENS * B * S;
  rc = mobody_dyn_forward(dyn_blob, nullptr, 0, S, A, obs, act, B, use_trg, mean, stream);       // inference mode: z = mu (:1126-1129), exact fp32
  if (rc) return rc;
  // reward head of member e on [s, a, mean_e]   (:1137: encode_reward(obs.repeat(7), act.repeat(7), mean))
  Mlp3FwdArgs m{};
  m.src[0] = obs; m.ld[0] = S; m.n[0] = S;
  m.src[1] = act; m.ld[1] = A; m.n[1] = A;
  m.src[2] = mean; m.ld[2] = S; m.n[2] = S; m.src_ms[2] = B * S;
  const MobodyLayer &l1 = L.layer[MOBODY_DL_RW1], &l2 = L.layer[MOBODY_DL_RW2], &l3 = L.layer[MOBODY_DL_RW3];
  m.w1 = dyn_blob + l1.w_off; m.b1 = dyn_blob + l1.b_off; m.sw1 = (long long)l1.Kp * l1.Np; m.sb1 = l1.Np;
  m.w2 = dyn_blob + l2.w_off; m.b2 = dyn_blob + l2.b_off; m.sw2 = (long long)l2.Kp * l2.Np; m.sb2 = l2.Np;
  m.w3 = dyn_blob + l3.w_off; m.b3 = dyn_blob + l3.b_off; m.sw3 = (long long)l3.Kp * l3.Np; m.sb3 = l3.Np;
  m.Kp1 = l1.Kp; m.Np3 = l3.Np; m.nout = 1; m.rows = B;
  m.out = r_mu; m.out_mstride = B; m.out_ld = 1;
  m.out_mode = 0; m.max_action = 1.f;
  rc = launch_mlp3_fwd(m, NENS, ACT_SWISH, as_stream(stream));
  if (rc) return rc;
  hipLaunchKernelGGL(k_pre_validate, dim3(NENS), dim3(256), 0, as_stream(stream), mean, next_obs, r_mu, rew, (long long)B, S, out);
  MB_LAUNCH_OK("k_pre_validate");
  return 0;
}
