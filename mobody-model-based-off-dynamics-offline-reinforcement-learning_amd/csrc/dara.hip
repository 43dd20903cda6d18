// DARA domain-classifier pieces used inside MOBODY (algo/offline_offline/mobody.py:11-33 Classifier,
// :146-181 update_classifier, :354-381 one-off warm-up + reward penalty), plus the generic
// "gradients of one packed MLP from dz3 and saved activations" entry the classifier update is built from.
//
// Reference quirks reproduced: the heads output softmax PROBABILITIES, the loss applies
// cross_entropy (log_softmax) to those probabilities, and the penalty applies softmax to them once more.
// Row-wise kernels: HBM streaming over a few floats per row.
#include "common.h"
#include "layers.h"
#include "rng.h"
#include "train.h"

namespace mobody {

constexpr uint32_t STREAM_CLS_SAS = 4, STREAM_CLS_SA = 5;

struct DaraInArgs {
  const float *s, *a, *s2;
  const float *noise_sas, *noise_sa;     // explicit unit normals [N][2S+A], [N][S+A] or null -> device generator
  long long N;
  int S, A;
  float std;
  uint32_t seed, call;
  float *x_sas, *x_sa;                   // [N][2S+A], [N][S+A]
};

// classifier inputs with additive Gaussian noise  (Classifier.forward with_noise=True, mobody.py:19-31)
__global__ __launch_bounds__(256) void k_dara_inputs(DaraInArgs a) {
  const int W3 = 2 * a.S + a.A, W2 = a.S + a.A;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= a.N * W3) return;
  const long long row = gid / W3;
  const int c = (int)(gid - row * W3);
  const float v = c < a.S ? a.s[row * a.S + c] : (c < W2 ? a.a[row * a.A + (c - a.S)] : a.s2[row * a.S + (c - W2)]);
  float e = 0.f;
  if (a.std != 0.f) e = a.noise_sas ? a.noise_sas[gid] : rng_normal_at(a.seed, STREAM_CLS_SAS, a.call, (uint64_t)gid);
  a.x_sas[gid] = v + e * a.std;
  if (c < W2) {
    const long long g2 = row * W2 + c;
    float e2 = 0.f;
    if (a.std != 0.f) e2 = a.noise_sa ? a.noise_sa[g2] : rng_normal_at(a.seed, STREAM_CLS_SA, a.call, (uint64_t)g2);
    a.x_sa[g2] = v + e2 * a.std;
  }
}

__device__ __forceinline__ void softmax2(float z0, float z1, float& p0, float& p1) {
  const float m = fmaxf(z0, z1);
  const float e0 = expf(z0 - m), e1 = expf(z1 - m);
  const float inv = 1.f / (e0 + e1);
  p0 = e0 * inv; p1 = e1 * inv;
}

// loss = CE(softmax(z)) with CE's own log_softmax (double softmax); returns -log q[label] and dL/dz
__device__ __forceinline__ float ce_on_probs(float z0, float z1, int label, float scale, float& dz0, float& dz1) {
  float p0, p1, q0, q1;
  softmax2(z0, z1, p0, p1);                   // head output (torch.nn.Softmax, :25,31)
  softmax2(p0, p1, q0, q1);                   // F.cross_entropy's softmax over the probabilities (:169-170)
  const float g0 = q0 - (label == 0 ? 1.f : 0.f), g1 = q1 - (label == 1 ? 1.f : 0.f);      // dL/dp
  const float dot = g0 * p0 + g1 * p1;
  dz0 = scale * p0 * (g0 - dot);               // through the first softmax
  dz1 = scale * p1 * (g1 - dot);
  return -logf(label == 0 ? q0 : q1);
}

struct DaraLossArgs {
  const float *z_sas, *z_sa;        // logits [N][2]
  const int32_t* labels;            // [N] or null -> rows < n_src are label 0, the rest label 1
  long long N, n_src;
  int Np3;
  float *dz_sas, *dz_sa;            // [N][Np3]
  float* lossp;                     // [blocks][2] partial sums (sa, sas)
};

__device__ __forceinline__ float block_sum256(float v, float* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void k_dara_loss(DaraLossArgs a) {
  __shared__ float sm[4];
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float l_sa = 0.f, l_sas = 0.f;
  if (row < a.N) {
    const int label = a.labels ? a.labels[row] : (row < a.n_src ? 0 : 1);
    const float scale = 1.f / (float)a.N;
    float d0, d1;
    l_sas = ce_on_probs(a.z_sas[2 * row], a.z_sas[2 * row + 1], label, scale, d0, d1);
    float* o = a.dz_sas + row * a.Np3;
    o[0] = d0; o[1] = d1;
    for (int k = 2; k < a.Np3; ++k) o[k] = 0.f;
    l_sa = ce_on_probs(a.z_sa[2 * row], a.z_sa[2 * row + 1], label, scale, d0, d1);
    o = a.dz_sa + row * a.Np3;
    o[0] = d0; o[1] = d1;
    for (int k = 2; k < a.Np3; ++k) o[k] = 0.f;
  }
  l_sa = block_sum256(l_sa, sm);
  l_sas = block_sum256(l_sas, sm);
  if (threadIdx.x == 0) { a.lossp[2 * blockIdx.x] = l_sa; a.lossp[2 * blockIdx.x + 1] = l_sas; }
}

__global__ __launch_bounds__(256) void k_dara_loss_final(const float* lossp, int nparts, long long N, float* loss_out) {
  __shared__ float sm[4];
  float s0 = 0.f, s1 = 0.f;
  for (int k = threadIdx.x; k < nparts; k += blockDim.x) { s0 += lossp[2 * k]; s1 += lossp[2 * k + 1]; }
  s0 = block_sum256(s0, sm);
  s1 = block_sum256(s1, sm);
  if (threadIdx.x == 0) { loss_out[0] = s0 / (float)N; loss_out[1] = s1 / (float)N; }     // (loss_sa, loss_sas)
}

// reward += coef * clamp(log p~_sas[1] - log p~_sa[1] - log p~_sas[0] + log p~_sa[0], -10, 10),
// p~ = softmax(head probabilities) + 1e-10                                      (mobody.py:373-379)
__global__ __launch_bounds__(256) void k_dara_penalty(const float* z_sas, const float* z_sa, long long n, float coef,
                                                      float* reward, float* delta_out) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  float p0, p1, s0, s1, t0, t1, u0, u1;
  softmax2(z_sas[2 * row], z_sas[2 * row + 1], p0, p1);
  softmax2(p0, p1, s0, s1);
  softmax2(z_sa[2 * row], z_sa[2 * row + 1], t0, t1);
  softmax2(t0, t1, u0, u1);
  float d = logf(s1 + 1e-10f) - logf(u1 + 1e-10f) - logf(s0 + 1e-10f) + logf(u0 + 1e-10f);
  d = fminf(fmaxf(d, -10.f), 10.f);
  if (delta_out) delta_out[row] = d;
  if (reward) reward[row] += coef * d;
}

// ---- generic MLP gradient (dz3 + saved activations -> gradient blob) ----
struct BwdWs { float *dz2, *dz1, *dbp, *slabs; long long slab_stride, total; int nsplit, ntiles, tile_rows; };

static void carve_bwd(const MobodyMlpLayout& L, long long rows, float* base, BwdWs& w) {
  long long off = 0;
  auto take = [&](long long n) { float* p = base ? base + off : nullptr; off += (n + 3) & ~3LL; return p; };
  w.dz2 = take((long long)L.members * rows * HID);
  w.dz1 = take((long long)L.members * rows * HID);
  w.tile_rows = pick_tile_rows(rows, L.members);
  w.ntiles = (int)cdiv(rows, w.tile_rows);
  w.nsplit = wgrad_nsplit(rows, L.members);
  w.dbp = take((long long)w.ntiles * L.members * (2 * HID + L.Np3));
  w.slab_stride = (L.total_floats + 3) & ~3LL;
  w.slabs = take(w.slab_stride * w.nsplit);
  w.total = off;
}

}  // namespace mobody
using namespace mobody;

extern "C" int64_t mobody_mlp3_backward_workspace(int in_dim, int out_dim, int members, int64_t rows) {
  MobodyMlpLayout L;
  if (mobody_mlp_layout(in_dim, out_dim, members, &L) || rows < 1) return -1;
  BwdWs w;
  carve_bwd(L, rows, nullptr, w);
  return w.total;
}

extern "C" int mobody_mlp3_backward(const float* blob_T, int in_dim, int out_dim, int members, const float* dz3,
                                    const float* x, const float* h1, const float* h2, int64_t rows, float* grad,
                                    float* workspace, void* stream) {
  MobodyMlpLayout L;
  int rc = mobody_mlp_layout(in_dim, out_dim, members, &L);
  if (rc) return rc;
  MB_REQUIRE(rows >= 1, "mobody_mlp3_backward: rows < 1");
  MB_REQUIRE(blob_T && dz3 && x && h1 && h2 && grad && workspace, "mobody_mlp3_backward: null pointer");
  BwdWs w;
  carve_bwd(L, rows, workspace, w);
  hipStream_t st = as_stream(stream);
  Mlp3BwdArgs b{};
  b.dz3 = dz3; b.h1 = h1; b.h2 = h2; b.wt = blob_T; b.t_mstride = L.t_member_floats;
  b.w3t = L.w3t; b.w2t = L.w2t; b.w1t = L.w1t; b.Np3 = L.Np3; b.Np1t = L.Np1t; b.rows = rows;
  b.dz2 = w.dz2; b.dz1 = w.dz1; b.dbp = w.dbp;
  rc = launch_mlp3_bwd(b, members, false, w.tile_rows, st);
  if (rc) return rc;
  WgradArgs g{};
  g.rows = rows; g.slabs = w.slabs; g.slab_stride = w.slab_stride; g.out_mstride = L.member_floats;
  g.nsplit = w.nsplit; g.members = L.members;
  const long long hs = rows * HID;
  g.job[0] = WgradJob{h1, hs, HID, HID, w.dz2, hs, HID, HID, L.w2, HID, HID, HID, 0, 1, 0, 0};
  g.job[1] = WgradJob{x, 0, L.Kp1, L.Kp1, w.dz1, hs, HID, HID, L.w1, HID, L.Kp1, HID, 0, 1, 0, 0};
  g.job[2] = WgradJob{dz3, rows * L.Np3, L.Np3, L.Np3, h2, hs, HID, HID, L.w3, L.Np3, L.Np3, HID, 1, 0, 0, 0};
  rc = launch_wgrad(g, st);
  if (rc) return rc;
  GradReduceArgs r{L, w.slabs, w.slab_stride, w.nsplit, w.dbp, w.ntiles, grad};
  return launch_grad_reduce(r, st);
}

extern "C" int mobody_dara_inputs(const float* s, const float* a, const float* s2, int64_t N, int S, int A, float std,
                                  const float* noise_sas, const float* noise_sa, uint32_t seed, uint32_t call,
                                  float* x_sas, float* x_sa, void* stream) {
  MB_REQUIRE(N >= 0 && S >= 1 && A >= 1, "mobody_dara_inputs: bad sizes");
  if (N == 0) return 0;
  MB_REQUIRE(s && a && s2 && x_sas && x_sa, "mobody_dara_inputs: null pointer");
  DaraInArgs d{s, a, s2, noise_sas, noise_sa, N, S, A, std, seed, call, x_sas, x_sa};
  hipLaunchKernelGGL(k_dara_inputs, dim3((unsigned)cdiv(N * (2 * S + A), 256)), dim3(256), 0, as_stream(stream), d);
  MB_LAUNCH_OK("k_dara_inputs");
  return 0;
}

extern "C" int mobody_dara_loss_grad(const float* z_sas, const float* z_sa, const int32_t* labels, int64_t N,
                                     int64_t n_src, float* dz_sas, float* dz_sa, float* loss_out, float* lossp_ws,
                                     void* stream) {
  MB_REQUIRE(N >= 1, "mobody_dara_loss_grad: N < 1");
  MB_REQUIRE(z_sas && z_sa && dz_sas && dz_sa && loss_out && lossp_ws, "mobody_dara_loss_grad: null pointer");
  MobodyMlpLayout L;
  mobody_mlp_layout(1, 2, 1, &L);
  const int nb = (int)cdiv(N, 256);
  DaraLossArgs d{z_sas, z_sa, labels, N, n_src, L.Np3, dz_sas, dz_sa, lossp_ws};
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_dara_loss, dim3(nb), dim3(256), 0, st, d);
  MB_LAUNCH_OK("k_dara_loss");
  hipLaunchKernelGGL(k_dara_loss_final, dim3(1), dim3(256), 0, st, lossp_ws, nb, (long long)N, loss_out);
  MB_LAUNCH_OK("k_dara_loss_final");
  return 0;
}

extern "C" int mobody_dara_penalty(const float* z_sas, const float* z_sa, int64_t n, float coef, float* reward,
                                   float* delta_out, void* stream) {
  MB_REQUIRE(n >= 0, "mobody_dara_penalty: n < 0");
  if (n == 0) return 0;
  MB_REQUIRE(z_sas && z_sa && (reward || delta_out), "mobody_dara_penalty: null pointer");
  hipLaunchKernelGGL(k_dara_penalty, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), z_sas, z_sa,
                     (long long)n, coef, reward, delta_out);
  MB_LAUNCH_OK("k_dara_penalty");
  return 0;
}
