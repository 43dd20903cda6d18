// Ensemble-dynamics imagined transition (the "rollout kernel" family).
//
//   k_dyn_fwd      mean[e,b,:] = forward_trg/forward_src(obs, act)   mobody_module.py:315-330
//                  one workgroup = 64 rows x ONE member, the 9 ensemble layers
//                  (zs1-3, za1-2, transition1-3) chained through one LDS image;
//                  grid = (ceil(B/64), 7) so even B = 4096 fills the chip (448 WGs).
//   k_dyn_sample   ensemble std, Gaussian sample of the elite member, pairwise-diff
//                  penalty, termination predicate (one thread per (row, state dim), whole rows per
//                  workgroup)                                       mobody_dynamics.py:218-256,
//                                                                    terminal_funs.py:10-113
//   reward head    generic fused MLP forward (Swish) on [s,a,s']     mobody_module.py:295-302
//   k_dyn_finalize reward = mean_e r_mu - coef*penalty               mobody_dynamics.py:236,261-263
//
// Roofline: k_dyn_fwd / reward head are MFMA-f32 bound (3.36 MFLOP per transition at
// S=17,A=6 against 240 algorithmic bytes); k_dyn_sample / k_dyn_finalize are HBM-bound
// streaming kernels over [E,B,S] floats.
#include "common.h"
#include "layers_bf.h"
#include "rng.h"

namespace mobody {

struct DynFwdArgs {
  const float* blob;
  MobodyDynLayout L;
  const float* obs;
  const float* act;
  float* mean;        // [E][B][S]
  long long B;
  int use_trg;
  const unsigned short* planes;   // split-precision modes: 16-bit planes of zs2 | transition2 | reward_model2 (dyn_planes_off)
};

// bf16 planes of the three 256 x 256 ensemble layers: [layer 0..2 = zs2, transition2, reward_model2][member][3 planes][65536]
constexpr long long DYN_PLANE_MEMBER = 3LL * HID * HID;                       // bf16 elements per (layer, member)
__host__ __device__ inline long long dyn_planes_off(int layer, int member) { return ((long long)layer * NENS + member) * DYN_PLANE_MEMBER; }

// NT3: 16-column tiles of the transition head handled by the K-split narrow layer (Np == 16*NT3: 1, 2, 3 or 7), 0 = any width
// (row-split narrow_layer: half of the waves idle on a 32-row tile).
// PM: 0 = exact fp32 MFMA; 1..4 = the two 256 x 256 layers (zs2, transition2) on the split-precision core (tile_bf.h).
template <int MT, int NT3, int PM>
__global__ __launch_bounds__(NTHREADS, 2) void k_dyn_fwd(DynFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  constexpr int TB = 32 * MT;
  constexpr int PMX = PM > 0 ? PM : 1;
  char* Ps = reinterpret_cast<char*>(Xs);
  float* scr = reinterpret_cast<float*>(Ps + split_scr_offset<PMX, TB>());
  BfRing<PMX> bring;
  const int e = blockIdx.y;
  const long long row0 = (long long)blockIdx.x * TB;
  const int rows_here = (int)min((long long)TB, a.B - row0);
  const int S = a.L.S, A = a.L.A;
  const int lane = lane_id(), w = wave_id();
  auto Wp = [&](int l) { return a.blob + a.L.layer[l].w_off + (long long)e * a.L.layer[l].Kp * a.L.layer[l].Np; };
  auto Bp = [&](int l) { return a.blob + a.L.layer[l].b_off + (long long)e * a.L.layer[l].Np; };

  // ---- state encoder: zs = mu-half of zs3(Sw(zs2(Sw(zs1(s)))))   (encode_state :217-225) ----
  // every wide layer's first weight fragments are requested one phase early (see wide_prefetch)
  WideRing ring;
  wide_prefetch(Wp(MOBODY_DL_ZS1), a.L.layer[MOBODY_DL_ZS1].Kp, ring);
  tile_load(Xs, 0, a.obs + row0 * S, S, S, 0, rows_here, TB);
  tile_zero_cols(Xs, S, a.L.layer[MOBODY_DL_ZS1].Kp, TB);
  lds_barrier();
  if constexpr (PM > 0) {
    const s16x8* pz = reinterpret_cast<const s16x8*>(a.planes + dyn_planes_off(0, e));
    const int ez = wide_layer_to_planes<ACT_SWISH, MT, PMX, TB>(Xs, Ps, scr, Wp(MOBODY_DL_ZS1), Bp(MOBODY_DL_ZS1),
                                                                a.L.layer[MOBODY_DL_ZS1].Kp, ring,
                                                                [&] { bf_prefetch<PMX>(pz, bring); });
    bf_layer<ACT_SWISH, MT, PMX, TB>(Xs, Ps, ez, pz, Bp(MOBODY_DL_ZS2), bring, [] {});
  } else {
    wide_layer<ACT_SWISH, MT>(Xs, Wp(MOBODY_DL_ZS1), Bp(MOBODY_DL_ZS1), a.L.layer[MOBODY_DL_ZS1].Kp, ring, NoExtra{},
                              [&] { wide_prefetch(Wp(MOBODY_DL_ZS2), HID, ring); });
    wide_layer<ACT_SWISH, MT>(Xs, Wp(MOBODY_DL_ZS2), Bp(MOBODY_DL_ZS2), HID, ring, NoExtra{}, [] {});
  }

  // From here to the latent sum every wave works on its own 16 rows: no barriers needed (waves without rows idle).
  if (16 * w < TB) {
    const int i = lane & 15, q = lane >> 4;
    float* myrow = Xs + (16 * w + 4 * q) * LDX;       // rows 16w+4q+r, r = 0..3 (C/D map of 16x16 MFMA)
    f32x4 zs[1];
    narrow_gemm<1>(Xs, Wp(MOBODY_DL_ZS3), HID, 16, 0, zs, TB);
    {
      const float b = Bp(MOBODY_DL_ZS3)[i];
#pragma unroll
      for (int r = 0; r < 4; ++r) { zs[0][r] += b; myrow[r * LDX + i] = zs[0][r]; }
    }
    // ---- action encoder on [zs, a]   (encode_trg_action :258-271 / encode_src_action :245-256) ----
    const int la1 = a.use_trg ? MOBODY_DL_ZA_TRG1 : MOBODY_DL_ZA_SRC1;
    const int la2 = a.use_trg ? MOBODY_DL_ZA_TRG2 : MOBODY_DL_ZA_SRC2;
    const int Kza = a.L.layer[la1].Kp;
    for (int idx = lane; idx < 16 * (Kza - LATENT); idx += 64) {
      const int r = idx / (Kza - LATENT), c = idx - r * (Kza - LATENT);
      const int row = 16 * w + r;
      const float v = a.act[(row0 + min(row, rows_here - 1)) * A + min(c, A - 1)];      // unconditional, clamped
      Xs[row * LDX + LATENT + c] = (c < A && row < rows_here) ? v : 0.f;
    }
    f32x4 g[2];
    narrow_gemm<2>(Xs, Wp(la1), Kza, 32, 0, g, TB);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const float b = Bp(la1)[16 * n + i];
#pragma unroll
      for (int r = 0; r < 4; ++r) myrow[r * LDX + 16 * n + i] = activate<ACT_SWISH>(g[n][r] + b);
    }
    f32x4 za[1];
    narrow_gemm<1>(Xs, Wp(la2), 32, 16, 0, za, TB);
    {
      const float b = Bp(la2)[i];
#pragma unroll
      for (int r = 0; r < 4; ++r) myrow[r * LDX + i] = zs[0][r] + za[0][r] + b;     // z_ns = zs + za  (:319,327)
    }
  }
  lds_barrier();

  // ---- transition decoder   (encode_transition :287-293) ----
  wide_prefetch(Wp(MOBODY_DL_TR1), 16, ring);
  const s16x8* pt = PM > 0 ? reinterpret_cast<const s16x8*>(a.planes + dyn_planes_off(1, e)) : nullptr;
  int et = 0;
  if constexpr (PM > 0)
    et = wide_layer_to_planes<ACT_SWISH, MT, PMX, TB>(Xs, Ps, scr, Wp(MOBODY_DL_TR1), Bp(MOBODY_DL_TR1), 16, ring,
                                                      [&] { bf_prefetch<PMX>(pt, bring); });
  else
    wide_layer<ACT_SWISH, MT>(Xs, Wp(MOBODY_DL_TR1), Bp(MOBODY_DL_TR1), 16, ring, NoExtra{},
                              [&] { wide_prefetch(Wp(MOBODY_DL_TR2), HID, ring); });
  const float* b3 = Bp(MOBODY_DL_TR3);
  float* mean = a.mean + ((long long)e * a.B + row0) * S;
  // transition2 at the requested precision; `between` runs after its last MFMA (the output layer's early requests)
  auto layer_tr2 = [&](auto&& between) {
    if constexpr (PM > 0) bf_layer<ACT_SWISH, MT, PMX, TB>(Xs, Ps, et, pt, Bp(MOBODY_DL_TR2), bring, between);
    else wide_layer<ACT_SWISH, MT>(Xs, Wp(MOBODY_DL_TR2), Bp(MOBODY_DL_TR2), HID, ring, NoExtra{}, between);
  };
  if constexpr (NT3 > 0) {
    NarrowRegs<NT3> br;
    const int mycol = threadIdx.x % (16 * NT3);          // column of every output element this thread finishes
    float bias;
    layer_tr2([&] {
      narrow_prefetch<NT3>(Wp(MOBODY_DL_TR3), 16 * NT3, br);
      bias = b3[mycol < S ? mycol : 0];
    });
    if constexpr (NT3 > 2) {                             // wide heads (pen 48, ant 112 columns): 32-row passes, bias by column
      narrow_run_wide<TB / 16, NT3>(Xs, br, [&](int row, int col, float v) {
        if (row < rows_here && col < S) mean[row * S + col] = v + b3[col];
      });
    } else {
      narrow_run<TB / 16, NT3>(Xs, br, [&](int row, int col, float v) {
        if (row < rows_here && col < S) mean[row * S + col] = v + bias;
      });
    }
  } else {
    layer_tr2([] {});
    narrow_layer(Xs, Wp(MOBODY_DL_TR3), HID, a.L.layer[MOBODY_DL_TR3].Np, [&](int row, int col, float v) {
      if (row < rows_here && col < S) mean[row * S + col] = v + b3[col];
    }, TB);
  }
}

// ------------------------------------------------------------------------------------------
struct DynSampleArgs {
  const float* mean;         // [E][B][S]
  const float* noise;        // [E][B][S] or null
  const int32_t* elite_idx;  // [B] or null
  const uint8_t* alive;      // [B] or null
  int32_t elites[NENS];
  int n_elites;
  uint32_t seed, call;
  const long long* call_dev; // optional device word added to `call` (graph replay: the step counter lives on the device)
  long long B;
  int S, task;
  float* next_obs;           // [B][S]
  float* penalty;            // [B]
  uint8_t* terminal;         // [B]
  // multi-step rollout bookkeeping fused in (MOBODY.rollout mobody.py:635-653 without host compaction), both optional:
  uint8_t* keep;             // [B] alive_in && (!use_filter || penalty <= env_filter)
  uint8_t* alive_out;        // [B] alive_in && !terminal   (may alias `alive`)
  float env_filter;
  int use_filter;
};

__device__ __forceinline__ bool term_predicate(int task, const float* n, int S) {
  // terminal_funs.py; comparisons with NaN are false exactly as in NumPy
  bool in_box = true, finite = true, lt100_from1 = true;
  for (int d = 0; d < S; ++d) {
    const float x = n[d];
    in_box = in_box && (x > -100.f) && (x < 100.f);
    finite = finite && isfinite(x);
    if (d >= 1) lt100_from1 = lt100_from1 && (x < 100.f);
  }
  switch (task) {
    case MOBODY_TERM_HALFCHEETAH: return !in_box;                                                   // :10-16
    case MOBODY_TERM_HOPPER: return !(finite && lt100_from1 && (n[0] > 0.7f) && (fabsf(n[1]) < 0.2f));   // :18-30
    case MOBODY_TERM_ANT: return !(finite && (n[0] >= 0.2f) && (n[0] <= 1.0f));                      // :39-61
    case MOBODY_TERM_WALKER2D:                                                                       // :63-75
      return !(in_box && (n[0] > 0.8f) && (n[0] < 2.0f) && (n[1] > -1.0f) && (n[1] < 1.0f));
    case MOBODY_TERM_HUMANOID: return (n[0] < 1.0f) || (n[0] > 2.0f);                                // :98-104
    case MOBODY_TERM_PEN: return n[26] < 0.075f;                                                     // :106-113
    default: return false;
  }
}

// One thread per (row, state dim) element, a workgroup owns rpb = 256 / S whole rows: the member means, the noise and
// next_obs are read / written at consecutive addresses by consecutive lanes.  (The first version used one thread per
// row walking its S floats: every store instruction scattered 64 dwords over 64 rows and WRITE_SIZE showed 10x the
// algorithmic bytes.)  The per-row sums over d (penalty) go through LDS and are added in increasing d by one thread
// per (row, member): same order as a serial loop, no atomics.
__global__ __launch_bounds__(256) void k_dyn_sample(DynSampleArgs a, int rpb) {
  __shared__ float s_nxt[256];
  __shared__ float s_t2[256 * NENS];
  __shared__ float s_sq[64 * NENS];
  const int S = a.S;
  const int tid = threadIdx.x;
  const int r = tid / S, d = tid - r * S;
  const long long row0 = (long long)blockIdx.x * rpb;
  const long long b = row0 + r;
  const uint32_t call = a.call + (a.call_dev != nullptr ? (uint32_t)a.call_dev[0] : 0u);
  if (r < rpb && b < a.B) {
    int e_sel;
    if (a.elite_idx) e_sel = a.elite_idx[b];
    else e_sel = a.elites[rng_index_at(a.seed, STREAM_ELITE, call, (uint64_t)b, (uint32_t)a.n_elites)];
    float mval[NENS], avg = 0.f, msel = 0.f;
#pragma unroll
    for (int e = 0; e < NENS; ++e) {
      mval[e] = a.mean[((long long)e * a.B + b) * S + d];
      avg += mval[e];
      msel = (e == e_sel) ? mval[e] : msel;
    }
    avg *= (1.f / NENS);
    float var = 0.f;
#pragma unroll
    for (int e = 0; e < NENS; ++e) { const float t = mval[e] - avg; var += t * t; s_t2[tid * NENS + e] = t * t; }
    const float sd = sqrtf(var * (1.f / (NENS - 1)));                      // torch.std: unbiased (:218)
    float eps;
    if (a.noise) eps = a.noise[((long long)e_sel * a.B + b) * S + d];
    else eps = rng_normal_at(a.seed, STREAM_NOISE, call, (uint64_t)b * S + d);
    const float v = msel + eps * sd;                                       // :220-226
    a.next_obs[b * S + d] = v;
    s_nxt[tid] = v;
  }
  __syncthreads();
  for (int t = tid; t < rpb * NENS; t += 256) {                            // (row, member): sum over d < S-1, in order
    const int rr = t / NENS, e = t - rr * NENS;
    float sq = 0.f;
    for (int dd = 0; dd < S - 1; ++dd) sq += s_t2[(rr * S + dd) * NENS + e];
    s_sq[t] = sq;
  }
  __syncthreads();
  if (tid < rpb && row0 + tid < a.B) {
    const long long bb = row0 + tid;
    // torch.amax propagates NaN (one non-finite member makes every norm of the row NaN); fmaxf would drop it and
    // the row would pass `penalty <= env_filter` with NaN next_obs, where the reference's comparisons are all False
    float pmax = 0.f;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < NENS; ++e) { const float q = s_sq[tid * NENS + e]; bad = bad || (q != q); pmax = fmaxf(pmax, q); }
    a.penalty[bb] = bad ? __builtin_nanf("") : sqrtf(pmax);                // :246-249 (last state dim dropped)
    bool done = term_predicate(a.task, s_nxt + tid * S, S);
    const bool was_alive = a.alive ? a.alive[bb] != 0 : true;
    if (!was_alive) done = true;
    a.terminal[bb] = done ? 1 : 0;
    const float pen = a.penalty[bb];
    if (a.keep) a.keep[bb] = (was_alive && (!a.use_filter || pen <= a.env_filter)) ? 1 : 0;     // NaN penalty: dropped (:648-653)
    if (a.alive_out) a.alive_out[bb] = (was_alive && !done) ? 1 : 0;
  }
}

struct DynFinalArgs {
  const float* r_mu;     // [E][B]
  const float* penalty;  // [B]
  float* reward;         // [B]
  float* raw_reward;     // [B] or null
  long long B;
  float coef;            // penalty_coef if (penalty_coef && use_penalty) else 0
};

__global__ __launch_bounds__(256) void k_dyn_finalize(DynFinalArgs a) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < NENS; ++e) s += a.r_mu[(long long)e * a.B + b];
  const float raw = s * (1.f / NENS);                                     // reward.mean(0) :236
  if (a.raw_reward) a.raw_reward[b] = raw;
  a.reward[b] = (a.coef != 0.f) ? raw - a.coef * a.penalty[b] : raw;       // :261-263
}

template <int MT, int NT3, int NPL>
static int launch_dyn_fwd_t(const DynFwdArgs& a, hipStream_t st) {
  constexpr size_t lds = split_lds_bytes<(NPL > 0 ? NPL : 1), 32 * MT>();
  static bool once = false;
  if (!once) {
    int rc = allow_big_lds(k_dyn_fwd<MT, NT3, NPL>, lds);
    if (rc) return rc;
    once = true;
  }
  ProfScope prof(PROF_DYN_FWD, st);
  hipLaunchKernelGGL((k_dyn_fwd<MT, NT3, NPL>), dim3((unsigned)cdiv(a.B, 32 * MT), NENS), dim3(NTHREADS), lds, st, a);
  MB_LAUNCH_OK("k_dyn_fwd");
  return 0;
}
template <int MT, int NPL>
static int launch_dyn_fwd_nt(const DynFwdArgs& a, int nt3, hipStream_t st) {
  return nt3 == 1 ? launch_dyn_fwd_t<MT, 1, NPL>(a, st) : nt3 == 2 ? launch_dyn_fwd_t<MT, 2, NPL>(a, st)
       : nt3 == 3 ? launch_dyn_fwd_t<MT, 3, NPL>(a, st) : nt3 == 7 ? launch_dyn_fwd_t<MT, 7, NPL>(a, st) : launch_dyn_fwd_t<MT, 0, NPL>(a, st);
}

static int launch_dyn_fwd(const float* blob, const MobodyDynLayout& L, const float* obs, const float* act, long long B,
                          int use_trg, float* mean, const float* planes, int prec, hipStream_t st) {
  DynFwdArgs a{blob, L, obs, act, mean, B, use_trg, reinterpret_cast<const unsigned short*>(planes)};
  // 64-row tiles for fp32: the nine-layer chain has a wave-local narrow section in which a 32-row tile idles half of
  // the waves (measured 102 vs 88 TFLOP/s at 50 000 rows); MOBODY_DYN_TILE_ROWS=32 selects the short tile.
  static const int forced = tune_int("MOBODY_DYN_TILE_ROWS", 0);
  const int np = L.layer[MOBODY_DL_TR3].Np;
  const int nt3 = np == 16 ? 1 : np == 32 ? 2 : np == 48 ? 3 : np == 112 ? 7 : 0;      // walker/hopper/cheetah, pen, ant heads; else generic
  if (prec == 0) return forced == 32 ? launch_dyn_fwd_nt<1, 0>(a, nt3, st) : launch_dyn_fwd_nt<2, 0>(a, nt3, st);
  // split-precision modes: the planes of a 64-row tile are 32 / 64 / 96 KB for 1 / 2 / 3 terms -> the three-term mode
  // runs 32-row tiles (48 KB, three workgroups per CU)
  const bool tall = forced == 64 || (forced != 32 && prec != 3);
  if (prec == 1) return tall ? launch_dyn_fwd_nt<2, 1>(a, nt3, st) : launch_dyn_fwd_nt<1, 1>(a, nt3, st);
  if (prec == 2) return tall ? launch_dyn_fwd_nt<2, 2>(a, nt3, st) : launch_dyn_fwd_nt<1, 2>(a, nt3, st);
  if (prec == 4) return tall ? launch_dyn_fwd_nt<2, 4>(a, nt3, st) : launch_dyn_fwd_nt<1, 4>(a, nt3, st);
  return tall ? launch_dyn_fwd_nt<2, 3>(a, nt3, st) : launch_dyn_fwd_nt<1, 3>(a, nt3, st);
}

// zs2 / transition2 / reward_model2 of every member -> their three bf16 planes (precision 0-3) or two fp16 planes of
// w * 2^F16_WSHIFT (precision 4)
__global__ __launch_bounds__(256) void k_dyn_planes(const float* blob, MobodyDynLayout L, short* planes, int precision) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 3LL * NENS * HID * HID) return;
  const int layer = (int)(gid / ((long long)NENS * HID * HID));
  const long long rem = gid - (long long)layer * NENS * HID * HID;
  const int e = (int)(rem / (HID * HID)), el = (int)(rem % (HID * HID)), k = el / HID, n = el % HID;
  const int li = layer == 0 ? MOBODY_DL_ZS2 : layer == 1 ? MOBODY_DL_TR2 : MOBODY_DL_RW2;
  const float w = blob[L.layer[li].w_off + (long long)e * HID * HID + wide_idx(k, n)];
  short* pl = planes + dyn_planes_off(layer, e);
  if (precision == 4) {
    short t[2];
    split_terms<4>(w * exp2i(F16_WSHIFT), t);
#pragma unroll
    for (int p = 0; p < 2; ++p) pl[bf_plane_idx(p, k, n)] = t[p];
  } else {
    short t[3];
    split_terms<3>(w, t);
#pragma unroll
    for (int p = 0; p < 3; ++p) pl[bf_plane_idx(p, k, n)] = t[p];
  }
}

}  // namespace mobody

using namespace mobody;

extern "C" int64_t mobody_dyn_planes_floats(void) { return 3LL * NENS * DYN_PLANE_MEMBER / 2; }

extern "C" int mobody_dyn_planes(const float* dyn_blob, int S, int A, float* planes, int precision, void* stream) {
  MobodyDynLayout L;
  int rc = mobody_dyn_layout(S, A, &L);
  if (rc) return rc;
  MB_REQUIRE(dyn_blob && planes, "mobody_dyn_planes: null pointer");
  MB_REQUIRE(precision >= 0 && precision <= 4, "mobody_dyn_planes: precision must be 0..4");
  hipLaunchKernelGGL(k_dyn_planes, dim3((unsigned)cdiv(3LL * NENS * HID * HID, 256)), dim3(256), 0, as_stream(stream), dyn_blob, L,
                     reinterpret_cast<short*>(planes), precision);
  MB_LAUNCH_OK("k_dyn_planes");
  return 0;
}

static int check_dyn_prec(const char* who, int precision, const float* planes) {
  MB_REQUIRE(precision >= 0 && precision <= 4, "%s: precision must be 0 (f32), 1 (bf16), 2 (bf16x2), 3 (bf16x3) or 4 (f16x2)", who);
  MB_REQUIRE(precision == 0 || planes, "%s: the split-precision modes need the plane blob (mobody_dyn_planes)", who);
  return 0;
}

extern "C" int mobody_dyn_forward(const float* dyn_blob, const float* dyn_planes, int precision, int S, int A, const float* obs,
                                  const float* act, int64_t B, int use_trg, float* mean, void* stream) {
  MobodyDynLayout L;
  int rc = mobody_dyn_layout(S, A, &L);
  if (rc) return rc;
  MB_REQUIRE(B >= 0, "mobody_dyn_forward: B < 0");
  if (B == 0) return 0;                      // empty batch: nothing to do (pointers may be null)
  MB_REQUIRE(dyn_blob && obs && act && mean, "mobody_dyn_forward: null pointer");
  rc = check_dyn_prec("mobody_dyn_forward", precision, dyn_planes);
  if (rc) return rc;
  return launch_dyn_fwd(dyn_blob, L, obs, act, B, use_trg, mean, dyn_planes, precision, as_stream(stream));
}

extern "C" int64_t mobody_dyn_step_workspace(int S, int A, int64_t B) {
  (void)A;
  return (int64_t)NENS * B * S + (int64_t)NENS * B;    // ensemble means + per-member reward means
}

// mopo_blob != null: the MOPO ablation (config['mopo'], mobody_module.py:114-118,218-219,251-254,264-266,288-289) --
// mean[e] = obs + MLP_e([obs, act]) with the 7-member Swish MLP (S+A -> 256 -> 256 -> S) za_src1..3 packed as
// mobody_mlp_layout(S + A, S, 7); encoders and decoder are bypassed, forward_trg == forward_src.  Everything after the means
// (std, sample, reward head, penalty, termination) is the same code.
static int dyn_step_impl(const float* dyn_blob, const float* dyn_planes, const float* mopo_blob, const float* mopo_blob_T,
                         int precision, int S, int A, int task, const float* obs, const float* act,
                         int64_t B, const float* noise, const int32_t* elite_idx, const uint8_t* alive,
                         const int32_t* elites, int n_elites, uint32_t seed, uint32_t call, const int64_t* call_dev, float penalty_coef, int use_penalty,
                         int use_trg, float* next_obs, float* reward, uint8_t* terminal, float* penalty,
                         float* raw_reward, float* mean_out, float* workspace, uint8_t* keep, uint8_t* alive_out,
                         float env_filter, int use_filter, void* stream) {
  MobodyDynLayout L;
  int rc = mobody_dyn_layout(S, A, &L);
  if (rc) return rc;
  MB_REQUIRE(B >= 0, "mobody_dyn_step: B < 0");
  if (B == 0) return 0;                      // empty batch: nothing to do (pointers may be null)
  MB_REQUIRE(dyn_blob && obs && act && next_obs && reward && terminal && penalty && workspace, "mobody_dyn_step: null pointer");
  MB_REQUIRE(task >= MOBODY_TERM_NEVER && task <= MOBODY_TERM_PEN, "mobody_dyn_step: unknown termination id %d", task);
  MB_REQUIRE(task != MOBODY_TERM_PEN || S > 26, "mobody_dyn_step: pen predicate needs S > 26");
  MB_REQUIRE(elite_idx != nullptr || (elites != nullptr && n_elites >= 1 && n_elites <= NENS),
             "mobody_dyn_step: need elite_idx or 1..7 elites");
  hipStream_t st = as_stream(stream);
  float* mean = mean_out ? mean_out : workspace;
  float* r_mu = workspace + (int64_t)NENS * B * S;
  rc = check_dyn_prec("mobody_dyn_step", precision, dyn_planes);
  if (rc) return rc;
  if (mopo_blob != nullptr) {
    MobodyMlpLayout ML;
    rc = mobody_mlp_layout(S + A, S, NENS, &ML);
    if (rc) return rc;
    MB_REQUIRE(precision == 0 || mopo_blob_T != nullptr, "mobody_mopo_step: the split-precision modes need the T blob of the MLP");
    Mlp3FwdArgs f{};
    f.src[0] = obs; f.ld[0] = S; f.n[0] = S;
    f.src[1] = act; f.ld[1] = A; f.n[1] = A;
    f.w1 = mopo_blob + ML.w1; f.b1 = mopo_blob + ML.b1; f.w2 = mopo_blob + ML.w2; f.b2 = mopo_blob + ML.b2;
    f.w3 = mopo_blob + ML.w3; f.b3 = mopo_blob + ML.b3;
    f.sw1 = f.sb1 = f.sw2 = f.sb2 = f.sw3 = f.sb3 = ML.member_floats;
    f.Kp1 = ML.Kp1; f.Np3 = ML.Np3; f.nout = S; f.rows = B;
    f.out = mean; f.out_mstride = B * S; f.out_ld = S;
    f.out_mode = 0; f.max_action = 1.f;
    f.resid = obs; f.resid_ld = S;
    if (precision == 0) {
      rc = launch_mlp3_fwd(f, NENS, ACT_SWISH, st);
    } else {
      f.w2_planes = reinterpret_cast<const unsigned short*>(mopo_blob_T + ML.w2p);
      f.planes_ms = 2 * ML.t_member_floats;
      rc = launch_mlp3_fwd_bf(f, NENS, Mlp3FwdArgs{}, 0, ACT_SWISH, precision, st);
    }
  } else {
    rc = launch_dyn_fwd(dyn_blob, L, obs, act, B, use_trg, mean, dyn_planes, precision, st);
  }
  if (rc) return rc;

  DynSampleArgs sa{};
  sa.mean = mean; sa.noise = noise; sa.elite_idx = elite_idx; sa.alive = alive;
  for (int k = 0; k < NENS; ++k) sa.elites[k] = (elites && k < n_elites) ? elites[k] : 0;
  sa.n_elites = n_elites; sa.seed = seed; sa.call = call; sa.call_dev = (const long long*)call_dev; sa.B = B; sa.S = S; sa.task = task;
  sa.next_obs = next_obs; sa.penalty = penalty; sa.terminal = terminal;
  sa.keep = keep; sa.alive_out = alive_out; sa.env_filter = env_filter; sa.use_filter = use_filter;
  const int rpb = 256 / S < 64 ? 256 / S : 64;         // whole rows per workgroup (S <= 256 is checked by the layout)
  hipLaunchKernelGGL(k_dyn_sample, dim3((unsigned)cdiv(B, rpb)), dim3(256), 0, st, sa, rpb);
  MB_LAUNCH_OK("k_dyn_sample");

  // reward head on [s, a, s'] shared by the 7 members  (mobody_dynamics.py:235)
  Mlp3FwdArgs m{};
  m.src[0] = obs; m.ld[0] = S; m.n[0] = S;
  m.src[1] = act; m.ld[1] = A; m.n[1] = A;
  m.src[2] = next_obs; m.ld[2] = S; m.n[2] = S;
  const MobodyLayer &l1 = L.layer[MOBODY_DL_RW1], &l2 = L.layer[MOBODY_DL_RW2], &l3 = L.layer[MOBODY_DL_RW3];
  m.w1 = dyn_blob + l1.w_off; m.b1 = dyn_blob + l1.b_off; m.sw1 = (long long)l1.Kp * l1.Np; m.sb1 = l1.Np;
  m.w2 = dyn_blob + l2.w_off; m.b2 = dyn_blob + l2.b_off; m.sw2 = (long long)l2.Kp * l2.Np; m.sb2 = l2.Np;
  m.w3 = dyn_blob + l3.w_off; m.b3 = dyn_blob + l3.b_off; m.sw3 = (long long)l3.Kp * l3.Np; m.sb3 = l3.Np;
  m.Kp1 = l1.Kp; m.Np3 = l3.Np; m.nout = 1; m.rows = B;
  m.out = r_mu; m.out_mstride = B; m.out_ld = 1;
  m.out_mode = 0; m.max_action = 1.f;
  if (precision == 0) {
    rc = launch_mlp3_fwd(m, NENS, ACT_SWISH, st);
  } else {
    m.w2_planes = reinterpret_cast<const unsigned short*>(dyn_planes) + dyn_planes_off(2, 0);
    m.planes_ms = DYN_PLANE_MEMBER;
    rc = launch_mlp3_fwd_bf(m, NENS, Mlp3FwdArgs{}, 0, ACT_SWISH, precision, st);
  }
  if (rc) return rc;

  DynFinalArgs fa{r_mu, penalty, reward, raw_reward, B, (penalty_coef != 0.f && use_penalty) ? penalty_coef : 0.f};
  hipLaunchKernelGGL(k_dyn_finalize, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, st, fa);
  MB_LAUNCH_OK("k_dyn_finalize");
  return 0;
}

extern "C" int mobody_dyn_step(const float* dyn_blob, const float* dyn_planes, int precision, int S, int A, int task,
                               const float* obs, const float* act,
                               int64_t B, const float* noise, const int32_t* elite_idx, const uint8_t* alive,
                               const int32_t* elites, int n_elites, uint32_t seed, uint32_t call, const int64_t* call_dev,
                               float penalty_coef, int use_penalty,
                               int use_trg, float* next_obs, float* reward, uint8_t* terminal, float* penalty,
                               float* raw_reward, float* mean_out, float* workspace, void* stream) {
  return dyn_step_impl(dyn_blob, dyn_planes, nullptr, nullptr, precision, S, A, task, obs, act, B, noise, elite_idx, alive, elites, n_elites, seed,
                       call, call_dev, penalty_coef, use_penalty, use_trg, next_obs, reward, terminal, penalty, raw_reward, mean_out, workspace,
                       nullptr, nullptr, 0.f, 0, stream);
}

extern "C" int mobody_mopo_step(const float* dyn_blob, const float* dyn_planes, const float* mopo_blob, const float* mopo_blob_T,
                                int precision, int S, int A, int task, const float* obs, const float* act, int64_t B,
                                const float* noise, const int32_t* elite_idx, const uint8_t* alive, const int32_t* elites,
                                int n_elites, uint32_t seed, uint32_t call, float penalty_coef, int use_penalty, float* next_obs,
                                float* reward, uint8_t* terminal, float* penalty, float* raw_reward, float* mean_out,
                                float* workspace, void* stream) {
  MB_REQUIRE(B == 0 || mopo_blob != nullptr, "mobody_mopo_step: null MLP blob");
  return dyn_step_impl(dyn_blob, dyn_planes, mopo_blob, mopo_blob_T, precision, S, A, task, obs, act, B, noise, elite_idx, alive, elites,
                       n_elites, seed, call, nullptr, penalty_coef, use_penalty, 1, next_obs, reward, terminal, penalty, raw_reward, mean_out,
                       workspace, nullptr, nullptr, 0.f, 0, stream);
}

// ---- whole H-step imagined rollout on the device (MOBODY.rollout + add_batch, mobody.py:596-657, utils.py:43-92) ----
namespace mobody {
struct RolloutWs {
  float *obs[2], *act, *reward, *penalty, *dyn;
  uint8_t *terminal, *keep, *alive;
  int32_t* scan;
  long long total;
};
static void rollout_carve(int S, int A, long long B, float* base, RolloutWs& w) {
  long long off = 0;
  auto take = [&](long long n) { float* p = base ? base + off : nullptr; off += (n + 3) & ~3LL; return p; };
  w.obs[0] = take(B * S); w.obs[1] = take(B * S); w.act = take(B * A); w.reward = take(B); w.penalty = take(B);
  w.dyn = take((long long)NENS * B * S + (long long)NENS * B);
  w.terminal = (uint8_t*)take((B + 3) / 4); w.keep = (uint8_t*)take((B + 3) / 4); w.alive = (uint8_t*)take((B + 3) / 4);
  w.scan = (int32_t*)take(B + 1040);
  w.total = off;
}
}  // namespace mobody

extern "C" int64_t mobody_rollout_workspace(int S, int A, int64_t B) {
  RolloutWs w;
  rollout_carve(S, A, B, nullptr, w);
  return w.total;
}

extern "C" int mobody_rollout(const float* dyn_blob, const float* dyn_planes, const float* actor_blob, const float* actor_blob_T,
                              int precision, int S, int A, int task, float max_action,
                              const float* init_obs, int64_t B, int H, const int32_t* elites, int n_elites, uint32_t seed,
                              uint32_t call0, float penalty_coef, int use_penalty, int use_trg, float env_filter,
                              int filter_bad_rollout, const MobodyBufferView* ring, int64_t cap, int64_t* ptr_size, float* workspace,
                              void* stream) {
  MB_REQUIRE(B >= 0 && H >= 0, "mobody_rollout: bad sizes");
  if (B == 0 || H == 0) return 0;
  MB_REQUIRE(B <= cap, "mobody_rollout: %lld rows per step overflow the ring of %lld twice", (long long)B, (long long)cap);
  MB_REQUIRE(dyn_blob && actor_blob && init_obs && elites && ring && ring->state && ring->action && ring->next_state && ring->reward && ring->not_done &&
                 ptr_size && workspace, "mobody_rollout: null pointer");
  MobodyMlpLayout La;
  int rc = mobody_mlp_layout(S, A, 1, &La);
  if (rc) return rc;
  RolloutWs w;
  rollout_carve(S, A, B, workspace, w);
  hipStream_t st = as_stream(stream);
  // the appends' arrival ticket (first words of the scan region) has to start at zero; every append leaves it at zero
  if (hipMemsetAsync(w.scan, 0, 8 * sizeof(int32_t), st) != hipSuccess) return fail(MOBODY_E_LAUNCH, "mobody_rollout: memset failed");
  const float* obs = init_obs;
  for (int t = 0; t < H; ++t) {
    float* nxt = w.obs[t & 1];
    // a = pi(s)  (select_action, mobody.py:612)
    Mlp3FwdArgs p{};
    p.src[0] = obs; p.ld[0] = S; p.n[0] = S;
    p.w1 = actor_blob + La.w1; p.b1 = actor_blob + La.b1; p.w2 = actor_blob + La.w2; p.b2 = actor_blob + La.b2;
    p.w3 = actor_blob + La.w3; p.b3 = actor_blob + La.b3;
    p.sw1 = p.sb1 = p.sw2 = p.sb2 = p.sw3 = p.sb3 = La.member_floats;
    p.Kp1 = La.Kp1; p.Np3 = La.Np3; p.nout = A; p.rows = B; p.out = w.act; p.out_mstride = B * A; p.out_ld = A;
    p.out_mode = 1; p.max_action = max_action;
    if (precision == 0) {
      rc = launch_mlp3_fwd(p, 1, ACT_RELU, st);
    } else {
      MB_REQUIRE(actor_blob_T, "mobody_rollout: the split-precision modes need the actor's T blob");
      p.w2_planes = reinterpret_cast<const unsigned short*>(actor_blob_T + La.w2p);
      p.planes_ms = 2 * La.t_member_floats;
      rc = launch_mlp3_fwd_bf(p, 1, Mlp3FwdArgs{}, 0, ACT_RELU, precision, st);
    }
    if (rc) return rc;
    // one imagined transition for every row; rows that terminated earlier keep their index and are flagged (alive mask);
    // the penalty filter and the alive update are formed in the sample kernel
    rc = dyn_step_impl(dyn_blob, dyn_planes, nullptr, nullptr, precision, S, A, task, obs, w.act, B, nullptr, nullptr, t == 0 ? nullptr : w.alive, elites, n_elites, seed,
                       call0 + (uint32_t)t, nullptr, penalty_coef, use_penalty, use_trg, nxt, w.reward, w.terminal, w.penalty, nullptr, nullptr,
                       w.dyn, w.keep, w.alive, env_filter, filter_bad_rollout, stream);
    if (rc) return rc;
    rc = launch_ring_append(*ring, cap, (long long*)ptr_size, S, A, obs, w.act, nxt,
                            w.reward, w.terminal, w.keep, B, w.scan, st);
    if (rc) return rc;
    obs = nxt;
  }
  return 0;
}

// ---- stand-alone predicate / rollout bookkeeping -------------------------------------------------
namespace mobody {
__global__ __launch_bounds__(256) void k_termination(int task, const float* next_obs, long long B, int S, uint8_t* done) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) done[b] = term_predicate(task, next_obs + b * S, S) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_rollout_mask(const uint8_t* alive_in, const uint8_t* terminal, const float* penalty,
                                                      float env_filter, int use_filter, long long B, uint8_t* keep,
                                                      uint8_t* alive_out) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const bool alive = alive_in ? alive_in[b] != 0 : true;
  if (keep) keep[b] = (alive && (!use_filter || penalty[b] <= env_filter)) ? 1 : 0;     // mobody.py:648-653
  if (alive_out) alive_out[b] = (alive && !terminal[b]) ? 1 : 0;                         // :635-639
}
}  // namespace mobody

extern "C" int mobody_termination(int task, const float* next_obs, int64_t B, int S, uint8_t* done, void* stream) {
  MB_REQUIRE(B >= 0 && S >= 1, "mobody_termination: bad sizes");
  if (B == 0) return 0;
  MB_REQUIRE(next_obs && done, "mobody_termination: null pointer");
  MB_REQUIRE(task >= MOBODY_TERM_NEVER && task <= MOBODY_TERM_PEN, "mobody_termination: unknown termination id %d", task);
  MB_REQUIRE(task != MOBODY_TERM_PEN || S > 26, "mobody_termination: pen predicate needs S > 26");
  MB_REQUIRE(S >= 2 || task == MOBODY_TERM_NEVER || task == MOBODY_TERM_HALFCHEETAH || task == MOBODY_TERM_ANT || task == MOBODY_TERM_HUMANOID,
             "mobody_termination: predicate needs S >= 2");
  hipLaunchKernelGGL(k_termination, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, as_stream(stream), task, next_obs, (long long)B, S, done);
  MB_LAUNCH_OK("k_termination");
  return 0;
}

extern "C" int mobody_rollout_mask(const uint8_t* alive_in, const uint8_t* terminal, const float* penalty, float env_filter,
                                   int use_filter, int64_t B, uint8_t* keep, uint8_t* alive_out, void* stream) {
  MB_REQUIRE(B >= 0, "mobody_rollout_mask: B < 0");
  if (B == 0) return 0;
  MB_REQUIRE(terminal && penalty, "mobody_rollout_mask: null pointer");
  hipLaunchKernelGGL(k_rollout_mask, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, as_stream(stream), alive_in, terminal, penalty,
                     env_filter, use_filter, (long long)B, keep, alive_out);
  MB_LAUNCH_OK("k_rollout_mask");
  return 0;
}
