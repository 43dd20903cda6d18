// Error channel + packed-weight layouts of the C ABI (include/mobody_hip.h).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace mobody {

std::string& last_error() {
  static thread_local std::string e;
  return e;
}

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

// ---- profiling state (the only process-global state of the library; off by default) ----
struct ProfEvent { int id; hipEvent_t a, b; };
static std::vector<ProfEvent>& prof_pool() { static std::vector<ProfEvent> p; return p; }
static bool g_prof_on = false;
static size_t g_prof_used = 0;

ProfScope::ProfScope(int id, hipStream_t stream) : slot(-1), st(stream) {
  if (!g_prof_on || g_prof_used >= prof_pool().size()) return;
  slot = (int)g_prof_used++;
  prof_pool()[slot].id = id;
  (void)hipEventRecord(prof_pool()[slot].a, st);
}
ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(prof_pool()[slot].b, st);
}

}  // namespace mobody

using namespace mobody;

#ifdef MOBODY_TRACE
namespace mobody { __device__ unsigned long long g_trace[TRACE_BLOCKS * TRACE_SLOTS]; }
extern "C" int mobody_debug_trace(unsigned long long* host_out, int n) {      // diagnostic builds only, not in the header
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mobody::g_trace), sizeof(unsigned long long) * n);
}
#endif

extern "C" int mobody_prof_begin(int max_events) {
  MB_REQUIRE(max_events > 0 && max_events <= (1 << 20), "mobody_prof_begin: bad capacity");
  auto& p = prof_pool();
  while ((int)p.size() < max_events) {
    ProfEvent e{};
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return fail(MOBODY_E_LAUNCH, "hipEventCreate failed");
    p.push_back(e);
  }
  g_prof_used = 0;
  g_prof_on = true;
  return 0;
}

extern "C" int mobody_prof_end(double* ms_by_id, int64_t* count_by_id, int n_ids) {
  g_prof_on = false;
  MB_REQUIRE(ms_by_id && count_by_id && n_ids >= PROF_COUNT, "mobody_prof_end: need %d slots", (int)PROF_COUNT);
  for (int i = 0; i < n_ids; ++i) { ms_by_id[i] = 0.0; count_by_id[i] = 0; }
  for (size_t k = 0; k < g_prof_used; ++k) {
    ProfEvent& e = prof_pool()[k];
    if (hipEventSynchronize(e.b) != hipSuccess) return fail(MOBODY_E_LAUNCH, "hipEventSynchronize failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) return fail(MOBODY_E_LAUNCH, "hipEventElapsedTime failed");
    ms_by_id[e.id] += ms;
    count_by_id[e.id] += 1;
  }
  g_prof_used = 0;
  return 0;
}

extern "C" const char* mobody_last_error(void) { return last_error().c_str(); }
extern "C" int mobody_abi_version(void) { return MOBODY_ABI_VERSION; }

// Layers of MOBODYModule the hot path evaluates (mobody_module.py:97-184, mopo=0, latent_reward=0).
// Only the halves of zs3 / za_*2 / reward_model3 that inference uses are packed (mu of the
// latent heads :217-225,258-271; reward mean :295-302); transition3 is padded to 16 columns.
extern "C" int mobody_dyn_layout(int S, int A, MobodyDynLayout* out) {
  MB_REQUIRE(out != nullptr, "mobody_dyn_layout: out is null");
  MB_REQUIRE(S >= 2 && S <= 240 && A >= 1 && A <= 64, "mobody_dyn_layout: unsupported S=%d A=%d (S in [2,240], A in [1,64])", S, A);
  MB_REQUIRE(round_up(2 * S + A, 8) <= 256, "mobody_dyn_layout: reward-head input 2S+A=%d exceeds 256", 2 * S + A);
  memset(out, 0, sizeof(*out));
  out->S = S; out->A = A; out->E = NENS;
  struct D { int id, in, out, Kp, Np; };
  const int L = LATENT, H = HID;
  const D dims[MOBODY_DL_COUNT] = {
      {MOBODY_DL_ZS1, S, H, round_up(S, 8), H},
      {MOBODY_DL_ZS2, H, H, H, H},
      {MOBODY_DL_ZS3, H, L, H, 16},
      {MOBODY_DL_ZA_SRC1, L + A, 32, round_up(L + A, 8), 32},
      {MOBODY_DL_ZA_SRC2, 32, L, 32, 16},
      {MOBODY_DL_ZA_TRG1, L + A, 32, round_up(L + A, 8), 32},
      {MOBODY_DL_ZA_TRG2, 32, L, 32, 16},
      {MOBODY_DL_TR1, L, H, 16, H},
      {MOBODY_DL_TR2, H, H, H, H},
      {MOBODY_DL_TR3, H, S, H, round_up(S, 16)},
      {MOBODY_DL_RW1, 2 * S + A, H, round_up(2 * S + A, 8), H},
      {MOBODY_DL_RW2, H, H, H, H},
      {MOBODY_DL_RW3, H, 1, H, 16},
  };
  int64_t off = 0;
  for (int i = 0; i < MOBODY_DL_COUNT; ++i) {
    MobodyLayer& l = out->layer[dims[i].id];
    l.in_dim = dims[i].in; l.out_dim = dims[i].out; l.Kp = dims[i].Kp; l.Np = dims[i].Np;
    l.w_off = off; off += (int64_t)NENS * l.Kp * l.Np;
    l.b_off = off; off += (int64_t)NENS * l.Np;
    off = (off + 3) & ~(int64_t)3;   // keep every matrix 16-byte aligned
  }
  out->total_floats = off;
  return 0;
}

extern "C" int mobody_mlp_layout(int in_dim, int out_dim, int members, MobodyMlpLayout* out) {
  MB_REQUIRE(out != nullptr, "mobody_mlp_layout: out is null");
  MB_REQUIRE(in_dim >= 1 && in_dim <= 256 && out_dim >= 1 && out_dim <= 128 && members >= 1 && members <= 8,
             "mobody_mlp_layout: unsupported in=%d out=%d members=%d", in_dim, out_dim, members);
  memset(out, 0, sizeof(*out));
  out->in_dim = in_dim; out->out_dim = out_dim; out->members = members;
  out->Kp1 = round_up(in_dim, 8); out->Np3 = round_up(out_dim, 16); out->Np1t = round_up(in_dim, 16);
  int64_t o = 0;
  out->w1 = o; o += (int64_t)out->Kp1 * HID;
  out->b1 = o; o += HID;
  out->w2 = o; o += (int64_t)HID * HID;
  out->b2 = o; o += HID;
  out->w3 = o; o += (int64_t)HID * out->Np3;
  out->b3 = o; o += out->Np3;
  out->member_floats = o; out->total_floats = o * members;
  int64_t t = 0;
  out->w3t = t; t += (int64_t)out->Np3 * HID;
  out->w2t = t; t += (int64_t)HID * HID;
  out->w1t = t; t += (int64_t)HID * out->Np1t;
  out->w2p = t; t += 3 * (int64_t)HID * HID / 2;        // three bf16 planes of W2 (2 bf16 per float slot)
  out->w2tp = t; t += 3 * (int64_t)HID * HID / 2;       // ... and of W2^T
  out->t_member_floats = t; out->t_total_floats = t * members;
  return 0;
}
