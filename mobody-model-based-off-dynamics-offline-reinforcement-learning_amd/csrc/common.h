// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>

#include "../../include/mobody_hip.h"
#include "tile.h"

namespace mobody {

std::string& last_error();
int fail(int code, const char* fmt, ...);

#define MB_REQUIRE(cond, ...)                                   \
  do {                                                          \
    if (!(cond)) return ::mobody::fail(MOBODY_E_ARG, __VA_ARGS__); \
  } while (0)

#define MB_LAUNCH_OK(what)                                                                   \
  do {                                                                                       \
    hipError_t e_ = hipGetLastError();                                                       \
    if (e_ != hipSuccess) return ::mobody::fail(MOBODY_E_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

constexpr size_t TILE_LDS_BYTES = (size_t)BM * LDX * sizeof(float);   // 66,560 B: one activation image

// Opt a kernel into > 64 KiB of dynamic LDS once per process.
template <class K>
inline int allow_big_lds(K kernel, size_t bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bytes);
  if (e != hipSuccess) return fail(MOBODY_E_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  return 0;
}

// Optional per-kernel timing (bench.py): when enabled, the launch helpers bracket each launch of a kernel
// family with a HIP event pair on the launch stream.  Disabled (one branch) in normal operation.
enum ProfId { PROF_MLP_FWD = 0, PROF_MLP_BWD, PROF_WGRAD, PROF_DYN_FWD, PROF_DYN_TAIL, PROF_ROWWISE, PROF_OPTIM, PROF_REPLAY, PROF_COUNT };
struct ProfScope {
  int slot;
  hipStream_t st;
  ProfScope(int id, hipStream_t stream);
  ~ProfScope();
};


inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ring append (replay.hip): block scan (+ commit of the new {ptr, size} by the last block) and the row scatter
int launch_ring_append(const MobodyBufferView& ring, long long cap, long long* ptr_size, int S, int A, const float* obs,
                       const float* act, const float* next_obs, const float* reward, const uint8_t* terminal,
                       const uint8_t* keep, long long M, int32_t* scan_ws, hipStream_t st);
__host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// device-side view of one packed MLP (member 0 pointers + strides), built from MobodyMlpLayout
struct MlpView {
  const float *w1, *b1, *w2, *b2, *w3, *b3;
  int64_t member_floats;
  int Kp1, Np3, in_dim, out_dim;
};
inline MlpView mlp_view(const float* blob, const MobodyMlpLayout& L) {
  MlpView v;
  v.w1 = blob + L.w1; v.b1 = blob + L.b1; v.w2 = blob + L.w2; v.b2 = blob + L.b2; v.w3 = blob + L.w3; v.b3 = blob + L.b3;
  v.member_floats = L.member_floats; v.Kp1 = L.Kp1; v.Np3 = L.Np3; v.in_dim = L.in_dim; v.out_dim = L.out_dim;
  return v;
}

// Phase timeline of the fused MLP kernels (diagnostic builds only: MOBODY_TRACE=1 python build.py --force).
// TR(k) stores the 100 MHz wall clock of workgroup (blockIdx.y, blockIdx.x) at phase k; tools/trace_mlp.py reads it.
#ifdef MOBODY_TRACE
constexpr int TRACE_BLOCKS = 16384, TRACE_SLOTS = 8;
extern __device__ unsigned long long g_trace[TRACE_BLOCKS * TRACE_SLOTS];
#define TR(k)                                                                                                     \
  do {                                                                                                            \
    if (threadIdx.x == 0)                                                                                         \
      g_trace[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) % TRACE_BLOCKS) * TRACE_SLOTS + (k)] = wall_clock64(); \
  } while (0)
#else
#define TR(k)
#endif

// Tuning knobs (tile heights, launch groupings ... the MOBODY_* environment variables the sweeps of DESIGN section 5 were
// run with) exist in the diagnostic build only; the product library never reads the environment.
inline int tune_int(const char* name, int dflt) {
#ifdef MOBODY_TRACE
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

}  // namespace mobody
