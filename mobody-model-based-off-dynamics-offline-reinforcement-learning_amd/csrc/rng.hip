// Stand-alone draws from the device generator (used by tests to read back exactly the
// noise / indices the fused kernels consume, and by the host mirror for replay sampling).
#include "common.h"
#include "rng.h"

namespace mobody {

__global__ void k_rng_normal(uint32_t seed, uint32_t sid, uint32_t call, long long n, float* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = rng_normal_at(seed, sid, call, (uint64_t)i);
}
__global__ void k_rng_index(uint32_t seed, uint32_t sid, uint32_t call, long long n, uint32_t bound, int32_t* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)rng_index_at(seed, sid, call, (uint64_t)i, bound);
}

}  // namespace mobody
using namespace mobody;

extern "C" int mobody_rng_normal(uint32_t seed, uint32_t stream_id, uint32_t call, int64_t n, float* out, void* stream) {
  MB_REQUIRE(out && n >= 0, "mobody_rng_normal: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_rng_normal, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), seed, stream_id, call, (long long)n, out);
  MB_LAUNCH_OK("k_rng_normal");
  return 0;
}

extern "C" int mobody_rng_index(uint32_t seed, uint32_t stream_id, uint32_t call, int64_t n, uint32_t bound, int32_t* out,
                                void* stream) {
  MB_REQUIRE(out && n >= 0 && bound >= 1, "mobody_rng_index: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_rng_index, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), seed, stream_id, call, (long long)n, bound, out);
  MB_LAUNCH_OK("k_rng_index");
  return 0;
}

namespace mobody {
__global__ void k_sample_indices(uint32_t seed, uint32_t sid, const long long* counter, long long call_offset, long long n,
                                 const long long* size, int32_t* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t call = (uint32_t)((counter ? counter[0] : 0) + call_offset);
  const long long sz = size[0];
  out[i] = (int32_t)rng_index_at(seed, sid, call, (uint64_t)i, (uint32_t)(sz > 0 ? sz : 1));
}
__global__ void k_counter_add(long long* c, int n, long long inc) { if ((int)threadIdx.x < n) c[threadIdx.x] += inc; }
}  // namespace mobody

extern "C" int mobody_sample_indices(uint32_t seed, uint32_t stream_id, const int64_t* counter, int64_t call_offset, int64_t n,
                                     const int64_t* size, int32_t* out, void* stream) {
  MB_REQUIRE(n >= 0, "mobody_sample_indices: n < 0");
  if (n == 0) return 0;
  MB_REQUIRE(size && out, "mobody_sample_indices: null pointer");
  hipLaunchKernelGGL(k_sample_indices, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), seed, stream_id,
                     (const long long*)counter, (long long)call_offset, (long long)n, (const long long*)size, out);
  MB_LAUNCH_OK("k_sample_indices");
  return 0;
}

extern "C" int mobody_counter_add(int64_t* counter, int n, int64_t inc, void* stream) {
  MB_REQUIRE(counter, "mobody_counter_add: null pointer");
  MB_REQUIRE(n >= 1 && n <= 64, "mobody_counter_add: n=%d outside 1..64", n);
  hipLaunchKernelGGL(k_counter_add, dim3(1), dim3(64), 0, as_stream(stream), (long long*)counter, n, (long long)inc);
  MB_LAUNCH_OK("k_counter_add");
  return 0;
}
