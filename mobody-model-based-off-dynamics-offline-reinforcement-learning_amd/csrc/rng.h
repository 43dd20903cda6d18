// Philox4x32-10 counter-based generator (Salmon et al., SC'11) -- device side.
// CPU twin: oracle/mobody_oracle.py philox4x32 / rng_normal / rng_index (known-answer tested).
// Replaces, in throughput mode, the reference's three host/device RNG streams
// (torch.normal mobody_dynamics.py:220, np.random.choice mobody_module.py:355-357,
//  np.random.randint utils.py:128); parity mode passes explicit noise/indices instead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mobody {

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

__host__ __device__ inline float rng_u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f; }

// element i of the unit-normal stream (seed, stream_id, call): counter (i>>2, call, 0, 0), Box-Muller lane i&3
__device__ inline float rng_normal_at(uint32_t seed, uint32_t stream_id, uint32_t call, uint64_t i) {
  const U4 r = philox4x32_10((uint32_t)(i >> 2), call, 0u, 0u, seed, stream_id);
  const int lane = (int)(i & 3);
  const float u1 = rng_u01(lane < 2 ? r.x : r.z);
  const float u2 = rng_u01(lane < 2 ? r.y : r.w);
  const float rad = sqrtf(-2.0f * logf(u1));
  const float ang = 6.283185307179586f * u2;
  return (lane & 1) ? rad * sinf(ang) : rad * cosf(ang);
}

// element i of the index stream: word i&3 of counter (i>>2, call, 0, 0); (x*bound)>>32
__device__ inline uint32_t rng_index_at(uint32_t seed, uint32_t stream_id, uint32_t call, uint64_t i, uint32_t bound) {
  const U4 r = philox4x32_10((uint32_t)(i >> 2), call, 0u, 0u, seed, stream_id);
  const int lane = (int)(i & 3);
  const uint32_t x = lane == 0 ? r.x : lane == 1 ? r.y : lane == 2 ? r.z : r.w;
  return (uint32_t)(((uint64_t)x * bound) >> 32);
}

constexpr uint32_t STREAM_NOISE = 1, STREAM_ELITE = 2, STREAM_SAMPLE = 3;

}  // namespace mobody
