// Device-resident replay data movement (all HBM-bound, 4*(2S+A+2) bytes per row each way):
//   k_gather        minibatch assembly: rows idx_k of up to three SoA buffers, concatenated
//                   src | tar | fake          (ReplayBuffer.sample utils.py:127-148 + torch.cat mobody.py:516-529)
//   ring append     stream-compact the kept rows (penalty filter mobody.py:468,648-653) and write them
//                   into the ring with add_batch's single-wrap arithmetic (utils.py:43-92);
//                   three small kernels: per-block scan, scan of block totals, scatter (+ ptr/size update).
#include "common.h"
#include "rng.h"

namespace mobody {

struct GatherArgs {
  MobodyBufferView bufs[3];
  const int32_t* idx[3];    // explicit row indices, or null -> drawn on the fly from the device generator
  long long start[4];       // row offsets of each source inside the output, start[nbuf] = N
  int nbuf, S, A;
  float *state, *action, *next_state, *reward, *not_done;
  // device-RNG mode (idx[k] == null): index i of source k = philox(seed[k], STREAM_SAMPLE, call)[i] * size >> 32,
  // call = (counter ? counter[0] : 0) + call_offset[k], size read from the device word size[k][0]
  uint32_t seed[3];
  long long call_offset[3];
  const long long* counter;
  const long long* size[3];
};

// One ROW per 16-lane group (16 rows per workgroup): the source index -- an explicit index or one Philox draw -- is formed
// once per row by the group's first lane and broadcast, then the group's lanes copy the five SoA pieces of the row with
// consecutive addresses.  (The first version ran one thread per FLOAT: an integer division and a full Philox-10 per
// element, 42 per row at S=17/A=6 -- the kernel was ALU bound at 1.0 TB/s.)
__global__ __launch_bounds__(256) void k_gather(GatherArgs a) {
  const int lane = threadIdx.x & 15;
  const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const long long N = a.start[a.nbuf];
  const bool ok = row < N;
  const long long r = ok ? row : 0;
  int k = 0;
  if (a.nbuf > 1 && r >= a.start[1]) k = 1;
  if (a.nbuf > 2 && r >= a.start[2]) k = 2;
  long long src = 0;
  if (lane == 0) {
    if (a.idx[k] != nullptr) {
      src = a.idx[k][r - a.start[k]];
    } else {
      const uint32_t call = (uint32_t)((a.counter ? a.counter[0] : 0) + a.call_offset[k]);
      const long long sz = a.size[k][0];
      src = rng_index_at(a.seed[k], STREAM_SAMPLE, call, (uint64_t)(r - a.start[k]), (uint32_t)(sz > 0 ? sz : 1));
    }
  }
  src = __shfl(src, threadIdx.x & 48, 64);            // lane 0 of this 16-lane group (groups are 16-aligned inside the wave)
  if (!ok) return;
  const int S = a.S, A = a.A;
  const MobodyBufferView& b = a.bufs[k];
  for (int c = lane; c < S; c += 16) {
    a.state[row * S + c] = b.state[src * S + c];
    a.next_state[row * S + c] = b.next_state[src * S + c];
  }
  for (int c = lane; c < A; c += 16) a.action[row * A + c] = b.action[src * A + c];
  if (lane == 0) a.reward[row] = b.reward[src];
  if (lane == 1) a.not_done[row] = b.not_done[src];
}

// ---- ring append in two launches ------------------------------------------------------------------------------------
//  k_scan_blocks  per-1024-row exclusive scan of the keep flags (pos[i]) and block totals (tops[b]); the LAST block to
//                 finish (atomic ticket) sums the totals, snapshots the ring's old {ptr, size} into the workspace and
//                 commits the new ones (add_batch's single-wrap arithmetic, utils.py:43-92) -- all integer, deterministic.
//  k_ring_scatter one row per 16-lane group: destination = old ptr + (rows kept before this one), wrapped once.
// Workspace (int32): pos[M] | tops[nblocks] | meta[8] = {ticket, K, old_ptr lo/hi, ...}.  The ticket is left at 0.
constexpr int SCAN_BLOCK = 1024;

struct ScanArgs {
  const uint8_t* keep;
  long long M, cap;
  int32_t *pos, *tops, *meta;
  long long* ptr_size;
  int nblocks;
};

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_blocks(ScanArgs a) {
  __shared__ int32_t sm[SCAN_BLOCK];
  __shared__ int last;
  const long long i = (long long)blockIdx.x * SCAN_BLOCK + threadIdx.x;
  const int32_t f = (i < a.M) ? (a.keep ? (a.keep[i] != 0) : 1) : 0;
  sm[threadIdx.x] = f;
  __syncthreads();
  for (int o = 1; o < SCAN_BLOCK; o <<= 1) {        // Hillis-Steele inclusive scan
    const int32_t v = (threadIdx.x >= (unsigned)o) ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += v;
    __syncthreads();
  }
  if (i < a.M) a.pos[i] = sm[threadIdx.x] - f;
  if (threadIdx.x == SCAN_BLOCK - 1) {
    a.tops[blockIdx.x] = sm[threadIdx.x];
    __threadfence();                                 // the total is visible before the ticket is taken
    last = atomicAdd(&a.meta[0], 1) == a.nblocks - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  int32_t s = 0;                                     // this block finished last: K = sum of the block totals
  for (int b = threadIdx.x; b < a.nblocks; b += SCAN_BLOCK) s += a.tops[b];
  __syncthreads();
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = SCAN_BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const long long K = sm[0];
    const long long ptr = a.ptr_size[0], size = a.ptr_size[1];
    a.meta[1] = (int32_t)K;
    a.meta[2] = (int32_t)(ptr & 0xFFFFFFFFLL); a.meta[3] = (int32_t)(ptr >> 32);
    const long long end = (ptr + K < a.cap) ? ptr + K : a.cap;
    const long long used = end - ptr;
    long long nptr = end % a.cap;
    const long long nsize = size + used < a.cap ? size + used : a.cap;
    if (nptr == 0) nptr = K - used;                  // utils.py:74-91
    a.ptr_size[0] = nptr;
    a.ptr_size[1] = nsize;
    a.meta[0] = 0;                                   // ticket ready for the next call
  }
}

struct RingArgs {
  float *b_state, *b_action, *b_next_state, *b_reward, *b_not_done;
  long long cap;
  int S, A;
  const float *obs, *act, *next_obs, *reward;
  const uint8_t *terminal, *keep;
  long long M;
  const int32_t *pos, *tops, *meta;
};

__device__ __forceinline__ long long ring_dst(long long j, long long ptr, long long K, long long cap) {
  const long long end = (ptr + K < cap) ? ptr + K : cap;       // utils.py:67-71
  const long long used = end - ptr;
  return j < used ? ptr + j : j - used;                        // second segment starts at 0 (:82-87)
}

__global__ __launch_bounds__(256) void k_ring_scatter(RingArgs a) {
  __shared__ int32_t sm[4];
  const int lane = threadIdx.x & 15;
  const long long row0 = (long long)blockIdx.x * 16;           // 16 rows per workgroup, all inside one scan block
  const int sb = (int)(row0 / SCAN_BLOCK);
  int32_t part = 0;                                            // rows kept in the scan blocks before this one
  for (int b = threadIdx.x; b < sb; b += 256) part += a.tops[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = part;
  __syncthreads();
  const long long base = (long long)sm[0] + sm[1] + sm[2] + sm[3];
  const long long i = row0 + (threadIdx.x >> 4);
  if (i >= a.M || (a.keep && !a.keep[i])) return;
  const long long K = a.meta[1];
  const long long ptr = ((long long)(uint32_t)a.meta[2]) | ((long long)a.meta[3] << 32);
  const long long d = ring_dst(base + a.pos[i], ptr, K, a.cap);
  const int S = a.S, A = a.A;
  for (int c = lane; c < S; c += 16) {
    a.b_state[d * S + c] = a.obs[i * S + c];
    a.b_next_state[d * S + c] = a.next_obs[i * S + c];
  }
  for (int c = lane; c < A; c += 16) a.b_action[d * A + c] = a.act[i * A + c];
  if (lane == 0) a.b_reward[d] = a.reward[i];
  if (lane == 1) a.b_not_done[d] = 1.f - (float)(a.terminal[i] != 0);
}

int launch_ring_append(float* b_state, float* b_action, float* b_next_state, float* b_reward, float* b_not_done, long long cap,
                       long long* ptr_size, int S, int A, const float* obs, const float* act, const float* next_obs,
                       const float* reward, const uint8_t* terminal, const uint8_t* keep, long long M, int32_t* scan_ws,
                       hipStream_t st) {
  const int nblocks = (int)cdiv(M, SCAN_BLOCK);
  int32_t* pos = scan_ws;
  int32_t* tops = scan_ws + M;
  int32_t* meta = tops + nblocks;
  ScanArgs sa{keep, M, cap, pos, tops, meta, ptr_size, nblocks};
  if (hipMemsetAsync(meta, 0, 8 * sizeof(int32_t), st) != hipSuccess) return fail(MOBODY_E_LAUNCH, "ring append: memset failed");
  hipLaunchKernelGGL(k_scan_blocks, dim3(nblocks), dim3(SCAN_BLOCK), 0, st, sa);
  MB_LAUNCH_OK("k_scan_blocks");
  RingArgs a{b_state, b_action, b_next_state, b_reward, b_not_done, cap, S, A, obs, act, next_obs, reward, terminal, keep, M,
             pos, tops, meta};
  hipLaunchKernelGGL(k_ring_scatter, dim3((unsigned)cdiv(M, 16)), dim3(256), 0, st, a);
  MB_LAUNCH_OK("k_ring_scatter");
  return 0;
}

}  // namespace mobody
using namespace mobody;

extern "C" int mobody_gather_batch(const MobodyBufferView* bufs, const int32_t* const* idx, const int64_t* counts,
                                   int nbuf, int S, int A, float* state, float* action, float* next_state,
                                   float* reward, float* not_done, void* stream) {
  MB_REQUIRE(bufs && idx && counts && nbuf >= 1 && nbuf <= 3, "mobody_gather_batch: need 1..3 source buffers");
  MB_REQUIRE(S >= 1 && A >= 1, "mobody_gather_batch: bad dims");
  GatherArgs a{};
  long long N = 0;
  for (int k = 0; k < nbuf; ++k) {
    MB_REQUIRE(counts[k] >= 0, "mobody_gather_batch: negative count");
    MB_REQUIRE(counts[k] == 0 || (idx[k] && bufs[k].state && bufs[k].action && bufs[k].next_state && bufs[k].reward && bufs[k].not_done),
               "mobody_gather_batch: null pointer in source %d", k);
    a.bufs[k] = bufs[k]; a.idx[k] = idx[k]; a.start[k] = N; N += counts[k];
  }
  a.start[nbuf] = N; a.nbuf = nbuf; a.S = S; a.A = A;
  if (N == 0) return 0;
  MB_REQUIRE(state && action && next_state && reward && not_done, "mobody_gather_batch: null output");
  a.state = state; a.action = action; a.next_state = next_state; a.reward = reward; a.not_done = not_done;
  hipLaunchKernelGGL(k_gather, dim3((unsigned)cdiv(N, 16)), dim3(256), 0, as_stream(stream), a);
  MB_LAUNCH_OK("k_gather");
  return 0;
}

extern "C" int mobody_ring_append(float* b_state, float* b_action, float* b_next_state, float* b_reward,
                                  float* b_not_done, int64_t cap, int64_t* ptr_size, int S, int A, const float* obs,
                                  const float* act, const float* next_obs, const float* reward,
                                  const uint8_t* terminal, const uint8_t* keep, int64_t M, int32_t* scan_ws,
                                  void* stream) {
  MB_REQUIRE(M >= 0 && cap >= 1, "mobody_ring_append: bad sizes");
  if (M == 0) return 0;
  MB_REQUIRE(M <= cap, "mobody_ring_append: batch of %lld rows overflows the ring of %lld twice (add_batch would raise)", (long long)M, (long long)cap);
  MB_REQUIRE(M <= (int64_t)SCAN_BLOCK * SCAN_BLOCK, "mobody_ring_append: at most %d rows per call", SCAN_BLOCK * SCAN_BLOCK);
  MB_REQUIRE(b_state && b_action && b_next_state && b_reward && b_not_done && ptr_size && obs && act && next_obs && reward &&
                 terminal && scan_ws, "mobody_ring_append: null pointer");
  return launch_ring_append(b_state, b_action, b_next_state, b_reward, b_not_done, cap, (long long*)ptr_size, S, A, obs, act,
                            next_obs, reward, terminal, keep, M, scan_ws, as_stream(stream));
}

extern "C" int mobody_gather_batch_rng(const MobodyBufferView* bufs, const int64_t* counts, int nbuf, int S, int A,
                                       const uint32_t* seeds, const int64_t* call_offsets, const int64_t* counter,
                                       const int64_t* const* sizes, float* state, float* action, float* next_state,
                                       float* reward, float* not_done, void* stream) {
  MB_REQUIRE(bufs && counts && seeds && call_offsets && sizes && nbuf >= 1 && nbuf <= 3, "mobody_gather_batch_rng: need 1..3 source buffers");
  MB_REQUIRE(S >= 1 && A >= 1, "mobody_gather_batch_rng: bad dims");
  GatherArgs a{};
  long long N = 0;
  for (int k = 0; k < nbuf; ++k) {
    MB_REQUIRE(counts[k] >= 0, "mobody_gather_batch_rng: negative count");
    MB_REQUIRE(counts[k] == 0 || (sizes[k] && bufs[k].state && bufs[k].action && bufs[k].next_state && bufs[k].reward && bufs[k].not_done),
               "mobody_gather_batch_rng: null pointer in source %d", k);
    a.bufs[k] = bufs[k]; a.idx[k] = nullptr; a.start[k] = N; N += counts[k];
    a.seed[k] = seeds[k]; a.call_offset[k] = call_offsets[k]; a.size[k] = (const long long*)sizes[k];
  }
  a.start[nbuf] = N; a.nbuf = nbuf; a.S = S; a.A = A; a.counter = (const long long*)counter;
  if (N == 0) return 0;
  MB_REQUIRE(state && action && next_state && reward && not_done, "mobody_gather_batch_rng: null output");
  a.state = state; a.action = action; a.next_state = next_state; a.reward = reward; a.not_done = not_done;
  hipLaunchKernelGGL(k_gather, dim3((unsigned)cdiv(N, 16)), dim3(256), 0, as_stream(stream), a);
  MB_LAUNCH_OK("k_gather");
  return 0;
}
