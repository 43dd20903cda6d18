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

__global__ __launch_bounds__(256) void k_gather(GatherArgs a) {
  const int W = 2 * a.S + a.A + 2;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long N = a.start[a.nbuf];
  if (gid >= N * W) return;
  const long long row = gid / W;
  const int c = (int)(gid - row * W);
  int k = 0;
  if (a.nbuf > 1 && row >= a.start[1]) k = 1;
  if (a.nbuf > 2 && row >= a.start[2]) k = 2;
  long long src;
  if (a.idx[k] != nullptr) {
    src = a.idx[k][row - a.start[k]];
  } else {
    const uint32_t call = (uint32_t)((a.counter ? a.counter[0] : 0) + a.call_offset[k]);
    const long long sz = a.size[k][0];
    src = rng_index_at(a.seed[k], STREAM_SAMPLE, call, (uint64_t)(row - a.start[k]), (uint32_t)(sz > 0 ? sz : 1));
  }
  const int S = a.S, A = a.A;
  if (c < S) a.state[row * S + c] = a.bufs[k].state[src * S + c];
  else if (c < S + A) a.action[row * A + (c - S)] = a.bufs[k].action[src * A + (c - S)];
  else if (c < 2 * S + A) a.next_state[row * S + (c - S - A)] = a.bufs[k].next_state[src * S + (c - S - A)];
  else if (c == 2 * S + A) a.reward[row] = a.bufs[k].reward[src];
  else a.not_done[row] = a.bufs[k].not_done[src];
}

// ---- exclusive scan of keep flags: pos[i] = #kept rows before i; tops[b] = kept rows in block b ----
constexpr int SCAN_BLOCK = 1024;

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_blocks(const uint8_t* keep, long long M, int32_t* pos, int32_t* tops) {
  __shared__ int32_t sm[SCAN_BLOCK];
  const long long i = (long long)blockIdx.x * SCAN_BLOCK + threadIdx.x;
  const int32_t f = (i < M) ? (keep ? (keep[i] != 0) : 1) : 0;
  sm[threadIdx.x] = f;
  __syncthreads();
  for (int o = 1; o < SCAN_BLOCK; o <<= 1) {        // Hillis-Steele inclusive scan
    const int32_t v = (threadIdx.x >= (unsigned)o) ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += v;
    __syncthreads();
  }
  if (i < M) pos[i] = sm[threadIdx.x] - f;
  if (threadIdx.x == SCAN_BLOCK - 1) tops[blockIdx.x] = sm[threadIdx.x];
}

// single block: exclusive scan of the block totals in place; tops[nblocks] = total kept
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_tops(int32_t* tops, int nblocks) {
  __shared__ int32_t sm[SCAN_BLOCK];
  const int32_t f = ((int)threadIdx.x < nblocks) ? tops[threadIdx.x] : 0;
  sm[threadIdx.x] = f;
  __syncthreads();
  for (int o = 1; o < SCAN_BLOCK; o <<= 1) {
    const int32_t v = (threadIdx.x >= (unsigned)o) ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += v;
    __syncthreads();
  }
  if ((int)threadIdx.x < nblocks) tops[threadIdx.x] = sm[threadIdx.x] - f;
  if (threadIdx.x == SCAN_BLOCK - 1) tops[nblocks] = sm[threadIdx.x];
}

struct RingArgs {
  float *b_state, *b_action, *b_next_state, *b_reward, *b_not_done;
  long long cap;
  long long* ptr_size;
  int S, A;
  const float *obs, *act, *next_obs, *reward;
  const uint8_t *terminal, *keep;
  long long M;
  const int32_t *pos, *tops;
  int nblocks;
};

__device__ __forceinline__ long long ring_dst(long long j, long long ptr, long long K, long long cap) {
  const long long end = (ptr + K < cap) ? ptr + K : cap;       // utils.py:67-71
  const long long used = end - ptr;
  return j < used ? ptr + j : j - used;                        // second segment starts at 0 (:82-87)
}

__global__ __launch_bounds__(256) void k_ring_scatter(RingArgs a) {
  const int W = 2 * a.S + a.A + 2;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= a.M * W) return;
  const long long i = gid / W;
  const int c = (int)(gid - i * W);
  if (a.keep && !a.keep[i]) return;
  const long long j = (long long)a.pos[i] + a.tops[i / SCAN_BLOCK];
  const long long K = a.tops[a.nblocks];
  const long long d = ring_dst(j, a.ptr_size[0], K, a.cap);
  const int S = a.S, A = a.A;
  if (c < S) a.b_state[d * S + c] = a.obs[i * S + c];
  else if (c < S + A) a.b_action[d * A + (c - S)] = a.act[i * A + (c - S)];
  else if (c < 2 * S + A) a.b_next_state[d * S + (c - S - A)] = a.next_obs[i * S + (c - S - A)];
  else if (c == 2 * S + A) a.b_reward[d] = a.reward[i];
  else a.b_not_done[d] = 1.f - (float)(a.terminal[i] != 0);
}

__global__ void k_ring_commit(long long* ptr_size, long long cap, const int32_t* tops, int nblocks) {
  const long long K = tops[nblocks];
  const long long ptr = ptr_size[0], size = ptr_size[1];
  const long long end = (ptr + K < cap) ? ptr + K : cap;
  const long long used = end - ptr;
  long long nptr = end % cap;
  long long nsize = size + used < cap ? size + used : cap;
  if (nptr == 0) nptr = K - used;                               // utils.py:74-91
  ptr_size[0] = nptr;
  ptr_size[1] = nsize;
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
  const long long total = N * (2 * S + A + 2);
  hipLaunchKernelGGL(k_gather, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream), a);
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
  hipStream_t st = as_stream(stream);
  const int nblocks = (int)cdiv(M, SCAN_BLOCK);
  int32_t* pos = scan_ws;
  int32_t* tops = scan_ws + M;
  hipLaunchKernelGGL(k_scan_blocks, dim3(nblocks), dim3(SCAN_BLOCK), 0, st, keep, (long long)M, pos, tops);
  MB_LAUNCH_OK("k_scan_blocks");
  hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(SCAN_BLOCK), 0, st, tops, nblocks);
  MB_LAUNCH_OK("k_scan_tops");
  RingArgs a{b_state, b_action, b_next_state, b_reward, b_not_done, cap, (long long*)ptr_size, S, A, obs, act, next_obs,
             reward, terminal, keep, M, pos, tops, nblocks};
  const long long total = M * (2 * S + A + 2);
  hipLaunchKernelGGL(k_ring_scatter, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, a);
  MB_LAUNCH_OK("k_ring_scatter");
  hipLaunchKernelGGL(k_ring_commit, dim3(1), dim3(1), 0, st, (long long*)ptr_size, (long long)cap, tops, nblocks);
  MB_LAUNCH_OK("k_ring_commit");
  return 0;
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
  const long long total = N * (2 * S + A + 2);
  hipLaunchKernelGGL(k_gather, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream), a);
  MB_LAUNCH_OK("k_gather");
  return 0;
}
