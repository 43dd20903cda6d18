// Device-resident replay data movement (all HBM-bound, 4*(2S+A+2) bytes per row each way):
//   k_gather        minibatch assembly: rows idx_k of up to three buffers, concatenated
//                   src | tar | fake          (ReplayBuffer.sample utils.py:127-148 + torch.cat mobody.py:516-529)
//   ring append     stream-compact the kept rows (penalty filter mobody.py:468,648-653) and write them
//                   into the ring with add_batch's single-wrap arithmetic (utils.py:43-92);
//                   two kernels: block scan (+ ptr/size commit) and the row scatter.
// Storage layouts (MobodyBufferView.pitch): the ROW-INTERLEAVED ring ("packed": one row = state | action | next_state |
// reward | not_done contiguous, rows `pitch` floats apart, pitch a multiple of 16 floats = 64 bytes) is what the mirror's
// ReplayBuffer allocates -- a random row is then three aligned 64-byte sectors (S=17, A=6: 168 of 192 bytes useful) read or
// written with 16-byte accesses; five separate arrays (pitch 0, the reference's field-per-array shape) cost five random
// pieces of 4..68 bytes per row, three times the fetch requests of the ring at S=17.  Both kernels stage 16 rows per workgroup in LDS so the
// batch side (contiguous [N][S] / [N][A] / [N] arrays) is read and written fully coalesced.
#include <stdlib.h>

#include "common.h"
#include "rng.h"

namespace mobody {

constexpr int ROWS_WG = 16;

__host__ __device__ inline bool view_packed(const MobodyBufferView& b, int S, int A) {
  return b.pitch > 0 && b.pitch % 16 == 0 && b.pitch >= 2LL * S + A + 2 && (reinterpret_cast<uintptr_t>(b.state) & 15) == 0 &&
         b.action == b.state + S && b.next_state == b.action + A && b.reward == b.next_state + S && b.not_done == b.reward + 1;
}

struct GatherArgs {
  MobodyBufferView bufs[3];
  int packed[3];            // view_packed(bufs[k])
  const int32_t* idx[3];    // explicit row indices, or null -> drawn on the fly from the device generator
  long long start[4];       // row offsets of each source inside the output, start[nbuf] = N
  int nbuf, S, A, WS;       // WS = staged floats per row (2S+A+2 rounded up to 4)
  float *state, *action, *next_state, *reward, *not_done;
  // device-RNG mode (idx[k] == null): index i of source k = philox(seed[k], STREAM_SAMPLE, call)[i] * size >> 32,
  // call = (counter ? counter[0] : 0) + call_offset[k], size read from the device word size[k][0]
  uint32_t seed[3];
  long long call_offset[3];
  const long long* counter;
  const long long* size[3];
  long long* bump[4];       // device words incremented by one thread (never `counter`): graph replay advances its step counts here
  int nbump;
};

// e / n for 0 <= e < 2^16 and 1 <= n <= 2^16 through one multiply-high with ceil(2^32 / n) (exact in that range; a runtime
// integer division costs ~20 vector instructions per element of the staging loops)
__host__ __device__ inline uint32_t div_magic(int n) { return (uint32_t)((0x100000000ULL + (uint32_t)n - 1) / (uint32_t)n); }
__device__ __forceinline__ int fast_div(int e, uint32_t magic, int n) { return n == 1 ? e : (int)__umulhi((uint32_t)e, magic); }

// field block f of `rows` staged rows -> contiguous output rows (all 256 threads, consecutive addresses)
__device__ __forceinline__ void stage_to_batch(const float* stage, int WS, int off, int n, int rows, float* out) {
  const uint32_t magic = div_magic(n);
  for (int e = threadIdx.x; e < rows * n; e += 256) {
    const int r = fast_div(e, magic, n), c = e - r * n;
    out[e] = stage[r * WS + off + c];
  }
}

// Source row of output row r (and which buffer it comes from): an explicit index, or one Philox draw.
__device__ __forceinline__ long long gather_src(const GatherArgs& a, long long r, int& k) {
  k = 0;
  if (a.nbuf > 1 && r >= a.start[1]) k = 1;
  if (a.nbuf > 2 && r >= a.start[2]) k = 2;
  if (a.idx[k] != nullptr) return a.idx[k][r - a.start[k]];
  const uint32_t call = (uint32_t)((a.counter ? a.counter[0] : 0) + a.call_offset[k]);
  const long long sz = a.size[k][0];
  return rng_index_at(a.seed[k], STREAM_SAMPLE, call, (uint64_t)(r - a.start[k]), (uint32_t)(sz > 0 ? sz : 1));
}

// Row-interleaved sources (every buffer a packed ring): 16 * P rows per workgroup, P rows per 16-lane group.
//   1. lanes 0 .. P-1 of every 16-lane group form the group's P source rows in parallel (one explicit index or one
//      Philox-10 each) and hand the row pointers to the group by shuffle;
//   2. every 16-lane group issues the 16-byte loads of ALL its P rows (P * NQ independent loads per lane, NQ = 16-byte
//      chunks of a row per lane) before the first LDS write: a random 192-byte row is latency-, not bandwidth-bound, so what
//      counts is rows in flight per CU -- 64-row workgroups x 8 resident = 512 (the one-row-per-group version held 128
//      and reached 2.0 TB/s at a million rows);
//   3. the five output blocks leave LDS coalesced.
// Loads are unconditional from clamped rows / chunks (a conditional load branches and drains vmcnt per element).
template <int P, int NQ>
__global__ __launch_bounds__(256) void k_gather_rows(GatherArgs a) {
  extern __shared__ __attribute__((aligned(16))) float stage[];     // [16 P][WS]
  constexpr int ROWS = ROWS_WG * P;
  const int lane = threadIdx.x & 15, g = threadIdx.x >> 4;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int k = 0; k < a.nbump; ++k) a.bump[k][0] += 1;
  const long long row0 = (long long)blockIdx.x * ROWS;
  const long long N = a.start[a.nbuf];
  const int S = a.S, A = a.A, WS = a.WS, nq = WS >> 2;
  // lane p < P of a group forms the pointer of the group's row p; the group takes it by shuffle (no LDS, no barrier)
  const float* mine = nullptr;
  if (lane < P) {
    const long long r = min(row0 + ROWS_WG * lane + g, N - 1);
    int k;
    const long long src = gather_src(a, r, k);
    mine = a.bufs[k].state + src * a.bufs[k].pitch;
  }
  // (plain vector types and global-address-space pointers: with HIP's float4 struct and a pointer rebuilt from two shuffled
  //  words the compiler emitted flat loads and kept the rows in SCRATCH memory -- 131 us per million rows instead of 85)
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) v4f* gptr_t;
  v4f v[P][NQ];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const unsigned long long up = (unsigned long long)mine;
    const unsigned lo = (unsigned)__shfl((int)(unsigned)up, (threadIdx.x & 48) + p, 64), hi = (unsigned)__shfl((int)(unsigned)(up >> 32), (threadIdx.x & 48) + p, 64);
    gptr_t rp = (gptr_t)(((unsigned long long)hi << 32) | lo);
#pragma unroll
    for (int j = 0; j < NQ; ++j) v[p][j] = rp[min(lane + 16 * j, nq - 1)];
  }
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int j = 0; j < NQ; ++j)
      if (lane + 16 * j < nq) reinterpret_cast<v4f*>(stage + (ROWS_WG * p + g) * WS)[lane + 16 * j] = v[p][j];
  __syncthreads();
  const int rows = (int)min((long long)ROWS, N - row0);
  stage_to_batch(stage, WS, 0, S, rows, a.state + row0 * S);
  stage_to_batch(stage, WS, S, A, rows, a.action + row0 * A);
  stage_to_batch(stage, WS, S + A, S, rows, a.next_state + row0 * S);
  stage_to_batch(stage, WS, 2 * S + A, 1, rows, a.reward + row0);
  stage_to_batch(stage, WS, 2 * S + A + 1, 1, rows, a.not_done + row0);
}

// General form (any mix of row-interleaved rings and the reference's five separate arrays): 16 rows per workgroup, one row
// per 16-lane group; the source index is formed once per row by the group's first lane and broadcast; the group copies its
// row into LDS (packed ring: 16-byte loads of the whole row; separate arrays: the five pieces), then the workgroup writes
// the five output blocks coalesced.  (The first version ran one thread per FLOAT: an integer division and a full
// Philox-10 per element.)
__global__ __launch_bounds__(256) void k_gather(GatherArgs a) {
  extern __shared__ __attribute__((aligned(16))) float stage[];     // [16][WS]
  const int lane = threadIdx.x & 15, g = threadIdx.x >> 4;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int k = 0; k < a.nbump; ++k) a.bump[k][0] += 1;
  const long long row0 = (long long)blockIdx.x * ROWS_WG;
  const long long row = row0 + g;
  const long long N = a.start[a.nbuf];
  const bool ok = row < N;
  const long long r = ok ? row : 0;
  int k = 0;
  long long src = 0;
  if (lane == 0) src = gather_src(a, r, k);
  src = __shfl(src, threadIdx.x & 48, 64);
  k = __shfl(k, threadIdx.x & 48, 64);
  const int S = a.S, A = a.A, WS = a.WS;
  float* mine = stage + g * WS;
  if (ok) {
    const MobodyBufferView& b = a.bufs[k];
    if (a.packed[k]) {
      const float4* rp = reinterpret_cast<const float4*>(b.state + src * b.pitch);
      for (int q = lane; 4 * q < WS; q += 16) reinterpret_cast<float4*>(mine)[q] = rp[q];
    } else {
      const long long ps = b.pitch ? b.pitch : S, pa = b.pitch ? b.pitch : A, p1 = b.pitch ? b.pitch : 1;
      for (int c = lane; c < S; c += 16) {
        mine[c] = b.state[src * ps + c];
        mine[S + A + c] = b.next_state[src * ps + c];
      }
      for (int c = lane; c < A; c += 16) mine[S + c] = b.action[src * pa + c];
      if (lane == 0) mine[2 * S + A] = b.reward[src * p1];
      if (lane == 1) mine[2 * S + A + 1] = b.not_done[src * p1];
    }
  }
  __syncthreads();
  const int rows = (int)min((long long)ROWS_WG, N - row0);
  stage_to_batch(stage, WS, 0, S, rows, a.state + row0 * S);
  stage_to_batch(stage, WS, S, A, rows, a.action + row0 * A);
  stage_to_batch(stage, WS, S + A, S, rows, a.next_state + row0 * S);
  stage_to_batch(stage, WS, 2 * S + A, 1, rows, a.reward + row0);
  stage_to_batch(stage, WS, 2 * S + A + 1, 1, rows, a.not_done + row0);
}

template <int P, int NQ>
static void launch_gather_rows(const GatherArgs& a, long long N, hipStream_t st) {
  const size_t lds = (size_t)ROWS_WG * P * a.WS * sizeof(float);
  hipLaunchKernelGGL((k_gather_rows<P, NQ>), dim3((unsigned)cdiv(N, ROWS_WG * P)), dim3(256), lds, st, a);
}

static int launch_gather(const GatherArgs& a, long long N, hipStream_t st) {
  const size_t lds = (size_t)ROWS_WG * a.WS * sizeof(float);
  if (lds > 64 * 1024) return fail(MOBODY_E_ARG, "gather: rows of %d floats do not fit the LDS stage", a.WS);
  bool all_packed = true;
  for (int k = 0; k < a.nbuf; ++k) all_packed = all_packed && (a.packed[k] || a.start[k + 1] == a.start[k]);
  const int nq = (a.WS / 4 + 15) / 16;               // 16-byte chunks of a row per lane
  if (all_packed && nq <= 4) {
    // rows in flight: four per group for short rows (S = 17: 11 chunks), fewer as the rows grow (ant: 58 chunks, 59 KB of LDS at 64 rows)
    // (measured at a million rows of S = 17: 2 / 4 / 8 rows per group 135 / 115 / 132 us before the shuffle hand-off)
    if (nq == 1) launch_gather_rows<4, 1>(a, N, st);
    else if (nq == 2) launch_gather_rows<2, 2>(a, N, st);
    else if (nq == 3) launch_gather_rows<1, 3>(a, N, st);
    else launch_gather_rows<1, 4>(a, N, st);
    MB_LAUNCH_OK("k_gather_rows");
    return 0;
  }
  hipLaunchKernelGGL(k_gather, dim3((unsigned)cdiv(N, ROWS_WG)), dim3(256), lds, st, a);
  MB_LAUNCH_OK("k_gather");
  return 0;
}

// ---- ring append in two launches ------------------------------------------------------------------------------------
//  k_scan_blocks  per-4096-row exclusive scan of the keep flags (pos[i]) and block totals (tops[b]); the LAST block to
//                 finish (atomic ticket) turns the totals into their exclusive prefix, snapshots the ring's old {ptr, size}
//                 into the workspace and commits the new ones (add_batch's single-wrap arithmetic, utils.py:43-92) -- all
//                 integer, deterministic.
//  k_ring_scatter one row per 16-lane group: destination = old ptr + (rows kept before this one), wrapped once.
// Workspace (int32): meta[8] = {ticket, K, old_ptr lo/hi, ...} | tops[1032] | pos[M].  The ticket must be 0 on entry and is left at 0.
constexpr int SCAN_PER_THREAD = 16;
constexpr int SCAN_BLOCK = 1024 * SCAN_PER_THREAD;   // rows per scan block

struct ScanArgs {
  const uint8_t* keep;
  long long M, cap;
  int32_t *pos, *tops, *meta;
  long long* ptr_size;
  int nblocks;
};

// inclusive scan of one int per thread over the 1024-thread workgroup: wave scans by shuffles, the 16 wave totals in LDS
__device__ __forceinline__ int32_t block_scan_1024(int32_t v, int32_t* wtot) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t u = __shfl_up(v, o, 64);
    if (lane >= o) v += u;
  }
  __syncthreads();                                   // wtot free (previous use read)
  if (lane == 63) wtot[w] = v;
  __syncthreads();
  int32_t before = 0;
#pragma unroll
  for (int k = 0; k < 15; ++k) before += (k < w) ? wtot[k] : 0;
  return v + before;
}

// 1024 threads x 16 consecutive rows = one 16384-row scan block.  One ticket atomic per block: ~40 ns each on one address,
// so a million rows cost 2.5 us of tickets (10 us with 4096-row blocks, 40 us with 1024-row blocks); a thread takes its
// sixteen flags with one 16-byte load and leaves its sixteen positions with four 16-byte stores.
__global__ __launch_bounds__(1024) void k_scan_blocks(ScanArgs a) {
  __shared__ int32_t wtot[16];
  __shared__ int last;
  constexpr int T = SCAN_PER_THREAD;
  const long long i0 = (long long)blockIdx.x * SCAN_BLOCK + T * threadIdx.x;
  int32_t f[T];
  if (a.keep != nullptr && i0 + T <= a.M && ((reinterpret_cast<uintptr_t>(a.keep) & 15) == 0)) {
    const uint4 w = *reinterpret_cast<const uint4*>(a.keep + i0);
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int j = 0; j < T; ++j) f[j] = ((ws[j >> 2] >> (8 * (j & 3))) & 0xffu) != 0;
  } else {
#pragma unroll
    for (int j = 0; j < T; ++j) f[j] = (i0 + j < a.M) ? (a.keep ? (a.keep[i0 + j] != 0) : 1) : 0;
  }
  int32_t mine4 = 0;
#pragma unroll
  for (int j = 0; j < T; ++j) mine4 += f[j];
  const int32_t inc = block_scan_1024(mine4, wtot);
  int32_t run = inc - mine4;
  if (i0 + T <= a.M) {
#pragma unroll
    for (int q = 0; q < T / 4; ++q) {
      int4 o;
      o.x = run; run += f[4 * q]; o.y = run; run += f[4 * q + 1]; o.z = run; run += f[4 * q + 2]; o.w = run; run += f[4 * q + 3];
      reinterpret_cast<int4*>(a.pos + i0)[q] = o;
    }
  } else {
#pragma unroll
    for (int j = 0; j < T; ++j) {
      if (i0 + j < a.M) a.pos[i0 + j] = run;
      run += f[j];
    }
  }
  if (threadIdx.x == 1023) {
    a.tops[blockIdx.x] = inc;
    __threadfence();                                 // the total is visible before the ticket is taken
    last = atomicAdd(&a.meta[0], 1) == a.nblocks - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  // this block finished last: exclusive scan of the block totals in place (nblocks <= 1024 = one per thread), K = their sum
  const int32_t mine = (int)threadIdx.x < a.nblocks ? a.tops[threadIdx.x] : 0;
  const int32_t pre = block_scan_1024(mine, wtot);
  if ((int)threadIdx.x < a.nblocks) a.tops[threadIdx.x] = pre - mine;
  if (threadIdx.x == 1023) {
    const long long K = pre;
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
  MobodyBufferView ring;                                       // written
  int packed;
  long long cap;
  int S, A, WS;                                                // WS = staged floats per row: the pitch (packed) or 2S+A+2 up to 4
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

// contiguous batch rows -> field block of the LDS stage (all 256 threads, consecutive addresses)
__device__ __forceinline__ void batch_to_stage(float* stage, int WS, int off, int n, int rows, const float* in) {
  const uint32_t magic = div_magic(n);
  for (int e = threadIdx.x; e < rows * n; e += 256) {
    const int r = fast_div(e, magic, n), c = e - r * n;
    stage[r * WS + off + c] = in[e];
  }
}

// Two-phase form of batch_to_stage: the first STAGE_U * 256 elements of a field block are LOADED by ld() (all requests of
// all fields go out before any LDS store: a load followed by its dependent store costs one HBM round trip per element, ~11
// of them in sequence per thread for a 64-row tile) and stored by st(); whatever is left of a long block takes the plain loop.
constexpr int STAGE_U = 5;
struct StageRegs { float v[STAGE_U]; };
__device__ __forceinline__ void stage_ld(StageRegs& r, int n, int rows, const float* in) {
  const int total = rows * n;
#pragma unroll
  for (int u = 0; u < STAGE_U; ++u) r.v[u] = in[min((int)threadIdx.x + 256 * u, total - 1)];
}
__device__ __forceinline__ void stage_st(const StageRegs& r, float* stage, int WS, int off, int n, int rows, const float* in) {
  const uint32_t magic = div_magic(n);
  const int total = rows * n;
#pragma unroll
  for (int u = 0; u < STAGE_U; ++u) {
    const int e = threadIdx.x + 256 * u;
    if (e < total) { const int rr = fast_div(e, magic, n); stage[rr * WS + off + (e - rr * n)] = r.v[u]; }
  }
  for (int e = threadIdx.x + 256 * STAGE_U; e < total; e += 256) {
    const int rr = fast_div(e, magic, n);
    stage[rr * WS + off + (e - rr * n)] = in[e];
  }
}

// P = rows per 16-lane group (16 * P rows per workgroup): 4 for long batches of short rows -- 1M-row appends 195 -> ~120 us.
template <int P>
__global__ __launch_bounds__(256) void k_ring_scatter(RingArgs a) {
  extern __shared__ __attribute__((aligned(16))) float stage[];     // [16 * P][WS]
  const int lane = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long long row0 = (long long)blockIdx.x * (ROWS_WG * P);     // all inside one scan block (4096 % (16 P) == 0)
  const int S = a.S, A = a.A, WS = a.WS, W = 2 * S + A + 2;
  const int rows = (int)min((long long)ROWS_WG * P, a.M - row0);
  // everything the destination addresses depend on is requested FIRST (clamped, unconditional), so these round trips
  // overlap the staging of the batch rows instead of following the barrier
  const long long base = a.tops[row0 / SCAN_BLOCK];                  // rows kept in the scan blocks before this one
  const long long K = a.meta[1];
  const long long ptr = ((long long)(uint32_t)a.meta[2]) | ((long long)a.meta[3] << 32);
  int32_t pos[P];
  bool kept[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const long long i = row0 + ROWS_WG * p + g, ic = min(i, a.M - 1);
    pos[p] = a.pos[ic];
    kept[p] = i < a.M && (a.keep == nullptr || a.keep[ic] != 0);
  }
  StageRegs rs, ra, rn;
  stage_ld(rs, S, rows, a.obs + row0 * S);
  stage_ld(ra, A, rows, a.act + row0 * A);
  stage_ld(rn, S, rows, a.next_obs + row0 * S);
  const int tr = min((int)threadIdx.x, rows - 1);
  const float rew = a.reward[row0 + tr];
  const float ndone = 1.f - (float)(a.terminal[row0 + tr] != 0);
  stage_st(rs, stage, WS, 0, S, rows, a.obs + row0 * S);
  stage_st(ra, stage, WS, S, A, rows, a.act + row0 * A);
  stage_st(rn, stage, WS, S + A, S, rows, a.next_obs + row0 * S);
  if ((int)threadIdx.x < rows) { stage[threadIdx.x * WS + 2 * S + A] = rew; stage[threadIdx.x * WS + 2 * S + A + 1] = ndone; }
  for (int e = threadIdx.x; e < rows * (WS - W); e += 256) {   // padding of the row: zeros (whole sectors are written)
    const int r = e / (WS - W), c = e - r * (WS - W);
    stage[r * WS + W + c] = 0.f;
  }
  __syncthreads();
  const MobodyBufferView& b = a.ring;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (!kept[p]) continue;
    const long long d = ring_dst(base + pos[p], ptr, K, a.cap);
    const float* mine = stage + (ROWS_WG * p + g) * WS;
    if (a.packed) {
      float4* rp = reinterpret_cast<float4*>(const_cast<float*>(b.state) + d * b.pitch);
      for (int q = lane; 4 * q < WS; q += 16) rp[q] = reinterpret_cast<const float4*>(mine)[q];
    } else {
      const long long ps = b.pitch ? b.pitch : S, pa = b.pitch ? b.pitch : A, p1 = b.pitch ? b.pitch : 1;
      float* bs = const_cast<float*>(b.state); float* bn = const_cast<float*>(b.next_state); float* ba = const_cast<float*>(b.action);
      for (int c = lane; c < S; c += 16) {
        bs[d * ps + c] = mine[c];
        bn[d * ps + c] = mine[S + A + c];
      }
      for (int c = lane; c < A; c += 16) ba[d * pa + c] = mine[S + c];
      if (lane == 0) const_cast<float*>(b.reward)[d * p1] = mine[2 * S + A];
      if (lane == 1) const_cast<float*>(b.not_done)[d * p1] = mine[2 * S + A + 1];
    }
  }
}

int launch_ring_append(const MobodyBufferView& ring, long long cap, long long* ptr_size, int S, int A, const float* obs,
                       const float* act, const float* next_obs, const float* reward, const uint8_t* terminal,
                       const uint8_t* keep, long long M, int32_t* scan_ws, hipStream_t st) {
  const int nblocks = (int)cdiv(M, SCAN_BLOCK);
  // meta[8] | tops[<= 1024 (+8 pad)] | pos[M].  meta[0] is the scan's arrival ticket: zero on entry (the caller zeroes the
  // workspace once, when it allocates it), left at zero by the block that arrives last -- no memset launch per append.
  int32_t* meta = scan_ws;
  int32_t* tops = scan_ws + 8;
  int32_t* pos = scan_ws + 1040;
  ScanArgs sa{keep, M, cap, pos, tops, meta, ptr_size, nblocks};
  hipLaunchKernelGGL(k_scan_blocks, dim3(nblocks), dim3(1024), 0, st, sa);
  MB_LAUNCH_OK("k_scan_blocks");
  const int packed = view_packed(ring, S, A);
  const int WS = packed ? (int)ring.pitch : (2 * S + A + 2 + 3) & ~3;
  const int P = M >= 32768 && WS <= 64 ? 4 : 1;     // (8 rows per group: 118 us against 107 us per million rows)
  const size_t lds = (size_t)ROWS_WG * P * WS * sizeof(float);
  if (lds > 64 * 1024) return fail(MOBODY_E_ARG, "ring append: rows of %d floats do not fit the LDS stage", WS);
  RingArgs a{ring, packed, cap, S, A, WS, obs, act, next_obs, reward, terminal, keep, M, pos, tops, meta};
  if (P == 4) hipLaunchKernelGGL(k_ring_scatter<4>, dim3((unsigned)cdiv(M, ROWS_WG * 4)), dim3(256), lds, st, a);
  else hipLaunchKernelGGL(k_ring_scatter<1>, dim3((unsigned)cdiv(M, ROWS_WG)), dim3(256), lds, st, a);
  MB_LAUNCH_OK("k_ring_scatter");
  return 0;
}

static int check_view(const char* who, const MobodyBufferView& b, int S, int A) {
  MB_REQUIRE(b.state && b.action && b.next_state && b.reward && b.not_done, "%s: null pointer in a buffer view", who);
  MB_REQUIRE(b.pitch == 0 || b.pitch >= S, "%s: pitch %lld smaller than a state row", who, (long long)b.pitch);
  return 0;
}

}  // namespace mobody
using namespace mobody;

extern "C" int64_t mobody_ring_pitch(int S, int A) { return S >= 1 && A >= 1 ? ((2LL * S + A + 2 + 15) / 16) * 16 : -1; }

static int gather_common(const char* who, GatherArgs& a, const MobodyBufferView* bufs, const int64_t* counts, int nbuf, int S, int A,
                         float* state, float* action, float* next_state, float* reward, float* not_done, long long& N) {
  MB_REQUIRE(S >= 1 && A >= 1, "%s: bad dims", who);
  N = 0;
  for (int k = 0; k < nbuf; ++k) {
    MB_REQUIRE(counts[k] >= 0, "%s: negative count", who);
    if (counts[k] > 0) { int rc = check_view(who, bufs[k], S, A); if (rc) return rc; }
    a.bufs[k] = bufs[k]; a.packed[k] = view_packed(bufs[k], S, A); a.start[k] = N; N += counts[k];
  }
  a.start[nbuf] = N; a.nbuf = nbuf; a.S = S; a.A = A; a.WS = (2 * S + A + 2 + 3) & ~3;
  if (N == 0) return 0;
  MB_REQUIRE(state && action && next_state && reward && not_done, "%s: null output", who);
  a.state = state; a.action = action; a.next_state = next_state; a.reward = reward; a.not_done = not_done;
  return 0;
}

extern "C" int mobody_gather_batch(const MobodyBufferView* bufs, const int32_t* const* idx, const int64_t* counts,
                                   int nbuf, int S, int A, float* state, float* action, float* next_state,
                                   float* reward, float* not_done, void* stream) {
  MB_REQUIRE(bufs && idx && counts && nbuf >= 1 && nbuf <= 3, "mobody_gather_batch: need 1..3 source buffers");
  GatherArgs a{};
  long long N;
  int rc = gather_common("mobody_gather_batch", a, bufs, counts, nbuf, S, A, state, action, next_state, reward, not_done, N);
  if (rc || N == 0) return rc;
  for (int k = 0; k < nbuf; ++k) {
    MB_REQUIRE(counts[k] == 0 || idx[k], "mobody_gather_batch: null index array of source %d", k);
    a.idx[k] = idx[k];
  }
  return launch_gather(a, N, as_stream(stream));
}

extern "C" int mobody_ring_append(const MobodyBufferView* ring, int64_t cap, int64_t* ptr_size, int S, int A, const float* obs,
                                  const float* act, const float* next_obs, const float* reward,
                                  const uint8_t* terminal, const uint8_t* keep, int64_t M, int32_t* scan_ws,
                                  void* stream) {
  MB_REQUIRE(M >= 0 && cap >= 1, "mobody_ring_append: bad sizes");
  if (M == 0) return 0;
  MB_REQUIRE(M <= cap, "mobody_ring_append: batch of %lld rows overflows the ring of %lld twice (add_batch would raise)", (long long)M, (long long)cap);
  MB_REQUIRE(M <= (int64_t)SCAN_BLOCK * 1024, "mobody_ring_append: at most %d rows per call", SCAN_BLOCK * 1024);
  MB_REQUIRE(ring && ptr_size && obs && act && next_obs && reward && terminal && scan_ws, "mobody_ring_append: null pointer");
  MB_REQUIRE(S >= 1 && A >= 1, "mobody_ring_append: bad dims");
  int rc = check_view("mobody_ring_append", *ring, S, A);
  if (rc) return rc;
  return launch_ring_append(*ring, cap, (long long*)ptr_size, S, A, obs, act, next_obs, reward, terminal, keep, M, scan_ws,
                            as_stream(stream));
}

extern "C" int mobody_gather_batch_rng(const MobodyBufferView* bufs, const int64_t* counts, int nbuf, int S, int A,
                                       const uint32_t* seeds, const int64_t* call_offsets, const int64_t* counter,
                                       const int64_t* const* sizes, float* state, float* action, float* next_state,
                                       float* reward, float* not_done, int64_t* const* bump, int nbump, void* stream) {
  MB_REQUIRE(bufs && counts && seeds && call_offsets && sizes && nbuf >= 1 && nbuf <= 3, "mobody_gather_batch_rng: need 1..3 source buffers");
  MB_REQUIRE(nbump >= 0 && nbump <= 4 && (nbump == 0 || bump), "mobody_gather_batch_rng: at most 4 words to advance");
  GatherArgs a{};
  long long N;
  int rc = gather_common("mobody_gather_batch_rng", a, bufs, counts, nbuf, S, A, state, action, next_state, reward, not_done, N);
  if (rc || N == 0) return rc;
  for (int k = 0; k < nbuf; ++k) {
    MB_REQUIRE(counts[k] == 0 || sizes[k], "mobody_gather_batch_rng: null size word of source %d", k);
    a.idx[k] = nullptr; a.seed[k] = seeds[k]; a.call_offset[k] = call_offsets[k]; a.size[k] = (const long long*)sizes[k];
  }
  a.counter = (const long long*)counter;
  for (int k = 0; k < nbump; ++k) {
    MB_REQUIRE(bump[k] && bump[k] != counter, "mobody_gather_batch_rng: bump word %d is null or the call counter itself", k);
    a.bump[k] = (long long*)bump[k];
  }
  a.nbump = nbump;
  return launch_gather(a, N, as_stream(stream));
}
