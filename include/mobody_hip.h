/* mobody_hip.h -- C ABI of the MI355X-native MOBODY hot path (libmobody_hip.so, gfx950).
 *
 * The reference (github guoyihonggyh/MOBODY-...) is pure Python/PyTorch and has no FFI
 * of its own; these entry points are what a binding for its hot path binds instead of
 * the ATen op sequences at the cited reference sites (paths relative to the reference):
 *
 *   mobody_dyn_forward   <- MOBODYModule.forward_trg/forward_src  algo/dynamics/mobody_module.py:315-330
 *   mobody_dyn_step      <- MOBODYEnsembleDynamics.step           algo/dynamics/mobody_dynamics.py:193-265
 *                           (+ termination predicates             algo/mb_utils/terminal_funs.py:10-149)
 *   mobody_mlp3_forward  <- Policy / DoubleQFunc / ValueFunc fwd  algo/offline_offline/mobody.py:35-83
 *   mobody_rollout       <- MOBODY.rollout + add_batch            algo/offline_offline/mobody.py:596-657, algo/utils.py:43-92
 *   mobody_gather_batch  <- ReplayBuffer.sample x3 + torch.cat    algo/utils.py:127-148, mobody.py:399-400,516-529
 *   mobody_ring_append   <- ReplayBuffer.add_batch (+ filter)     algo/utils.py:43-92, mobody.py:468,648-653
 *   mobody_critic_step   <- update_q_functions + backward         mobody.py:189-208,544-547
 *   mobody_actor_forward / mobody_actor_backward
 *                        <- update_policy + bc_loss + backward    mobody.py:246-276,314-345,555-572
 *   mobody_adam_polyak   <- Adam.step + update_target             mobody.py:127-131,183-187,548,552,573
 *   mobody_pretrain_*    <- MOBODYEnsembleDynamics.learn / validate  algo/dynamics/mobody_dynamics.py:300-384,594-653,1113-1140
 *   mobody_rng_*         <- torch.normal / np.random.choice / np.random.randint draws
 *                           (mobody_dynamics.py:220, mobody_module.py:355-357, utils.py:128)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch `tensor.data_ptr()`), fp32
 *     row-major contiguous unless stated; the caller owns every buffer; the library
 *     allocates nothing and keeps no state between calls (no context object needed);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no call
 *     synchronises; calls are graph-capturable;
 *   - every function returns 0 on success, a negative MOBODY_E* code otherwise and
 *     never throws; mobody_last_error() gives the text (thread local);
 *   - not thread safe per stream: one caller per stream.
 *   - network weights are passed as PACKED blobs whose layout is computed by
 *     mobody_dyn_layout / mobody_mlp_layout (zero padded [K_pad][N_pad] per member; matrices with
 *     N_pad == 256 are stored K-interleaved by four, element (k,n) at ((k/4)*256 + n)*4 + k%4, narrow
 *     ones row major); the host-side mirror (mobody_amd/packing.py) packs/unpacks the reference's
 *     state_dict tensors;
 *   - every launch of a call is ordered on the caller's `stream`; the library owns no stream of its own.
 */
#ifndef MOBODY_HIP_H
#define MOBODY_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOBODY_ABI_VERSION 6
#define MOBODY_E_ARG (-1)      /* bad argument (dims, null pointer, unsupported size) */
#define MOBODY_E_LAUNCH (-2)   /* hipLaunch / runtime error */
#define MOBODY_E_UNSUPPORTED (-3)

#define MOBODY_HIDDEN 256
#define MOBODY_ENSEMBLE 7
#define MOBODY_LATENT 16

/* termination predicate ids (terminal_funs.py:123-149 dispatch, resolved on the host) */
enum { MOBODY_TERM_NEVER = 0, MOBODY_TERM_HALFCHEETAH = 1, MOBODY_TERM_HOPPER = 2, MOBODY_TERM_ANT = 3,
       MOBODY_TERM_WALKER2D = 4, MOBODY_TERM_HUMANOID = 5, MOBODY_TERM_PEN = 6 };

/* ---- packed layouts -------------------------------------------------------------------- */
typedef struct MobodyLayer {
  int32_t in_dim, out_dim;   /* logical dims actually used by the hot path */
  int32_t Kp, Np;            /* padded dims of the packed matrix W[member][Kp][Np] */
  int64_t w_off, b_off;      /* float offsets of member 0 inside the blob; bias is [member][Np] */
} MobodyLayer;

enum { MOBODY_DL_ZS1 = 0, MOBODY_DL_ZS2, MOBODY_DL_ZS3, MOBODY_DL_ZA_SRC1, MOBODY_DL_ZA_SRC2, MOBODY_DL_ZA_TRG1,
       MOBODY_DL_ZA_TRG2, MOBODY_DL_TR1, MOBODY_DL_TR2, MOBODY_DL_TR3, MOBODY_DL_RW1, MOBODY_DL_RW2, MOBODY_DL_RW3,
       MOBODY_DL_COUNT };

typedef struct MobodyDynLayout {
  int32_t S, A, E, _pad;
  MobodyLayer layer[MOBODY_DL_COUNT];
  int64_t total_floats;
} MobodyDynLayout;

/* 3-layer MLP (in -> 256 -> 256 -> out), `members` independent copies (twin-Q = 2).
 * Parameter blob, per member: W1[Kp1][256] b1[256] W2[256][256] b2[256] W3[256][Np3] b3[Np3]
 * (W stored [in][out], i.e. the transpose of nn.Linear.weight).  The same layout is used for
 * gradients and both Adam moments.  The "T" blob holds the transposes the backward pass
 * streams as MFMA B operands: W3T[Np3][256] W2T[256][256] W1T[256][Np1t] per member, followed by the 16-bit planes of
 * W2 and W2^T that the split-precision modes stream.  Precision modes (`precision` arguments, MobodyHyper.precision):
 *   0 "f32"     exact fp32 MFMA (v_mfma_f32_32x32x2_f32), the reference's arithmetic
 *   1 "bf16"    one bf16 plane, one product (~3e-3)
 *   2 "bf16x2"  two bf16 planes, three products (~6e-6)
 *   3 "bf16x3"  three bf16 planes, six products: fp32-grade
 *   4 "f16x2"   two fp16 planes (w * 2^8 in the weight planes, per-tile power-of-two scales on the activation side),
 *               three products: fp32-grade at half the MFMA work of bf16x3
 * Modes 0-3 share one plane format (three bf16 planes); mode 4 keeps its two fp16 planes in the first two plane slots.
 * Whoever writes planes (mobody_mlp_transpose, the Adam entry points, mobody_dyn_planes) is told the mode. */
typedef struct MobodyMlpLayout {
  int32_t in_dim, out_dim, members, Kp1, Np3, Np1t;
  int64_t w1, b1, w2, b2, w3, b3;   /* float offsets inside one member */
  int64_t member_floats, total_floats;
  int64_t w3t, w2t, w1t;            /* float offsets inside one member of the T blob */
  int64_t t_member_floats, t_total_floats;
  int64_t w2p, w2tp;                /* float offsets (inside one member of the T blob) of the planes of W2 and W2^T:
                                       [3 planes][32][256][8] 16-bit each, x = x0 + x1 (+ x2) (split-precision modes) */
} MobodyMlpLayout;

const char* mobody_last_error(void);
int mobody_abi_version(void);
int mobody_dyn_layout(int S, int A, MobodyDynLayout* out);
int mobody_mlp_layout(int in_dim, int out_dim, int members, MobodyMlpLayout* out);

/* ---- optional per-kernel timing (measurement only; the one piece of process-global state) ----
 * Between prof_begin and prof_end every launch of the heavy kernel families is bracketed by a HIP
 * event pair on its launch stream.  prof_end SYNCHRONISES, then returns the summed milliseconds
 * and launch counts per family id: 0 mlp3_fwd, 1 mlp3_bwd, 2 wgrad, 3 dyn_fwd (4..7 reserved). */
#define MOBODY_PROF_IDS 8
int mobody_prof_begin(int max_events);
int mobody_prof_end(double* ms_by_id, int64_t* count_by_id, int n_ids);

/* ---- counter based RNG (Philox4x32-10; CPU twin: oracle/mobody_oracle.py rng_*) ---------- */
int mobody_rng_normal(uint32_t seed, uint32_t stream_id, uint32_t call, int64_t n, float* out, void* stream);
int mobody_rng_index(uint32_t seed, uint32_t stream_id, uint32_t call, int64_t n, uint32_t bound, int32_t* out,
                     void* stream);

/* ---- ensemble dynamics ----------------------------------------------------------------- */
/* mean[E][B][S] = forward_trg/forward_src(obs, act) in inference mode. */
int mobody_dyn_forward(const float* dyn_blob, const float* dyn_planes, int precision, int S, int A, const float* obs,
                       const float* act, int64_t B, int use_trg, float* mean, void* stream);
/* dyn_planes / precision (here and in mobody_dyn_step / mobody_rollout): 0 = exact fp32 MFMA (planes may be NULL);
 * 1 bf16, 2 bf16x2, 3 bf16x3, 4 f16x2 = the three 256 x 256 layers of every member (zs2, transition2, reward_model2) on the
 * split-precision core, streaming the planes mobody_dyn_planes built from the blob FOR THAT MODE
 * (mobody_dyn_planes_floats() floats: [3 layers][7 members][3 planes][65536] 16-bit). */
int64_t mobody_dyn_planes_floats(void);
int mobody_dyn_planes(const float* dyn_blob, int S, int A, float* planes, int precision, void* stream);

/* floats of scratch mobody_dyn_step needs for a batch of B rows */
int64_t mobody_dyn_step_workspace(int S, int A, int64_t B);

/* One imagined transition for B rows (A.1 of SURVEY.md).
 *   noise      [E][B][S] unit normals, or NULL -> generated on device from (seed, call + call_dev[0])
 *   call_dev   optional device int64 word added to `call` (NULL = 0): a captured HIP graph keeps its step counter on the
 *              device, so every replay draws fresh noise without new kernel arguments
 *   elite_idx  [B] member id per row,   or NULL -> elites[philox % n_elites]
 *   alive      [B] optional uint8 mask (NULL = all alive); dead rows are computed but flagged
 *              terminal=1 so an on-device multi-step rollout can keep fixed row indices
 *   elites     HOST array of n_elites member ids (MOBODYModule.elites), used when elite_idx is NULL
 *   outputs    next_obs[B][S], reward[B], terminal[B] (uint8), penalty[B], raw_reward[B] (nullable)
 *   mean_out   optional [E][B][S] copy of the ensemble means (info['samples']) */
int mobody_dyn_step(const float* dyn_blob, const float* dyn_planes, int precision, int S, int A, int task,
                    const float* obs, const float* act, int64_t B,
                    const float* noise, const int32_t* elite_idx, const uint8_t* alive, const int32_t* elites,
                    int n_elites, uint32_t seed,
                    uint32_t call, const int64_t* call_dev, float penalty_coef, int use_penalty, int use_trg, float* next_obs,
                    float* reward,
                    uint8_t* terminal, float* penalty, float* raw_reward, float* mean_out, float* workspace,
                    void* stream);

/* The same step for the MOPO ablation (config['mopo'] = 1, mobody_module.py:114-118,218-219,251-254,264-266,288-289): the
 * ensemble means are obs + MLP_e([obs, act]) with the 7-member Swish MLP za_src1..3 (S+A -> 256 -> 256 -> S) packed as
 * mobody_mlp_layout(S + A, S, 7) (mopo_blob; mopo_blob_T from mobody_mlp_transpose, needed when precision != 0);
 * encoders / decoder are bypassed and forward_trg == forward_src, so there is no use_trg.  The reward head is read
 * from dyn_blob exactly as in mobody_dyn_step; every other argument has the same meaning. */
int mobody_mopo_step(const float* dyn_blob, const float* dyn_planes, const float* mopo_blob, const float* mopo_blob_T,
                     int precision, int S, int A, int task, const float* obs, const float* act, int64_t B,
                     const float* noise, const int32_t* elite_idx, const uint8_t* alive, const int32_t* elites,
                     int n_elites, uint32_t seed, uint32_t call, float penalty_coef, int use_penalty, float* next_obs,
                     float* reward, uint8_t* terminal, float* penalty, float* raw_reward, float* mean_out,
                     float* workspace, void* stream);

/* ---- replay buffer views (used by the rollout below and by the gather / append entry points) ---- */
typedef struct MobodyBufferView {   /* ReplayBuffer fields, algo/utils.py:19-23 */
  const float* state; const float* action; const float* next_state; const float* reward; const float* not_done;
  int64_t pitch;   /* floats between consecutive rows of EVERY field; 0 = five separate contiguous arrays
                      ([rows][S], [rows][A], [rows][S], [rows], [rows]: the reference's shape) */
} MobodyBufferView;
/* Row-interleaved ring (what the mirror's ReplayBuffer allocates): one allocation of [rows][pitch] floats with
 * pitch = mobody_ring_pitch(S, A) = 2S+A+2 rounded up to 16 floats (64 bytes), a row laid out
 * state[S] | action[A] | next_state[S] | reward | not_done | zero padding, i.e. state = base, action = base + S,
 * next_state = base + S + A, reward = base + 2S + A, not_done = reward + 1.  A view of exactly this shape with a
 * 16-byte aligned base takes the kernels' fast path (a random row = whole aligned 64-byte sectors, 16-byte accesses);
 * any other view (separate arrays, other pitches) is handled piecewise. */
int64_t mobody_ring_pitch(int S, int A);

/* The whole H-step imagined rollout on the device, appended to a ring buffer (MOBODY.rollout + add_batch,
 * mobody.py:596-657, utils.py:43-92): per step a = pi(s) (actor blob, mobody_mlp_layout(S, A, 1)), one fused ensemble
 * step with device-Philox noise / elite picks at call id call0 + t, the penalty filter (`penalty <= env_filter` when
 * filter_bad_rollout) and the alive update formed in the sample kernel, and a two-launch stream-compacting append of the
 * kept rows: 7 launches per step, no host work and no synchronisation between steps.  Rows keep their index across
 * steps; a row that terminated is computed but never appended again (the reference drops it from the batch, :635-639).
 * NB quirk Q1: MOBODY.rollout passes its `use_trg` argument in step()'s `use_penalty` slot -- callers mirror that here.
 * workspace: mobody_rollout_workspace(S, A, B) floats. */
int64_t mobody_rollout_workspace(int S, int A, int64_t B);
int mobody_rollout(const float* dyn_blob, const float* dyn_planes, const float* actor_blob, const float* actor_blob_T,
                   int precision, int S, int A, int task, float max_action,
                   const float* init_obs, int64_t B, int H, const int32_t* elites, int n_elites, uint32_t seed, uint32_t call0,
                   float penalty_coef, int use_penalty, int use_trg, float env_filter, int filter_bad_rollout,
                   const MobodyBufferView* ring, int64_t cap, int64_t* ptr_size, float* workspace, void* stream);

/* Termination predicate alone: done[B] (uint8) = terminal_fn(next_obs[B][S])  (terminal_funs.py:10-121). */
int mobody_termination(int task, const float* next_obs, int64_t B, int S, uint8_t* done, void* stream);

/* Bookkeeping of an on-device multi-step rollout (MOBODY.rollout mobody.py:635-653 without host
 * compaction): keep[b] = alive_in[b] && (!use_filter || penalty[b] <= env_filter);
 * alive_out[b] = alive_in[b] && !terminal[b].  alive_in == NULL means all alive; alive_out may alias alive_in. */
int mobody_rollout_mask(const uint8_t* alive_in, const uint8_t* terminal, const float* penalty, float env_filter,
                        int use_filter, int64_t B, uint8_t* keep, uint8_t* alive_out, void* stream);

/* Replay indices drawn on the device (np.random.randint(0, size, n), utils.py:128, throughput mode):
 * out[i] = philox(seed, stream_id, call = (uint32)(counter[0] + call_offset))[i] % size[0]-ish
 * (multiply-high), with `counter` and `size` read from DEVICE int64 words so a captured graph
 * advances without new kernel arguments.  size[0] must be >= 1. */
int mobody_sample_indices(uint32_t seed, uint32_t stream_id, const int64_t* counter, int64_t call_offset, int64_t n,
                          const int64_t* size, int32_t* out, void* stream);

/* counter[i] += inc for i < n (device int64 words, n <= 64): one launch advances the RNG call id and both Adam
 * step counts of a captured train() step */
int mobody_counter_add(int64_t* counter, int n, int64_t inc, void* stream);

/* ---- 3-layer MLP forward (ReLU) -------------------------------------------------------- */
/* x = concat(src0[rows][n0], src1[rows][n1]) (src1 may be NULL), n0+n1 == in_dim.
 * out_mode 0: out[m][rows][out_dim] raw;  1: max_action*tanh(.) (Policy.forward mobody.py:68-72).
 * save_x [rows][Kp1], save_h1/save_h2 [members][rows][256] are optional (backward inputs). */
int mobody_mlp3_forward(const float* blob, const float* blob_T, int precision, int in_dim, int out_dim, int members,
                        const float* src0, int n0, const float* src1, int n1, int64_t rows, int out_mode,
                        float max_action, float* out, float* save_x, float* save_h1, float* save_h2, void* stream);
/* blob_T / precision: 0 = exact fp32 MFMA (blob_T may be NULL); 1..4 = the split-precision modes, which stream W2's
 * planes from the T blob (mobody_mlp_transpose builds them, the Adam kernels keep them current -- both for the SAME mode). */

/* ---- replay gather / ring append ------------------------------------------------------- */
/* Concatenate rows idx_k of up to three buffers (src | tar | fake order, mobody.py:525-529)
 * into one minibatch: state[N][S] action[N][A] next_state[N][S] reward[N] not_done[N]. */
int mobody_gather_batch(const MobodyBufferView* bufs, const int32_t* const* idx, const int64_t* counts, int nbuf,
                        int S, int A, float* state, float* action, float* next_state, float* reward, float* not_done,
                        void* stream);

/* Same gather with the row indices drawn on the device (np.random.randint(0, size, n), utils.py:128, throughput
 * mode): index i of source k = philox(seeds[k], stream 3, call)[i] * size >> 32 with
 * call = (counter ? counter[0] : 0) + call_offsets[k] and size read from the DEVICE word sizes[k][0]
 * (`seeds`, `call_offsets`, `counts` and the `sizes` pointer array are host arrays).  Identical draws to
 * mobody_rng_index / mobody_sample_indices for the same (seed, call).  `bump`: host array of nbump <= 4 device int64
 * words (none of them `counter`) that one thread of the launch increments by one (a captured step advances its Adam
 * step counts here); nbump = 0: none. */
int mobody_gather_batch_rng(const MobodyBufferView* bufs, const int64_t* counts, int nbuf, int S, int A,
                            const uint32_t* seeds, const int64_t* call_offsets, const int64_t* counter,
                            const int64_t* const* sizes, float* state, float* action, float* next_state,
                            float* reward, float* not_done, int64_t* const* bump, int nbump, void* stream);

/* Append the rows with keep[i] != 0 (NULL = all), in order, to the ring `ring` of `cap` rows (written through the
 * view's pointers) at *ptr_size (device int64[2] = {ptr, size}), reproducing add_batch's single-wrap arithmetic
 * (utils.py:43-92); not_done = 1 - terminal; the row-interleaved ring also gets its row padding zeroed.
 * `scan_ws` needs (M + 1040) int32 whose FIRST EIGHT must be zero on entry (zero the workspace once when allocating it;
 * every call leaves them zero: they hold the scan's arrival ticket, and a memset launch per append would cost more than
 * a short append itself).  Two launches: a block scan whose last block commits the new {ptr, size}, and the row scatter. */
int mobody_ring_append(const MobodyBufferView* ring, int64_t cap, int64_t* ptr_size, int S, int A, const float* obs,
                       const float* act, const float* next_obs, const float* reward, const uint8_t* terminal,
                       const uint8_t* keep, int64_t M, int32_t* scan_ws, void* stream);

/* ---- training step --------------------------------------------------------------------- */
typedef struct MobodyTrainDims {
  int32_t S, A;
  int64_t N;          /* rows of the mixed critic batch on this rank */
  int64_t Nt;         /* leading rows that form the "true" BC batch (src|tar) */
  int64_t N_global;   /* sum of N over data-parallel ranks (== N on one GPU) */
  int64_t Nt_global;
} MobodyTrainDims;

typedef struct MobodyHyper {
  float gamma, tau, max_action, weight, bc_coef;
  int32_t q_weighted, scale_q;
  int32_t precision;   /* MFMA mode of the 256 x 256 GEMMs of the forward and backward passes and of the 256 x 256
                          weight-gradient job: 0 exact fp32 (the reference's arithmetic), 1 bf16, 2 bf16x2, 3 bf16x3, 4 f16x2;
                          the update entry points also write the nets' W2 planes in this mode's format */
} MobodyHyper;

/* floats of scratch the training calls need.  The SAME workspace has to be handed to mobody_actor_forward and the
 * following mobody_actor_backward: it carries pi(s), the Q values, the saved activations and the ReLU sign words
 * between the two calls (the data-parallel caller all-reduces `stats` in between). */
int64_t mobody_train_workspace(const MobodyTrainDims* d);

/* Critic loss + gradients (A.2): grad_q (MobodyMlpLayout(S+A,1,2) layout) and loss_out[0] = L_Q of
 * the LOCAL rows scaled by 1/N_global (sum over ranks == global loss). */
int mobody_critic_step(const MobodyTrainDims* d, const MobodyHyper* h, const float* actor_blob, const float* actor_blob_T,
                       const float* q_blob, const float* q_blob_T, const float* qtarg_blob, const float* qtarg_blob_T,
                       const float* state, const float* action, const float* next_state, const float* reward,
                       const float* not_done, const float* q_next, float* grad_q, float* loss_out, float* workspace,
                       int policy_forward, void* stream);
/* actor_blob_T / qtarg_blob_T are only read when h->precision != 0 (their W2 planes); NULL is fine at precision 0. */
/* policy_forward != 0 (needs q_next == NULL): the target-Q launch also evaluates pi(s) with its saves for the coming
 * mobody_actor_forward(..., policy_ready = 1) on the same workspace -- the actor does not change in between, and
 * the merged launch fills the chip better than the two it replaces.
 * q_next: NULL -> min target-Q(s', pi(s')) is computed here (update_q_functions, mobody.py:189-208); non-NULL ->
 * [N] bootstrap values supplied by the caller, V(s') in the advantage variant (update_q_functions_1, :210-229;
 * actor_blob / qtarg_blob / next_state may then be NULL). */

/* Single-GPU form of mobody_critic_step + mobody_adam_polyak (mobody.py:540-552): the gradient reduction applies the
 * Adam step (1-based t, or a device word t_dev) and the Polyak update of qtarg_blob (tau from `h`) itself, so the
 * gradient blob is never written -- one launch and one gradient round trip fewer.  Bit-identical to the two calls. */
int mobody_critic_update(const MobodyTrainDims* d, const MobodyHyper* h, const float* actor_blob, const float* actor_blob_T,
                         float* q_blob, float* q_blob_T, float* qtarg_blob, float* qtarg_blob_T, const float* state,
                         const float* action, const float* next_state, const float* reward, const float* not_done,
                         const float* q_next, float* m, float* v, int64_t t, const int64_t* t_dev, float lr,
                         float* loss_out, float* workspace, int policy_forward, int64_t* bump, void* stream);
/* qtarg_blob_T (nullable): the target net's T blob; when given, the W2 planes of the target follow the Polyak update.
 * bump (nullable, != t_dev): a device int64 word the optimizer launch increments by one -- a captured step advances the
 * RNG call id it has already consumed here instead of in a launch of its own. */

/* Actor phase, part 1: forwards + the two batch statistics stats[0]=sum|min Q(s,pi(s))|,
 * stats[1]=sum|min Q(s_t,a_t)| over LOCAL rows (all-reduce them across ranks before part 2). */
int mobody_actor_forward(const MobodyTrainDims* d, const MobodyHyper* h, const float* actor_blob, const float* actor_blob_T,
                         const float* q_blob, const float* q_blob_T, const float* state, const float* action, float* stats,
                         float* workspace, int policy_ready, void* stream);
/* policy_ready != 0: pi(s) and its saves are already in the workspace (mobody_critic_step(..., policy_forward = 1)). */

/* Actor phase, part 2: grad_actor (MobodyMlpLayout(S,A,1)) and loss_out[0]=L_pi, [1]=L_BC (local share). */
int mobody_actor_backward(const MobodyTrainDims* d, const MobodyHyper* h, const float* actor_blob,
                          const float* actor_blob_T, const float* q_blob, const float* q_blob_T, const float* state,
                          const float* action, const float* stats, const float* v_true, float* grad_actor,
                          float* loss_out, float* workspace, void* stream);
/* v_true: NULL -> BC weights exp(3*q_b/mean|q_b|); [Nt] V(s_true) -> exp(3*(q_b - V)) (config['advantage'], :255-256). */

/* mobody_critic_update in two calls: phase 1 enqueues its forwards (none of them reads `reward`), phase 2 the backward, the
 * weight gradients and the reduction / optimizer step.  Between the two the caller may join a stream that rewrites `reward`
 * (penalty_type 'par', mobody.py:428-434: an ensemble step on the source rows that otherwise sits in front of the critic). */
int mobody_critic_update_phase(const MobodyTrainDims* d, const MobodyHyper* h, const float* actor_blob, const float* actor_blob_T,
                               float* q_blob, float* q_blob_T, float* qtarg_blob, float* qtarg_blob_T, const float* state,
                               const float* action, const float* next_state, const float* reward, const float* not_done,
                               const float* q_next, float* m, float* v, int64_t t, const int64_t* t_dev, float lr, float* loss_out,
                               float* workspace, int policy_forward, int64_t* bump, int phase, void* stream);

/* Single-GPU form of mobody_actor_backward + mobody_adam_polyak (mobody.py:554-578), as mobody_critic_update. */
int mobody_actor_update(const MobodyTrainDims* d, const MobodyHyper* h, float* actor_blob, float* actor_blob_T,
                        const float* q_blob, const float* q_blob_T, const float* state, const float* action,
                        const float* stats, const float* v_true, float* m, float* v, int64_t t, const int64_t* t_dev,
                        float lr, float* loss_out, float* workspace, void* stream);

/* Expectile loss of the V function (update_v_function, mobody.py:231-242): adv = min(qt[0],qt[1]) - v;
 * dz3[N][16] column 0 = dL_V/dV (1/N_global scaling), loss_out[0] = local share of L_V; lossp_ws: ceil(N/256) floats. */
int mobody_value_loss_grad(const float* qt, const float* v, int64_t N, int64_t N_global, float* dz3, float* loss_out,
                           float* lossp_ws, void* stream);

/* Adam (torch defaults b1=.9 b2=.999 eps=1e-8) on a packed blob, 1-based step t; optional Polyak
 * target update target = tau*p + (1-tau)*target (tau < 0 or target == NULL: skip); refreshes the
 * transposed blob used by the backward kernels (blob_T may be NULL). grad_scale multiplies the
 * gradient first (1/world for an all-reduced SUM). */
int mobody_adam_polyak(int in_dim, int out_dim, int members, float* blob, float* blob_T, const float* grad, float* m,
                       float* v, float* target, float* target_T, int64_t t, float lr, float tau, float grad_scale,
                       int precision, void* stream);
/* target_T (nullable): T blob of the target net, whose W2 planes then follow the Polyak update (split-precision modes).
 * precision: format of the W2 planes written into blob_T / target_T (the mode the nets are evaluated in). */

/* Same, with the 1-based step count read from DEVICE memory (t_dev[0]) so that a captured HIP graph advances
 * without new kernel arguments (bias corrections are formed in double on the device). */
int mobody_adam_polyak_dev(int in_dim, int out_dim, int members, float* blob, float* blob_T, const float* grad,
                           float* m, float* v, float* target, float* target_T, const int64_t* t_dev, float lr, float tau,
                           float grad_scale, int precision, void* stream);

/* PAR reward shaping: reward[i] -= coef * mean_d (next_state_true[i][d] - next_state_model[i][d])^2  (mobody.py:428-434) */
int mobody_par_penalty(const float* next_state_true, const float* next_state_model, float* reward, float coef,
                       int64_t n, int S, void* stream);

/* ---- generic gradient of one packed MLP + DARA classifier pieces (mobody.py:11-33,146-181,354-381) ---- */
/* grad (parameter-blob layout) from dz3[members][rows][Np3] and the activations mobody_mlp3_forward saved. */
int64_t mobody_mlp3_backward_workspace(int in_dim, int out_dim, int members, int64_t rows);
int mobody_mlp3_backward(const float* blob_T, int in_dim, int out_dim, int members, const float* dz3, const float* x,
                         const float* h1, const float* h2, int64_t rows, float* grad, float* workspace, void* stream);

/* Classifier inputs x_sas[N][2S+A] = [s,a,s'] + std*eps, x_sa[N][S+A] = [s,a] + std*eps' (eps explicit or device
 * Philox streams 4/5 at (seed, call); std = 0 -> no noise). */
int mobody_dara_inputs(const float* s, const float* a, const float* s2, int64_t N, int S, int A, float std,
                       const float* noise_sas, const float* noise_sa, uint32_t seed, uint32_t call, float* x_sas,
                       float* x_sa, void* stream);

/* loss_out = (loss_sa, loss_sas) = mean cross_entropy(softmax(logits), label) with the reference's double softmax;
 * dz_*[N][16] = d(loss)/d(logits) (columns 2..15 zero).  labels NULL -> rows < n_src are 0, the rest 1.
 * lossp_ws: 2*ceil(N/256) floats. */
int mobody_dara_loss_grad(const float* z_sas, const float* z_sa, const int32_t* labels, int64_t N, int64_t n_src,
                          float* dz_sas, float* dz_sa, float* loss_out, float* lossp_ws, void* stream);

/* delta = clamp(log p~_sas[1] - log p~_sa[1] - log p~_sas[0] + log p~_sa[0], -10, 10), p~ = softmax(softmax(logits)) + 1e-10;
 * reward[i] += coef * delta (reward may be NULL), delta_out optional. */
int mobody_dara_penalty(const float* z_sas, const float* z_sa, int64_t n, float coef, float* reward, float* delta_out,
                        void* stream);

/* ---- dynamics pre-training (MOBODYEnsembleDynamics.learn / validate, algo/dynamics/mobody_dynamics.py:300-384,
 *      594-653, 1113-1140; Swish/reparameterisation mobody_module.py:9-15,237-243) ----------------------------------
 * All trained parameters of the 7-member ensemble live in ONE blob: three MobodyMlpLayout regions with 7 members
 * (state encoder zs1-3: S -> 256 -> 256 -> 32 = mu | logvar; decoder transition1-3: 16 -> 256 -> 256 -> S; reward head
 * reward_model1-3: 2S+A -> 256 -> 256 -> 2) and the two action encoders (za_src*, za_trg*), per member
 * W1[16+A][32] b1[32] W2[32][16] b2[16] row major (only the mu half of za_*2 takes part in the loss).  Gradients and
 * both Adam moments use the same layout; the T blob holds the three regions' transposes. */
typedef struct MobodyPretrainLayout {
  int32_t S, A, za_in, _pad;
  MobodyMlpLayout enc, tr, rw;
  int64_t off_enc, off_tr, off_rw, off_za_src, off_za_trg;    /* float offsets inside the parameter blob */
  int64_t za_w1, za_b1, za_w2, za_b2, za_member_floats;       /* inside one member of an action encoder */
  int64_t total_floats;
  int64_t t_off_enc, t_off_tr, t_off_rw, t_total_floats;      /* T blob */
} MobodyPretrainLayout;
int mobody_pretrain_layout(int S, int A, MobodyPretrainLayout* out);
/* `precision` of the four pre-training entry points: 0 = exact fp32 MFMA, 4 = "f16x2" (the 256 x 256 layers of the three
 * 7-member nets on the split core, h1 / dz2 handed to the weight-gradient GEMM as fp16 planes; other modes are refused).
 * The T blob carries the W2 / W2^T planes of the mode it was built for; Adam keeps them current. */
int mobody_pretrain_transpose(int S, int A, const float* blob, float* blob_T, int precision, void* stream);
int64_t mobody_pretrain_workspace(int S, int A, int64_t b);

/* Bootstrap gather of one batch (learn() slices train_obss[:, k*bs:(k+1)*bs] of the per-member bootstrapped arrays,
 * :604-612): member e takes dataset rows idx[e][start + r], r < b.  idx is a DEVICE int32 [7][n_idx] matrix.
 * Outputs: xenc[7][2b][S] (s rows, then s' rows), act[7][b][A], rew[7][b]. */
int mobody_pretrain_gather(const float* state, const float* action, const float* next_state, const float* reward,
                           const int32_t* idx, int64_t n_idx, int64_t start, const int64_t* start_dev, int64_t b, int S,
                           int A, float* xenc, float* act, float* rew, void* stream);
/* start_dev (nullable): DEVICE int64 batch counter, start += start_dev[0] * b, so a captured graph walks the index matrix
 * batch by batch (advance it with mobody_counter_add); reads are clamped to the matrix. */

/* Loss and gradients of one learn() batch (zero_grad + loss.backward, :594-642).  b rows per member on this rank,
 * b_global = rows per member over all data-parallel ranks (gradients / losses are local shares of the global means:
 * SUM all-reduce `grad` before mobody_pretrain_adam).  noise6 [6][7][b][16] = the six reparameterisation draws in the
 * reference's order z1(s) z2(s') z3(s) z4(s') z5(s) z6(s), noise7 [7][b][S] = the fake-next-state draw; NULL -> device
 * Philox streams 16..22 at (seed, call).  grad: blob layout; the action encoder that is not used this step
 * (za_trg* on source batches, za_src* on target ones) is left untouched.
 * loss_out[5] = (loss, transition_loss, encoder_loss, recon_loss, kl_loss). */
int mobody_pretrain_grads(int S, int A, int64_t b, int64_t b_global, int use_trg, float encoder_loss_coef,
                          const float* blob, const float* blob_T, const float* xenc, const float* act, const float* rew,
                          const float* noise6, const float* noise7, uint32_t seed, uint32_t call, float* grad,
                          float* loss_out, float* workspace, int precision, float transition_coef, float reward_coef,
                          void* stream);
/* transition_coef / reward_coef: weights of transition_loss and reward_loss in this call's loss (1, 1 = learn()).
 * inverse_sep_reward_loss = 1 runs learn() with reward_coef 0 and learn_sep_reward (:482-519) with encoder_loss_coef 0,
 * transition_coef 0, reward_coef 1. */

/* Single-GPU form of mobody_pretrain_grads + mobody_pretrain_adam: every gradient reduction applies the Adam step of the
 * elements it has just reduced (no gradient blob, four launches fewer).  t_dev (nullable): DEVICE int64[2] = {t_main,
 * t_za} read instead of the host counts; call_dev (nullable): DEVICE int64 word added to `call` -- both for graph replay.
 * loss_acc (nullable): DEVICE float[5], loss_acc += loss_out in the step's last launch (learn()'s running sums, :630-650). */
int mobody_pretrain_update(int S, int A, int64_t b, int use_trg, float encoder_loss_coef, float* blob, float* blob_T,
                           const float* xenc, const float* act, const float* rew, const float* noise6, const float* noise7,
                           uint32_t seed, uint32_t call, const int64_t* call_dev, float* m, float* v, int64_t t_main,
                           int64_t t_za, const int64_t* t_dev, float lr, float* loss_out, float* loss_acc, float* workspace,
                           int precision, void* stream);

/* torch.optim.Adam step on the blob (and its T blob): the three MLP regions use the 1-based step count t_main, the
 * action encoder of this step's domain t_za; the other action encoder is skipped (its .grad is None in the reference,
 * so its Adam state does not advance). */
int mobody_pretrain_adam(int S, int A, int use_trg, float* blob, float* blob_T, const float* grad, float* m, float* v,
                         int64_t t_main, int64_t t_za, float lr, float grad_scale, int precision, int net_mask, int64_t t_rw,
                         void* stream);
/* net_mask: bit 0 state encoder, bit 1 decoder, bit 2 reward head -- a net without a gradient in this step is skipped and
 * keeps its step count (7 and t_rw = t_main: every net, one count -- learn()); the reward head steps with its own t_rw. */

/* Adam step of ONE action encoder only (blob layout, its own 1-based step count): the second encoder of a learn_src_trg step
 * (config train_together = 1, :521-590 -- the summed source + target loss moves both, mobody_pretrain_adam steps one). */
int mobody_pretrain_za_adam(int S, int A, int use_trg, float* blob, const float* grad, float* m, float* v, int64_t t_za,
                            float lr, float grad_scale, void* stream);

/* validate() (:1113-1140) on an inference blob (mobody_dyn_layout): out[0..6] = per-member mean_{b,d}(mean_e - s')^2,
 * out[7..13] = per-member mean_b (r_mu_e(s, a, mean_e) - r)^2.  Workspace floats: mobody_dyn_validate_workspace. */
int64_t mobody_dyn_validate_workspace(int S, int A, int64_t B);
int mobody_dyn_validate(const float* dyn_blob, int S, int A, const float* obs, const float* act, const float* next_obs,
                        const float* rew, int64_t B, int use_trg, float* out, float* workspace, void* stream);

/* (Re)build the transposed blob from a parameter blob (after loading a checkpoint); `precision` selects the format of
 * the W2 / W2^T planes it writes (the mode the net will be evaluated in). */
int mobody_mlp_transpose(int in_dim, int out_dim, int members, const float* blob, float* blob_T, int precision,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOBODY_HIP_H */
