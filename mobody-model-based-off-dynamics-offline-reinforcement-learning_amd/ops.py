"""Thin torch-tensor front end of the C ABI (allocation + pointer plumbing only).

Every function here enqueues HIP kernels from libmobody_hip.so on torch's current stream;
none of them computes anything in PyTorch.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, cur_stream, load, ptr


def _f32(t, device=None):
    t = torch.as_tensor(t, dtype=torch.float32)
    if device is not None:
        t = t.to(device)
    return t.contiguous()


def rng_normal(seed, stream_id, call, n, device):
    out = torch.empty(n, dtype=torch.float32, device=device)
    check(load().mobody_rng_normal(seed, stream_id, call, n, ptr(out), cur_stream()), "mobody_rng_normal")
    return out


def rng_index(seed, stream_id, call, n, bound, device):
    out = torch.empty(n, dtype=torch.int32, device=device)
    check(load().mobody_rng_index(seed, stream_id, call, n, bound, ptr(out), cur_stream()), "mobody_rng_index")
    return out


def prec_id(mfma):
    """'f32' | 'bf16' | 'bf16x2' | 'bf16x3' | 'f16x2' (or the integer id) -> MobodyHyper.precision."""
    return mfma if isinstance(mfma, int) else _lib.PRECISIONS[mfma]


def default_mfma():
    """MFMA mode when a config does not name one: exact fp32, unless MOBODY_MFMA says otherwise (how the whole parity
    suite is re-run in a split-precision mode: `MOBODY_MFMA=bf16x3 pytest -m gpu`)."""
    import os
    return os.environ.get("MOBODY_MFMA", "f32")


def dyn_planes(blob, S, A, out=None, precision=3):
    """16-bit planes of zs2 / transition2 / reward_model2 in the format of the split-precision mode `precision`
    (modes 1-3 share the three bf16 planes; 'f16x2' has its own two fp16 planes)."""
    pl = out if out is not None else torch.empty(load().mobody_dyn_planes_floats(), dtype=torch.float32, device=blob.device)
    check(load().mobody_dyn_planes(ptr(blob), S, A, ptr(pl), prec_id(precision), cur_stream()), "mobody_dyn_planes")
    return pl


def dyn_forward(blob, S, A, obs, act, use_trg=True, planes=None, precision=0):
    obs, act = _f32(obs), _f32(act)
    B = obs.shape[0]
    mean = torch.empty(7, B, S, dtype=torch.float32, device=obs.device)
    check(load().mobody_dyn_forward(ptr(blob), ptr(planes), prec_id(precision), S, A, ptr(obs), ptr(act), B, int(use_trg),
                                    ptr(mean), cur_stream()), "mobody_dyn_forward")
    return mean


def dyn_step(blob, S, A, task_id, obs, act, noise=None, elite_idx=None, alive=None, elites=(0, 1, 2, 3, 4), seed=0,
             call=0, penalty_coef=0.0, use_penalty=True, use_trg=True, want_mean=False, workspace=None, out=None,
             planes=None, precision=0, mopo=None, call_dev=None):
    """Returns dict(next_obs[B,S], reward[B,1], terminal uint8[B,1], penalty[B,1], raw_reward[B,1], mean?).
    mopo = (blob, blob_T) of the 7-member MLP za_src1..3: the MOPO ablation's step (mobody_mopo_step).
    call_dev: device int64[1] added to `call` (graph replay)."""
    obs, act = _f32(obs), _f32(act)
    dev, B = obs.device, obs.shape[0]
    if noise is not None:
        noise = _f32(noise, dev)
        assert noise.shape == (7, B, S)
    if elite_idx is not None:
        elite_idx = torch.as_tensor(elite_idx).to(device=dev, dtype=torch.int32).contiguous()
    need = load().mobody_dyn_step_workspace(S, A, B)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 1), dtype=torch.float32, device=dev)
    o = out or {}
    nxt = o.get("next_obs", None)
    if nxt is None:
        nxt = torch.empty(B, S, dtype=torch.float32, device=dev)
    rew = o["reward"] if "reward" in o else torch.empty(B, 1, dtype=torch.float32, device=dev)
    term = o["terminal"] if "terminal" in o else torch.empty(B, 1, dtype=torch.uint8, device=dev)
    pen = o["penalty"] if "penalty" in o else torch.empty(B, 1, dtype=torch.float32, device=dev)
    raw = o["raw_reward"] if "raw_reward" in o else torch.empty(B, 1, dtype=torch.float32, device=dev)
    mean = torch.empty(7, B, S, dtype=torch.float32, device=dev) if want_mean else None
    el = (C.c_int32 * len(elites))(*[int(e) for e in elites])
    if mopo is not None:
        check(load().mobody_mopo_step(ptr(blob), ptr(planes), ptr(mopo[0]), ptr(mopo[1]), prec_id(precision), S, A, task_id, ptr(obs),
                                      ptr(act), B, ptr(noise), ptr(elite_idx), ptr(alive), el, len(elites), seed, call,
                                      float(penalty_coef), int(bool(use_penalty)), ptr(nxt), ptr(rew), ptr(term), ptr(pen), ptr(raw),
                                      ptr(mean), ptr(workspace), cur_stream()), "mobody_mopo_step")
        res = dict(next_obs=nxt, reward=rew, terminal=term, penalty=pen, raw_reward=raw)
        if want_mean:
            res["mean"] = mean
        return res
    check(load().mobody_dyn_step(ptr(blob), ptr(planes), prec_id(precision), S, A, task_id, ptr(obs), ptr(act), B, ptr(noise), ptr(elite_idx),
                                 ptr(alive), el, len(elites), seed, call, ptr(call_dev), float(penalty_coef), int(bool(use_penalty)),
                                 int(bool(use_trg)), ptr(nxt), ptr(rew), ptr(term), ptr(pen), ptr(raw), ptr(mean),
                                 ptr(workspace), cur_stream()), "mobody_dyn_step")
    res = dict(next_obs=nxt, reward=rew, terminal=term, penalty=pen, raw_reward=raw)
    if want_mean:
        res["mean"] = mean
    return res


def mlp3_forward(blob, in_dim, out_dim, members, src0, src1=None, out_mode=0, max_action=1.0, save=False, blob_T=None,
                 precision=0):
    """out[members, rows, out_dim] (+ saved (x, h1, h2) when save=True)."""
    src0 = _f32(src0)
    rows, n0 = src0.shape
    n1 = 0
    if src1 is not None:
        src1 = _f32(src1)
        n1 = src1.shape[1]
    dev = src0.device
    out = torch.empty(members, rows, out_dim, dtype=torch.float32, device=dev)
    sx = sh1 = sh2 = None
    if save:
        L = _lib.mlp_layout(in_dim, out_dim, members)
        sx = torch.empty(rows, L.Kp1, dtype=torch.float32, device=dev)
        sh1 = torch.empty(members, rows, 256, dtype=torch.float32, device=dev)
        sh2 = torch.empty(members, rows, 256, dtype=torch.float32, device=dev)
    check(load().mobody_mlp3_forward(ptr(blob), ptr(blob_T), prec_id(precision), in_dim, out_dim, members, ptr(src0), n0, ptr(src1), n1, rows, out_mode,
                                     float(max_action), ptr(out), ptr(sx), ptr(sh1), ptr(sh2), cur_stream()),
          "mobody_mlp3_forward")
    return (out, sx, sh1, sh2) if save else out


# ------------------------------------------------------------------------------------------------
# training step
# ------------------------------------------------------------------------------------------------
def train_dims(S, A, N, Nt, N_global=None, Nt_global=None):
    return _lib.MobodyTrainDims(S, A, N, Nt, N if N_global is None else N_global, Nt if Nt_global is None else Nt_global)


def hyper(cfg):
    return _lib.MobodyHyper(float(cfg["gamma"]), float(cfg["tau"]), float(cfg["max_action"]), float(cfg["weight"]),
                            float(cfg["bc_coef"]), int(bool(cfg["q_weighted"])), int(bool(cfg["scale_Q"])),
                            prec_id(cfg.get("mfma", default_mfma())))


def train_workspace(dims, device):
    n = load().mobody_train_workspace(C.byref(dims))
    if n < 0:
        raise _lib.MobodyError("mobody_train_workspace: " + load().mobody_last_error().decode())
    return torch.empty(n, dtype=torch.float32, device=device)


def mlp_transpose(blob, in_dim, out_dim, members, out=None, precision=0):
    """T blob of a packed MLP; `precision` = the MFMA mode whose W2 plane format it carries."""
    L = _lib.mlp_layout(in_dim, out_dim, members)
    bt = out if out is not None else torch.empty(L.t_total_floats, dtype=torch.float32, device=blob.device)
    assert bt.numel() == L.t_total_floats
    check(load().mobody_mlp_transpose(in_dim, out_dim, members, ptr(blob), ptr(bt), prec_id(precision), cur_stream()),
          "mobody_mlp_transpose")
    return bt


def critic_step(dims, hyp, actor_blob, q_blob, q_blob_T, qtarg_blob, batch, grad_q, loss_out, ws, q_next=None,
                policy_forward=False, actor_blob_T=None, qtarg_blob_T=None):
    s, a, s2, r, nd = batch
    check(load().mobody_critic_step(C.byref(dims), C.byref(hyp), ptr(actor_blob), ptr(actor_blob_T), ptr(q_blob), ptr(q_blob_T),
                                    ptr(qtarg_blob), ptr(qtarg_blob_T), ptr(s), ptr(a), ptr(s2), ptr(r), ptr(nd), ptr(q_next), ptr(grad_q),
                                    ptr(loss_out), ptr(ws), int(bool(policy_forward)), cur_stream()), "mobody_critic_step")


def actor_forward(dims, hyp, actor_blob, q_blob, state, action, stats, ws, policy_ready=False, actor_blob_T=None, q_blob_T=None):
    check(load().mobody_actor_forward(C.byref(dims), C.byref(hyp), ptr(actor_blob), ptr(actor_blob_T), ptr(q_blob), ptr(q_blob_T), ptr(state),
                                      ptr(action), ptr(stats), ptr(ws), int(bool(policy_ready)), cur_stream()),
          "mobody_actor_forward")


def actor_backward(dims, hyp, actor_blob, actor_blob_T, q_blob, q_blob_T, state, action, stats, grad_actor, loss_out,
                   ws, v_true=None):
    check(load().mobody_actor_backward(C.byref(dims), C.byref(hyp), ptr(actor_blob), ptr(actor_blob_T), ptr(q_blob),
                                       ptr(q_blob_T), ptr(state), ptr(action), ptr(stats), ptr(v_true), ptr(grad_actor),
                                       ptr(loss_out), ptr(ws), cur_stream()), "mobody_actor_backward")


def critic_update(dims, hyp, actor_blob, q_blob, q_blob_T, qtarg_blob, batch, m, v, t, lr, loss_out, ws, q_next=None,
                  t_dev=None, policy_forward=False, actor_blob_T=None, qtarg_blob_T=None, bump=None, phase=0):
    """critic_step + Adam + Polyak in the fused single-GPU form (t: host step count, or t_dev: device int64[1]);
    bump: device int64[1] word (not t_dev) the optimizer launch increments by one.  phase 1 / 2: only the forwards / only the
    backward + update (mobody_critic_update_phase: the caller joins whatever rewrites `reward` in between)."""
    s, a, s2, r, nd = batch
    args = (C.byref(dims), C.byref(hyp), ptr(actor_blob), ptr(actor_blob_T), ptr(q_blob), ptr(q_blob_T),
            ptr(qtarg_blob), ptr(qtarg_blob_T), ptr(s), ptr(a), ptr(s2), ptr(r), ptr(nd), ptr(q_next), ptr(m),
            ptr(v), int(t), ptr(t_dev), float(lr), ptr(loss_out), ptr(ws), int(bool(policy_forward)), ptr(bump))
    if phase:
        check(load().mobody_critic_update_phase(*args, int(phase), cur_stream()), "mobody_critic_update_phase")
    else:
        check(load().mobody_critic_update(*args, cur_stream()), "mobody_critic_update")


def actor_update(dims, hyp, actor_blob, actor_blob_T, q_blob, q_blob_T, state, action, stats, m, v, t, lr, loss_out, ws,
                 v_true=None, t_dev=None):
    check(load().mobody_actor_update(C.byref(dims), C.byref(hyp), ptr(actor_blob), ptr(actor_blob_T), ptr(q_blob),
                                     ptr(q_blob_T), ptr(state), ptr(action), ptr(stats), ptr(v_true), ptr(m), ptr(v),
                                     int(t), ptr(t_dev), float(lr), ptr(loss_out), ptr(ws), cur_stream()),
          "mobody_actor_update")


def value_loss_grad(qt, v, n_global):
    """qt [2,N] target twin-Q(s,a), v [N] -> (dz3[N,16], loss[1]) of the expectile V loss."""
    N = v.numel()
    dz3 = torch.empty(N, 16, dtype=torch.float32, device=v.device)
    loss = torch.empty(1, dtype=torch.float32, device=v.device)
    lossp = torch.empty((N + 255) // 256, dtype=torch.float32, device=v.device)
    check(load().mobody_value_loss_grad(ptr(qt), ptr(v), N, int(n_global), ptr(dz3), ptr(loss), ptr(lossp), cur_stream()),
          "mobody_value_loss_grad")
    return dz3, loss


def adam_polyak(in_dim, out_dim, members, blob, blob_T, grad, m, v, target, t, lr, tau=-1.0, grad_scale=1.0, target_T=None,
                precision=0):
    check(load().mobody_adam_polyak(in_dim, out_dim, members, ptr(blob), ptr(blob_T), ptr(grad), ptr(m), ptr(v),
                                    ptr(target), ptr(target_T), int(t), float(lr), float(tau), float(grad_scale),
                                    prec_id(precision), cur_stream()),
          "mobody_adam_polyak")


# ------------------------------------------------------------------------------------------------
# replay data movement
# ------------------------------------------------------------------------------------------------
class RingView:
    """A row-interleaved replay ring: `store` is one [rows][pitch] fp32 tensor, a row = state | action | next_state | reward |
    not_done | padding (include/mobody_hip.h, mobody_ring_pitch).  Iterating yields the five field views (strided)."""

    def __init__(self, store, S, A):
        assert store.is_contiguous() and store.dtype == torch.float32 and store.shape[1] >= 2 * S + A + 2
        self.store, self.S, self.A = store, int(S), int(A)
        self.device = store.device

    def fields(self):
        S, A, st = self.S, self.A, self.store
        return st[:, :S], st[:, S:S + A], st[:, S + A:2 * S + A], st[:, 2 * S + A:2 * S + A + 1], st[:, 2 * S + A + 1:2 * S + A + 2]

    def __iter__(self):
        return iter(self.fields())

    def __getitem__(self, k):
        return self.fields()[k]


def ring_pitch(S, A):
    return int(load().mobody_ring_pitch(int(S), int(A)))


def buffer_view(b):
    """MobodyBufferView of a RingView or of a 5-tuple of separate contiguous tensors (state, action, next_state, reward, not_done)."""
    if isinstance(b, RingView):
        base, S, A = b.store.data_ptr(), b.S, b.A
        return _lib.MobodyBufferView(base, base + 4 * S, base + 4 * (S + A), base + 4 * (2 * S + A), base + 4 * (2 * S + A + 1), b.store.shape[1])
    return _lib.MobodyBufferView(*[ptr(t) for t in b], 0)


def _view_device(b):
    return b.device if isinstance(b, RingView) else b[0].device


def gather_batch(buffers, indices, S, A, out=None):
    """buffers: list of RingView or (state, action, next_state, reward, not_done) device tensors; indices: int32 device
    tensors. Returns the concatenated minibatch (state[N,S], action[N,A], next_state[N,S], reward[N,1], not_done[N,1])."""
    n = len(buffers)
    dev = _view_device(buffers[0])
    views = (_lib.MobodyBufferView * n)(*[buffer_view(b) for b in buffers])
    idx = [i.to(device=dev, dtype=torch.int32).contiguous() for i in indices]
    iptr = (C.c_void_p * n)(*[ptr(i) if i.numel() else None for i in idx])
    cnt = (C.c_int64 * n)(*[i.numel() for i in idx])
    N = sum(i.numel() for i in idx)
    if out is None:
        out = (torch.empty(N, S, device=dev), torch.empty(N, A, device=dev), torch.empty(N, S, device=dev),
               torch.empty(N, 1, device=dev), torch.empty(N, 1, device=dev))
    check(load().mobody_gather_batch(views, iptr, cnt, n, S, A, *[ptr(t) for t in out], cur_stream()),
          "mobody_gather_batch")
    return out


_scan_ws = {}


def ring_append(buf, cap, ptr_size, S, A, obs, act, next_obs, reward, terminal, keep=None):
    """buf = RingView or (state, action, next_state, reward, not_done) ring tensors; ptr_size int64[2] device tensor."""
    M = obs.shape[0]
    if M == 0:
        return
    dev = obs.device
    scan = _scan_ws.get(dev)
    if scan is None or scan.numel() < M + 1040:      # zeroed once: the append keeps its ticket words at zero itself
        scan = _scan_ws[dev] = torch.zeros(max(M + 1040, 2 * (scan.numel() if scan is not None else 0)), dtype=torch.int32, device=dev)
    check(load().mobody_ring_append(C.byref(buffer_view(buf)), cap, ptr(ptr_size), S, A, ptr(obs), ptr(act), ptr(next_obs),
                                    ptr(reward), ptr(terminal), ptr(keep), M, ptr(scan), cur_stream()),
          "mobody_ring_append")


# ------------------------------------------------------------------------------------------------
# small device-side helpers used by the host mirror
# ------------------------------------------------------------------------------------------------
def termination(task_id, next_obs):
    B, S = next_obs.shape
    done = torch.empty(B, 1, dtype=torch.uint8, device=next_obs.device)
    check(load().mobody_termination(task_id, ptr(next_obs), B, S, ptr(done), cur_stream()), "mobody_termination")
    return done


def rollout_mask(alive_in, terminal, penalty, env_filter, use_filter, keep, alive_out):
    B = terminal.shape[0]
    check(load().mobody_rollout_mask(ptr(alive_in), ptr(terminal), ptr(penalty), float(env_filter), int(bool(use_filter)),
                                     B, ptr(keep), ptr(alive_out), cur_stream()), "mobody_rollout_mask")


def sample_indices(seed, stream_id, counter, call_offset, n, size_dev, out=None):
    if out is None:
        out = torch.empty(n, dtype=torch.int32, device=size_dev.device)
    check(load().mobody_sample_indices(seed, stream_id, ptr(counter), call_offset, n, ptr(size_dev), ptr(out),
                                       cur_stream()), "mobody_sample_indices")
    return out


def counter_add(counter, inc=1):
    """counter: device int64[n]; every word += inc."""
    check(load().mobody_counter_add(ptr(counter), counter.numel(), inc, cur_stream()), "mobody_counter_add")


def par_penalty(next_state_true, next_state_model, reward, coef):
    n, S = next_state_true.shape
    check(load().mobody_par_penalty(ptr(next_state_true), ptr(next_state_model), ptr(reward), float(coef), n, S,
                                    cur_stream()), "mobody_par_penalty")


def adam_polyak_dev(in_dim, out_dim, members, blob, blob_T, grad, m, v, target, t_dev, lr, tau=-1.0, grad_scale=1.0,
                    target_T=None, precision=0):
    check(load().mobody_adam_polyak_dev(in_dim, out_dim, members, ptr(blob), ptr(blob_T), ptr(grad), ptr(m), ptr(v),
                                        ptr(target), ptr(target_T), ptr(t_dev), float(lr), float(tau), float(grad_scale),
                                        prec_id(precision), cur_stream()), "mobody_adam_polyak_dev")


def gather_batch_rng(buffers, counts, seeds, call_offsets, counter, sizes, S, A, out, bump=()):
    """Gather with device-drawn indices: buffers = list of RingView / 5-tuples, sizes = list of device int64[1] views,
    counter = device int64[1] or None.  `out` = (state, action, next_state, reward, not_done) destination tensors.
    bump: up to four device int64[1] words (not `counter`) the launch increments by one."""
    n = len(buffers)
    views = (_lib.MobodyBufferView * n)(*[buffer_view(b) for b in buffers])
    cnt = (C.c_int64 * n)(*[int(c) for c in counts])
    sd = (C.c_uint32 * n)(*[int(s) & 0xFFFFFFFF for s in seeds])
    off = (C.c_int64 * n)(*[int(o) for o in call_offsets])
    sz = (C.c_void_p * n)(*[ptr(s) for s in sizes])
    bp = (C.c_void_p * max(1, len(bump)))(*[ptr(t) for t in bump])
    check(load().mobody_gather_batch_rng(views, cnt, n, S, A, sd, off, ptr(counter), sz, *[ptr(t) for t in out], bp, len(bump),
                                         cur_stream()), "mobody_gather_batch_rng")
    return out


# ------------------------------------------------------------------------------------------------
# generic MLP gradient + DARA classifier pieces
# ------------------------------------------------------------------------------------------------
def mlp3_backward(blob_T, in_dim, out_dim, members, dz3, x, h1, h2, grad, ws=None):
    rows = x.shape[0]
    need = load().mobody_mlp3_backward_workspace(in_dim, out_dim, members, rows)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=x.device)
    check(load().mobody_mlp3_backward(ptr(blob_T), in_dim, out_dim, members, ptr(dz3), ptr(x), ptr(h1), ptr(h2), rows,
                                      ptr(grad), ptr(ws), cur_stream()), "mobody_mlp3_backward")
    return ws


def dara_inputs(s, a, s2, std, noise_sas=None, noise_sa=None, seed=0, call=0):
    N, S = s.shape
    A = a.shape[1]
    x_sas = torch.empty(N, 2 * S + A, dtype=torch.float32, device=s.device)
    x_sa = torch.empty(N, S + A, dtype=torch.float32, device=s.device)
    check(load().mobody_dara_inputs(ptr(s), ptr(a), ptr(s2), N, S, A, float(std), ptr(noise_sas), ptr(noise_sa), seed,
                                    call, ptr(x_sas), ptr(x_sa), cur_stream()), "mobody_dara_inputs")
    return x_sas, x_sa


def dara_loss_grad(z_sas, z_sa, n_src, labels=None):
    """z_*: logits [N,2].  Returns (dz_sas[N,16], dz_sa[N,16], loss[2] = (loss_sa, loss_sas))."""
    N = z_sas.shape[0]
    dev = z_sas.device
    dz_sas = torch.empty(N, 16, dtype=torch.float32, device=dev)
    dz_sa = torch.empty(N, 16, dtype=torch.float32, device=dev)
    loss = torch.empty(2, dtype=torch.float32, device=dev)
    lossp = torch.empty(2 * ((N + 255) // 256), dtype=torch.float32, device=dev)
    if labels is not None:
        labels = torch.as_tensor(labels).to(device=dev, dtype=torch.int32).contiguous()
    check(load().mobody_dara_loss_grad(ptr(z_sas), ptr(z_sa), ptr(labels), N, int(n_src), ptr(dz_sas), ptr(dz_sa),
                                       ptr(loss), ptr(lossp), cur_stream()), "mobody_dara_loss_grad")
    return dz_sas, dz_sa, loss


def dara_penalty(z_sas, z_sa, coef, reward=None, want_delta=False):
    n = z_sas.shape[0]
    delta = torch.empty(n, 1, dtype=torch.float32, device=z_sas.device) if want_delta or reward is None else None
    check(load().mobody_dara_penalty(ptr(z_sas), ptr(z_sa), n, float(coef), ptr(reward), ptr(delta), cur_stream()),
          "mobody_dara_penalty")
    return delta


# ------------------------------------------------------------------------------------------------
# dynamics pre-training
# ------------------------------------------------------------------------------------------------
def pretrain_transpose(blob, S, A, out=None, precision=0):
    """T blob of the pre-training parameter blob; `precision` (0 / "f32" or 4 / "f16x2") = the mode it will be trained in."""
    L = _lib.pretrain_layout(S, A)
    bt = out if out is not None else torch.zeros(L.t_total_floats, dtype=torch.float32, device=blob.device)
    check(load().mobody_pretrain_transpose(S, A, ptr(blob), ptr(bt), prec_id(precision), cur_stream()), "mobody_pretrain_transpose")
    return bt


def pretrain_workspace(S, A, b, device):
    n = load().mobody_pretrain_workspace(S, A, b)
    if n < 0:
        raise _lib.MobodyError("mobody_pretrain_workspace: " + load().mobody_last_error().decode())
    return torch.empty(n, dtype=torch.float32, device=device)


def pretrain_gather(state, action, next_state, reward, idx, start, b, out=None, start_dev=None):
    """idx: device int32 [7, n_idx].  Returns (xenc[7,2b,S], act[7,b,A], rew[7,b])."""
    S, A = state.shape[1], action.shape[1]
    dev = state.device
    assert idx.dtype == torch.int32 and idx.dim() == 2 and idx.shape[0] == 7 and idx.is_contiguous()
    xenc, act, rew = out or (torch.empty(7, 2 * b, S, dtype=torch.float32, device=dev),
                             torch.empty(7, b, A, dtype=torch.float32, device=dev),
                             torch.empty(7, b, dtype=torch.float32, device=dev))
    check(load().mobody_pretrain_gather(ptr(state), ptr(action), ptr(next_state), ptr(reward), ptr(idx), idx.shape[1], start,
                                        ptr(start_dev), b, S, A, ptr(xenc), ptr(act), ptr(rew), cur_stream()),
          "mobody_pretrain_gather")
    return xenc, act, rew


def pretrain_grads(S, A, b, use_trg, encoder_loss_coef, blob, blob_T, xenc, act, rew, grad, loss_out, ws, noise6=None,
                   noise7=None, seed=0, call=0, b_global=None, precision=0, transition_coef=1.0, reward_coef=1.0):
    check(load().mobody_pretrain_grads(S, A, b, b if b_global is None else b_global, int(bool(use_trg)),
                                       float(encoder_loss_coef), ptr(blob), ptr(blob_T), ptr(xenc), ptr(act), ptr(rew),
                                       ptr(noise6), ptr(noise7), seed, call, ptr(grad), ptr(loss_out), ptr(ws),
                                       prec_id(precision), float(transition_coef), float(reward_coef), cur_stream()), "mobody_pretrain_grads")


def pretrain_update(S, A, b, use_trg, encoder_loss_coef, blob, blob_T, xenc, act, rew, m, v, t_main, t_za, lr, loss_out, ws,
                    noise6=None, noise7=None, seed=0, call=0, call_dev=None, t_dev=None, precision=0, loss_acc=None):
    check(load().mobody_pretrain_update(S, A, b, int(bool(use_trg)), float(encoder_loss_coef), ptr(blob), ptr(blob_T), ptr(xenc),
                                        ptr(act), ptr(rew), ptr(noise6), ptr(noise7), seed, call, ptr(call_dev), ptr(m), ptr(v),
                                        t_main, t_za, ptr(t_dev), float(lr), ptr(loss_out), ptr(loss_acc), ptr(ws), prec_id(precision), cur_stream()),
          "mobody_pretrain_update")


def pretrain_adam(S, A, use_trg, blob, blob_T, grad, m, v, t_main, t_za, lr, grad_scale=1.0, precision=0, net_mask=7, t_rw=None):
    """net_mask: bit 0 state encoder, 1 decoder, 2 reward head (a net without a gradient this step is skipped); t_rw: the reward
    head's own step count (default: t_main)."""
    check(load().mobody_pretrain_adam(S, A, int(bool(use_trg)), ptr(blob), ptr(blob_T), ptr(grad), ptr(m), ptr(v), t_main,
                                      t_za, float(lr), float(grad_scale), prec_id(precision), int(net_mask),
                                      t_main if t_rw is None else t_rw, cur_stream()), "mobody_pretrain_adam")


def pretrain_za_adam(S, A, use_trg, blob, grad, m, v, t_za, lr, grad_scale=1.0):
    """Adam step of one action encoder only (the second one of a learn_src_trg step)."""
    check(load().mobody_pretrain_za_adam(S, A, int(bool(use_trg)), ptr(blob), ptr(grad), ptr(m), ptr(v), t_za, float(lr),
                                         float(grad_scale), cur_stream()), "mobody_pretrain_za_adam")


def dyn_validate(blob, S, A, obs, act, next_obs, rew, use_trg, ws=None):
    """validate(): out[0:7] per-member transition MSE, out[7:14] per-member reward MSE (device tensor)."""
    B = obs.shape[0]
    need = load().mobody_dyn_validate_workspace(S, A, B)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=obs.device)
    out = torch.empty(14, dtype=torch.float32, device=obs.device)
    check(load().mobody_dyn_validate(ptr(blob), S, A, ptr(_f32(obs)), ptr(_f32(act)), ptr(_f32(next_obs)),
                                     ptr(_f32(rew).reshape(-1).contiguous()), B, int(bool(use_trg)), ptr(out), ptr(ws),
                                     cur_stream()), "mobody_dyn_validate")
    return out


def rollout(dyn_blob, actor_blob, S, A, task_id, max_action, init_obs, H, elites, seed, call0, penalty_coef, use_penalty,
            use_trg, env_filter, filter_bad_rollout, buf, cap, ptr_size, ws=None, dyn_planes=None, actor_blob_T=None,
            precision=0):
    """H-step on-device rollout of `init_obs` appended to the ring `buf` (mobody_rollout).  Returns the workspace."""
    B = init_obs.shape[0]
    need = load().mobody_rollout_workspace(S, A, B)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1), dtype=torch.float32, device=init_obs.device)
    el = (C.c_int32 * len(elites))(*[int(e) for e in elites])
    check(load().mobody_rollout(ptr(dyn_blob), ptr(dyn_planes), ptr(actor_blob), ptr(actor_blob_T), prec_id(precision), S, A,
                                task_id, float(max_action), ptr(_f32(init_obs)), B, int(H),
                                el, len(elites), seed, call0, float(penalty_coef), int(bool(use_penalty)), int(bool(use_trg)),
                                float(env_filter), int(bool(filter_bad_rollout)), C.byref(buffer_view(buf)), cap, ptr(ptr_size),
                                ptr(ws), cur_stream()), "mobody_rollout")
    return ws
