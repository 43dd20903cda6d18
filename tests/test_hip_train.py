"""HIP training step (critic / actor / Adam+Polyak) vs the oracle and the reference's golden vectors.

Per-step comparison from identical state (never after many steps).  Tolerances are written at
each assert:
  * losses: q_loss 1e-5 relative; pi_loss / bc_loss 5e-5 relative + 2e-5 absolute when they are
    evaluated AFTER the critic's Adam step (see the Adam note below: they inherit O(lr) differences
    of a few critic weights), 2e-6 when evaluated on identical weights (tools/diag_grad_error.py);
  * gradients: |hip - ref| <= 1e-5 * max|g_network| + 1e-5*|g|  (north_star's 1e-5, relative to the
    scale of the network's gradient; bias sums of +-terms cancel, so the tensor's own max is not the scale.  tools/diag_grad_error.py measured, against an fp64 evaluation of the same
    step, 1e-8..3.3e-7 of max|g| for the fp32 reference and 1e-8..3.8e-7 for the HIP path: the two
    fp32 paths differ by summation order only);
  * parameters after Adam: Adam divides by sqrt(v)+1e-8, so an entry whose gradient is ~1e-8 turns
    rounding noise into an O(lr) difference -- 99.5 % of the entries must agree to 1e-5 rel + 1e-6
    abs and every entry to 0.1*lr; the Adam kernel itself is pinned to 1e-6 by test_adam_polyak_kernel.
"""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import mobody_oracle as O

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-5, atol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a.astype(np.float64), b.astype(np.float64), rtol=rtol, atol=atol)


def params_close(a, b, lr, max_frac=0.1):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    tight = d <= 1e-6 + 1e-5 * np.abs(b)
    assert tight.mean() >= 0.995, f"only {tight.mean():.4f} of the entries within 1e-5"
    assert d.max() <= max_frac * lr, f"max deviation {d.max():.3e} exceeds {max_frac}*lr"


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


from mobody_amd.engine import Engine  # noqa: E402  (the C-ABI driver ships with the package)


@pytest.mark.parametrize("tag", ["default", "noqw", "noscale", "nofake", "bc05"])
def test_train_step_vs_reference_golden(tag, mfma, dev):
    g = gu.load(f"g7_train_{tag}")
    S, A, bs = int(g["S"]), int(g["A"]), int(g["bs"])
    cfg = gu.policy_cfg(S, A, **gu.G7_VARIANTS[tag])
    pa, pq, _ = gu.policy_params(int(g["seed"]), S, A)
    eng = Engine(S, A, pa, pq, dev)
    batch, n_true = gu.g7_batch(cfg, bs, S, A)
    for step in (1, 2):
        out = eng.step(batch, n_true, cfg)
        close(out["q_loss"], g["q_loss"][step - 1], rtol=1e-5, atol=0)
        close(out["pi_loss"], g["pi_loss"][step - 1], rtol=5e-5, atol=2e-5)
        close(out["bc_loss"], g["bc_loss"][step - 1], rtol=5e-5, atol=2e-5)
        for nm, blob in (("q", eng.gq), ("actor", eng.ga)):
            ks = [k for k in g if k.startswith(f"s{step}_{nm}_g::")]
            scale = max(float(np.abs(g[k]).max()) for k in ks)
            for k, v in eng.unpack(blob, nm).items():
                close(gu.sub(v.cpu().numpy()), g[f"s{step}_{nm}_g::{k}"], rtol=1e-5, atol=1e-5 * scale)
        for nm, blob in (("q", eng.q), ("actor", eng.actor), ("qt", eng.qt)):
            for k, v in eng.unpack(blob, "actor" if nm == "actor" else "q").items():
                params_close(gu.sub(v.cpu().numpy()), g[f"s{step}_{nm}_p::{k}"], cfg["critic_lr"])


@pytest.mark.parametrize("S,A,N,Nt", [(17, 6, 640, 512), (17, 6, 333, 200), (111, 8, 192, 128), (45, 24, 130, 65)])
def test_train_step_vs_oracle_shapes(S, A, N, Nt, mfma, dev):
    cfg = gu.policy_cfg(S, A)
    pa, pq, pv = gu.policy_params(77, S, A)
    batch = gu.gi.batch(5, N, S, A)
    st = O.TrainState(pa, pq, pv)
    want = O.train_step(st, batch, Nt, cfg)
    eng = Engine(S, A, pa, pq, dev)
    got = eng.step(batch, Nt, cfg)
    close(got["q_loss"], float(want["q_loss"]), rtol=2e-5, atol=0)
    close(got["pi_loss"], float(want["pi_loss"]), rtol=5e-5, atol=2e-5)
    for nm, blob, grads in (("q", eng.gq, want["q_grads"]), ("actor", eng.ga, want["actor_grads"])):
        scale = max(float(gw.abs().max()) for gw in grads.values())
        for k, v in eng.unpack(blob, nm).items():
            close(v, grads[k].numpy(), rtol=1e-5, atol=1e-5 * scale)
    for nm, blob, params in (("q", eng.q, st.q), ("actor", eng.actor, st.actor), ("q", eng.qt, st.q_targ)):
        for k, v in eng.unpack(blob, nm).items():
            params_close(v, params[k], cfg["critic_lr"])


def test_train_step_without_true_rows_runs(mfma, dev):
    """Nt = 0 (no BC rows): the reference would average an empty tensor (NaN); the kernels define L_BC = 0."""
    S, A = 17, 6
    cfg = gu.policy_cfg(S, A)
    pa, pq, _ = gu.policy_params(77, S, A)
    out = Engine(S, A, pa, pq, dev).step(gu.gi.batch(5, 70, S, A), 0, cfg)
    assert np.isfinite(out["q_loss"]) and np.isfinite(out["pi_loss"]) and out["bc_loss"] == 0.0


def test_data_parallel_shards_sum_to_full_batch(mfma, dev):
    """N-GPU == 1-GPU by construction: run two half batches with N_global = N and sum the gradient blobs."""
    from mobody_amd import ops
    S, A, N, Nt = 17, 6, 256, 192
    cfg = gu.policy_cfg(S, A)
    pa, pq, _ = gu.policy_params(78, S, A)
    batch = gu.gi.batch(6, N, S, A)
    full = Engine(S, A, pa, pq, dev)
    full.step(batch, Nt, cfg, apply=False)
    # shard rows so that each rank holds half of the true rows and half of the fake rows
    perm = np.concatenate([np.arange(0, Nt // 2), np.arange(Nt, Nt + (N - Nt) // 2),
                           np.arange(Nt // 2, Nt), np.arange(Nt + (N - Nt) // 2, N)])
    halves = [perm[:N // 2], perm[N // 2:]]
    gq = torch.zeros_like(full.gq); ga = torch.zeros_like(full.ga)
    engines, stats = [], torch.zeros(2, device=dev)
    hyp = ops.hyper(cfg)
    parts = []
    for hsel in halves:
        e = Engine(S, A, pa, pq, dev)
        b = [torch.as_tensor(x[hsel], dtype=torch.float32).to(dev).contiguous() for x in batch]
        d = ops.train_dims(S, A, N // 2, Nt // 2, N, Nt)
        ws = ops.train_workspace(d, dev)
        ops.critic_step(d, hyp, e.actor, e.q, e.q_T, e.qt, b, e.gq, e.loss[0:1], ws, actor_blob_T=e.actor_T, qtarg_blob_T=e.qt_T)
        gq += e.gq
        ops.actor_forward(d, hyp, e.actor, e.q, b[0], b[1], e.stats, ws, actor_blob_T=e.actor_T, q_blob_T=e.q_T)
        stats += e.stats
        parts.append((e, b, d, ws))
    for e, b, d, ws in parts:           # "all-reduced" statistics, then the backward halves
        ops.actor_backward(d, hyp, e.actor, e.actor_T, e.q, e.q_T, b[0], b[1], stats, e.ga, e.loss[1:3], ws)
        ga += e.ga
    torch.cuda.synchronize()
    close(gq, full.gq, rtol=1e-5, atol=1e-5 * float(full.gq.abs().max()))
    close(ga, full.ga, rtol=1e-5, atol=1e-5 * float(full.ga.abs().max()))
    close(sum(float(p[0].loss[0]) for p in parts), float(full.loss[0]), rtol=1e-5, atol=0)
    close(sum(float(p[0].loss[1]) for p in parts), float(full.loss[1]), rtol=1e-5, atol=1e-7)


def test_adam_polyak_kernel(dev):
    """mobody_adam_polyak vs the oracle's torch.optim.Adam restatement on identical gradients (3 steps)."""
    from mobody_amd import ops, _lib
    S, A = 17, 6
    L = _lib.mlp_layout(S + A, 1, 2)
    rng = np.random.default_rng(3)
    n = L.total_floats
    p0 = rng.standard_normal(n).astype(np.float32) * 0.1
    tg0 = rng.standard_normal(n).astype(np.float32) * 0.1
    p, tg = torch.from_numpy(p0).to(dev), torch.from_numpy(tg0).to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    pt = torch.empty(L.t_total_floats, device=dev)
    P, M, V, TG = torch.from_numpy(p0.copy()), torch.zeros(n), torch.zeros(n), torch.from_numpy(tg0.copy())
    for t in (1, 2, 3):
        g = (rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 0, n)).astype(np.float32)
        ops.adam_polyak(S + A, 1, 2, p, pt, torch.from_numpy(g).to(dev), m, v, tg, t, 3e-4, 0.005)
        O.adam_update(P, torch.from_numpy(g), M, V, t, 3e-4)
        TG.copy_(0.005 * P + 0.995 * TG)
        close(p, P, rtol=2e-6, atol=1e-7); close(tg, TG, rtol=2e-6, atol=1e-7)
        # the kernel uses torch's lerp form m + (1-b1)(g-m); the oracle uses b1*m + (1-b1)*g: 1-ulp-of-g apart
        close(m, M, rtol=2e-6, atol=1e-8); close(v, V, rtol=2e-6, atol=1e-20)
    # the transposed blob mirrors the updated parameters (256-wide matrices use the interleaved storage)
    from mobody_amd import packing
    W2 = packing.wide_unpack(p[L.w2:L.w2 + 65536], 256)
    close(packing.wide_unpack(pt[L.w2t:L.w2t + 65536], 256), W2.t(), rtol=0, atol=0)
    W3 = p[L.w3:L.w3 + 256 * L.Np3].view(256, L.Np3)
    close(packing.wide_unpack(pt[L.w3t:L.w3t + 256 * L.Np3], L.Np3), W3.t(), rtol=0, atol=0)
    W1 = packing.wide_unpack(p[L.w1:L.w1 + L.Kp1 * 256], L.Kp1)
    close(pt[L.w1t:L.w1t + 256 * L.Np1t].view(256, L.Np1t)[:, :L.Kp1], W1.t(), rtol=0, atol=0)


@pytest.mark.parametrize("S,A,N,Nt", [(17, 6, 640, 512), (45, 24, 130, 65)])
def test_fused_update_is_bit_identical_to_step_plus_adam(S, A, N, Nt, mfma, dev):
    """mobody_critic_update / mobody_actor_update (Adam + Polyak inside the gradient reduction) vs the separate
    gradient + mobody_adam_polyak calls: same parameters, moments, target and transposed blobs, bit for bit."""
    from mobody_amd import ops
    cfg = gu.policy_cfg(S, A)
    pa, pq, _ = gu.policy_params(31, S, A)
    batch = gu.gi.batch(8, N, S, A)
    ref, fus = Engine(S, A, pa, pq, dev), Engine(S, A, pa, pq, dev)
    b = [torch.as_tensor(x, dtype=torch.float32).to(dev).contiguous() for x in batch]
    dims, hyp = ops.train_dims(S, A, N, Nt), ops.hyper(cfg)
    ws = ops.train_workspace(dims, dev)
    for step in (1, 2, 3):
        ref.step(batch, Nt, cfg)
        ops.critic_update(dims, hyp, fus.actor, fus.q, fus.q_T, fus.qt, b, fus.mq, fus.vq, step, cfg["critic_lr"], fus.loss[0:1], ws, actor_blob_T=fus.actor_T, qtarg_blob_T=fus.qt_T)
        ops.actor_forward(dims, hyp, fus.actor, fus.q, b[0], b[1], fus.stats, ws, actor_blob_T=fus.actor_T, q_blob_T=fus.q_T)
        ops.actor_update(dims, hyp, fus.actor, fus.actor_T, fus.q, fus.q_T, b[0], b[1], fus.stats, fus.ma, fus.va, step,
                         cfg["actor_lr"], fus.loss[1:3], ws)
        torch.cuda.synchronize()
        for name in ("q", "q_T", "qt", "mq", "vq", "actor", "actor_T", "ma", "va", "loss"):
            assert torch.equal(getattr(ref, name), getattr(fus, name)), (step, name)


def test_policy_forward_riding_with_the_target_q_launch_is_bit_identical(mfma, dev):
    """mobody_critic_step(policy_forward=1) + mobody_actor_forward(policy_ready=1) == the default placement of pi(s)."""
    from mobody_amd import ops
    S, A, N, Nt = 17, 6, 333, 200
    cfg = gu.policy_cfg(S, A)
    pa, pq, _ = gu.policy_params(41, S, A)
    b = [torch.as_tensor(x, dtype=torch.float32).to(dev).contiguous() for x in gu.gi.batch(9, N, S, A)]
    dims, hyp = ops.train_dims(S, A, N, Nt), ops.hyper(cfg)
    outs = []
    for ride in (False, True):
        e = Engine(S, A, pa, pq, dev)
        ws = ops.train_workspace(dims, dev)
        ops.critic_step(dims, hyp, e.actor, e.q, e.q_T, e.qt, b, e.gq, e.loss[0:1], ws, policy_forward=ride, actor_blob_T=e.actor_T, qtarg_blob_T=e.qt_T)
        ops.actor_forward(dims, hyp, e.actor, e.q, b[0], b[1], e.stats, ws, policy_ready=ride, actor_blob_T=e.actor_T, q_blob_T=e.q_T)
        ops.actor_backward(dims, hyp, e.actor, e.actor_T, e.q, e.q_T, b[0], b[1], e.stats, e.ga, e.loss[1:3], ws)
        torch.cuda.synchronize()
        outs.append((e.gq.clone(), e.ga.clone(), e.loss.clone(), e.stats.clone()))
    import os
    for x, y in zip(*outs):
        if os.environ.get("MOBODY_FWD_SHAPE"):   # single launches on another kernel shape
            close(x, y, rtol=1e-5, atol=1e-6 * float(y.abs().max()))
        else:
            assert torch.equal(x, y)
