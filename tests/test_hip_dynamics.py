"""HIP ensemble-dynamics path vs the oracle and the reference's golden vectors (GPU box only).

Tolerance: north_star asks 1e-5 fp32; the MFMA f32 chain sums in a different k order than
MKL, so outputs of O(1) magnitude are compared with atol=rtol=1e-5 (stated per assert).
"""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import mobody_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


def close(a, b, **kw):
    kw = {**TOL, **kw}
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a.astype(np.float64), b.astype(np.float64), **kw)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("tag", ["walker", "ant", "pen"])
def test_dyn_forward_and_step_vs_golden(tag, mfma, dev):
    from mobody_amd import ops, packing, _lib
    g = gu.load(f"g234_dynamics_{tag}")
    S, A = int(g["S"]), int(g["A"])
    p = gu.dyn_params_for(g)
    blob = packing.pack_dynamics(p, S, A, dev)
    kw = gu.dyn_kw(blob, S, A, mfma)
    obs, act = torch.from_numpy(g["obs"]).to(dev), torch.from_numpy(g["act"]).to(dev)
    close(ops.dyn_forward(blob, S, A, obs, act, True, **kw), g["mean_trg"])
    close(ops.dyn_forward(blob, S, A, obs, act, False, **kw), g["mean_src"])
    task = _lib.TERM_IDS[O.resolve_task(str(g["task"]))]
    for up in (1, 0):
        for ut in (1, 0):
            k = f"step_p{up}_t{ut}_"
            r = ops.dyn_step(blob, S, A, task, obs, act, noise=g[k + "eps"], elite_idx=g[k + "idx"], penalty_coef=0.1,
                             use_penalty=bool(up), use_trg=bool(ut), want_mean=True, **kw)
            close(r["mean"], g[k + "samples"])
            close(r["next_obs"], g[k + "next_obs"])
            close(r["penalty"], g[k + "penalty"])
            close(r["raw_reward"], g[k + "raw_reward"])
            close(r["reward"], g[k + "reward"])
            # the predicate is discontinuous: rows whose decisive coordinate sits within 1e-5 of a
            # threshold may legitimately flip; none of the fixture rows do
            assert (r["terminal"].cpu().numpy().astype(bool) == g[k + "terminal"]).all()


@pytest.mark.parametrize("B", [1, 63, 64, 65, 200])
def test_dyn_step_ragged_batches_vs_oracle(B, mfma, dev):
    from mobody_amd import ops, packing
    S, A = 17, 6
    p = gu.gi.dyn_params(7, S, A)
    p["transition3.bias"][:, 0, 0] += np.float32(0.85)
    blob = packing.pack_dynamics(p, S, A, dev)
    rng = np.random.default_rng(B)
    obs = gu.gi.walker_like_obs(rng, B, S); act = rng.uniform(-1, 1, (B, A)).astype(np.float32)
    eps = rng.standard_normal((7, B, S)).astype(np.float32); idx = rng.integers(0, 5, B)
    with torch.no_grad():
        want = O.dyn_step(O.to_torch(p), obs, act, eps, idx, "walker2d-medium-v2", penalty_coef=0.25)
    got = ops.dyn_step(blob, S, A, 4, torch.from_numpy(obs).to(dev), torch.from_numpy(act).to(dev), noise=eps,
                       elite_idx=idx, penalty_coef=0.25, **gu.dyn_kw(blob, S, A, mfma))
    for k in ("next_obs", "reward", "penalty", "raw_reward"):
        close(got[k], want[k])
    assert (got["terminal"].cpu().numpy().astype(bool) == want["terminal"]).all()


def test_dyn_step_empty_batch(dev):
    from mobody_amd import ops, packing
    blob = packing.pack_dynamics(gu.gi.dyn_params(7, 17, 6), 17, 6, dev)
    r = ops.dyn_step(blob, 17, 6, 4, torch.zeros(0, 17, device=dev), torch.zeros(0, 6, device=dev),
                     noise=np.zeros((7, 0, 17), np.float32), elite_idx=np.zeros(0, np.int64))
    assert r["next_obs"].shape == (0, 17)


def test_termination_predicates_vs_golden(dev):
    """The predicate FUSED in k_dyn_sample sees the G5 rows themselves (boundary values 0.8, 2.0, +-1, +-100, NaN, Inf).
    Model: all weights zero and transition3.bias = c_e * 1 with c = (-1,-1,-1,0,1,1,1), so every mean is its member's
    constant, the ensemble mean is exactly 0 and the unbiased std exactly 1 (6/6); with elite member 3 (mean 0) and
    noise[3] = the fixture row, next_obs = 0 + row * 1 = the row, bit for bit -- then terminal must equal the
    reference's flag for that row."""
    from mobody_amd import ops, packing, _lib
    g = gu.load("g5_termination")
    c = np.array([-1, -1, -1, 0, 1, 1, 1], np.float32)
    for t in sorted({k.split("::")[0] for k in g if "::" in k}):
        n = g[t + "::next_obs"]
        B, S = n.shape
        A = 6
        p = {k: np.zeros_like(v) for k, v in gu.gi.dyn_params(3, S, A).items()}
        p["transition3.bias"] = np.broadcast_to(c[:, None, None], (7, 1, S)).astype(np.float32).copy()
        blob = packing.pack_dynamics(p, S, A, dev)
        noise = np.zeros((7, B, S), np.float32)
        noise[3] = n
        r = ops.dyn_step(blob, S, A, _lib.TERM_IDS[O.resolve_task(t)], torch.zeros(B, S, device=dev),
                         torch.zeros(B, A, device=dev), noise=noise, elite_idx=np.full(B, 3, np.int64))
        got = r["next_obs"].cpu().numpy()
        assert np.array_equal(got, n, equal_nan=True), t                    # the kernel's rows ARE the fixture rows
        assert (r["terminal"].cpu().numpy().astype(bool) == g[t + "::done"]).all(), t


def test_device_rng_matches_cpu_twin(dev):
    from mobody_amd import ops
    z = ops.rng_normal(123, 1, 7, 4099, dev).cpu().numpy()
    close(z, O.rng_normal(123, 1, 7, 4099), rtol=1e-5, atol=2e-6)
    i = ops.rng_index(123, 2, 7, 4099, 5, dev).cpu().numpy()
    assert (i == O.rng_index(123, 2, 7, 4099, 5)).all()
    i = ops.rng_index(9, 3, 0, 1000, 1000000, dev).cpu().numpy()
    assert (i == O.rng_index(9, 3, 0, 1000, 1000000)).all()


def test_dyn_step_device_rng_mode(dev):
    """noise=None/elite_idx=None: the kernel's own draws equal the stand-alone generator, and the
    step equals the oracle fed with those draws."""
    from mobody_amd import ops, packing
    S, A, B = 17, 6, 130
    p = gu.gi.dyn_params(7, S, A)
    p["transition3.bias"][:, 0, 0] += np.float32(0.85)
    blob = packing.pack_dynamics(p, S, A, dev)
    rng = np.random.default_rng(1)
    obs = gu.gi.walker_like_obs(rng, B, S); act = rng.uniform(-1, 1, (B, A)).astype(np.float32)
    elites = (1, 2, 4, 5, 6)
    got = ops.dyn_step(blob, S, A, 4, torch.from_numpy(obs).to(dev), torch.from_numpy(act).to(dev), elites=elites,
                       seed=42, call=3, penalty_coef=0.1)
    z = ops.rng_normal(42, 1, 3, B * S, dev).cpu().numpy().reshape(B, S)
    idx = np.asarray(elites)[ops.rng_index(42, 2, 3, B, 5, dev).cpu().numpy()]
    eps = np.broadcast_to(z, (7, B, S)).copy()
    with torch.no_grad():
        want = O.dyn_step(O.to_torch(p), obs, act, eps, idx, "walker2d-medium-v2", penalty_coef=0.1)
    for k in ("next_obs", "reward", "penalty"):
        close(got[k], want[k])


@pytest.mark.parametrize("S,A", [(17, 6), (111, 8), (45, 24)])
def test_mlp3_forward_actor_and_twin_q(S, A, dev):
    from mobody_amd import ops, packing
    pa, pq, _ = gu.policy_params(401, S, A)
    rows = 150
    s, a, _, _, _ = gu.gi.batch(11, rows, S, A)
    ab = packing.pack_mlp([{k[len("network."):]: v for k, v in pa.items()}], S, A, dev)
    qb = packing.pack_mlp(pq, S + A, 1, dev, prefixes=["network1.", "network2."])
    sd, ad = torch.from_numpy(s).to(dev), torch.from_numpy(a).to(dev)
    with torch.no_grad():
        want_pi = O.actor(O.to_torch(pa), O.T(s), 1.0)
        w1, w2 = O.twin_q(O.to_torch(pq), O.T(s), O.T(a))
    pi = ops.mlp3_forward(ab, S, A, 1, sd, out_mode=1, max_action=1.0)
    close(pi[0], want_pi)
    q, sx, h1, h2 = ops.mlp3_forward(qb, S + A, 1, 2, sd, ad, save=True)
    close(q[0], w1); close(q[1], w2)
    # saved activations are the post-ReLU hidden layers
    x = torch.cat([O.T(s), O.T(a)], 1)
    P = O.to_torch(pq)
    hh1 = torch.relu(torch.nn.functional.linear(x, P["network2.network.0.weight"], P["network2.network.0.bias"]))
    close(h1[1], hh1)
    close(sx[:, :S + A], x)
    # unpack(pack(p)) round trip
    back = packing.unpack_mlp(qb, S + A, 1, 2)
    for m, pre in enumerate(("network1.", "network2.")):
        for k, v in back[m].items():
            close(v, pq[pre + k], rtol=0, atol=0)


def test_nan_member_poisons_penalty_and_row_is_dropped(dev):
    """One non-finite ensemble member: torch.amax propagates the NaN norm (mobody_dynamics.py:246-249), so the
    reference's `penalty <= env_filter` (mobody.py:649) and `penalty < env_filter` (:466) are False and the row never
    reaches the fake buffer.  The kernel's max must not swallow the NaN (fmaxf would)."""
    from mobody_amd import ops, packing
    S, A, B = 17, 6, 70
    p = gu.gi.dyn_params(7, S, A)
    p["transition3.bias"][:, 0, 0] += np.float32(0.85)
    p["transition3.bias"][2, 0, 5] = np.nan
    blob = packing.pack_dynamics(p, S, A, dev)
    rng = np.random.default_rng(3)
    obs = gu.gi.walker_like_obs(rng, B, S); act = rng.uniform(-1, 1, (B, A)).astype(np.float32)
    eps = rng.standard_normal((7, B, S)).astype(np.float32); idx = rng.integers(0, 5, B)
    with torch.no_grad():
        want = O.dyn_step(O.to_torch(p), obs, act, eps, idx, "walker2d-medium-v2", penalty_coef=0.1)
    assert np.isnan(want["penalty"].numpy()).all()
    got = ops.dyn_step(blob, S, A, 4, torch.from_numpy(obs).to(dev), torch.from_numpy(act).to(dev), noise=eps,
                       elite_idx=idx, penalty_coef=0.1)
    assert torch.isnan(got["penalty"]).all()
    keep = torch.empty(B, dtype=torch.uint8, device=dev); alive = torch.empty(B, dtype=torch.uint8, device=dev)
    ops.rollout_mask(None, got["terminal"], got["penalty"], 1e9, True, keep, alive)
    assert int(keep.sum()) == 0
