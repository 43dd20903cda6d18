"""Split-precision MFMA modes (BASELINE configs[1]: "bf16 MFMA inputs / fp32 accumulate") of the 256 x 256 layers: the
actor / twin-Q / target-Q forwards and backwards of train() and its 256 x 256 weight-gradient job, zs2 / transition2 /
reward_model2 of the ensemble step.  Layer 1, every narrow layer, reductions and the optimizer stay exact fp32.

    mfma      terms / products      measured deviation from the fp32 kernels (relative to max |output|)
    'f16x2'   2 fp16 / 3            ~3e-7   -> held to the SAME tolerances as the fp32 path (1e-5, north_star); bench default
    'bf16x3'  3 bf16 / 6            ~5e-7   -> the same
    'bf16x2'  2 bf16 / 3            ~6e-6   -> 5e-5 here
    'bf16'    1 bf16 / 1            ~3e-3   -> 3e-2 here
The golden-vector suites (test_hip_train / _mirror / _dynamics / _pretrain / _fullsize) run in 'f32' AND in the bench's
mode through the `mfma` fixture of conftest.py; this file holds what is specific to the split modes."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import mobody_oracle as O

pytestmark = pytest.mark.gpu
TOL = {"f16x2": 1e-5, "bf16x3": 1e-5, "bf16x2": 5e-5, "bf16": 3e-2}
FP32_GRADE = ("f16x2", "bf16x3")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rel(a, b):
    a = a.detach().cpu().numpy().astype(np.float64) if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().numpy().astype(np.float64) if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3", "bf16x2", "bf16"])
@pytest.mark.parametrize("S,A,rows", [(17, 6, 333), (111, 8, 64), (45, 24, 97)])
def test_mlp3_forward_modes_vs_oracle(mode, S, A, rows, dev):
    from mobody_amd import ops, packing
    pa, pq, _ = gu.policy_params(401, S, A)
    s, a, _, _, _ = gu.gi.batch(11, rows, S, A)
    ab = packing.pack_mlp([{k[len("network."):]: v for k, v in pa.items()}], S, A, dev)
    qb = packing.pack_mlp(pq, S + A, 1, dev, prefixes=["network1.", "network2."])
    aT, qT = ops.mlp_transpose(ab, S, A, 1, precision=mode), ops.mlp_transpose(qb, S + A, 1, 2, precision=mode)
    sd, ad = torch.from_numpy(s).to(dev), torch.from_numpy(a).to(dev)
    with torch.no_grad():
        want_pi = O.actor(O.to_torch(pa), O.T(s), 1.0)
        w1, w2 = O.twin_q(O.to_torch(pq), O.T(s), O.T(a))
    pi = ops.mlp3_forward(ab, S, A, 1, sd, out_mode=1, max_action=1.0, blob_T=aT, precision=mode)
    q, sx, h1, h2 = ops.mlp3_forward(qb, S + A, 1, 2, sd, ad, save=True, blob_T=qT, precision=mode)
    assert rel(pi[0], want_pi) <= TOL[mode]
    assert rel(q[0], w1) <= TOL[mode] and rel(q[1], w2) <= TOL[mode]
    # the saved activations (what the exact-fp32 backward consumes) come from the same pass
    x = torch.cat([O.T(s), O.T(a)], 1)
    P = O.to_torch(pq)
    hh1 = torch.relu(torch.nn.functional.linear(x, P["network1.network.0.weight"], P["network1.network.0.bias"]))
    assert rel(h1[0], hh1) <= 1e-5                                           # layer 1 is exact fp32 in every mode
    hh2 = torch.relu(torch.nn.functional.linear(hh1, P["network1.network.2.weight"], P["network1.network.2.bias"]))
    assert rel(h2[0], hh2) <= TOL[mode]


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3", "bf16x2", "bf16"])
@pytest.mark.parametrize("tag", ["walker", "ant", "pen"])
def test_dyn_step_modes_vs_reference_golden(mode, tag, dev):
    from mobody_amd import ops, packing, _lib
    g = gu.load(f"g234_dynamics_{tag}")
    S, A = int(g["S"]), int(g["A"])
    blob = packing.pack_dynamics(gu.dyn_params_for(g), S, A, dev)
    planes = ops.dyn_planes(blob, S, A, precision=mode)
    obs, act = torch.from_numpy(g["obs"]).to(dev), torch.from_numpy(g["act"]).to(dev)
    assert rel(ops.dyn_forward(blob, S, A, obs, act, True, planes=planes, precision=mode), g["mean_trg"]) <= TOL[mode]
    assert rel(ops.dyn_forward(blob, S, A, obs, act, False, planes=planes, precision=mode), g["mean_src"]) <= TOL[mode]
    k = "step_p1_t1_"
    r = ops.dyn_step(blob, S, A, _lib.TERM_IDS[O.resolve_task(str(g["task"]))], obs, act, noise=g[k + "eps"], elite_idx=g[k + "idx"],
                     penalty_coef=0.1, planes=planes, precision=mode)
    for key in ("next_obs", "reward", "raw_reward"):
        assert rel(r[key], g[k + key]) <= TOL[mode], key
    assert rel(r["penalty"], g[k + "penalty"]) <= 20 * TOL[mode]              # a difference of means: relative error amplifies
    if mode in FP32_GRADE:
        assert (r["terminal"].cpu().numpy().astype(bool) == g[k + "terminal"]).all()


@pytest.mark.parametrize("mode", FP32_GRADE)
def test_train_step_fp32_grade_modes_meet_the_fp32_parity_bar(mode, dev):
    """G7 'default' (two train() steps of the reference) with every 256 x 256 GEMM on the split core: the same
    assertions and tolerances as tests/test_hip_train.py::test_train_step_vs_reference_golden."""
    from mobody_amd.engine import Engine
    from test_hip_train import close, params_close
    g = gu.load("g7_train_default")
    S, A, bs = int(g["S"]), int(g["A"]), int(g["bs"])
    cfg = gu.policy_cfg(S, A, mfma=mode)
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
    # the planes every forward streamed were kept current by the optimizer kernels: rebuilding them changes nothing
    from mobody_amd import _lib, ops
    assert torch.equal(ops.mlp_transpose(eng.q, S + A, 1, 2, precision=mode), eng.q_T)
    assert torch.equal(ops.mlp_transpose(eng.actor, S, A, 1, precision=mode), eng.actor_T)
    L = _lib.mlp_layout(S + A, 1, 2)                     # target net: only its W2 planes follow the Polyak update
    fresh = ops.mlp_transpose(eng.qt, S + A, 1, 2, precision=mode).view(2, L.t_member_floats)
    assert torch.equal(fresh[:, L.w2p:L.w2tp], eng.qt_T.view(2, L.t_member_floats)[:, L.w2p:L.w2tp])


@pytest.mark.parametrize("mode,tol", [("bf16x2", 2e-4), ("bf16", 5e-2)])
def test_train_step_lower_modes_vs_oracle(mode, tol, dev):
    """bf16x2 / bf16 forwards: losses and gradients of one step against the oracle at the tolerance the mode achieves."""
    from mobody_amd.engine import Engine
    S, A, N, Nt = 17, 6, 640, 512
    cfg = gu.policy_cfg(S, A, mfma=mode)
    pa, pq, pv = gu.policy_params(77, S, A)
    batch = gu.gi.batch(5, N, S, A)
    st = O.TrainState(pa, pq, pv)
    want = O.train_step(st, batch, Nt, cfg, apply=False)
    eng = Engine(S, A, pa, pq, dev)
    out = eng.step(batch, Nt, cfg, apply=False)
    assert abs(out["q_loss"] - float(want["q_loss"])) <= tol * abs(float(want["q_loss"]))
    assert abs(out["pi_loss"] - float(want["pi_loss"])) <= 5 * tol * abs(float(want["pi_loss"]))
    for nm, blob, key in (("q", eng.gq, "q_grads"), ("actor", eng.ga, "actor_grads")):
        scale = max(float(v.abs().max()) for v in want[key].values())
        for k, v in eng.unpack(blob, nm).items():
            d = float((v.cpu() - want[key][k]).abs().max())
            assert d <= 10 * tol * scale, (nm, k, d, scale)


@pytest.mark.parametrize("split", FP32_GRADE)
def test_mirror_graph_steps_fp32_grade_modes_track_the_fp32_run(split, dev):
    """The mirror in a fp32-grade split mode (device RNG, graph replay, refresh through mobody_rollout at step 1): six train() calls
    stay within 2e-4 of the same run in exact fp32 (identical seeds -> identical minibatches; the step-1 rollout rows differ
    by the mode's ~5e-7), and the W2 planes every forward streams were kept current by the fused optimizer kernels."""
    from mobody_amd import ops, synthetic
    from mobody_amd.algo import utils
    from mobody_amd.algo.call_algo import call_algo
    from mobody_amd.algo.dynamics.mobody_dynamics import MOBODYEnsembleDynamics
    from mobody_amd.algo.dynamics.mobody_module import MOBODYModule
    from mobody_amd.algo.mb_utils.terminal_funs import get_termination_fn
    S, A, bs, task = 17, 6, 256, "walker2d-medium-v2"
    res = {}
    for mode in ("f32", split):
        cfg = gu.policy_cfg(S, A, rng="device", seed=3, mfma=mode, graph=1, batch_size=bs)
        torch.manual_seed(0); np.random.seed(0)
        pol = call_algo("mobody", cfg, 3, dev)
        src = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=4000, rng="device", seed=100), 4000, task, 0)
        tar = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=500, rng="device", seed=200), 500, task, 50)
        torch.manual_seed(1)
        model = synthetic.alive_dynamics(MOBODYModule(S, A, 256, 7, 5, device=dev, config=cfg), task)
        pol.dynamics = MOBODYEnsembleDynamics(cfg, model, None, None, get_termination_fn(task), penalty_coef=0.1, rng="device", seed=9)
        for _ in range(6):
            pol.train(src, tar, bs, None, None)
        torch.cuda.synchronize()
        assert pol._graph is not None and pol.fake_replay_buffer.size > 0 and pol.precision == ops.prec_id(mode)
        res[mode] = {k: v.cpu() for k, v in list(pol.policy.state_dict().items()) + list(pol.q_funcs.state_dict().items())}
        res[mode]["fake_size"] = pol.fake_replay_buffer.size
        assert torch.equal(ops.mlp_transpose(pol.q_funcs.blob, S + A, 1, 2, precision=mode), pol.q_funcs.blob_T)
        assert torch.equal(ops.mlp_transpose(pol.policy.blob, S, A, 1, precision=mode), pol.policy.blob_T)
        L = pol.q_funcs.layout
        fresh = ops.mlp_transpose(pol.target_q_funcs.blob, S + A, 1, 2, precision=mode).view(2, L.t_member_floats)
        assert torch.equal(fresh[:, L.w2p:L.w2tp], pol.target_q_funcs.blob_T.view(2, L.t_member_floats)[:, L.w2p:L.w2tp])
    assert abs(res["f32"]["fake_size"] - res[split]["fake_size"]) <= 3      # a row at the filter / termination edge may flip
    for k in res["f32"]:
        if k != "fake_size":
            np.testing.assert_allclose(res[split][k].numpy(), res["f32"][k].numpy(), rtol=0, atol=2e-4, err_msg=k)
