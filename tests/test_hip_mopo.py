"""The MOPO ablation (config['mopo'] = 1, mobody_module.py:114-118,218-219,251-254,264-266,288-289) on the HIP path:
means = s + MLP_e([s, a]) for both models, then the unchanged step.  Fixture g18 is the reference run with that flag."""
import numpy as np
import pytest
import torch

import golden_util as gu
from test_hip_mirror import close, feed, make_dynamics

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("mode", ["f32", "f16x2", "bf16x3"])
@pytest.mark.parametrize("tag", ["walker", "ant"])
def test_mopo_forward_and_step_vs_reference_golden(tag, mode, dev):
    g = gu.load(f"g18_mopo_{tag}")
    S, A, task = int(g["S"]), int(g["A"]), str(g["task"])
    cfg = gu.policy_cfg(S, A, mopo=1, mfma=mode)
    dyn = make_dynamics(gu.mopo_params_for(g, tag), S, A, task, dev, cfg)
    m = dyn.model
    assert m.mopo and tuple(m.state_dict()["za_src3.weight"].shape) == (7, 256, S)
    obs, act = torch.from_numpy(g["obs"]).to(dev), torch.from_numpy(g["act"]).to(dev)
    close(m.forward_trg(obs, act)[0], g["mean_trg"]); close(m.forward_src(obs, act)[0], g["mean_src"])
    for up in (1, 0):
        for ut in (1, 0):
            k = f"step_p{up}_t{ut}_"
            feed(dyn, [g[k + "eps"]])
            np.random.seed(int(g["seed"]))                      # the golden run drew its elite ids from this NumPy state
            no, rw, term, info = dyn.step(obs, act, bool(up), bool(ut))
            close(no, g[k + "next_obs"]); close(rw, g[k + "reward"]); close(info["penalty"], g[k + "penalty"])
            close(info["raw_reward"], g[k + "raw_reward"])
            assert (term == g[k + "terminal"]).all()
    with pytest.raises(NotImplementedError):                       # pre-training of the ablation is not part of the build
        m.train_state()


def test_mopo_rollout_and_refresh_through_the_mirror(dev):
    """MOBODY.rollout on the mopo model == the reference's 3-step rollout; a device-RNG refresh takes the host loop."""
    from mobody_amd.algo.offline_offline.mobody import MOBODY
    from mobody_amd import synthetic
    from mobody_amd.algo import utils
    g = gu.load("g18_mopo_walker")
    S, A, task = int(g["S"]), int(g["A"]), str(g["task"])
    cfg = gu.policy_cfg(S, A, mopo=1, env_filter=float(g["env_filter"]))
    pol = MOBODY(cfg, dev)
    pa, _, _ = gu.policy_params(int(g["actor_seed"]), S, A)
    pol.policy.load_state_dict({k: torch.from_numpy(v) for k, v in pa.items()})
    pol.dynamics = make_dynamics(gu.mopo_params_for(g, "walker"), S, A, task, dev, cfg)
    n = int(g["n_steps"])
    feed(pol.dynamics, [g[f"roll_eps{t}"] for t in range(n)])
    np.random.seed(78)
    res, info = pol.rollout(torch.from_numpy(g["obs"]).to(dev), 3, True)
    assert info["num_transitions"] == int(g["num_transitions"])
    for k in ("obss", "next_obss", "actions", "rewards", "terminals", "penalty"):
        assert tuple(res[k].shape) == g["roll_" + k].shape, k
        close(res[k], g["roll_" + k], rtol=2e-5, atol=2e-5)
    # device-RNG mode: train() step 1 refreshes the fake buffer through the host loop (mobody_rollout has no mopo form)
    cfg2 = gu.policy_cfg(S, A, mopo=1, rng="device", seed=3)
    pol2 = MOBODY(cfg2, dev)
    pol2.dynamics = make_dynamics(gu.mopo_params_for(g, "walker"), S, A, task, dev, cfg2, rng="device", seed=4)
    src = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=60000, rng="device", seed=1), 60000, task, 0)
    tar = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=3000, rng="device", seed=2), 3000, task, 1)
    pol2.train(src, tar, 64, None, None)
    assert pol2.fake_replay_buffer.size > 0 and torch.isfinite(pol2.fake_replay_buffer.state[:pol2.fake_replay_buffer.size]).all()
    assert all(v == v for v in pol2.losses())
