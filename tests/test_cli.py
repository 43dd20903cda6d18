"""The CLI mirror (mobody_amd/train_mobody.py) against the reference's own argparse table (fixture g10, extracted
from the reference source text by tests/golden/make_cli_golden.py), and one short end-to-end run on the GPU."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_flags_match_the_reference_table():
    from mobody_amd import train_mobody as tm
    want = {f["flag"]: f for f in json.load(open(os.path.join(ROOT, "tests", "golden", "g10_cli_flags.json")))}
    got = {a.option_strings[0]: a for a in tm.build_parser()._actions if a.option_strings and a.option_strings[0] != "-h"}
    extra = {"--synthetic", "--rng", "--src_rows", "--tar_rows", "--log_every", "--dynamics_max_epochs", "--scalars", "--src_data", "--tar_data"}   # additions of this build
    assert set(got) - extra == set(want)
    for flag, f in want.items():
        a = got[flag]
        if f["action"] == "store_true":
            assert a.const is True and a.default is False, flag
            continue
        assert a.default == f["default"] and type(a.default) is type(f["default"]), (flag, a.default, f["default"])
        assert (a.type.__name__ if a.type is not None else None) == f["type"], flag


def test_config_merge_precedence():
    """yaml <- --params JSON <- CLI-derived keys (train_mobody.py:407-416, 470-531)."""
    from mobody_amd import train_mobody as tm
    args = tm.build_parser().parse_args(["--policy", "MOBODY", "--env", "walker2d-friction", "--shift_level", "2.0",
                                         "--params", '{"batch_size": 64, "bc_coef": 9.0}', "--bc_coef", "0.5",
                                         "--penalty_type", "none", "--scale_q", "0"])
    cfg = tm.build_config(args, 17, 6, 1.0)
    assert cfg["batch_size"] == 64                      # --params overrides the yaml
    assert cfg["bc_coef"] == 0.5                        # CLI-derived keys override --params
    assert cfg["shift_level"] == 2.0 and cfg["state_dim"] == 17 and cfg["action_dim"] == 6
    assert cfg["penalty_type"] == "none" and cfg["scale_Q"] == 0 and cfg["weight"] == 2.5 and cfg["tau"] == 0.005
    with pytest.raises(NotImplementedError):
        tm.domain_of("reacher-x")


@pytest.mark.gpu
def test_cli_runs_end_to_end_on_synthetic_buffers(tmp_path, capsys):
    import torch
    from mobody_amd import train_mobody as tm
    pol = tm.main(["--policy", "MOBODY", "--env", "walker2d_friction", "--shift_level", "2.0", "--mode", "3", "--seed", "1",
                   "--synthetic", "1", "--rng", "device", "--penalty_type", "none", "--src_rows", "20000", "--tar_rows", "2000",
                   "--src_rollout_batch_size", "4000", "--trg_rollout_batch_size", "1000", "--max_step", "12",
                   "--params", '{"batch_size": 256, "max_step": 12, "eval_freq": 10}', "--log_every", "6",
                   "--dir", str(tmp_path), "--save-model", "--eval_freq", "10"])
    out = capsys.readouterr().out
    assert pol.total_it == 12 and "step 12:" in out and "grad-steps/s" in out
    q, pi, bc = pol.losses()
    assert all(map(lambda v: v == v and abs(v) < 1e6, (q, pi, bc)))
    assert pol.fake_replay_buffer.size > 0                       # the step-1 refresh filled the fake buffer
    models = os.path.join(str(tmp_path), "MOBODY", "walker2d-friction-srcdatatype-medium-tardatatype-medium-2.0", "r1", "models")
    assert sorted(os.listdir(models)) == ["model_actor", "model_actor_optimizer", "model_critic", "model_critic_optimizer"]
    sd = torch.load(os.path.join(models, "model_actor"), weights_only=True)
    assert sorted(sd) == sorted(f"network.network.{i}.{w}" for i in (0, 2, 4) for w in ("weight", "bias"))
    # the writer stream (SummaryWriter surface): the model-error pair on the eval cadence (step 10), finite values
    rows = [l.strip().split(",") for l in open(os.path.join(os.path.dirname(models), "tb", "scalars.csv"))][1:]
    tags = {(r[0], int(r[1])) for r in rows}
    assert ("test/model error next_obs", 10) in tags and ("test/model error reward", 10) in tags
    assert all(float(r[2]) == float(r[2]) for r in rows)


@pytest.mark.gpu
def test_cli_pretrains_saves_and_reloads_the_dynamics(tmp_path, capsys):
    """--train_dynamics 1: MOBODYEnsembleDynamics.train on the buffers, saved in the reference's directory scheme
    (train_mobody.py:817-877); a second run with --train_dynamics 0 finds and loads it (:848-851)."""
    import torch
    from mobody_amd import train_mobody as tm
    common = ["--policy", "MOBODY", "--env", "walker2d_friction", "--shift_level", "2.0", "--mode", "3", "--seed", "2",
              "--synthetic", "1", "--rng", "device", "--penalty_type", "none", "--src_rows", "3000", "--tar_rows", "900",
              "--params", '{"batch_size": 128, "max_step": 3}', "--max_step", "3", "--dir", str(tmp_path),
              "--dynamics_path", str(tmp_path / "dyn")]
    os.makedirs(tmp_path / "dyn" / "walker2d-friction", exist_ok=True)
    pol = tm.main(common + ["--train_dynamics", "1", "--dynamics_max_epochs", "2"])
    d = pol.dynamics
    assert d.total_steps == 2 * (10 + 3 * 3) and len(d.history) == 2           # 2400 / 256 = 10 source, 720 / 256 = 3 target batches
    assert d.history[1]["trg_holdout"] < d.history[0]["trg_holdout"]           # it learns
    save = tmp_path / "dyn" / "walker2d-friction" / "srcdatatype-medium-tardatatype-medium-2.0"
    assert sorted(os.listdir(save)) == ["dynamics.pth", "mu.npy", "std.npy"]
    sd = torch.load(save / "dynamics.pth", weights_only=True)
    assert "zs1.saved_weight" in sd and "za_de_trg2.bias" in sd and sd["elites"].shape == (5,)
    pol2 = tm.main(common + ["--train_dynamics", "0"])
    assert "pretrained dynamics loaded" in capsys.readouterr().out
    sd2 = pol2.dynamics.model.state_dict()
    assert all(torch.equal(sd[k].to(sd2[k].device), sd2[k]) for k in sd)


def test_merged_config_matches_the_reference_mapping():
    """C1 pin: the reference's `config.update({...})` literal (train_mobody.py:470-531) as data -- every
    key and the expression it is bound to (fixture g10b, extracted with ast) -- evaluated on the parsed args must equal
    what build_config produces, key for key; the yaml layer equals the reference's shipped yaml files."""
    from mobody_amd import train_mobody as tm
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "g10b_config_merge.json")))
    argv = ["--policy", "MOBODY", "--env", "walker2d-friction", "--shift_level", "2.0", "--bc_coef", "0.25", "--scale_q", "0",
            "--rollout_from_src", "1", "--env_filter", "3.5", "--trg_ratio", "0.5", "--penalty_type", "dara",
            "--src_rollout_length", "4", "--params", '{"batch_size": 64}']
    args = tm.build_parser().parse_args(argv)
    cfg = tm.build_config(args, 17, 6, 1.0)
    scope = dict(args=args, state_dim=17, action_dim=6, max_action=1.0, shift_level=2.0, int=int)
    want = {k: eval(expr, {"__builtins__": {}}, scope) for k, expr in g["update"]}      # plain attribute / int() expressions
    for k, v in want.items():
        assert cfg[k] == v and type(cfg[k]) is type(v), (k, cfg[k], v)
    y = g["yaml"]["mujoco/mobody/walker2d.yaml"]
    extra = set(cfg) - set(want) - set(y)
    assert extra == {"rng", "seed"}, extra                     # this build's two additions, nothing else
    for k, v in y.items():
        if k not in want and k != "batch_size":
            assert cfg[k] == v, (k, cfg[k], v)
    assert cfg["batch_size"] == 64
    for name in ("ant", "halfcheetah", "hopper"):             # the built-in yaml copy covers all four shipped files
        yy = g["yaml"][f"mujoco/mobody/{name}.yaml"]
        assert {k: v for k, v in yy.items() if k != "eval_freq"} == {k: v for k, v in y.items() if k != "eval_freq"}


@pytest.mark.gpu
def test_cli_real_dataset_mode_without_simulators(tmp_path, capsys):
    """--synthetic 0: source transitions from an .npz in d4rl.qlearning_dataset's layout, target transitions from the raw
    arrays of an ODRL file through dataset/call_dataset.py's shift-by-one transformation (train_mobody.py:548-557)."""
    import numpy as np
    from mobody_amd import train_mobody as tm
    rng = np.random.default_rng(5)
    S, A, n, m = 17, 6, 3000, 801
    mu = np.zeros(S, np.float32); mu[0] = 1.25
    obs = (mu + 0.1 * rng.standard_normal((n, S))).astype(np.float32)
    np.savez(tmp_path / "src.npz", observations=obs, actions=rng.uniform(-1, 1, (n, A)).astype(np.float32),
             next_observations=(obs + 0.01 * rng.standard_normal((n, S))).astype(np.float32),
             rewards=rng.standard_normal(n).astype(np.float32), terminals=np.zeros(n, bool))
    tobs = (mu + 0.1 * rng.standard_normal((m, S))).astype(np.float32)
    np.savez(tmp_path / "tar.npz", observations=tobs, actions=rng.uniform(-1, 1, (m, A)).astype(np.float32),
             rewards=rng.standard_normal((m, 1)).astype(np.float32), terminals=np.zeros(m, bool), timeouts=np.zeros(m, bool))
    pol = tm.main(["--policy", "MOBODY", "--env", "walker2d-friction", "--shift_level", "2.0", "--mode", "3", "--seed", "2",
                   "--synthetic", "0", "--src_data", str(tmp_path / "src.npz"), "--tar_data", str(tmp_path / "tar.npz"),
                   "--penalty_type", "none", "--src_rollout_batch_size", "500", "--trg_rollout_batch_size", "200",
                   "--max_step", "6", "--eval_freq", "5",
                   "--params", '{"batch_size": 64, "max_step": 6, "eval_freq": 5}', "--log_every", "3", "--dir", str(tmp_path)])
    out = capsys.readouterr().out
    assert "datasets: 3000 source / 800 target transitions" in out        # N-1 target rows (shift by one)
    assert pol.total_it == 6 and all(v == v for v in pol.losses())
    with pytest.raises(ValueError):
        tm.main(["--policy", "MOBODY", "--env", "ant-friction", "--mode", "3", "--synthetic", "0",
                 "--src_data", str(tmp_path / "src.npz"), "--tar_data", str(tmp_path / "tar.npz")])


def test_scalar_log_and_dataset_loading_on_the_host(tmp_path):
    """The CLI's CSV writer (SummaryWriter.add_scalar surface) and the --synthetic 0 dataset loader, without a GPU."""
    import argparse
    import numpy as np
    import torch
    from mobody_amd import train_mobody as tm
    w = tm.ScalarLog(str(tmp_path / "tb" / "scalars.csv"))
    w.add_scalar("train/q1", torch.tensor(1.5), 5000)
    w.add_scalar("test/model error reward", 0.25, global_step=10)
    w.close()
    rows = [l.strip().split(",") for l in open(w.path)]
    assert rows == [["tag", "step", "value"], ["train/q1", "5000", "1.5"], ["test/model error reward", "10", "0.25"]]
    rng = np.random.default_rng(0)
    n, m, S, A = 20, 11, 17, 6
    np.savez(tmp_path / "src.npz", observations=rng.standard_normal((n, S)), actions=rng.standard_normal((n, A)),
             next_observations=rng.standard_normal((n, S)), rewards=rng.standard_normal(n), terminals=np.zeros(n, bool))
    tobs = rng.standard_normal((m, S)).astype(np.float32)
    np.savez(tmp_path / "tar.npz", observations=tobs, actions=rng.standard_normal((m, A)), rewards=rng.standard_normal((m, 1)),
             terminals=np.zeros(m, bool), timeouts=np.zeros(m, bool))
    args = argparse.Namespace(src_data=str(tmp_path / "src.npz"), tar_data=str(tmp_path / "tar.npz"), env="walker2d-friction",
                              shift_level=2.0, tartype="medium")
    src, tar = tm.load_datasets(args)
    assert src["observations"].shape == (n, S) and tar["observations"].shape == (m - 1, S)
    assert (tar["next_observations"] == tobs[1:]).all() and tar["rewards"].shape == (m - 1,)       # shift by one, [N,1] flattened
    args.src_data = None
    with pytest.raises(NotImplementedError):
        tm.load_datasets(args)


class _FakeDyn:
    """Records what build_dynamics asks of the dynamics object (no GPU)."""

    def __init__(self, load_fails=False):
        self.calls, self.load_fails, self.optim = [], load_fails, None

    def load(self, path):
        self.calls.append(("load", path))
        if self.load_fails:
            raise RuntimeError("corrupt checkpoint")

    def train(self, src, trg, writer=None, buffer=None, max_epochs=None):
        self.calls.append(("train", writer, max_epochs))

    def save(self, path):
        assert os.path.isdir(path)
        self.calls.append(("save", path))


class _FakeRb:
    def sample_all(self):
        return ()


def _bd(tmp_path, monkeypatch, argv, explicit, dyn=None, make=None):
    """Run build_dynamics in tmp_path (the default tree `pretrained_dynamics/` is relative to the working directory)."""
    from mobody_amd import synthetic, train_mobody as tm
    monkeypatch.chdir(tmp_path)
    shifted = []
    monkeypatch.setattr(synthetic, "alive_dynamics", lambda model, task: shifted.append(task))
    args = tm.build_parser().parse_args(["--policy", "MOBODY", "--env", "walker2d-friction", "--shift_level", "2.0"] + argv)
    if make:
        os.makedirs(make)
    dyn = dyn or _FakeDyn()
    res = tm.build_dynamics(args, dyn, object(), _FakeRb(), _FakeRb(), "W", "walker2d-medium-v2", explicit_synthetic=explicit)
    return res, dyn.calls, shifted


LEAF = "srcdatatype-medium-tardatatype-medium-2.0"


def test_build_dynamics_default_flags_train_and_save_under_the_default_tree(tmp_path, monkeypatch):
    """train_mobody.py:842-877 with --dynamics_path None, --train_dynamics 0 and nothing on disk: train, then save under
    pretrained_dynamics/<env>/srcdatatype-...; the writer is handed to dynamics.train; no 'alive' shift on real data."""
    res, calls, shifted = _bd(tmp_path, monkeypatch, [], explicit=False)
    want = os.path.join("pretrained_dynamics", "walker2d-friction", LEAF)
    assert res == "trained" and calls == [("train", "W", None), ("save", want)] and shifted == []


def test_build_dynamics_loads_the_default_tree_when_present(tmp_path, monkeypatch):
    want = os.path.join("pretrained_dynamics", "walker2d-friction", LEAF)
    res, calls, _ = _bd(tmp_path, monkeypatch, [], explicit=False, make=os.path.join(str(tmp_path), want))
    assert res == "loaded" and calls == [("load", want)]
    # a failing load falls through to training and saving (the reference's try / except, :848-864)
    res, calls, _ = _bd(tmp_path, monkeypatch, [], explicit=False, dyn=_FakeDyn(load_fails=True))
    assert res == "trained" and [c[0] for c in calls] == ["load", "train", "save"]
    # --train_dynamics 1 ignores what is on disk (:847) and saves under --dynamics_path when one is given (:867-872)
    res, calls, _ = _bd(tmp_path, monkeypatch, ["--train_dynamics", "1", "--dynamics_path", str(tmp_path / "dp")], explicit=False)
    assert res == "trained" and calls == [("train", "W", None), ("save", os.path.join(str(tmp_path / "dp"), "walker2d-friction", LEAF))]


def test_build_dynamics_with_a_dynamics_path(tmp_path, monkeypatch):
    """:819-840: --dynamics_path P --train_dynamics 0 loads P/<env>/srcdatatype-... when it exists, else trains and saves there."""
    p = os.path.join(str(tmp_path / "dp"), "walker2d-friction", LEAF)
    res, calls, _ = _bd(tmp_path, monkeypatch, ["--dynamics_path", str(tmp_path / "dp"), "--dynamics_max_epochs", "3"], explicit=False)
    assert res == "trained" and calls == [("train", "W", 3), ("save", p)]
    res, calls, _ = _bd(tmp_path, monkeypatch, ["--dynamics_path", str(tmp_path / "dp")], explicit=False)
    assert res == "loaded" and calls == [("load", p)]


def test_build_dynamics_random_model_only_under_explicit_synthetic(tmp_path, monkeypatch):
    res, calls, shifted = _bd(tmp_path, monkeypatch, [], explicit=True)
    assert res == "random" and calls == [] and shifted == ["walker2d-medium-v2"]
    res, calls, shifted = _bd(tmp_path, monkeypatch, ["--train_dynamics", "1"], explicit=True)
    assert res == "trained" and [c[0] for c in calls] == ["train", "save"] and shifted == ["walker2d-medium-v2"]


@pytest.mark.gpu
def test_cli_default_flags_pretrain_into_the_default_tree(tmp_path, monkeypatch, capsys):
    """The reference's default path end to end on dataset files: no --dynamics_path, --train_dynamics 0, nothing on disk ->
    the ensemble is pre-trained on the two buffers and saved under pretrained_dynamics/<env>/...; a second run loads it."""
    import numpy as np
    import torch
    from mobody_amd import train_mobody as tm
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(7)
    S, A, n, m = 17, 6, 2000, 601
    mu = np.zeros(S, np.float32); mu[0] = 1.25
    obs = (mu + 0.1 * rng.standard_normal((n, S))).astype(np.float32)
    np.savez(tmp_path / "src.npz", observations=obs, actions=rng.uniform(-1, 1, (n, A)).astype(np.float32),
             next_observations=(obs + 0.01 * rng.standard_normal((n, S))).astype(np.float32),
             rewards=rng.standard_normal(n).astype(np.float32), terminals=np.zeros(n, bool))
    tobs = (mu + 0.1 * rng.standard_normal((m, S))).astype(np.float32)
    np.savez(tmp_path / "tar.npz", observations=tobs, actions=rng.uniform(-1, 1, (m, A)).astype(np.float32),
             rewards=rng.standard_normal((m, 1)).astype(np.float32), terminals=np.zeros(m, bool), timeouts=np.zeros(m, bool))
    argv = ["--policy", "MOBODY", "--env", "walker2d-friction", "--shift_level", "2.0", "--mode", "3", "--seed", "3",
            "--synthetic", "0", "--src_data", str(tmp_path / "src.npz"), "--tar_data", str(tmp_path / "tar.npz"),
            "--penalty_type", "none", "--src_rollout_batch_size", "300", "--trg_rollout_batch_size", "100", "--max_step", "2",
            "--params", '{"batch_size": 64, "max_step": 2}', "--dir", str(tmp_path / "logs"), "--dynamics_max_epochs", "1"]
    pol = tm.main(argv)
    out = capsys.readouterr().out
    save = tmp_path / "pretrained_dynamics" / "walker2d-friction" / LEAF
    assert "dynamics trained and saved" in out and sorted(os.listdir(save)) == ["dynamics.pth", "mu.npy", "std.npy"]
    assert pol.dynamics.total_steps > 0 and pol.total_it == 2
    rows = [l.split(",")[0] for l in open(tmp_path / "logs" / "MOBODY" / f"walker2d-friction-{LEAF}" / "r3" / "tb" / "scalars.csv")]
    assert "trg_loss/dynamics_holdout_loss" in rows                       # dynamics.train received the writer (:829)
    sd = torch.load(save / "dynamics.pth", weights_only=True)
    pol2 = tm.main(argv)
    assert "pretrained dynamics loaded" in capsys.readouterr().out
    sd2 = pol2.dynamics.model.state_dict()
    assert all(torch.equal(sd[k].to(sd2[k].device), sd2[k]) for k in sd)
