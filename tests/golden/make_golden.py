"""Generate tests/golden/*.npz by running the REAL reference (read-only at /root/reference).

Runs only in the build container (the reference never travels to the GPU box);
the .npz files it writes are plain numbers: explicit small inputs (noise, indices,
batches) and the reference's outputs.  Large inputs (network weights) are NOT
stored: tests regenerate them from `gen_inputs.py` and verify the stored checksum.

Harness-side shims (no reference file is modified):
  * `algo.mb_utils.logger` is pre-seeded with a stub (its `tensorboard` import is the
    only thing that blocks `algo.dynamics.mobody_dynamics`).
  * `torch.normal` / `np.random.choice` are wrapped while `step()` runs so the unit
    noise eps[E,B,S] and the elite ids [B] are explicit fixture inputs
    (`torch.normal(0,std)` is replaced by `eps*std`).

Usage:  python tests/golden/make_golden.py        (writes next to this file)
"""
import os
import sys
import warnings
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
_stub = types.ModuleType("algo.mb_utils.logger")
_stub.Logger = object
sys.modules["algo.mb_utils.logger"] = _stub

import gen_inputs as gi  # noqa: E402
from algo.dynamics.mobody_module import MOBODYModule, EnsembleLinear  # noqa: E402
from algo.dynamics.mobody_dynamics import MOBODYEnsembleDynamics  # noqa: E402
from algo.offline_offline.mobody import MOBODY  # noqa: E402
from algo.mb_utils.terminal_funs import get_termination_fn  # noqa: E402
from algo import utils as ref_utils  # noqa: E402

torch.set_num_threads(1)
DYN_CFG = dict(mopo=0, latent_reward=0, encoder_loss_coef=1, domain_loss_coef=0.0, cycle_loss_coef=0.3)


def sub(x):
    """Fixture thinning for big tensors (tests apply the same rule to their side)."""
    x = np.asarray(x)
    f = x.reshape(-1)
    return f[::13].copy() if f.size > 4096 else f.copy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


class RngTap:
    """Explicit noise / elite ids for MOBODYEnsembleDynamics.step (mobody_dynamics.py:220,224)."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.eps, self.idx = [], []

    def __enter__(self):
        self._n, self._c = torch.normal, np.random.choice

        def normal(mean=None, std=None, **kw):
            e = torch.from_numpy(self.rng.standard_normal(tuple(std.shape)).astype(np.float32))
            self.eps.append(e.numpy().copy())
            return mean + e * std

        def choice(a, size=None, **kw):
            i = self._c(a, size=size, **kw)
            self.idx.append(np.asarray(i).copy())
            return i

        torch.normal, np.random.choice = normal, choice
        return self

    def __exit__(self, *a):
        torch.normal, np.random.choice = self._n, self._c


def load_dyn(S, A, seed, alive_dim, alive_val):
    p = gi.dyn_params(seed, S, A)
    p["transition3.bias"][:, 0, alive_dim] += np.float32(alive_val)
    m = MOBODYModule(S, A, 256, 7, 5, device="cpu", config=dict(DYN_CFG))
    sd = m.state_dict()
    for k, v in p.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(v)
    m.load_state_dict(sd)
    return m, p


# --------------------------------------------------------------------------- G1
def g1():
    rng = np.random.default_rng(11)
    lin = EnsembleLinear(17, 32, 7)
    W = gi._w(rng, (7, 17, 32), 0.2); b = gi._w(rng, (7, 1, 32), 0.1)
    lin.weight.data.copy_(torch.from_numpy(W)); lin.bias.data.copy_(torch.from_numpy(b))
    x2 = rng.standard_normal((8, 17)).astype(np.float32)
    x3 = rng.standard_normal((7, 8, 17)).astype(np.float32)
    with torch.no_grad():
        y2 = lin(torch.from_numpy(x2)).numpy(); y3 = lin(torch.from_numpy(x3)).numpy()
    save("g1_ensemble_linear", W=W, b=b, x2=x2, x3=x3, y2=y2, y3=y3)


# ----------------------------------------------------------------------- G2-G4
SHAPES = [  # (tag, S, A, B, task, alive_dim, alive_val, seed)
    ("walker", 17, 6, 48, "walker2d-medium-v2", 0, 0.85, 101),
    ("ant", 111, 8, 16, "ant-medium-v2", 0, 0.25, 102),
    ("pen", 45, 24, 16, "pen-human-v1", 26, 0.075, 103),
]


def g234():
    for tag, S, A, B, task, ad, av, seed in SHAPES:
        m, p = load_dyn(S, A, seed, ad, av)
        rng = np.random.default_rng(seed + 1000)
        obs = gi.walker_like_obs(rng, B, S); act = rng.uniform(-1, 1, (B, A)).astype(np.float32)
        nxt = gi.walker_like_obs(rng, B, S)
        m.inference()
        with torch.no_grad():
            mt, zmu, zlv = m.forward_trg(torch.from_numpy(obs), torch.from_numpy(act))
            ms, _, _ = m.forward_src(torch.from_numpy(obs), torch.from_numpy(act))
            rmu, rlv = m.encode_reward(torch.from_numpy(obs), torch.from_numpy(act), torch.from_numpy(nxt))
        out = dict(S=S, A=A, seed=seed, alive_dim=ad, alive_val=av, task=task, wsum=gi.checksum(p), obs=obs, act=act,
                   nxt=nxt, mean_trg=mt.numpy(), mean_src=ms.numpy(), zs_mu=zmu.numpy(), zs_logvar=zlv.numpy(),
                   r_mu=rmu.numpy(), r_logvar=rlv.numpy())
        dyn = MOBODYEnsembleDynamics(dict(DYN_CFG), m, None, None, get_termination_fn(task), penalty_coef=0.1)
        for up in (True, False):
            for ut in (True, False):
                np.random.seed(seed)
                with RngTap(seed + 7) as tap:
                    no, rw, term, info = dyn.step(torch.from_numpy(obs), torch.from_numpy(act), up, ut)
                k = f"step_p{int(up)}_t{int(ut)}_"
                out.update({k + "eps": tap.eps[0], k + "idx": tap.idx[0], k + "next_obs": no.numpy(),
                            k + "reward": rw.numpy(), k + "terminal": term, k + "penalty": info["penalty"].numpy(),
                            k + "raw_reward": info["raw_reward"].numpy(), k + "samples": info["samples"].numpy()})
        print(tag, "terminated rows:", int(out["step_p1_t1_terminal"].sum()), "/", B)
        save(f"g234_dynamics_{tag}", **out)


# --------------------------------------------------------------------------- G5
def g5():
    rng = np.random.default_rng(5)
    out = {}
    tasks = ["halfcheetah-medium-v2", "hopper-medium-v2", "walker2d-medium-v2", "ant-medium-v2",
             "halfcheetahvel-x", "antangle-x", "humanoid-x", "pen-human-v1", "door-human-v1", "pendulum-x"]
    special = np.array([0.8, 2.0, 1.0, -1.0, 100.0, -100.0, 0.7, 0.2, -0.2, 0.075, 99.99, -99.99, 0.0, 1.5, 1.25,
                        np.nan, np.inf, -np.inf, 0.8000001, 1.9999999, 0.2000001, 0.6999999], np.float32)
    for t in tasks:
        S = 45 if "pen" in t or "door" in t else 17
        n = gi.walker_like_obs(rng, 96, S)
        # sprinkle boundary values on the coordinates the predicates look at
        cols = [0, 1, 2, 26] if S == 45 else [0, 1, 2, 16]
        for r in range(96):
            if r % 3:
                n[r, cols[rng.integers(len(cols))]] = special[rng.integers(len(special))]
            if r % 7 == 0:
                n[r, 0] = special[rng.integers(len(special))]
        o = gi.walker_like_obs(rng, 96, S); a = rng.uniform(-1, 1, (96, 6)).astype(np.float32)
        d = get_termination_fn(t)(o, a, n)
        out[t + "::next_obs"] = n
        out[t + "::done"] = np.asarray(d).astype(bool).reshape(96, 1)
    try:
        get_termination_fn("reacher-x")
        out["unknown_raises"] = np.array(0)
    except TypeError:
        out["unknown_raises"] = np.array(1)
    save("g5_termination", **out)


# ------------------------------------------------------------------- policy cfg
def policy_cfg(S, A, **over):
    cfg = dict(gamma=0.99, tau=0.005, update_interval=2, state_dim=S, action_dim=A, penalty_type="none",
               hidden_sizes=256, max_action=1.0, critic_lr=3e-4, actor_lr=3e-4, gaussian_noise_std=1.0,
               penalize_fake=0, src_ratio=1, trg_ratio=1, src_rollout_length=1, trg_rollout_length=1,
               use_src_sa_to_get_target_next_state=1, env_filter=10.0, rollout_from_src=0, fake_batch_scale=0.5,
               advantage=0, scale_Q=1, weight=2.5, bc_coef=1.0, q_weighted=1, filter_bad_rollout=1,
               penalty_coef=0.1, **DYN_CFG)
    cfg.update(over)
    return cfg


def load_mlp(mod, p):
    sd = mod.state_dict()
    for k, v in p.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(v)
    mod.load_state_dict(sd)


def make_policy(cfg, seed):
    S, A = cfg["state_dim"], cfg["action_dim"]
    pol = MOBODY(cfg, torch.device("cpu"))
    pa = {"network." + k: v for k, v in gi.mlp_params(seed, S, A).items()}
    pq = {}
    for j, sd_ in ((1, seed + 1), (2, seed + 2)):
        pq.update({f"network{j}." + k: v for k, v in gi.mlp_params(sd_, S + A, 1).items()})
    pv = {"network." + k: v for k, v in gi.mlp_params(seed + 3, S, 1).items()}
    load_mlp(pol.policy, pa); load_mlp(pol.q_funcs, pq); load_mlp(pol.target_q_funcs, pq); load_mlp(pol.v_func, pv)
    return pol, pa, pq, pv


# --------------------------------------------------------------------------- G6
def g6():
    S, A, B = 17, 6, 40
    m, p = load_dyn(S, A, 201, 0, 0.85)
    for H, use_trg, tag in ((1, True, "h1"), (5, True, "h5"), (3, False, "h3_nopen")):
        cfg = policy_cfg(S, A)
        pol, pa, _, _ = make_policy(cfg, 301)
        dyn = MOBODYEnsembleDynamics(cfg, m, None, None, get_termination_fn("walker2d-medium-v2"), penalty_coef=0.1)
        pol.dynamics = dyn
        rng = np.random.default_rng(202)
        init = gi.walker_like_obs(rng, B, S)
        # pick env_filter at the median step-1 penalty so the filter really drops rows
        with RngTap(1) as tap, torch.no_grad():
            a0 = pol.select_action(torch.from_numpy(init), pol.policy, cuda=True).reshape(-1, A)
            _, _, _, info = dyn.step(torch.from_numpy(init), a0)
        cfg["env_filter"] = float(np.median(info["penalty"].numpy()))
        np.random.seed(77)
        with RngTap(203) as tap:
            tr, inf = pol.rollout(torch.from_numpy(init), H, use_trg)
        out = dict(S=S, A=A, H=H, use_trg=int(use_trg), dyn_seed=201, actor_seed=301, alive_val=0.85,
                   wsum_dyn=gi.checksum(p), wsum_actor=gi.checksum(pa), env_filter=cfg["env_filter"], init=init,
                   n_steps=len(tap.eps), num_transitions=inf["num_transitions"], reward_mean=inf["reward_mean"])
        for t, (e, i) in enumerate(zip(tap.eps, tap.idx)):
            out[f"eps{t}"] = e; out[f"idx{t}"] = i
        for k, v in tr.items():
            out["out_" + k] = v.numpy()
        print("rollout", tag, "steps", len(tap.eps), "rows", [e.shape[1] for e in tap.eps], "kept", len(tr["obss"]))
        save(f"g6_rollout_{tag}", **out)


# --------------------------------------------------------------------------- G7
class FixedRB:
    """Stands in for ReplayBuffer.sample (utils.py:127-148) with preset rows; counts calls."""

    def __init__(self, rows):
        self.rows = [torch.from_numpy(np.asarray(r)) for r in rows]
        self.size = len(self.rows[0]); self.calls = []

    def sample(self, n):
        self.calls.append(n)
        assert n <= self.size
        return tuple(r[:n].clone() for r in self.rows)

    def add_batch(self, b):
        pass


def g7():
    S, A, bs = 17, 6, 32
    variants = dict(default={}, noqw=dict(q_weighted=0), adv=dict(advantage=1), noscale=dict(scale_Q=0),
                    nofake=dict(fake_batch_scale=0), par=dict(penalty_type="par"), bc05=dict(bc_coef=0.05, trg_ratio=0.5))
    for tag, over in variants.items():
        cfg = policy_cfg(S, A, **over)
        pol, pa, pq, pv = make_policy(cfg, 401)
        src = FixedRB(gi.batch(501, 64, S, A)); tar = FixedRB(gi.batch(502, 64, S, A)); fake = FixedRB(gi.batch(503, 64, S, A))
        pol.fake_replay_buffer = fake
        out = dict(S=S, A=A, bs=bs, seed=401, wsum_actor=gi.checksum(pa), wsum_q=gi.checksum(pq), wsum_v=gi.checksum(pv),
                   cfg_keys=np.array(sorted(over)), cfg_vals=np.array([str(over[k]) for k in sorted(over)]))
        if tag == "par":
            m, p = load_dyn(S, A, 201, 0, 0.85)
            pol.dynamics = MOBODYEnsembleDynamics(cfg, m, None, None, get_termination_fn("walker2d-medium-v2"), penalty_coef=0.1)
            out["wsum_dyn"] = gi.checksum(p)
        rec = dict(q_loss=[], pi_loss=[], bc_loss=[])
        for nm, key in (("update_q_functions", "q_loss"), ("update_q_functions_1", "q_loss"), ("update_policy", "pi_loss"),
                        ("update_policy_1", "pi_loss"), ("bc_loss", "bc_loss")):
            def mk(orig, key):
                def f(*a, **k):
                    o = orig(*a, **k); rec[key].append(float(o.detach())); return o
                return f
            setattr(pol, nm, mk(getattr(pol, nm), key))
        pol.total_it = 1                       # -> 2,3: no rollout refresh, no DARA warm-up
        dummy = types.SimpleNamespace(log=lambda *a, **k: None)
        for step in (1, 2):
            np.random.seed(9)
            with RngTap(600 + step) as tap:
                pol.train(src, tar, bs, None, dummy)
            if tag == "par":
                out[f"par_eps{step}"] = tap.eps[0]; out[f"par_idx{step}"] = tap.idx[0]
            for nm, mod in (("q", pol.q_funcs), ("actor", pol.policy), ("qt", pol.target_q_funcs), ("v", pol.v_func)):
                for k, v in mod.state_dict().items():
                    out[f"s{step}_{nm}_p::{k}"] = sub(v.numpy())
            for nm, mod in (("q", pol.q_funcs), ("actor", pol.policy), ("v", pol.v_func)):
                for k, v in mod.named_parameters():
                    if v.grad is not None:
                        g = v.grad.numpy()
                        out[f"s{step}_{nm}_g::{k}"] = sub(g)
                        out[f"s{step}_{nm}_gsum::{k}"] = np.array([g.astype(np.float64).sum(), (g.astype(np.float64) ** 2).sum()])
        out["q_loss"] = np.array(rec["q_loss"]); out["pi_loss"] = np.array(rec["pi_loss"]); out["bc_loss"] = np.array(rec["bc_loss"])
        out["calls_src"] = np.array(src.calls); out["calls_tar"] = np.array(tar.calls); out["calls_fake"] = np.array(fake.calls)
        print("train", tag, "q_loss", rec["q_loss"], "pi_loss", rec["pi_loss"], "calls", src.calls, tar.calls, fake.calls)
        save(f"g7_train_{tag}", **out)


# --------------------------------------------------------------------------- G8
def g8():
    S, A, cap = 5, 2, 50
    out = {}
    rb = ref_utils.ReplayBuffer(S, A, "cpu", max_size=cap)
    rng = np.random.default_rng(8)
    log = []
    for ci, M in enumerate([20, 20, 20, 10, 50, 7, 43, 49, 3]):
        b = dict(obss=torch.from_numpy(rng.standard_normal((M, S)).astype(np.float32)),
                 next_obss=torch.from_numpy(rng.standard_normal((M, S)).astype(np.float32)),
                 actions=torch.from_numpy(rng.standard_normal((M, A)).astype(np.float32)),
                 rewards=torch.from_numpy(rng.standard_normal((M, 1)).astype(np.float32)),
                 terminals=torch.from_numpy((rng.uniform(size=(M, 1)) > 0.7).astype(np.float32)))
        for k, v in b.items():
            out[f"add{ci}_{k}"] = v.numpy()
        rb.add_batch(b)
        log.append((M, rb.ptr, rb.size))
        out[f"after{ci}_state"] = rb.state.numpy().copy(); out[f"after{ci}_action"] = rb.action.numpy().copy()
        out[f"after{ci}_next_state"] = rb.next_state.numpy().copy(); out[f"after{ci}_reward"] = rb.reward.numpy().copy()
        out[f"after{ci}_not_done"] = rb.not_done.numpy().copy()
    rb.add_batch(None)
    out["log"] = np.array(log)
    np.random.seed(123)
    smp = rb.sample(16)
    np.random.seed(123)
    out["sample_ind"] = np.random.randint(0, rb.size, size=16)
    for k, v in zip(("state", "action", "next_state", "reward", "not_done"), smp):
        out["sample_" + k] = v.numpy()
    # the index stream the driver sees for seed 0 (train_mobody.py:442 + mobody.py:399-400)
    np.random.seed(0)
    out["stream_seed0"] = np.concatenate([np.random.randint(0, 1000000, size=8), np.random.randint(0, 5000, size=8)])
    # convert_D4RL (utils.py:173-193)
    ds = dict(observations=rng.standard_normal((30, S)).astype(np.float32), actions=rng.standard_normal((30, A)).astype(np.float32),
              next_observations=rng.standard_normal((30, S)).astype(np.float32), rewards=rng.standard_normal(30).astype(np.float32),
              terminals=(rng.uniform(size=30) > 0.8))
    rb2 = ref_utils.ReplayBuffer(S, A, "cpu", max_size=cap)
    rb2.convert_D4RL(ds)
    for k, v in ds.items():
        out["d4rl_" + k] = v
    out["d4rl_size"] = np.array(rb2.size); out["d4rl_not_done"] = rb2.not_done.numpy(); out["d4rl_reward"] = rb2.reward.numpy()
    save("g8_replay", **out)


# --------------------------------------------------------------------------- G9
def g9():
    S, A, bs = 17, 6, 24
    cfg = policy_cfg(S, A, penalty_type="dara")
    pol, _, _, _ = make_policy(cfg, 401)
    pc = {}
    pc.update({"sa_classifier." + k: v for k, v in gi.mlp_params(701, S + A, 2).items()})
    pc.update({"sas_classifier." + k: v for k, v in gi.mlp_params(702, 2 * S + A, 2).items()})
    load_mlp(pol.classifier, pc)
    s, a, s2, _, _ = gi.batch(703, 64, S, A)
    with torch.no_grad():
        ps, pa_ = pol.classifier(torch.from_numpy(s), torch.from_numpy(a), torch.from_numpy(s2), with_noise=False)
        sp, ap = torch.softmax(ps, -1), torch.softmax(pa_, -1)
        ls, la = torch.log(sp + 1e-10), torch.log(ap + 1e-10)
        dr = (ls[:, 1:] - la[:, 1:] - ls[:, :1] + la[:, :1]).clamp(-10, 10)
    out = dict(S=S, A=A, seed_sa=701, seed_sas=702, wsum=gi.checksum(pc), s=s, a=a, s2=s2, probs_sas=ps.numpy(),
               probs_sa=pa_.numpy(), delta_r=dr.numpy())
    # one update_classifier step with recorded permutation / noise (mobody.py:146-181)
    src = FixedRB(gi.batch(704, 64, S, A)); tar = FixedRB(gi.batch(705, 64, S, A))
    taps = dict(perm=None, noise=[])
    o_perm, o_randn = torch.randperm, torch.randn_like

    def randperm(n, **k):
        p = o_perm(n, **k); taps["perm"] = p.numpy().copy(); return p

    def randn_like(x, **k):
        e = o_randn(x); taps["noise"].append(e.numpy().copy()); return e

    torch.randperm, torch.randn_like = randperm, randn_like
    try:
        torch.manual_seed(5)
        loss_sa, loss_sas = pol.update_classifier(src, tar, bs, None)
    finally:
        torch.randperm, torch.randn_like = o_perm, o_randn
    out.update(bs=bs, perm=taps["perm"], noise_sas=taps["noise"][0], noise_sa=taps["noise"][1], loss_sa=float(loss_sa),
               loss_sas=float(loss_sas))
    for k, v in pol.classifier.named_parameters():
        out["cls_g::" + k] = sub(v.grad.numpy()); out["cls_p::" + k] = sub(v.detach().numpy())
    save("g9_dara", **out)


def g9b():
    """update_classifier with penalize_fake=1 and a non-empty fake buffer (mobody.py:146-181): the reference draws
    src(bs), tar(bs), fake(bs), tar(2bs) and -- because its labels stay [0]*bs+[1]*bs and randperm runs over 2*bs
    entries -- trains on the src rows (label 0) and the FAKE rows (label 1) only."""
    S, A, bs = 17, 6, 24
    cfg = policy_cfg(S, A, penalty_type="dara", penalize_fake=1)
    pol, _, _, _ = make_policy(cfg, 401)
    pc = {}
    pc.update({"sa_classifier." + k: v for k, v in gi.mlp_params(701, S + A, 2).items()})
    pc.update({"sas_classifier." + k: v for k, v in gi.mlp_params(702, 2 * S + A, 2).items()})
    load_mlp(pol.classifier, pc)
    log = []

    class LogRB(FixedRB):
        def __init__(self, rows, name):
            super().__init__(rows); self.name = name

        def sample(self, n):
            log.append(f"{self.name}:{n}")
            return super().sample(n)

    src = LogRB(gi.batch(704, 64, S, A), "src"); tar = LogRB(gi.batch(705, 64, S, A), "tar")
    pol.fake_replay_buffer = LogRB(gi.batch(706, 64, S, A), "fake")
    taps = dict(perm=None, noise=[])
    o_perm, o_randn = torch.randperm, torch.randn_like

    def randperm(n, **k):
        q = o_perm(n, **k); taps["perm"] = q.numpy().copy(); return q

    def randn_like(x, **k):
        e = o_randn(x); taps["noise"].append(e.numpy().copy()); return e

    torch.randperm, torch.randn_like = randperm, randn_like
    try:
        torch.manual_seed(6)
        loss_sa, loss_sas = pol.update_classifier(src, tar, bs, None)
    finally:
        torch.randperm, torch.randn_like = o_perm, o_randn
    out = dict(S=S, A=A, bs=bs, seed_sa=701, seed_sas=702, wsum=gi.checksum(pc), perm=taps["perm"],
               noise_sas=taps["noise"][0], noise_sa=taps["noise"][1], loss_sa=float(loss_sa), loss_sas=float(loss_sas),
               draw_log=np.array(log))
    for k, v in pol.classifier.named_parameters():
        out["cls_g::" + k] = sub(v.grad.numpy()); out["cls_p::" + k] = sub(v.detach().numpy())
    print("penalize_fake draws", log, "perm len", len(taps["perm"]), "loss", float(loss_sa), float(loss_sas))
    save("g9_dara_penfake", **out)


# -------------------------------------------------------------------------- G11
class CudaAlias:
    """Harness-side alias 'cuda' -> cpu for the call sites the reference hard-codes (`.to('cuda')`, mobody.py:495-497,
    mobody_dynamics.py:105-107,610-613,1118-1121).  No reference file is modified."""

    def __enter__(self):
        self._to = torch.Tensor.to

        def to(t, *a, **k):
            a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
            if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
                k["device"] = "cpu"
            return self._to(t, *a, **k)

        torch.Tensor.to = to
        self._mto = torch.nn.Module.to

        def mto(mod, *a, **k):                       # Classifier(...).to('cuda') in data_augmentation (:688)
            a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
            return self._mto(mod, *a, **k)

        torch.nn.Module.to = mto
        return self

    def __exit__(self, *a):
        torch.Tensor.to = self._to
        torch.nn.Module.to = self._mto


REFRESH_MAP = {50000: 96, 2000: 40, 100: 12}     # hard-coded refresh sizes (mobody.py:442-443,484-485) -> fixture sizes


def shrink_samples(rb, name, log):
    """The reference hard-codes sample(50000)/sample(2000)/sample(100) in the refresh; the harness maps them to small
    counts (the mirror's test patches its module constants to the same numbers) and logs the order of the draws."""
    orig = rb.sample

    def sample(n):
        n = REFRESH_MAP.get(n, n)
        log.append((name, n))
        return orig(n)

    rb.sample = sample


def _ref_buffer(S, A, cap, rows):
    rb = ref_utils.ReplayBuffer(S, A, "cpu", max_size=cap)
    s, a, s2, r, nd = rows
    rb.add_batch(dict(obss=torch.from_numpy(s), next_obss=torch.from_numpy(s2), actions=torch.from_numpy(a),
                      rewards=torch.from_numpy(r), terminals=torch.from_numpy(1.0 - nd)))
    return rb


def g11():
    """First train() call (total_it 0 -> 1): the fake-buffer refresh of mobody.py:441-513 in the reference's order
    (src rollout -> add -> trg rollout -> add -> (s,a) relabel with strict '<' -> [rollout_from_src]) followed by
    the gradient step on src|tar|fake rows, through the reference's own ReplayBuffers (the fake ring wraps)."""
    S, A, bs = 17, 6, 32
    m, p = load_dyn(S, A, 201, 0, 0.85)
    for tag, over in (("default", {}), ("fromsrc", dict(rollout_from_src=1, rollout_from_src_length=2))):
        def run(env_filter, record):
            cfg = policy_cfg(S, A, src_rollout_length=2, trg_rollout_length=3, env_filter=env_filter, **over)
            pol, pa, pq, pv = make_policy(cfg, 401)
            pc = {}
            pc.update({"sa_classifier." + k: v for k, v in gi.mlp_params(701, S + A, 2).items()})
            pc.update({"sas_classifier." + k: v for k, v in gi.mlp_params(702, 2 * S + A, 2).items()})
            load_mlp(pol.classifier, pc)
            dyn = MOBODYEnsembleDynamics(cfg, m, None, None, get_termination_fn("walker2d-medium-v2"), penalty_coef=0.1)
            pol.dynamics = dyn
            log = []
            src = _ref_buffer(S, A, 300, gi.batch(801, 300, S, A)); shrink_samples(src, "src", log)
            tar = _ref_buffer(S, A, 120, gi.batch(802, 120, S, A)); shrink_samples(tar, "tar", log)
            pol.fake_replay_buffer = ref_utils.ReplayBuffer(S, A, "cpu", max_size=260)
            shrink_samples(pol.fake_replay_buffer, "fake", log)
            pens = []
            o_step = dyn.step

            def step(*a, **k):
                r = o_step(*a, **k); pens.append(r[3]["penalty"].numpy().copy()); return r

            dyn.step = step
            rec = dict(q_loss=[], pi_loss=[], bc_loss=[])
            for nm, key in (("update_q_functions", "q_loss"), ("update_policy", "pi_loss"), ("bc_loss", "bc_loss")):
                def mk(orig, key):
                    def f(*a, **k):
                        o = orig(*a, **k); rec[key].append(float(o.detach())); return o
                    return f
                setattr(pol, nm, mk(getattr(pol, nm), key))
            taps = dict(perm=[], noise=[])
            o_perm, o_randn = torch.randperm, torch.randn_like

            def randperm(n, **k):
                q = o_perm(n, **k); taps["perm"].append(q.numpy().copy()); return q

            def randn_like(x, **k):
                e = o_randn(x); taps["noise"].append(e.numpy().copy()); return e

            torch.randperm, torch.randn_like = randperm, randn_like
            dummy = types.SimpleNamespace(log=lambda *a, **k: None)
            try:
                np.random.seed(31); torch.manual_seed(31)
                with RngTap(900) as tap, CudaAlias():
                    pol.train(src, tar, bs, None, dummy)
            finally:
                torch.randperm, torch.randn_like = o_perm, o_randn
            return dict(pol=pol, tap=tap, pens=pens, log=log, rec=rec, taps=taps, cfg=cfg, pa=pa, pq=pq, pc=pc)

        dry = run(1e9, False)
        allp = np.unique(np.concatenate([x.ravel() for x in dry["pens"]]))
        mid = len(allp) // 2
        env_filter = float(0.5 * (float(allp[mid - 1]) + float(allp[mid])))      # between two penalties: well conditioned
        r = run(env_filter, True)
        pol, tap, fb = r["pol"], r["tap"], r["pol"].fake_replay_buffer
        out = dict(S=S, A=A, bs=bs, dyn_seed=201, alive_val=0.85, seed=401, wsum_dyn=gi.checksum(p),
                   wsum_actor=gi.checksum(r["pa"]), wsum_q=gi.checksum(r["pq"]), wsum_cls=gi.checksum(r["pc"]),
                   env_filter=env_filter, n_steps=len(tap.eps), fake_cap=260, np_seed=31,
                   refresh_src=REFRESH_MAP[50000], refresh_tar=REFRESH_MAP[2000], refresh_from_src_tar=REFRESH_MAP[100],
                   draw_log=np.array([f"{n}:{k}" for n, k in r["log"]]), fake_ptr=fb.ptr, fake_size=fb.size,
                   fake_state=fb.state.numpy(), fake_action=fb.action.numpy(), fake_next_state=fb.next_state.numpy(),
                   fake_reward=fb.reward.numpy(), fake_not_done=fb.not_done.numpy(),
                   q_loss=np.array(r["rec"]["q_loss"]), pi_loss=np.array(r["rec"]["pi_loss"]),
                   bc_loss=np.array(r["rec"]["bc_loss"]), total_it=pol.total_it,
                   cfg_keys=np.array(sorted(over)), cfg_vals=np.array([str(over[k]) for k in sorted(over)]))
        for t, (e, i) in enumerate(zip(tap.eps, tap.idx)):
            out[f"eps{t}"] = e; out[f"idx{t}"] = i; out[f"pen{t}"] = r["pens"][t]
        if r["taps"]["perm"]:
            out["cls_perm"] = r["taps"]["perm"][0]; out["cls_noise_sas"] = r["taps"]["noise"][0]; out["cls_noise_sa"] = r["taps"]["noise"][1]
            for k, v in pol.classifier.state_dict().items():
                out["cls_p::" + k] = sub(v.numpy())
        for nm, mod in (("q", pol.q_funcs), ("actor", pol.policy), ("qt", pol.target_q_funcs)):
            for k, v in mod.state_dict().items():
                out[f"s1_{nm}_p::{k}"] = sub(v.numpy())
        print("refresh", tag, "steps", len(tap.eps), "rows/step", [e.shape[1] for e in tap.eps], "draws", r["log"],
              "fake ptr/size", fb.ptr, fb.size, "env_filter", env_filter, "losses", r["rec"])
        save(f"g11_refresh_{tag}", **out)


# -------------------------------------------------------------------------- G12
def sub101(x):
    """Coarser thinning for the pre-training fixtures (the float64 sum / sum of squares of every gradient tensor is
    stored next to it, so the whole tensor is still pinned)."""
    f = np.asarray(x).reshape(-1)
    return f[::101].copy() if f.size > 65536 else f[::17].copy() if f.size > 256 else f.copy()


class NoiseTap:
    """Every torch.randn_like (reparameterisation noise mobody_module.py:239-240, fake-next-state noise
    mobody_dynamics.py:353) is replaced by the next draw of numpy default_rng(seed): tests regenerate the identical
    stream from the seed (gen_inputs.noise_stream), so no noise is stored."""

    def __init__(self, seed):
        self.rng, self.shapes = np.random.default_rng(seed), []

    def __enter__(self):
        self._r = torch.randn_like

        def randn_like(x, **k):
            self.shapes.append(tuple(x.shape))
            return torch.from_numpy(self.rng.standard_normal(tuple(x.shape)).astype(np.float32))

        torch.randn_like = randn_like
        return self

    def __exit__(self, *a):
        torch.randn_like = self._r


PRE_CFG = dict(DYN_CFG, no_vae=0, inverse_sep_reward_loss=0, train_together=0, train_with_src_threshold=1)


def make_dyn_trainer(S, A, seed, lr=1e-3, **cfg_over):
    m, p = load_dyn(S, A, seed, 0, 0.85)
    m.config = dict(PRE_CFG, **cfg_over)
    opt = torch.optim.Adam(m.parameters(), lr=lr)                 # train_mobody.py:801-804
    dyn = MOBODYEnsembleDynamics(dict(PRE_CFG, **cfg_over), m, opt, None, get_termination_fn("walker2d-medium-v2"), penalty_coef=0.1)
    dyn.total_steps = 0
    return dyn, m, p


def g12():
    """Dynamics pre-training, one optimizer step per `learn()` call (mobody_dynamics.py:594-653 with encoder_loss
    :300-330, transition_loss :337-347, reward_loss :349-384): the call sequence src, trg, src, trg pins the losses,
    every gradient, the per-parameter Adam step counts (za_src* only steps on source batches, za_trg* on target ones,
    decoders and saved_* never) and the post-step parameters."""
    # "walker_novae": the same walker run with config no_vae = 1 (:616-635: encoder_loss is neither evaluated nor added, its
    # three reported numbers are 0; only the transition / reward terms draw reparameterisation noise: 3 draws per step, not 7)
    for tag, S, A, b, seed, over in (("walker", 17, 6, 24, 211, {}), ("pen", 45, 24, 20, 213, {}), ("walker_novae", 17, 6, 24, 211, dict(no_vae=1))):
        dyn, m, p = make_dyn_trainer(S, A, seed, **over)
        out = dict(S=S, A=A, b=b, seed=seed, alive_val=0.85, wsum=gi.checksum(p), noise_seed=1200 + seed, lr=1e-3, no_vae=int(over.get("no_vae", 0)))
        with NoiseTap(1200 + seed) as tap, CudaAlias():
            for step, use_trg in enumerate((False, True, False, True)):
                rows = gi.pretrain_batch(3000 + 10 * seed + step, b, S, A)       # per-member rows [7,b,.]
                res = dyn.learn(use_trg, *[torch.from_numpy(x) for x in rows], b, 0.01)
                out[f"s{step}_losses"] = np.array(res, np.float64)               # loss, transition, encoder, recon, kl
                for k, v in m.named_parameters():
                    if v.grad is not None and not k.startswith(("max_", "min_", "elites")):
                        g = v.grad.numpy()
                        out[f"s{step}_g::{k}"] = sub101(g)
                        out[f"s{step}_gsum::{k}"] = np.array([g.astype(np.float64).sum(), (g.astype(np.float64) ** 2).sum()])
                    # za_de_*: decoder weights no loss reaches; MOBODYModule draws them from the unseeded global torch RNG, so
                    # they are not reproducible from this script and are left out (no test reads them)
                    if ".saved_" not in k and not k.startswith(("max_", "min_", "elites", "za_de_")):
                        out[f"s{step}_p::{k}"] = sub101(v.detach().numpy())
                out[f"s{step}_has_grad"] = np.array(sorted(k for k, v in m.named_parameters() if v.grad is not None))
        out["noise_shapes"] = np.array([",".join(map(str, sh)) for sh in tap.shapes])
        st = dyn.optim.state_dict()["state"]
        names = [k for k, _ in m.named_parameters()]
        out["adam_steps"] = np.array([f"{names[i]}={int(float(v['step']))}" for i, v in st.items()])
        print("pretrain", tag, [out[f"s{k}_losses"][0] for k in range(4)], "noise calls", len(tap.shapes))
        save(f"g12_pretrain_{tag}", **out)


def g12t():
    """config train_together = 1, one optimizer step per learn_src_trg() call (mobody_dynamics.py:521-590): a 24-row source
    batch and a 17-row target batch per step, two steps -- total loss, the target batch's transition / encoder / KL numbers,
    every gradient, post-step parameters, Adam step counts (both action encoders step)."""
    S, A, bs, bt, seed = 17, 6, 24, 17, 231
    dyn, m, p = make_dyn_trainer(S, A, seed, train_together=1)
    out = dict(S=S, A=A, bs=bs, bt=bt, seed=seed, alive_val=0.85, wsum=gi.checksum(p), noise_seed=1400 + seed, lr=1e-3)
    with NoiseTap(1400 + seed) as tap, CudaAlias():
        for step in range(2):
            src = gi.pretrain_batch(5000 + 10 * step, bs, S, A); trg = gi.pretrain_batch(5001 + 10 * step, bt, S, A)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")                                  # np.mean([]) of the never-filled recon list
                res = dyn.learn_src_trg(False, *[torch.from_numpy(x) for x in src], *[torch.from_numpy(x) for x in trg], bs, 0.01)
            out[f"s{step}_stats"] = np.array(res, np.float64)                    # total, trg transition, trg encoder, nan, trg kl
            for k, v in m.named_parameters():
                if v.grad is not None and not k.startswith(("max_", "min_", "elites")):
                    out[f"s{step}_g::{k}"] = sub101(v.grad.numpy())
                if ".saved_" not in k and not k.startswith(("max_", "min_", "elites", "za_de_")):
                    out[f"s{step}_p::{k}"] = sub101(v.detach().numpy())
            out[f"s{step}_has_grad"] = np.array(sorted(k for k, v in m.named_parameters() if v.grad is not None))
    out["noise_shapes"] = np.array([",".join(map(str, sh)) for sh in tap.shapes])
    st = dyn.optim.state_dict()["state"]
    names = [k for k, _ in m.named_parameters()]
    out["adam_steps"] = np.array([f"{names[i]}={int(float(v['step']))}" for i, v in st.items()])
    print("together", [out[f"s{k}_stats"] for k in range(2)], "noise calls", len(tap.shapes))
    save("g12_together_walker", **out)


def g12s():
    """config inverse_sep_reward_loss = 1: learn() without reward_loss (:637-641; the reward head has no gradient, Adam skips it)
    and learn_sep_reward (:482-519; reward losses of a source and a target batch only).  Sequence: learn(src), learn(trg),
    learn_sep_reward(src 24 rows | trg 17 rows), learn(trg) -- losses, every gradient, post-step parameters, per-parameter Adam
    step counts (the reward head ends at 1, everything else trained every step)."""
    S, A, bs, bt, seed = 17, 6, 24, 17, 241
    dyn, m, p = make_dyn_trainer(S, A, seed, inverse_sep_reward_loss=1)
    out = dict(S=S, A=A, bs=bs, bt=bt, seed=seed, alive_val=0.85, wsum=gi.checksum(p), noise_seed=1500 + seed, lr=1e-3)

    def snap(step):
        for k, v in m.named_parameters():
            if v.grad is not None and not k.startswith(("max_", "min_", "elites")):
                out[f"s{step}_g::{k}"] = sub101(v.grad.numpy())
            if ".saved_" not in k and not k.startswith(("max_", "min_", "elites", "za_de_")):
                out[f"s{step}_p::{k}"] = sub101(v.detach().numpy())
        out[f"s{step}_has_grad"] = np.array(sorted(k for k, v in m.named_parameters() if v.grad is not None))

    with NoiseTap(1500 + seed) as tap, CudaAlias():
        for step, kind in enumerate(("src", "trg", "sep", "trg")):
            for v in m.parameters():
                v.grad = None                                                    # (optim.zero_grad keeps zero tensors: make "no gradient" visible)
            if kind == "sep":
                src = gi.pretrain_batch(6000, bs, S, A); trg = gi.pretrain_batch(6001, bt, S, A)
                res = dyn.learn_sep_reward(*[torch.from_numpy(x) for x in src], *[torch.from_numpy(x) for x in trg], bs)
                out[f"s{step}_losses"] = np.array([res], np.float64)
            else:
                rows = gi.pretrain_batch(6100 + step, bs, S, A)
                res = dyn.learn(kind == "trg", *[torch.from_numpy(x) for x in rows], bs, 0.01)
                out[f"s{step}_losses"] = np.array(res, np.float64)
            snap(step)
    out["noise_shapes"] = np.array([",".join(map(str, sh)) for sh in tap.shapes])
    st = dyn.optim.state_dict()["state"]
    names = [k for k, _ in m.named_parameters()]
    out["adam_steps"] = np.array([f"{names[i]}={int(float(v['step']))}" for i, v in st.items()])
    print("inverse_sep", [out[f"s{k}_losses"] for k in range(4)], "noise calls", len(tap.shapes), [x for x in out["adam_steps"] if x.startswith("reward_model1.w")])
    save("g12_sepreward_walker", **out)


def g13(tag="g13_dyn_train", **cfg_over):
    """MOBODYEnsembleDynamics.train (mobody_dynamics.py:731-978) end to end on a tiny data set, max_epochs=2: holdout
    split (random_split), bootstrap indices (torch.randint), per-epoch learn(src) + 3 x learn(trg), validate, per-member
    early-stopping bookkeeping (update_save on > 1 % improvement), shuffle_rows, select_elites / set_elites / load_save."""
    S, A, bs = 17, 6, 32
    dyn, m, p = make_dyn_trainer(S, A, 221, **cfg_over)
    src = gi.batch(901, 150, S, A); trg = gi.batch(902, 90, S, A)
    rec = []
    o_val = dyn.validate

    def validate(*a, **k):
        r = o_val(*a, **k); rec.append(np.array([r[0], r[1]], np.float64)); return r

    dyn.validate = validate
    torch.manual_seed(41); np.random.seed(41)
    with NoiseTap(1300) as tap, CudaAlias():
        dyn.train(tuple(torch.from_numpy(x) for x in src), tuple(torch.from_numpy(x) for x in trg), max_epochs=2, batch_size=bs)
    sd = m.state_dict()
    out = dict(S=S, A=A, bs=bs, seed=221, alive_val=0.85, wsum=gi.checksum(p), noise_seed=1300, rng_seed=41, lr=1e-3,
               n_src=150, n_trg=90, validate=np.stack(rec), elites=sd["elites"].numpy(), n_noise=len(tap.shapes),
               total_steps=dyn.total_steps)
    for k, v in sd.items():
        if k.split(".")[0] in [n for n, _, _ in gi.dyn_layer_dims(S, A)]:
            out["sd::" + k] = sub101(v.numpy())
    print("train: elites", out["elites"], "validate calls", len(rec), "noise calls", len(tap.shapes), "steps", dyn.total_steps)
    print(np.stack(rec)[:, 0])
    save(tag, **out)


def g13s():
    """The g13 run with config inverse_sep_reward_loss = 1 (:935-941: one learn_sep_reward pass per epoch after the three target passes)."""
    g13("g13_dyn_train_sepreward", inverse_sep_reward_loss=1)


def g13a():
    """config train_with_src_threshold != 1 (data_augmentation, mobody_dynamics.py:685-729, and train()'s concatenation :797-812):
    a domain classifier is trained for 8 000 steps on batches of 256 + 256 rows, every source row whose sas head says "target"
    with probability above the threshold (after the reference's second softmax) joins the target TRAINING set, then train()
    runs as usual (max_epochs = 1 here).  The classifier's 8 000 chained noisy steps are not a parity target; the fixture holds
    the trained classifier, the threshold (placed in the widest gap of the sorted probabilities), the selected rows and what
    train() made of them -- the mirror is checked from the trained classifier on."""
    S, A, bs = 17, 6, 32
    import algo.dynamics.mobody_dynamics as md
    extra = dict(state_dim=S, action_dim=A, hidden_sizes=256, gaussian_noise_std=1.0, actor_lr=3e-4)
    src = gi.batch(901, 150, S, A); trg = gi.batch(902, 90, S, A)
    trg[0][:, 2] += 0.6; trg[2][:, 2] += 0.6                      # a shifted state dimension: the domains are separable to a degree

    def rb_of(rows):
        rb = ref_utils.ReplayBuffer(S, A, "cpu", max_size=len(rows[0]))
        rb.add_batch(dict(obss=torch.from_numpy(rows[0]), next_obss=torch.from_numpy(rows[2]), actions=torch.from_numpy(rows[1]),
                          rewards=torch.from_numpy(rows[3]), terminals=torch.from_numpy(rows[4])))
        return rb

    # pass 1: train the classifier exactly as data_augmentation does, read the probabilities, place the threshold in a gap
    dyn, m, p = make_dyn_trainer(S, A, 221, train_with_src_threshold=0.5, **extra)
    torch.manual_seed(43); np.random.seed(43)
    with NoiseTap(1600) as tap, CudaAlias():
        o_rb = md.utils.ReplayBuffer if hasattr(md, "utils") else None
        dyn.data_augmentation((rb_of(src), rb_of(trg)))
        cls_sd = {k: v.detach().clone() for k, v in dyn.classifier.state_dict().items()}
        with torch.no_grad():
            sas, _ = dyn.classifier(torch.from_numpy(src[0]), torch.from_numpy(src[1]), torch.from_numpy(src[2]), with_noise=False)
            probs = torch.softmax(sas, -1)[:, 1].numpy()
    sp = np.sort(probs)
    lo, hi = int(0.3 * len(sp)), int(0.8 * len(sp))
    j = lo + int(np.argmax(np.diff(sp[lo:hi])))
    thr = float(0.5 * (sp[j] + sp[j + 1]))
    print("probs range", sp[0], sp[-1], "threshold", thr, "gap", sp[j + 1] - sp[j], "selected", int((probs > thr).sum()))
    # pass 2: the whole train() with that threshold and the SAME trained classifier (data_augmentation's training loop is cut to
    # zero steps harness-side by handing it a classifier that is already trained)
    dyn, m, p = make_dyn_trainer(S, A, 221, train_with_src_threshold=thr, **extra)
    rec = []
    o_val = dyn.validate

    def validate(*a, **k):
        r = o_val(*a, **k); rec.append(np.array([r[0], r[1]], np.float64)); return r

    dyn.validate = validate
    o_upd = dyn.update_classifier
    state = dict(loaded=False)

    def upd(*a, **k):                                                # the 8 000 calls: load the trained weights once, train nothing
        if not state["loaded"]:
            dyn.classifier.load_state_dict(cls_sd); state["loaded"] = True
        z = torch.zeros(())
        return z, z

    dyn.update_classifier = upd
    torch.manual_seed(41); np.random.seed(41)
    with NoiseTap(1300) as tap, CudaAlias():
        dyn.train(tuple(torch.from_numpy(x) for x in src[:4]), tuple(torch.from_numpy(x) for x in trg[:4]), max_epochs=1, batch_size=bs,
                  buffer=(rb_of(src), rb_of(trg)))
    sim = dyn.src_replay_buffer_sim_trg
    out = dict(S=S, A=A, bs=bs, seed=221, alive_val=0.85, wsum=gi.checksum(p), noise_seed=1300, rng_seed=41, lr=1e-3, n_src=150, n_trg=90,
               threshold=thr, probs=probs.astype(np.float64), include=(probs > thr), n_added=int(sim.size),
               sim_state=sim.state[:sim.size].numpy(), validate=np.stack(rec), elites=m.state_dict()["elites"].numpy(),
               n_noise=len(tap.shapes), total_steps=dyn.total_steps, trg_shift=0.6)
    for k, v in cls_sd.items():
        out["cls::" + k] = v.numpy()
    print("augmentation: added", out["n_added"], "steps", dyn.total_steps, "noise calls", len(tap.shapes))
    save("g13_dyn_train_augment", **out)


def g13t():
    """The same run with config train_together = 1 (:853-880: per epoch learn(source) then ONE learn_src_trg pass, no
    reshuffle of the bootstrap indices)."""
    import warnings as w_
    with w_.catch_warnings():
        w_.simplefilter("ignore")
        g13("g13_dyn_train_together", train_together=1)


def g14():
    """Target-data ingestion, dataset/call_dataset.py:21-109 (`call_tar_dataset`): the HDF5 arrays -> transitions
    transformation.  The module imports gym, d4rl and h5py, none of which exist in this image; the harness pre-seeds
    sys.modules with minimal stand-ins (gym.make -> an object with _max_episode_steps, h5py.File -> an in-memory
    group over the fixture arrays) so that the reference's own loop runs on known arrays.  Inputs and the function's
    outputs are stored; nothing of the reference's text is."""
    class DS:
        def __init__(self, a): self.a = a
        def __getitem__(self, k): return self.a[k]

    cur = {}

    class File:
        def __init__(self, path, mode): self.path = path
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def visititems(self, fn):
            for k, v in cur.items():
                fn(k, DS(v))
        def __getitem__(self, k): return DS(cur[k])

    h5 = types.ModuleType("h5py"); h5.File = File; h5.Dataset = DS
    gymm = types.ModuleType("gym"); gymm.make = lambda name: types.SimpleNamespace(_max_episode_steps=10)
    for name, mod in (("h5py", h5), ("gym", gymm), ("d4rl", types.ModuleType("d4rl"))):
        sys.modules.setdefault(name, mod)
    from dataset.call_dataset import call_tar_dataset
    rng = np.random.default_rng(14)
    out = {}
    for tag, with_timeouts, rew2d in (("a", True, False), ("b", False, True)):
        N, S, A = 57, 17, 6
        cur.clear()
        cur.update(observations=rng.standard_normal((N, S)).astype(np.float64), actions=rng.uniform(-1, 1, (N, A)).astype(np.float32),
                   rewards=rng.standard_normal((N, 1) if rew2d else N).astype(np.float64),
                   terminals=(rng.uniform(size=N) > 0.9))
        if with_timeouts:
            cur["timeouts"] = (np.arange(N) % 10 == 9)
        res = call_tar_dataset("walker2d-friction", 2.0, "medium")
        for k, v in cur.items():
            out[f"{tag}_in_{k}"] = v.copy()
        for k, v in res.items():
            out[f"{tag}_out_{k}"] = np.asarray(v)
        print("ingest", tag, {k: (np.asarray(v).shape, np.asarray(v).dtype) for k, v in res.items()})
    save("g14_ingest", **out)


def g16():
    """Policy checkpoints written by the REFERENCE's own MOBODY.save (mobody.py:584-588) after two train() steps of the
    G7 default setup -> tests/golden/ckpt_ref/model_{actor,critic,actor_optimizer,critic_optimizer} (plain tensors,
    loadable with weights_only=True), plus what the reference computes in step 3 when it continues from them."""
    S, A, bs = 17, 6, 32
    cfg = policy_cfg(S, A)
    pol, pa, pq, pv = make_policy(cfg, 401)
    src = FixedRB(gi.batch(501, 64, S, A)); tar = FixedRB(gi.batch(502, 64, S, A)); fake = FixedRB(gi.batch(503, 64, S, A))
    pol.fake_replay_buffer = fake
    rec = dict(q_loss=[], pi_loss=[], bc_loss=[])
    for nm, key in (("update_q_functions", "q_loss"), ("update_policy", "pi_loss"), ("bc_loss", "bc_loss")):
        def mk(orig, key):
            def f(*a, **k):
                o = orig(*a, **k); rec[key].append(float(o.detach())); return o
            return f
        setattr(pol, nm, mk(getattr(pol, nm), key))
    pol.total_it = 1
    dummy = types.SimpleNamespace(log=lambda *a, **k: None)
    for _ in range(2):
        pol.train(src, tar, bs, None, dummy)
    d = os.path.join(HERE, "ckpt_ref")
    os.makedirs(d, exist_ok=True)
    pol.save(os.path.join(d, "model"))
    # continue the way a user of the reference would: a FRESH policy object (same initial weights, so its target
    # critic is the initial critic -- MOBODY.load does not restore target_q_funcs, :589-594) loads the files, then trains
    rec = dict(q_loss=[], pi_loss=[], bc_loss=[])
    pol, _, _, _ = make_policy(cfg, 401)
    pol.fake_replay_buffer = fake
    for nm, key in (("update_q_functions", "q_loss"), ("update_policy", "pi_loss"), ("bc_loss", "bc_loss")):
        def mk2(orig, key):
            def f(*a, **k):
                o = orig(*a, **k); rec[key].append(float(o.detach())); return o
            return f
        setattr(pol, nm, mk2(getattr(pol, nm), key))
    pol.load(os.path.join(d, "model"))
    pol.total_it = 3
    pol.train(src, tar, bs, None, dummy)
    out = dict(S=S, A=A, bs=bs, seed=401, q_loss=np.array(rec["q_loss"]), pi_loss=np.array(rec["pi_loss"]),
               bc_loss=np.array(rec["bc_loss"]))
    for nm, mod in (("q", pol.q_funcs), ("actor", pol.policy)):
        for k, v in mod.state_dict().items():
            out[f"s3_{nm}_p::{k}"] = sub(v.numpy())
    print("ckpt files", {f: os.path.getsize(os.path.join(d, f)) for f in sorted(os.listdir(d))}, "losses", rec["q_loss"])
    save("g16_ckpt_step3", **out)


# --------------------------------------------------------------------------- G17
class RotatingRB(FixedRB):
    """Preset rows, a different window of them per call: sample k returns rows [(k * 13) % (size - n) ...)."""

    def sample(self, n):
        k = len(self.calls)
        self.calls.append(n)
        o = (k * 13) % (self.size - n + 1)
        return tuple(r[o:o + n].clone() for r in self.rows)


def g17():
    """A 30-step train() trajectory (no refresh inside): per-step losses and the final parameters.  Pins that the mirror
    TRACKS the reference over many optimizer steps (Adam moments, bias corrections, Polyak target), not only steps 1-2."""
    S, A, bs, steps = 17, 6, 32, 30
    cfg = policy_cfg(S, A)
    pol, pa, pq, pv = make_policy(cfg, 411)
    src = RotatingRB(gi.batch(511, 96, S, A)); tar = RotatingRB(gi.batch(512, 96, S, A)); fake = RotatingRB(gi.batch(513, 96, S, A))
    pol.fake_replay_buffer = fake
    rec = dict(q_loss=[], pi_loss=[], bc_loss=[])
    for nm, key in (("update_q_functions", "q_loss"), ("update_policy", "pi_loss"), ("bc_loss", "bc_loss")):
        def mk(orig, key):
            def f(*a, **k):
                o = orig(*a, **k); rec[key].append(float(o.detach())); return o
            return f
        setattr(pol, nm, mk(getattr(pol, nm), key))
    pol.total_it = 1
    dummy = types.SimpleNamespace(log=lambda *a, **k: None)
    for _ in range(steps):
        pol.train(src, tar, bs, None, dummy)
    out = dict(S=S, A=A, bs=bs, steps=steps, seed=411, wsum_actor=gi.checksum(pa), wsum_q=gi.checksum(pq),
               q_loss=np.array(rec["q_loss"]), pi_loss=np.array(rec["pi_loss"]), bc_loss=np.array(rec["bc_loss"]),
               calls_src=np.array(src.calls), calls_tar=np.array(tar.calls), calls_fake=np.array(fake.calls))
    for nm, mod in (("q", pol.q_funcs), ("actor", pol.policy), ("qt", pol.target_q_funcs)):
        for k, v in mod.state_dict().items():
            out[f"final_{nm}_p::{k}"] = sub(v.numpy())
    print("trajectory q_loss", rec["q_loss"][0], "->", rec["q_loss"][-1], "pi_loss", rec["pi_loss"][0], "->", rec["pi_loss"][-1],
          "calls", src.calls[:3], tar.calls[:3], fake.calls[:3])
    save("g17_train_trajectory", **out)


# --------------------------------------------------------------------------- G18
def g18():
    """The MOPO ablation (config['mopo'] = 1): forward_trg / forward_src, step() in the four penalty / model flag
    combinations, and a 3-step rollout through MOBODY.rollout, at walker and ant shapes."""
    cfg_dyn = dict(DYN_CFG, mopo=1)
    for tag, S, A, B, task, seed in (("walker", 17, 6, 48, "walker2d-medium-v2", 801), ("ant", 111, 8, 40, "ant-medium-v2", 802)):
        p = gi.dyn_params(seed, S, A, mopo=True)
        p["za_src3.bias"][:, 0, 0] += np.float32(-0.35 if tag == "walker" else -0.3)      # a few rows leave the alive box
        m = MOBODYModule(S, A, 256, 7, 5, device="cpu", config=dict(cfg_dyn))
        sd = m.state_dict()
        for k, v in p.items():
            assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
            sd[k] = torch.from_numpy(v)
        m.load_state_dict(sd)
        m.inference()
        rng = np.random.default_rng(seed + 1000)
        obs = gi.walker_like_obs(rng, B, S) if tag == "walker" else (0.6 * (np.arange(S) == 0) + 0.1 * rng.standard_normal((B, S))).astype(np.float32)
        act = rng.uniform(-1, 1, (B, A)).astype(np.float32)
        with torch.no_grad():
            mt, _, _ = m.forward_trg(torch.from_numpy(obs), torch.from_numpy(act))
            ms, _, _ = m.forward_src(torch.from_numpy(obs), torch.from_numpy(act))
        out = dict(S=S, A=A, seed=seed, task=task, wsum=gi.checksum(p), obs=obs, act=act, mean_trg=mt.numpy(), mean_src=ms.numpy())
        dyn = MOBODYEnsembleDynamics(dict(cfg_dyn), m, None, None, get_termination_fn(task), penalty_coef=0.1)
        for up in (True, False):
            for ut in (True, False):
                np.random.seed(seed)
                with RngTap(seed + 7) as tap:
                    no, rw, term, info = dyn.step(torch.from_numpy(obs), torch.from_numpy(act), up, ut)
                k = f"step_p{int(up)}_t{int(ut)}_"
                out.update({k + "eps": tap.eps[0], k + "idx": tap.idx[0], k + "next_obs": no.numpy(), k + "reward": rw.numpy(),
                            k + "terminal": term, k + "penalty": info["penalty"].numpy(), k + "raw_reward": info["raw_reward"].numpy()})
        if tag == "walker":
            cfg = policy_cfg(S, A, mopo=1)
            pol, pa, _, _ = make_policy(cfg, 311)
            pol.dynamics = dyn
            cfg["env_filter"] = float(np.median(out["step_p1_t1_penalty"]))
            np.random.seed(78)
            with RngTap(204) as tap:
                tr, inf = pol.rollout(torch.from_numpy(obs), 3, True)
            out.update(actor_seed=311, wsum_actor=gi.checksum(pa), env_filter=cfg["env_filter"], n_steps=len(tap.eps),
                       num_transitions=inf["num_transitions"])
            for t, (e, i) in enumerate(zip(tap.eps, tap.idx)):
                out[f"roll_eps{t}"] = e; out[f"roll_idx{t}"] = i
            for k, v in tr.items():
                out["roll_" + k] = v.numpy()
            print("mopo rollout rows", [e.shape[1] for e in tap.eps], "kept", len(tr["obss"]))
        print("mopo", tag, "terminated rows:", int(out["step_p1_t1_terminal"].sum()), "/", B)
        save(f"g18_mopo_{tag}", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g234", "g5", "g6", "g7", "g8", "g9", "g9b", "g11", "g12", "g13", "g14", "g16", "g17", "g18"]
    for w in which:
        globals()[w]()
