"""Pin the CPU oracle (oracle/mobody_oracle.py) against golden vectors produced by the real reference.

CPU-only (-m "not gpu").  Tolerances: the oracle uses the same ATen CPU kernels as the
reference, so most outputs agree to ~1e-6; 2e-5 relative is asserted to stay robust to
BLAS blocking differences between hosts.
"""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import mobody_oracle as O

RT, AT = 2e-5, 2e-6


def close(a, b, rtol=RT, atol=AT):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


def test_g1_ensemble_linear():
    g = gu.load("g1_ensemble_linear")
    W, b = O.T(g["W"]), O.T(g["b"])
    close(O.ensemble_linear(O.T(g["x2"]), W, b), g["y2"])
    close(O.ensemble_linear(O.T(g["x3"]), W, b), g["y3"])


@pytest.mark.parametrize("tag", ["walker", "ant", "pen"])
def test_g234_dynamics(tag):
    g = gu.load(f"g234_dynamics_{tag}")
    p = O.to_torch(gu.dyn_params_for(g))
    obs, act, nxt = O.T(g["obs"]), O.T(g["act"]), O.T(g["nxt"])
    with torch.no_grad():
        mt, zmu, zlv = O.dyn_forward(p, obs, act, True)
        ms, _, _ = O.dyn_forward(p, obs, act, False)
        rmu, rlv = O.dyn_reward(p, obs, act, nxt)
    close(mt, g["mean_trg"]); close(ms, g["mean_src"]); close(zmu, g["zs_mu"]); close(zlv, g["zs_logvar"])
    close(rmu, g["r_mu"]); close(rlv, g["r_logvar"])
    task = str(g["task"])
    for up in (1, 0):
        for ut in (1, 0):
            k = f"step_p{up}_t{ut}_"
            with torch.no_grad():
                st = O.dyn_step(p, obs, act, g[k + "eps"], g[k + "idx"], task, penalty_coef=0.1,
                                use_penalty=bool(up), use_trg=bool(ut))
            close(st["next_obs"], g[k + "next_obs"]); close(st["reward"], g[k + "reward"])
            close(st["penalty"], g[k + "penalty"]); close(st["raw_reward"], g[k + "raw_reward"])
            close(st["mean"], g[k + "samples"])
            assert (st["terminal"] == g[k + "terminal"]).all()
    assert 0 < g["step_p1_t1_terminal"].sum() < len(obs)      # fixture exercises both outcomes


@pytest.mark.parametrize("tag", ["walker", "ant"])
def test_g18_mopo_ablation(tag):
    """config['mopo'] = 1: means = s + MLP_e([s, a]) for both models, then the unchanged step (and a 3-step rollout)."""
    g = gu.load(f"g18_mopo_{tag}")
    p = O.to_torch(gu.mopo_params_for(g, tag))
    obs, act = O.T(g["obs"]), O.T(g["act"])
    with torch.no_grad():
        close(O.dyn_forward(p, obs, act, True)[0], g["mean_trg"]); close(O.dyn_forward(p, obs, act, False)[0], g["mean_src"])
    assert (g["mean_trg"] == g["mean_src"]).all()
    task = str(g["task"])
    for up in (1, 0):
        for ut in (1, 0):
            k = f"step_p{up}_t{ut}_"
            with torch.no_grad():
                st = O.dyn_step(p, obs, act, g[k + "eps"], g[k + "idx"], task, penalty_coef=0.1, use_penalty=bool(up), use_trg=bool(ut))
            close(st["next_obs"], g[k + "next_obs"]); close(st["reward"], g[k + "reward"])
            close(st["penalty"], g[k + "penalty"]); close(st["raw_reward"], g[k + "raw_reward"])
            assert (st["terminal"] == g[k + "terminal"]).all()
    assert 0 < g["step_p1_t1_terminal"].sum() < len(obs)
    if tag == "walker":
        pa, _, _ = gu.policy_params(int(g["actor_seed"]), int(g["S"]), int(g["A"]))
        n = int(g["n_steps"])
        cfg = gu.policy_cfg(int(g["S"]), int(g["A"]), env_filter=float(g["env_filter"]))
        with torch.no_grad():
            tr, info = O.rollout(O.to_torch(pa), p, g["obs"], 3, [g[f"roll_eps{t}"] for t in range(n)],
                                 [g[f"roll_idx{t}"] for t in range(n)], task, cfg, penalty_coef=0.1)
        assert info["num_transitions"] == int(g["num_transitions"])
        for k in ("obss", "next_obss", "actions", "rewards", "terminals", "penalty"):
            assert tr[k].shape == g["roll_" + k].shape, k
            close(tr[k], g["roll_" + k], rtol=2e-5, atol=2e-5)


def test_g5_termination():
    g = gu.load("g5_termination")
    tasks = sorted({k.split("::")[0] for k in g if "::" in k})
    assert len(tasks) == 10
    for t in tasks:
        n = g[t + "::next_obs"]
        d = O.termination(t, n, None, n)
        assert d.shape == (len(n), 1) and d.dtype == bool
        assert (d == g[t + "::done"]).all(), t
    assert int(g["unknown_raises"]) == 1
    with pytest.raises(TypeError):
        O.termination("reacher-x", np.zeros((1, 3)), None, np.zeros((1, 3)))


@pytest.mark.parametrize("tag", ["h1", "h5", "h3_nopen"])
def test_g6_rollout(tag):
    g = gu.load(f"g6_rollout_{tag}")
    S, A = int(g["S"]), int(g["A"])
    p = O.to_torch(gu.dyn_params_for(g))
    pa, _, _ = gu.policy_params(int(g["actor_seed"]), S, A)
    cfg = gu.policy_cfg(S, A, env_filter=float(g["env_filter"]))
    n = int(g["n_steps"])
    with torch.no_grad():
        res, info = O.rollout(O.to_torch(pa), p, g["init"], int(g["H"]), [g[f"eps{t}"] for t in range(n)],
                              [g[f"idx{t}"] for t in range(n)], "walker2d-medium-v2", cfg,
                              use_trg=bool(int(g["use_trg"])), penalty_coef=0.1)
    assert info["num_transitions"] == int(g["num_transitions"])
    close(info["reward_mean"], float(g["reward_mean"]))
    for k in ("obss", "next_obss", "actions", "rewards", "terminals", "penalty"):
        assert res[k].shape == g["out_" + k].shape, k
        close(res[k], g["out_" + k])


@pytest.mark.parametrize("tag", list(gu.G7_VARIANTS))
def test_g7_train_step(tag):
    g = gu.load(f"g7_train_{tag}")
    S, A, bs = int(g["S"]), int(g["A"]), int(g["bs"])
    cfg = gu.policy_cfg(S, A, **gu.G7_VARIANTS[tag])
    pa, pq, pv = gu.policy_params(int(g["seed"]), S, A)
    st = O.TrainState(pa, pq, pv)
    dyn = None
    if tag == "par":
        dyn = O.to_torch(gu.dyn_params_for(dict(S=S, A=A, dyn_seed=201, alive_val=0.85, wsum_dyn=g["wsum_dyn"])))
    for step in (1, 2):
        override = None
        if tag == "par":     # mobody.py:428-434
            src = gu.gi.batch(501, 64, S, A)
            with torch.no_grad():
                stp = O.dyn_step(dyn, src[0][:bs], src[1][:bs], g[f"par_eps{step}"], g[f"par_idx{step}"],
                                 "walker2d-medium-v2", penalty_coef=0.1)
                pen = ((O.T(src[2][:bs]) - stp["next_obs"]) ** 2).mean(1, keepdim=True)
            override = src[3][:bs] - cfg["penalty_coef"] * pen.numpy()
        batch, n_true = gu.g7_batch(cfg, bs, S, A, override)
        out = O.train_step(st, batch, n_true, cfg)
        close(out["q_loss"], g["q_loss"][step - 1], rtol=1e-5)
        close(out["pi_loss"], g["pi_loss"][step - 1], rtol=1e-5)
        close(out["bc_loss"], g["bc_loss"][step - 1], rtol=1e-5)
        for nm, grads in (("q", out["q_grads"]), ("actor", out["actor_grads"])):
            for k, v in grads.items():
                key = f"s{step}_{nm}_g::{k}"
                close(gu.sub(v.numpy()), g[key], rtol=2e-4, atol=2e-7)
        for nm, params in (("q", st.q), ("actor", st.actor), ("qt", st.q_targ), ("v", st.v)):
            for k, v in params.items():
                close(gu.sub(v.numpy()), g[f"s{step}_{nm}_p::{k}"], rtol=1e-5, atol=1e-6)


def test_g17_thirty_step_trajectory():
    """The CPU restatement tracks the reference over 30 optimizer steps (Adam moments / bias corrections / Polyak)."""
    g = gu.load("g17_train_trajectory")
    S, A, bs, steps = int(g["S"]), int(g["A"]), int(g["bs"]), int(g["steps"])
    cfg = gu.policy_cfg(S, A)
    pa, pq, pv = gu.policy_params(int(g["seed"]), S, A)
    st = O.TrainState(pa, pq, pv)
    for k in range(steps):
        batch, n_true = gu.g17_batch(k, cfg, bs, S, A)
        out = O.train_step(st, batch, n_true, cfg)
        close(out["q_loss"], g["q_loss"][k], rtol=2e-5)
        close(out["pi_loss"], g["pi_loss"][k], rtol=5e-5, atol=5e-6)
        close(out["bc_loss"], g["bc_loss"][k], rtol=5e-5, atol=5e-6)
    for nm, params in (("q", st.q), ("actor", st.actor), ("qt", st.q_targ)):
        for k_, v in params.items():
            close(gu.sub(v.numpy()), g[f"final_{nm}_p::{k_}"], rtol=1e-4, atol=2e-6)


def test_g8_ring_append_and_sample():
    g = gu.load("g8_replay")
    cap, S, A = 50, 5, 2
    buf = dict(state=np.zeros((cap, S), np.float32), action=np.zeros((cap, A), np.float32),
               next_state=np.zeros((cap, S), np.float32), reward=np.zeros((cap, 1), np.float32),
               not_done=np.zeros((cap, 1), np.float32))
    ptr = size = 0
    for ci, (M, want_ptr, want_size) in enumerate(g["log"]):
        segs, ptr, size = O.ring_append_plan(ptr, size, cap, int(M))
        for dst, src, ln in segs:
            buf["state"][dst:dst + ln] = g[f"add{ci}_obss"][src:src + ln]
            buf["action"][dst:dst + ln] = g[f"add{ci}_actions"][src:src + ln]
            buf["next_state"][dst:dst + ln] = g[f"add{ci}_next_obss"][src:src + ln]
            buf["reward"][dst:dst + ln] = g[f"add{ci}_rewards"][src:src + ln]
            buf["not_done"][dst:dst + ln] = 1.0 - g[f"add{ci}_terminals"][src:src + ln]
        assert (ptr, size) == (int(want_ptr), int(want_size)), ci
        for k in buf:
            assert (buf[k] == g[f"after{ci}_{k}"]).all(), (ci, k)
    np.random.seed(123)
    ind = np.random.randint(0, size, size=16)
    assert (ind == g["sample_ind"]).all()
    for k in buf:
        assert (buf[k][ind] == g["sample_" + k]).all()
    np.random.seed(0)
    stream = np.concatenate([np.random.randint(0, 1000000, size=8), np.random.randint(0, 5000, size=8)])
    assert (stream == g["stream_seed0"]).all()
    assert int(g["d4rl_size"]) == 30
    close(g["d4rl_not_done"], 1.0 - g["d4rl_terminals"].reshape(-1, 1).astype(np.float32))


def test_g9_dara():
    g = gu.load("g9_dara")
    S, A, bs = int(g["S"]), int(g["A"]), int(g["bs"])
    pc = {}
    pc.update({"sa_classifier." + k: v for k, v in gu.gi.mlp_params(int(g["seed_sa"]), S + A, 2).items()})
    pc.update({"sas_classifier." + k: v for k, v in gu.gi.mlp_params(int(g["seed_sas"]), 2 * S + A, 2).items()})
    assert abs(gu.gi.checksum(pc) - float(g["wsum"])) < 1e-9 * abs(float(g["wsum"]))
    p = O.to_torch(pc)
    with torch.no_grad():
        ps, pa = O.classifier_probs(p, O.T(g["s"]), O.T(g["a"]), O.T(g["s2"]))
        close(ps, g["probs_sas"]); close(pa, g["probs_sa"])
        close(O.dara_delta_r(p, g["s"], g["a"], g["s2"]), g["delta_r"], rtol=1e-4, atol=1e-5)
    # one classifier update (mobody.py:146-181): rows src|tar, labels 0|1, permuted
    src = gu.gi.batch(704, 64, S, A); tar = gu.gi.batch(705, 64, S, A)
    perm = g["perm"]
    cat = [np.concatenate([src[i][:bs], tar[i][:bs]], 0)[perm] for i in range(3)]
    label = np.concatenate([np.zeros(bs), np.ones(bs)])[perm]
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss = O.classifier_loss(pr, cat[0], cat[1], cat[2], label, g["noise_sas"], g["noise_sa"], 1.0)
    close(float(loss), float(g["loss_sa"]) + float(g["loss_sas"]), rtol=1e-5)
    grads = torch.autograd.grad(loss, list(pr.values()))
    for (k, _), gr in zip(pr.items(), grads):
        close(gu.sub(gr.numpy()), g["cls_g::" + k], rtol=2e-4, atol=2e-7)


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors)."""
    r = O.philox4x32(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in r] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    r = O.philox4x32(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(x) for x in r] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    r = O.philox4x32(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0)
    assert [int(x) for x in r] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    z = O.rng_normal(1, 2, 3, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    i = O.rng_index(1, 2, 3, 100000, 5)
    assert i.min() == 0 and i.max() == 4 and abs(np.bincount(i)[0] / 1e5 - 0.2) < 0.01


def _g11_setup(g, S, A):
    """Buffers, weights and draws of a g11 fixture (make_golden.g11), oracle side."""
    p = O.to_torch(gu.dyn_params_for(g))
    pa, pq, _ = gu.policy_params(int(g["seed"]), S, A)
    assert abs(gu.gi.checksum(pa) - float(g["wsum_actor"])) < 1e-9 * abs(float(g["wsum_actor"]))
    pc = {}
    pc.update({"sa_classifier." + k: v for k, v in gu.gi.mlp_params(701, S + A, 2).items()})
    pc.update({"sas_classifier." + k: v for k, v in gu.gi.mlp_params(702, 2 * S + A, 2).items()})
    bufs = []
    for seed, cap in ((801, 300), (802, 120)):
        s, a, s2, r, nd = gu.gi.batch(seed, cap, S, A)
        rb = O.RingBuffer(S, A, cap)
        rb.add_batch(dict(obss=s, actions=a, next_obss=s2, rewards=r, terminals=1.0 - nd))
        bufs.append(rb)
    # elite ids are NOT passed: the oracle draws them from the NumPy global stream where the reference does, which is
    # what keeps the later np.random.randint index draws aligned (Appendix C of SURVEY.md)
    draws = [(g[f"eps{t}"], None) for t in range(int(g["n_steps"]))]
    return p, pa, pq, pc, bufs[0], bufs[1], draws


@pytest.mark.parametrize("tag", ["default", "fromsrc"])
def test_g11_refresh_step(tag):
    """First train() call: refresh order, strict '<' relabel filter, ring wrap, index-draw order, then the gradient step."""
    g = gu.load(f"g11_refresh_{tag}")
    S, A, bs = int(g["S"]), int(g["A"]), int(g["bs"])
    over = {k: (int(v) if v.lstrip("-").isdigit() else v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    cfg = gu.policy_cfg(S, A, src_rollout_length=2, trg_rollout_length=3, env_filter=float(g["env_filter"]), **over)
    p, pa, pq, pc, src, tar, draws = _g11_setup(g, S, A)
    fake = O.RingBuffer(S, A, int(g["fake_cap"]))
    sizes = (int(g["refresh_src"]), int(g["refresh_tar"]), int(g["refresh_from_src_tar"]))
    cls = {k: O.T(v).clone() for k, v in pc.items()}
    mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in cls.items()}

    def classifier_update(src_rb, tar_rb, batch_size):                      # update_classifier, mobody.py:146-181
        sb, tb = src_rb.sample(batch_size), tar_rb.sample(batch_size)
        perm = g["cls_perm"]
        cat = [torch.cat([sb[i], tb[i]], 0)[perm] for i in range(3)]
        label = np.concatenate([np.zeros(batch_size), np.ones(batch_size)])[perm]
        with torch.enable_grad():
            pr = {k: v.clone().requires_grad_(True) for k, v in cls.items()}
            loss = O.classifier_loss(pr, cat[0], cat[1], cat[2], label, g["cls_noise_sas"], g["cls_noise_sa"], cfg["gaussian_noise_std"])
            grads = torch.autograd.grad(loss, list(pr.values()))
        for (k, _), gr in zip(pr.items(), grads):
            O.adam_update(cls[k], gr, mom[k][0], mom[k][1], 1, cfg["actor_lr"])

    np.random.seed(int(g["np_seed"]))
    ns, nt = int(cfg["src_ratio"] * bs), int(cfg["trg_ratio"] * bs)
    sb, tb = src.sample(ns), tar.sample(nt)                                  # mobody.py:399-400 come first
    with torch.no_grad():
        O.refresh(O.to_torch(pa), p, src, tar, fake, cfg, "walker2d-medium-v2", draws, bs, sizes=sizes, penalty_coef=0.1,
                  classifier_update=classifier_update, cls_p=cls)
    assert (fake.ptr, fake.size) == (int(g["fake_ptr"]), int(g["fake_size"]))
    for f in O.RingBuffer.FIELDS:
        close(getattr(fake, f), g["fake_" + f], rtol=1e-5, atol=1e-6)
    if tag == "fromsrc":
        for k, v in cls.items():
            close(gu.sub(v.numpy()), g["cls_p::" + k], rtol=1e-5, atol=2e-6)
    fb = fake.sample(int(cfg["fake_batch_scale"] * bs))                       # :523
    batch = tuple(torch.cat([sb[i], tb[i], fb[i]], 0) for i in range(5))
    st = O.TrainState(pa, pq, None)
    out = O.train_step(st, batch, ns + nt, cfg)
    close(float(out["q_loss"]), g["q_loss"][0], rtol=1e-5, atol=0)
    close(float(out["pi_loss"]), g["pi_loss"][0], rtol=2e-5, atol=0)
    for k in st.q:
        close(gu.sub(st.q[k].numpy()), g["s1_q_p::" + k], rtol=1e-5, atol=1e-6)


def test_g9b_dara_penalize_fake():
    """penalize_fake=1: rows that reach the classifier are src (label 0) | fake (label 1), mobody.py:146-165."""
    g = gu.load("g9_dara_penfake")
    S, A, bs = int(g["S"]), int(g["A"]), int(g["bs"])
    assert [str(x) for x in g["draw_log"]] == [f"src:{bs}", f"tar:{bs}", f"fake:{bs}", f"tar:{2 * bs}"]
    pc = {}
    pc.update({"sa_classifier." + k: v for k, v in gu.gi.mlp_params(int(g["seed_sa"]), S + A, 2).items()})
    pc.update({"sas_classifier." + k: v for k, v in gu.gi.mlp_params(int(g["seed_sas"]), 2 * S + A, 2).items()})
    src = gu.gi.batch(704, 64, S, A); fake = gu.gi.batch(706, 64, S, A)
    perm = g["perm"]
    assert len(perm) == 2 * bs
    cat = [np.concatenate([src[i][:bs], fake[i][:bs]], 0)[perm] for i in range(3)]
    label = np.concatenate([np.zeros(bs), np.ones(bs)])[perm]
    pr = {k: v.clone().requires_grad_(True) for k, v in O.to_torch(pc).items()}
    loss = O.classifier_loss(pr, cat[0], cat[1], cat[2], label, g["noise_sas"], g["noise_sa"], 1.0)
    close(float(loss.detach()), float(g["loss_sa"]) + float(g["loss_sas"]), rtol=1e-5)
    for (k, _), gr in zip(pr.items(), torch.autograd.grad(loss, list(pr.values()))):
        close(gu.sub(gr.numpy()), g["cls_g::" + k], rtol=2e-4, atol=2e-7)


def _noise7(rng, b, S):
    return [rng.standard_normal((7, b, 16)).astype(np.float32) for _ in range(6)] + \
           [rng.standard_normal((7, b, S)).astype(np.float32)]


@pytest.mark.parametrize("tag", ["walker", "pen"])
def test_g12_dynamics_pretraining_steps(tag):
    """Four learn() steps src, trg, src, trg against the reference: losses, every gradient (thinned values + full-tensor
    float64 sums), which parameters receive a gradient, per-parameter Adam step counts, post-step parameters."""
    g = gu.load(f"g12_pretrain_{tag}")
    S, A, b, seed = int(g["S"]), int(g["A"]), int(g["b"]), int(g["seed"])
    p = gu.dyn_params_for(g)
    st = O.DynTrainState(p, lr=float(g["lr"]))
    rng = gu.gi.noise_stream(int(g["noise_seed"]))
    shapes = [tuple(int(x) for x in s.split(",")) for s in g["noise_shapes"]]
    assert shapes[:7] == [(7, b, 16)] * 6 + [(7, b, S)]            # the order of the reference's randn_like calls
    for step, use_trg in enumerate((False, True, False, True)):
        rows = gu.gi.pretrain_batch(3000 + 10 * seed + step, b, S, A)
        out = O.dyn_learn_step(st, *rows, _noise7(rng, b, S), use_trg)
        close(np.array(out["losses"]), g[f"s{step}_losses"], rtol=2e-5, atol=1e-6)
        has = sorted(k for k, v in out["grads"].items() if v is not None)
        assert has == [str(x) for x in g[f"s{step}_has_grad"]]
        scale = max(float(np.abs(g[k]).max()) for k in g if k.startswith(f"s{step}_g::"))
        for k in has:
            gr = out["grads"][k].numpy()
            close(gu.sub101(gr), g[f"s{step}_g::{k}"], rtol=1e-4, atol=1e-5 * scale)
            s64 = g[f"s{step}_gsum::{k}"]
            close((gr.astype(np.float64) ** 2).sum(), s64[1], rtol=1e-4, atol=1e-12)
        for k, v in st.p.items():
            close(gu.sub101(v.numpy()), g[f"s{step}_p::{k}"], rtol=1e-5, atol=2e-6)
    want = dict(x.split("=") for x in g["adam_steps"])
    assert {k: int(v) for k, v in want.items()} == st.t


def novae_noise(rng, b, S):
    """config no_vae = 1: the reference draws three tensors per step -- transition_loss's sample (z5 in the seven-draw
    order), reward_loss's (z6) and the fake-next-state noise; the four encoder_loss samples are never drawn (zeros here:
    every term they feed is weighted by 0)."""
    z = np.zeros((7, b, 16), np.float32)
    z5, z6 = (rng.standard_normal((7, b, 16)).astype(np.float32) for _ in range(2))
    return [z, z, z, z, z5, z6, rng.standard_normal((7, b, S)).astype(np.float32)]


def test_g12_pretraining_steps_no_vae():
    """The no_vae ablation (mobody_dynamics.py:616-635: loss = transition_loss + reward_loss, encoder_loss neither evaluated
    nor added, reported as 0) is the default step with encoder_loss weighted by 0: losses, gradients, Adam steps."""
    g = gu.load("g12_pretrain_walker_novae")
    assert int(g["no_vae"]) == 1
    S, A, b, seed = int(g["S"]), int(g["A"]), int(g["b"]), int(g["seed"])
    p = gu.dyn_params_for(g)
    st = O.DynTrainState(p, lr=float(g["lr"]))
    rng = gu.gi.noise_stream(int(g["noise_seed"]))
    shapes = [tuple(int(x) for x in s.split(",")) for s in g["noise_shapes"]]
    assert shapes[:3] == [(7, b, 16), (7, b, 16), (7, b, S)] and len(shapes) == 12
    for step, use_trg in enumerate((False, True, False, True)):
        rows = gu.gi.pretrain_batch(3000 + 10 * seed + step, b, S, A)
        out = O.dyn_learn_step(st, *rows, novae_noise(rng, b, S), use_trg, encoder_loss_coef=0.0)
        want = g[f"s{step}_losses"]
        # (`loss = transition_loss` aliases the tensor and `loss += reward_loss` adds in place, :631,641: the "transition"
        #  number the reference reports under no_vae is the total loss)
        close(np.array([out["losses"][0], out["losses"][0]]), want[:2], rtol=2e-5, atol=1e-6)
        assert list(want[2:]) == [0.0, 0.0, 0.0]
        has = [str(x) for x in g[f"s{step}_has_grad"]]
        scale = max(float(np.abs(g[k]).max()) for k in g if k.startswith(f"s{step}_g::"))
        for k in has:
            gr = out["grads"][k].numpy()
            close(gu.sub101(gr), g[f"s{step}_g::{k}"], rtol=1e-4, atol=1e-5 * scale)
        for k, v in st.p.items():
            close(gu.sub101(v.numpy()), g[f"s{step}_p::{k}"], rtol=1e-5, atol=2e-6)


def test_g12_pretraining_steps_train_together():
    """config train_together = 1: two learn_src_trg steps (mobody_dynamics.py:521-590) -- a source batch and a target batch in
    ONE loss (target encoder_loss weighted 1 x, not 5 x), one Adam step that moves both action encoders."""
    g = gu.load("g12_together_walker")
    S, A, bs, bt = int(g["S"]), int(g["A"]), int(g["bs"]), int(g["bt"])
    p = gu.dyn_params_for(g)
    st = O.DynTrainState(p, lr=float(g["lr"]))
    rng = gu.gi.noise_stream(int(g["noise_seed"]))
    shapes = [tuple(int(x) for x in s.split(",")) for s in g["noise_shapes"]]
    assert shapes[:14] == [(7, bs, 16)] * 6 + [(7, bs, S)] + [(7, bt, 16)] * 6 + [(7, bt, S)]     # source draws, then target draws
    for step in range(2):
        src = gu.gi.pretrain_batch(5000 + 10 * step, bs, S, A); trg = gu.gi.pretrain_batch(5001 + 10 * step, bt, S, A)
        out = O.dyn_learn_step_together(st, src, trg, _noise7(rng, bs, S), _noise7(rng, bt, S))
        want = g[f"s{step}_stats"]
        assert np.isnan(want[3])                                   # the reference averages a list it never fills
        close(np.array(out["losses"]), want[[0, 1, 2, 4]], rtol=2e-5, atol=1e-6)
        has = sorted(k for k, v in out["grads"].items() if v is not None)
        assert has == [str(x) for x in g[f"s{step}_has_grad"]]
        scale = max(float(np.abs(g[k]).max()) for k in g if k.startswith(f"s{step}_g::"))
        for k in has:
            close(gu.sub101(out["grads"][k].numpy()), g[f"s{step}_g::{k}"], rtol=1e-4, atol=1e-5 * scale)
        for k, v in st.p.items():
            close(gu.sub101(v.numpy()), g[f"s{step}_p::{k}"], rtol=1e-5, atol=2e-6)
    want_t = dict(x.split("=") for x in g["adam_steps"])
    assert {k: int(v) for k, v in want_t.items()} == st.t


def sep_noise_learn(rng, b, S):
    """inverse_sep_reward_loss = 1, learn(): encoder_loss's four draws and transition_loss's one; reward_loss is not evaluated
    (slots 5 and 6 carry zero weight: zeros)."""
    z = [rng.standard_normal((7, b, 16)).astype(np.float32) for _ in range(5)]
    return z + [np.zeros((7, b, 16), np.float32), np.zeros((7, b, S), np.float32)]


def sep_noise_reward(rng, b, S):
    """learn_sep_reward: per domain the forward's state sample (slot 5) and the fake-next-state noise (slot 6)."""
    z = np.zeros((7, b, 16), np.float32)
    z6 = rng.standard_normal((7, b, 16)).astype(np.float32)
    return [z, z, z, z, z, z6, rng.standard_normal((7, b, S)).astype(np.float32)]


def test_g12_pretraining_steps_inverse_sep_reward_loss():
    """config inverse_sep_reward_loss = 1: learn() without reward_loss (the reward head has no gradient and keeps its Adam count)
    and one learn_sep_reward step (reward losses of a source + a target batch only), sequence src, trg, sep, trg."""
    g = gu.load("g12_sepreward_walker")
    S, A, bs, bt = int(g["S"]), int(g["A"]), int(g["bs"]), int(g["bt"])
    p = gu.dyn_params_for(g)
    st = O.DynTrainState(p, lr=float(g["lr"]))
    rng = gu.gi.noise_stream(int(g["noise_seed"]))
    shapes = [tuple(int(x) for x in s.split(",")) for s in g["noise_shapes"]]
    assert shapes == [(7, bs, 16)] * 10 + [(7, bs, 16), (7, bs, S), (7, bt, 16), (7, bt, S)] + [(7, bs, 16)] * 5
    for step, kind in enumerate(("src", "trg", "sep", "trg")):
        if kind == "sep":
            src = gu.gi.pretrain_batch(6000, bs, S, A); trg = gu.gi.pretrain_batch(6001, bt, S, A)
            out = O.dyn_learn_step_sep_reward(st, src, trg, sep_noise_reward(rng, bs, S), sep_noise_reward(rng, bt, S))
            close(np.array(out["losses"]), g[f"s{step}_losses"], rtol=2e-5, atol=1e-6)
        else:
            rows = gu.gi.pretrain_batch(6100 + step, bs, S, A)
            out = O.dyn_learn_step(st, *rows, sep_noise_learn(rng, bs, S), kind == "trg", with_reward=False)
            close(np.array(out["losses"]), g[f"s{step}_losses"], rtol=2e-5, atol=1e-6)
        has = sorted(k for k, v in out["grads"].items() if v is not None)
        assert has == [str(x) for x in g[f"s{step}_has_grad"]]
        scale = max(float(np.abs(g[k]).max()) for k in g if k.startswith(f"s{step}_g::"))
        for k in has:
            close(gu.sub101(out["grads"][k].numpy()), g[f"s{step}_g::{k}"], rtol=1e-4, atol=1e-5 * scale)
        for k, v in st.p.items():
            close(gu.sub101(v.numpy()), g[f"s{step}_p::{k}"], rtol=1e-5, atol=2e-6)
    want_t = dict(x.split("=") for x in g["adam_steps"])
    assert {k: int(v) for k, v in want_t.items()} == st.t and st.t["reward_model1.weight"] == 1 and st.t["zs1.weight"] == 4
