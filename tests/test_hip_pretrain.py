"""Dynamics pre-training on the GPU (csrc/pretrain.hip through the C ABI) against the reference's golden vectors
(tests/golden/g12_pretrain_*.npz, produced by make_golden.g12 from MOBODYEnsembleDynamics.learn) and the oracle.

Tolerances (north_star: 1e-5 fp32):
  * losses: 2e-5 relative (sums of ~1e4 squared residuals in a different order than torch's);
  * gradients: |hip - ref| <= 1e-5 * max|g| of the tensor's sub-network + 1e-5 |g|  (the three sub-networks' gradient
    scales differ by orders of magnitude -- 100x reconstruction weight on the encoder/decoder, 0.01x on the source
    domain's reward head -- so each is judged against its own scale);
  * parameters after Adam: same rule as tests/test_hip_train.py (99.5 % within 1e-5, all within 0.1 * lr).
Every suite that takes the `mfma` fixture runs twice at the SAME tolerances: exact fp32 MFMA and "f16x2" (the 256 x 256
layers of the three nets on the split core, tests/conftest.py); the mirror picks the mode up through ops.default_mfma().
"""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import mobody_oracle as O

pytestmark = pytest.mark.gpu

SUBNET = dict(zs="enc", tr="tr", re="rw", za="za")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def close(a, b, rtol=1e-5, atol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a.astype(np.float64), b.astype(np.float64), rtol=rtol, atol=atol)


def noise7(rng, b, S):
    return [rng.standard_normal((7, b, 16)).astype(np.float32) for _ in range(6)] + \
           [rng.standard_normal((7, b, S)).astype(np.float32)]


class Trainer:
    """Minimal driver of the pre-training entry points of the C ABI."""

    def __init__(self, p, S, A, b, dev, lr=1e-3, prec="f32"):
        from mobody_amd import ops, packing
        self.ops, self.packing, self.S, self.A, self.b, self.dev, self.lr = ops, packing, S, A, b, dev, lr
        self.prec = prec                                  # "f32" (exact fp32 MFMA) or "f16x2": every entry point takes it
        self.blob = packing.pack_pretrain(p, S, A, dev)
        self.blob_T = ops.pretrain_transpose(self.blob, S, A, precision=prec)
        self.grad, self.m, self.v = (torch.zeros_like(self.blob) for _ in range(3))
        self.loss = torch.zeros(5, device=dev)
        self.ws = ops.pretrain_workspace(S, A, b, dev)
        self.t_main, self.t_za = 0, {False: 0, True: 0}
        self.zero2 = {k: torch.zeros(7, 32, 32, device=dev) for k in ("za_src2.weight", "za_trg2.weight")}
        self.zero2.update({k: torch.zeros(7, 1, 32, device=dev) for k in ("za_src2.bias", "za_trg2.bias")})

    def grads(self, rows, noise, use_trg, b_global=None, enc_coef=1.0):
        td = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(self.dev)
        s, a, s2, r = rows
        xenc = td(np.concatenate([s, s2], 1)); act = td(a); rew = td(r[..., 0])
        n6 = td(np.stack(noise[:6])); n7 = td(noise[6])
        self.ops.pretrain_grads(self.S, self.A, s.shape[1], use_trg, enc_coef, self.blob, self.blob_T, xenc, act, rew, self.grad,
                                self.loss, self.ws, noise6=n6, noise7=n7, b_global=b_global, precision=self.prec)
        torch.cuda.synchronize()
        return self.loss.cpu().numpy().copy()

    def apply(self, use_trg):
        self.t_main += 1; self.t_za[use_trg] += 1
        self.ops.pretrain_adam(self.S, self.A, use_trg, self.blob, self.blob_T, self.grad, self.m, self.v, self.t_main,
                               self.t_za[use_trg], self.lr, precision=self.prec)

    def unpack(self, blob, full_za2=None):
        into = {k: v.clone() for k, v in (full_za2 or self.zero2).items()}
        return self.packing.unpack_pretrain(blob, self.S, self.A, into=into)


@pytest.mark.parametrize("tag", ["walker", "pen"])
def test_pretrain_steps_vs_reference_golden(tag, dev, mfma):
    from test_hip_train import params_close
    g = gu.load(f"g12_pretrain_{tag}")
    S, A, b, seed = int(g["S"]), int(g["A"]), int(g["b"]), int(g["seed"])
    p = gu.dyn_params_for(g)
    tr = Trainer(p, S, A, b, dev, lr=float(g["lr"]), prec=mfma)
    full2 = {k: torch.from_numpy(p[k]).to(dev) for k in ("za_src2.weight", "za_trg2.weight", "za_src2.bias", "za_trg2.bias")}
    rng = gu.gi.noise_stream(int(g["noise_seed"]))
    for step, use_trg in enumerate((False, True, False, True)):
        rows = gu.gi.pretrain_batch(3000 + 10 * seed + step, b, S, A)
        losses = tr.grads(rows, noise7(rng, b, S), use_trg)
        close(losses, g[f"s{step}_losses"], rtol=2e-5, atol=1e-6)
        has = [str(x) for x in g[f"s{step}_has_grad"]]
        skip = "za_trg" if not use_trg else "za_src"
        assert not any(k.startswith(skip) for k in has)                 # the other domain's action encoder: .grad is None
        got = tr.unpack(tr.grad)
        scale = {}
        for k in has:
            sn = SUBNET[k[:2]]
            scale[sn] = max(scale.get(sn, 0.0), float(np.abs(g[f"s{step}_g::{k}"]).max()))
        for k in has:
            want = g[f"s{step}_g::{k}"]
            close(gu.sub101(got[k].cpu().numpy()), want, rtol=1e-5, atol=1e-5 * scale[SUBNET[k[:2]]])
            s64 = g[f"s{step}_gsum::{k}"]                               # whole-tensor pins: sum and sum of squares
            gk = got[k].double()
            close(float((gk * gk).sum()), s64[1], rtol=1e-4, atol=1e-30)
        tr.apply(use_trg)
        cur = tr.unpack(tr.blob, full2)
        for k in cur:
            params_close(gu.sub101(cur[k].cpu().numpy()), g[f"s{step}_p::{k}"], tr.lr)
    want_t = {x.split("=")[0]: int(x.split("=")[1]) for x in g["adam_steps"]}
    assert want_t["zs1.weight"] == tr.t_main and want_t["za_src1.weight"] == tr.t_za[False] and want_t["za_trg2.bias"] == tr.t_za[True]


def test_pretrain_no_vae_vs_reference_golden(dev, mfma):
    """config no_vae = 1 against the reference's own run with the flag (fixture g12_pretrain_walker_novae): the step with
    encoder_loss weighted by 0 -- gradients, post-Adam parameters; and the mirror's learn() reports the reference's five
    numbers (total, total again -- its in-place add aliases transition_loss --, 0, 0, 0)."""
    from test_hip_train import params_close
    from test_oracle_golden import novae_noise
    g = gu.load("g12_pretrain_walker_novae")
    S, A, b, seed = int(g["S"]), int(g["A"]), int(g["b"]), int(g["seed"])
    p = gu.dyn_params_for(g)
    tr = Trainer(p, S, A, b, dev, lr=float(g["lr"]), prec=mfma)
    dyn, m = _mirror_dynamics(p, S, A, dev, dict(no_vae=1))
    full2 = {k: torch.from_numpy(p[k]).to(dev) for k in ("za_src2.weight", "za_trg2.weight", "za_src2.bias", "za_trg2.bias")}
    rng, rng2 = gu.gi.noise_stream(int(g["noise_seed"])), gu.gi.noise_stream(int(g["noise_seed"]))
    td = lambda x: torch.from_numpy(x).to(dev)

    def noise(bb):
        nz = novae_noise(rng2, bb, S)
        return td(np.stack(nz[:6])), td(nz[6])

    dyn.train_noise_fn = noise
    for step, use_trg in enumerate((False, True, False, True)):
        rows = gu.gi.pretrain_batch(3000 + 10 * seed + step, b, S, A)
        losses = tr.grads(rows, novae_noise(rng, b, S), use_trg, enc_coef=0.0)
        want = g[f"s{step}_losses"]
        close(losses[0], want[0], rtol=2e-5, atol=1e-6)
        has = [str(x) for x in g[f"s{step}_has_grad"]]
        got = tr.unpack(tr.grad)
        scale = {}
        for k in has:
            sn = SUBNET[k[:2]]
            scale[sn] = max(scale.get(sn, 0.0), float(np.abs(g[f"s{step}_g::{k}"]).max()))
        for k in has:
            close(gu.sub101(got[k].cpu().numpy()), g[f"s{step}_g::{k}"], rtol=1e-5, atol=1e-5 * scale[SUBNET[k[:2]]])
        tr.apply(use_trg)
        cur = tr.unpack(tr.blob, full2)
        for k in cur:
            params_close(gu.sub101(cur[k].cpu().numpy()), g[f"s{step}_p::{k}"], tr.lr)
        stats = dyn.learn(use_trg, *[torch.from_numpy(x) for x in rows], b, 0.01)
        close(np.array(stats), want, rtol=2e-5, atol=1e-6)
    sd = m.state_dict()
    for k in cur:
        params_close(gu.sub101(sd[k].cpu().numpy()), g[f"s3_p::{k}"], tr.lr)


def test_pretrain_train_together_vs_reference_golden(dev, mfma):
    """config train_together = 1, step level (fixture g12_together_walker: two learn_src_trg calls of the reference, a
    24-row source batch + a 17-row target batch each): the mirror's learn_src_trg reports the reference's numbers and leaves
    its parameters."""
    from test_hip_train import params_close
    g = gu.load("g12_together_walker")
    S, A, bs, bt = int(g["S"]), int(g["A"]), int(g["bs"]), int(g["bt"])
    p = gu.dyn_params_for(g)
    dyn, m = _mirror_dynamics(p, S, A, dev, dict(train_together=1))
    rng = gu.gi.noise_stream(int(g["noise_seed"]))
    td = lambda x: torch.from_numpy(x).to(dev)

    def noise(b):
        nz = noise7(rng, b, S)
        return td(np.stack(nz[:6])), td(nz[6])

    dyn.train_noise_fn = noise
    for step in range(2):
        src = gu.gi.pretrain_batch(5000 + 10 * step, bs, S, A); trg = gu.gi.pretrain_batch(5001 + 10 * step, bt, S, A)
        stats = dyn.learn_src_trg(False, *[torch.from_numpy(x) for x in src], *[torch.from_numpy(x) for x in trg], bs, 0.01)
        want = g[f"s{step}_stats"]
        assert np.isnan(stats[3]) and np.isnan(want[3])
        close(np.array(stats)[[0, 1, 2, 4]], want[[0, 1, 2, 4]], rtol=2e-5, atol=1e-6)
        sd = m.state_dict()
        for k in g:
            if k.startswith(f"s{step}_p::"):
                params_close(gu.sub101(sd[k.split("::")[1]].cpu().numpy()), g[k], 1e-3)
    st = m.train_state()
    assert st["t_main"] == 2 and st["t_za"] == {False: 2, True: 2}


def test_pretrain_inverse_sep_reward_loss_vs_reference_golden(dev, mfma):
    """config inverse_sep_reward_loss = 1, step level (fixture g12_sepreward_walker: the reference's learn(src), learn(trg),
    learn_sep_reward, learn(trg)): the mirror's numbers and parameters; the reward head's Adam count ends at 1."""
    from test_hip_train import params_close
    from test_oracle_golden import sep_noise_learn, sep_noise_reward
    g = gu.load("g12_sepreward_walker")
    S, A, bs, bt = int(g["S"]), int(g["A"]), int(g["bs"]), int(g["bt"])
    p = gu.dyn_params_for(g)
    dyn, m = _mirror_dynamics(p, S, A, dev, dict(inverse_sep_reward_loss=1))
    rng = gu.gi.noise_stream(int(g["noise_seed"]))
    td = lambda x: torch.from_numpy(x).to(dev)
    kind_now = ["src"]

    def noise(b):
        nz = (sep_noise_reward if kind_now[0] == "sep" else sep_noise_learn)(rng, b, S)
        return td(np.stack(nz[:6])), td(nz[6])

    dyn.train_noise_fn = noise
    for step, kind in enumerate(("src", "trg", "sep", "trg")):
        kind_now[0] = kind
        if kind == "sep":
            src = gu.gi.pretrain_batch(6000, bs, S, A); trg = gu.gi.pretrain_batch(6001, bt, S, A)
            got = [dyn.learn_sep_reward(*[torch.from_numpy(x) for x in src], *[torch.from_numpy(x) for x in trg], bs)]
        else:
            rows = gu.gi.pretrain_batch(6100 + step, bs, S, A)
            got = dyn.learn(kind == "trg", *[torch.from_numpy(x) for x in rows], bs, 0.01)
        close(np.array(got), g[f"s{step}_losses"], rtol=2e-5, atol=1e-6)
        sd = m.state_dict()
        for k in g:
            if k.startswith(f"s{step}_p::"):
                params_close(gu.sub101(sd[k.split("::")[1]].cpu().numpy()), g[k], 1e-3)
    st = m.train_state()
    assert st["t_main"] == 4 and st["t_rw"] == 1 and st["t_za"] == {False: 2, True: 3}


def test_mirror_dynamics_train_sep_reward_vs_reference_golden(dev, mfma):
    """MOBODYEnsembleDynamics.train with inverse_sep_reward_loss = 1 end to end (fixture g13_dyn_train_sepreward: 32 optimizer
    steps, per epoch 4 + 3 x 3 learn() steps and 3 learn_sep_reward steps): counts, the four validate() results, elites."""
    from test_oracle_golden import sep_noise_learn, sep_noise_reward
    g = gu.load("g13_dyn_train_sepreward")
    S, A, bs = int(g["S"]), int(g["A"]), int(g["bs"])
    p = gu.dyn_params_for(g)
    dyn, m = _mirror_dynamics(p, S, A, dev, dict(inverse_sep_reward_loss=1))
    rng = gu.gi.noise_stream(int(g["noise_seed"]))
    td = lambda x: torch.from_numpy(x).to(dev)
    in_sep = [False]
    o_sep = dyn._learn_sep_reward_indexed

    def sep_indexed(*a, **k):
        in_sep[0] = True
        try:
            return o_sep(*a, **k)
        finally:
            in_sep[0] = False

    dyn._learn_sep_reward_indexed = sep_indexed
    draws = [0]

    def noise(b):
        nz = (sep_noise_reward if in_sep[0] else sep_noise_learn)(rng, b, S)
        draws[0] += 2 if in_sep[0] else 5
        return td(np.stack(nz[:6])), td(nz[6])

    dyn.train_noise_fn = noise
    src = gu.gi.batch(901, int(g["n_src"]), S, A); trg = gu.gi.batch(902, int(g["n_trg"]), S, A)
    torch.manual_seed(int(g["rng_seed"])); np.random.seed(int(g["rng_seed"]))
    dyn.train(tuple(torch.from_numpy(x) for x in src), tuple(torch.from_numpy(x) for x in trg), max_epochs=2, batch_size=bs)
    assert dyn.total_steps == int(g["total_steps"]) and draws[0] == int(g["n_noise"])
    want = g["validate"]
    got = []
    for h in dyn.history:
        got += [h["src_val"], h["trg_val"]]
    close(np.array(got), want[:, 0], rtol=1e-4, atol=1e-8)
    assert sorted(int(x) for x in m.elites.tolist()) == sorted(int(x) for x in g["elites"])


def test_mirror_dynamics_train_augmented_vs_reference_golden(dev, mfma):
    """config train_with_src_threshold != 1 (data_augmentation + train(), fixture g13_dyn_train_augment from the reference's own
    run): checked FROM THE TRAINED CLASSIFIER ON -- its 8 000 chained noisy steps are not a parity target (one step is, G9), so
    the fixture's classifier is loaded in place of training, exactly as the fixture's second pass did.  The probabilities of
    every source row (the double softmax), the selected rows, the enlarged target training set's effect on train(): step
    count, validate() results, elites."""
    from mobody_amd.algo.utils import ReplayBuffer
    g = gu.load("g13_dyn_train_augment")
    S, A, bs, thr = int(g["S"]), int(g["A"]), int(g["bs"]), float(g["threshold"])
    p = gu.dyn_params_for(g)
    dyn, m = _mirror_dynamics(p, S, A, dev, dict(train_with_src_threshold=thr))
    rng = gu.gi.noise_stream(int(g["noise_seed"]))
    td = lambda x: torch.from_numpy(x).to(dev)

    def noise(b):
        nz = noise7(rng, b, S)
        return td(np.stack(nz[:6])), td(nz[6])

    dyn.train_noise_fn = noise
    src = gu.gi.batch(901, int(g["n_src"]), S, A); trg = gu.gi.batch(902, int(g["n_trg"]), S, A)
    trg[0][:, 2] += float(g["trg_shift"]); trg[2][:, 2] += float(g["trg_shift"])

    def rb_of(rows):
        rb = ReplayBuffer(S, A, dev, max_size=len(rows[0]))
        rb.add_batch(dict(obss=torch.from_numpy(rows[0]), next_obss=torch.from_numpy(rows[2]), actions=torch.from_numpy(rows[1]),
                          rewards=torch.from_numpy(rows[3]), terminals=torch.from_numpy(rows[4])))
        return rb

    calls = [0]

    def upd(*a, **k):                                      # the 8 000 calls: load the reference-trained weights once, train nothing
        if calls[0] == 0:
            dyn.classifier.load_state_dict({k_[5:]: td(np.ascontiguousarray(g[k_])) for k_ in g if k_.startswith("cls::")})
        calls[0] += 1
        return None, None

    dyn.update_classifier = upd
    torch.manual_seed(int(g["rng_seed"])); np.random.seed(int(g["rng_seed"]))
    dyn.train(tuple(torch.from_numpy(x) for x in src[:4]), tuple(torch.from_numpy(x) for x in trg[:4]), max_epochs=1, batch_size=bs,
              buffer=(rb_of(src), rb_of(trg)))
    assert calls[0] == 8000
    close(dyn.augment_probs, g["probs"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(dyn.augment_include.cpu().numpy(), g["include"]) and dyn.src_replay_buffer_sim_trg.size == int(g["n_added"])
    assert torch.equal(dyn.src_replay_buffer_sim_trg.sample_all()[0].cpu(), torch.from_numpy(g["sim_state"]))
    assert dyn.total_steps == int(g["total_steps"]) and dyn._train_calls == int(g["n_noise"]) // 7
    got = []
    for h in dyn.history:
        got += [h["src_val"], h["trg_val"]]
    close(np.array(got), g["validate"][:, 0], rtol=1e-4, atol=1e-8)
    assert sorted(int(x) for x in m.elites.tolist()) == sorted(int(x) for x in g["elites"])


def test_data_augmentation_trains_its_classifier(dev):
    """The un-patched path: 8 000 classifier steps on the device, then the selection -- no parity target, only that it runs, that
    the classifier has learnt the (shifted) domains apart and that the threshold cuts the source rows."""
    from mobody_amd.algo.utils import ReplayBuffer
    S, A = 17, 6
    dyn, m = _mirror_dynamics(gu.gi.dyn_params(5, S, A), S, A, dev, dict(train_with_src_threshold=0.45))
    src = gu.gi.batch(901, 400, S, A); trg = gu.gi.batch(902, 200, S, A)
    trg[0][:, 2] += 2.0; trg[2][:, 2] += 2.0

    def rb_of(rows):
        rb = ReplayBuffer(S, A, dev, max_size=len(rows[0]), rng="device", seed=3)
        rb.add_batch(dict(obss=torch.from_numpy(rows[0]), next_obss=torch.from_numpy(rows[2]), actions=torch.from_numpy(rows[1]),
                          rewards=torch.from_numpy(rows[3]), terminals=torch.from_numpy(rows[4])))
        return rb

    dyn.data_augmentation((rb_of(src), rb_of(trg)))
    pr = dyn.augment_probs
    assert torch.isfinite(pr).all() and 0.26 < float(pr.min()) and float(pr.max()) < 0.74          # softmax of a probability pair
    s, a, s2 = (torch.from_numpy(x).to(dev) for x in (trg[0], trg[1], trg[2]))
    z, _ = dyn.classifier.logits(s.contiguous(), a.contiguous(), s2.contiguous(), False)
    pt = torch.softmax(torch.softmax(z[0], -1), -1)[:, 1]
    assert float(pt.mean()) > float(pr.mean()) + 0.1                                               # target rows look like target rows
    assert dyn.src_replay_buffer_sim_trg.size == int((pr > 0.45).sum()) < 400


def test_mirror_dynamics_train_together_vs_reference_golden(dev, mfma):
    """MOBODYEnsembleDynamics.train with train_together = 1 end to end (fixture g13_dyn_train_together: the reference's own
    run, 14 optimizer steps -- per epoch four learn() steps on the source rows, then three joint steps -- no reshuffle of the
    bootstrap matrices): step and noise-draw counts, the four validate() results, elites."""
    g = gu.load("g13_dyn_train_together")
    S, A, bs = int(g["S"]), int(g["A"]), int(g["bs"])
    p = gu.dyn_params_for(g)
    dyn, m = _mirror_dynamics(p, S, A, dev, dict(train_together=1))
    rng = gu.gi.noise_stream(int(g["noise_seed"]))
    td = lambda x: torch.from_numpy(x).to(dev)

    def noise(b):
        nz = noise7(rng, b, S)
        return td(np.stack(nz[:6])), td(nz[6])

    dyn.train_noise_fn = noise
    src = gu.gi.batch(901, int(g["n_src"]), S, A); trg = gu.gi.batch(902, int(g["n_trg"]), S, A)
    torch.manual_seed(int(g["rng_seed"])); np.random.seed(int(g["rng_seed"]))
    dyn.train(tuple(torch.from_numpy(x) for x in src), tuple(torch.from_numpy(x) for x in trg), max_epochs=2, batch_size=bs)
    assert dyn.total_steps == int(g["total_steps"]) and dyn._train_calls == int(g["n_noise"]) // 7
    want = g["validate"]
    got = []
    for h in dyn.history:
        got += [h["src_val"], h["trg_val"]]
    close(np.array(got), want[:, 0], rtol=1e-4, atol=1e-8)
    assert sorted(int(x) for x in m.elites.tolist()) == sorted(int(x) for x in g["elites"])


@pytest.mark.parametrize("S,A,b", [(17, 6, 1), (17, 6, 33), (17, 6, 256), (111, 8, 40), (45, 24, 65)])
def test_pretrain_grads_vs_oracle_shapes(S, A, b, dev, mfma):
    """Ragged / full batches and the ant / pen shapes against the oracle's autograd (same noise), source and target step."""
    p = gu.gi.dyn_params(5, S, A)
    tr = Trainer(p, S, A, b, dev, prec=mfma)
    rng = np.random.default_rng(b)
    for use_trg in (False, True):
        rows = gu.gi.pretrain_batch(77 + b, b, S, A)
        nz = noise7(rng, b, S)
        st = O.DynTrainState(p)
        want = O.dyn_learn_step(st, *rows, nz, use_trg, apply=False)
        tr.grad.zero_()
        losses = tr.grads(rows, nz, use_trg)
        close(losses, np.array(want["losses"]), rtol=2e-5, atol=1e-6)
        got = tr.unpack(tr.grad)
        scale = {}
        for k, v in want["grads"].items():
            if v is not None:
                scale[SUBNET[k[:2]]] = max(scale.get(SUBNET[k[:2]], 0.0), float(v.abs().max()))
        for k, v in want["grads"].items():
            if v is None:
                assert float(got[k].abs().max()) == 0.0, k             # untouched region stays zero
                continue
            close(got[k], v, rtol=1e-5, atol=1e-5 * scale[SUBNET[k[:2]]])


def test_pretrain_data_parallel_shards_sum_to_full_batch(dev, mfma):
    """Two ranks' shares (b rows each, b_global = 2b) add up to the gradient and losses of the 2b-row batch whenever the
    ensemble-coupled term is row-local (it is: the std runs over members, not rows)."""
    S, A, b = 17, 6, 48
    p = gu.gi.dyn_params(9, S, A)
    rows = gu.gi.pretrain_batch(5, 2 * b, S, A)
    rng = np.random.default_rng(0)
    nz = noise7(rng, 2 * b, S)
    full = Trainer(p, S, A, 2 * b, dev, prec=mfma)
    lf = full.grads(rows, nz, True)
    acc, lsum = torch.zeros_like(full.grad), np.zeros(5)
    for h in (0, 1):
        sl = slice(h * b, (h + 1) * b)
        sh = Trainer(p, S, A, b, dev, prec=mfma)
        lsum += sh.grads(tuple(x[:, sl] for x in rows), [x[:, sl] for x in nz], True, b_global=2 * b)
        acc += sh.grad
    close(lsum, lf, rtol=1e-5, atol=1e-6)
    scale = float(full.grad.abs().max())
    close(acc, full.grad, rtol=1e-5, atol=2e-6 * scale)


def test_pretrain_gather_and_device_noise(dev, mfma):
    """Bootstrap gather == fancy indexing; with noise=None the kernels draw Philox streams 16..22 themselves and the
    result equals the explicit-noise call fed with the CPU twin of those streams."""
    from mobody_amd import ops
    S, A, b, n = 17, 6, 40, 300
    s, a, s2, r, _ = gu.gi.batch(3, n, S, A)
    td = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    idx = np.random.default_rng(1).integers(0, n, (7, 120)).astype(np.int32)
    xenc, act, rew = ops.pretrain_gather(td(s), td(a), td(s2), td(r), td(idx), 37, b)
    sel = idx[:, 37:37 + b]
    assert torch.equal(xenc[:, :b].cpu(), torch.from_numpy(s[sel])) and torch.equal(xenc[:, b:].cpu(), torch.from_numpy(s2[sel]))
    assert torch.equal(act.cpu(), torch.from_numpy(a[sel])) and torch.equal(rew.cpu(), torch.from_numpy(r[sel][..., 0]))
    p = gu.gi.dyn_params(5, S, A)
    tr = Trainer(p, S, A, b, dev, prec=mfma)
    ops.pretrain_grads(S, A, b, True, 1.0, tr.blob, tr.blob_T, xenc, act, rew, tr.grad, tr.loss, tr.ws, seed=11, call=4, precision=mfma)
    torch.cuda.synchronize()
    l_dev, g_dev = tr.loss.cpu().numpy().copy(), tr.grad.clone()
    nz = [O.rng_normal(11, 16 + k, 4, 7 * b * 16).reshape(7, b, 16) for k in range(6)] + \
         [O.rng_normal(11, 22, 4, 7 * b * S).reshape(7, b, S)]
    tr.grad.zero_()
    l_exp = tr.grads((s[sel], a[sel], s2[sel], r[sel]), nz, True)
    close(l_dev, l_exp, rtol=2e-5, atol=1e-6)
    close(g_dev, tr.grad, rtol=1e-4, atol=2e-5 * float(tr.grad.abs().max()))


def test_dyn_validate_vs_oracle(dev):
    """validate() (mobody_dynamics.py:1113-1140): per-member transition and reward MSE on a holdout set."""
    from mobody_amd import ops, packing
    S, A, B = 17, 6, 137
    p = gu.gi.dyn_params(5, S, A)
    s, a, s2, r, _ = gu.gi.batch(8, B, S, A)
    blob = packing.pack_dynamics(p, S, A, dev)
    td = lambda x: torch.from_numpy(x).to(dev)
    for use_trg in (True, False):
        out = ops.dyn_validate(blob, S, A, td(s), td(a), td(s2), td(r), use_trg).cpu().numpy()
        pt = O.to_torch(p)
        with torch.no_grad():
            mean, _, _ = O.dyn_forward(pt, O.T(s), O.T(a), use_trg)
            tl = ((mean - O.T(s2)) ** 2).mean(dim=(1, 2))
            pr, _ = O.dyn_reward(pt, O.T(s).unsqueeze(0).repeat(7, 1, 1), O.T(a).unsqueeze(0).repeat(7, 1, 1), mean)
            rl = ((pr - O.T(r)) ** 2).mean(dim=(1, 2))
        close(out[:7], tl, rtol=1e-5, atol=1e-7)
        close(out[7:], rl, rtol=1e-5, atol=1e-7)


def _mirror_dynamics(p, S, A, dev, cfg_over=None):
    from mobody_amd.algo.dynamics.mobody_module import MOBODYModule
    from mobody_amd.algo.dynamics.mobody_dynamics import MOBODYEnsembleDynamics
    from mobody_amd.algo.mb_utils.terminal_funs import get_termination_fn
    cfg = gu.policy_cfg(S, A, **dict(dict(no_vae=0, inverse_sep_reward_loss=0, train_together=0, train_with_src_threshold=1,
                                          dynamics_lr=1e-3), **(cfg_over or {})))
    m = MOBODYModule(S, A, 256, 7, 5, device=dev, config=cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}, strict=False)
    return MOBODYEnsembleDynamics(cfg, m, None, None, get_termination_fn("walker2d-medium-v2"), penalty_coef=0.1), m


def test_mirror_dynamics_train_vs_reference_golden(dev, mfma):
    """MOBODYEnsembleDynamics.train end to end (fixture g13: the reference's own train() on 150 + 90 rows, max_epochs=2,
    batch 32): same holdout split / bootstrap / shuffle index streams (torch CPU generator + NumPy seeded as in the
    golden run), the reference's noise stream, 26 optimizer steps, 4 validate() calls, early-stopping bookkeeping,
    elites, load_save.  Validation losses 1e-4 relative after up to 26 chained steps (per-step parity is pinned at 1e-5
    by test_pretrain_steps_vs_reference_golden); final weights 5e-4 absolute (= 0.5 lr)."""
    g = gu.load("g13_dyn_train")
    S, A, bs = int(g["S"]), int(g["A"]), int(g["bs"])
    p = gu.dyn_params_for(g)
    dyn, m = _mirror_dynamics(p, S, A, dev)
    rng = gu.gi.noise_stream(int(g["noise_seed"]))
    td = lambda x: torch.from_numpy(x).to(dev)

    def noise(b):
        nz = noise7(rng, b, S)
        return td(np.stack(nz[:6])), td(nz[6])

    dyn.train_noise_fn = noise
    src = gu.gi.batch(901, int(g["n_src"]), S, A); trg = gu.gi.batch(902, int(g["n_trg"]), S, A)
    torch.manual_seed(int(g["rng_seed"])); np.random.seed(int(g["rng_seed"]))
    dyn.train(tuple(torch.from_numpy(x) for x in src), tuple(torch.from_numpy(x) for x in trg), max_epochs=2, batch_size=bs)
    assert dyn.total_steps == int(g["total_steps"]) and dyn._train_calls == int(g["n_noise"]) // 7
    want = g["validate"]                                   # [4 calls][transition | reward][7]
    got = []
    for h in dyn.history:
        got += [h["src_val"], h["trg_val"]]
    close(np.array(got), want[:, 0], rtol=1e-4, atol=1e-8)
    close(np.array(dyn.history[-1]["trg_reward_val"]), want[-1, 1], rtol=1e-4, atol=1e-8)
    # elites: same members; order may only differ between members whose holdout losses are within 1e-4 of each other
    el, wel = [int(x) for x in m.elites.tolist()], [int(x) for x in g["elites"]]
    assert sorted(el) == sorted(wel)
    fin = want[-1, 0]
    for a_, b_ in zip(el, wel):
        assert a_ == b_ or abs(fin[a_] - fin[b_]) <= 1e-4 * abs(fin[b_]), (el, wel)
    sd = m.state_dict()
    for k in g:
        if k.startswith("sd::"):
            d = np.abs(gu.sub101(sd[k[4:]].cpu().numpy()).astype(np.float64) - g[k])
            assert d.max() <= 5e-4, (k, d.max())
            assert (d <= 1e-5 + 1e-4 * np.abs(g[k])).mean() >= 0.98, (k, (d <= 1e-5 + 1e-4 * np.abs(g[k])).mean())
    # weight == saved_weight after load_save, and the inference path runs on the trained model
    assert torch.equal(sd["zs1.weight"], sd["zs1.saved_weight"])
    r = dyn.step_device(td(src[0][:8]), td(src[1][:8]))
    assert torch.isfinite(r["next_obs"]).all()


def test_mirror_learn_public_signature(dev, mfma):
    """learn(use_trg, obss[7,n,S], actions, next_obss, rewards, batch_size, logvar_loss_coef) -- the reference's own
    call shape (:594) -- equals the C-ABI driver fed with the same rows and noise."""
    S, A, b = 17, 6, 24
    p = gu.gi.dyn_params(5, S, A)
    dyn, m = _mirror_dynamics(p, S, A, dev)
    rows = gu.gi.pretrain_batch(5, 2 * b + 7, S, A)
    rng1, rng2 = np.random.default_rng(3), np.random.default_rng(3)
    td = lambda x: torch.from_numpy(x).to(dev)

    def noise(bb):
        nz = noise7(rng1, bb, S)
        return td(np.stack(nz[:6])), td(nz[6])

    dyn.train_noise_fn = noise
    stats = dyn.learn(True, *[torch.from_numpy(x) for x in rows], b, 0.01)
    tr = Trainer(p, S, A, b, dev, prec=mfma)
    ls = []
    for k in range(3):
        sl = slice(k * b, min((k + 1) * b, 2 * b + 7))
        bb = sl.stop - sl.start
        if bb != tr.b:
            tr = _retarget(tr, bb)
        ls.append(tr.grads(tuple(x[:, sl] for x in rows), noise7(rng2, bb, S), True))
        tr.apply(True)
    close(np.array(stats), np.mean(ls, 0), rtol=1e-5, atol=1e-6)
    got = m.state_dict()
    full2 = {k: torch.from_numpy(p[k]).to(dev) for k in ("za_src2.weight", "za_trg2.weight", "za_src2.bias", "za_trg2.bias")}
    for k, v in tr.unpack(tr.blob, full2).items():
        assert torch.equal(got[k], v), k


def _retarget(tr, b):
    """Same training state, workspace for another batch size."""
    tr.b = b
    tr.ws = tr.ops.pretrain_workspace(tr.S, tr.A, b, tr.dev)
    return tr


def test_pretrain_graph_replay_equals_eager_fused_steps(dev, mfma):
    """One pass of _learn_indexed (5 full batches + a ragged one, device-Philox noise): the captured-graph replay (batch
    offset, noise call id and Adam step counts advanced in device words) equals the eager fused steps; the only
    difference allowed is the device-side double pow of the Adam bias corrections (1 ulp of fp32), and both equal the
    unfused grads + Adam entry points."""
    from test_hip_train import params_close
    S, A, b = 17, 6, 32
    p = gu.gi.dyn_params(5, S, A)
    n = 400
    s, a, s2, r, _ = gu.gi.batch(4, n, S, A)
    td = lambda x: torch.from_numpy(x).to(dev)
    data = [td(s), td(a), td(s2), td(r)]
    idx = td(np.random.default_rng(2).integers(0, n, (7, 5 * b + 7)).astype(np.int32)).contiguous()
    out = {}
    for mode in ("graph", "eager"):
        dyn, m = _mirror_dynamics(p, S, A, dev, dict(train_graph=int(mode == "graph")))
        dyn.seed = 9
        st1 = dyn._learn_indexed(True, data, idx, b)
        st2 = dyn._learn_indexed(False, data, idx, b)
        assert (len(dyn._pre_graphs) == 2) == (mode == "graph")
        assert dyn.model.train_state()["t_main"] == 12 and dyn.model.train_state()["t_za"] == {False: 6, True: 6}
        out[mode] = (st1, st2, {k: v.cpu() for k, v in m.state_dict().items() if k in p})   # (saved_* / decoders: random init)
    for k in out["graph"][2]:
        np.testing.assert_allclose(out["graph"][2][k].numpy(), out["eager"][2][k].numpy(), rtol=2e-6, atol=1e-8, err_msg=k)
    close(np.array(out["graph"][0]), np.array(out["eager"][0]), rtol=1e-5, atol=1e-6)
    close(np.array(out["graph"][1]), np.array(out["eager"][1]), rtol=1e-5, atol=1e-6)
    # unfused entry points (what data-parallel ranks use) fed with the same Philox streams
    tr = Trainer(p, S, A, b, dev, prec=mfma)
    sel = idx.cpu().numpy()[:, :b]
    nz = [O.rng_normal(9 + 77, 16 + k, 1, 7 * b * 16).reshape(7, b, 16) for k in range(6)] + [O.rng_normal(9 + 77, 22, 1, 7 * b * S).reshape(7, b, S)]
    tr.grads((s[sel], a[sel], s2[sel], r[sel]), nz, True)
    tr.apply(True)
    dyn, m = _mirror_dynamics(p, S, A, dev, dict(train_graph=0))
    dyn.seed = 9
    dyn._learn_indexed(True, data, idx[:, :b].contiguous(), b)
    full2 = {k: torch.from_numpy(p[k]).to(dev) for k in ("za_src2.weight", "za_trg2.weight", "za_src2.bias", "za_trg2.bias")}
    got = m.state_dict()
    for k, v in tr.unpack(tr.blob, full2).items():
        # (the CPU twin of the Philox normals agrees with the device to ~2e-6, and Adam's first step is sign-like: an element
        #  whose gradient is ~1e-8 moves by anything up to lr -- the rule of the train-step tests: 99.5 % within 1e-5, all
        #  within 0.1 lr)
        params_close(got[k], v, 1e-3)


def _dp_train_worker(rank, world, port, tmp):
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")]
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = gu.load("g13_dyn_train")
    S, A, bs = int(g["S"]), int(g["A"]), int(g["bs"])
    dyn, m = _mirror_dynamics(gu.dyn_params_for(g), S, A, dev)
    rng = gu.gi.noise_stream(int(g["noise_seed"]))
    td = lambda x: torch.from_numpy(x).to(dev)

    def noise(b):                                       # called with the WHOLE batch's row count on every rank
        nz = noise7(rng, b, S)
        return td(np.stack(nz[:6])), td(nz[6])

    dyn.train_noise_fn = noise
    src = gu.gi.batch(901, int(g["n_src"]), S, A); trg = gu.gi.batch(902, int(g["n_trg"]), S, A)
    torch.manual_seed(int(g["rng_seed"]) + 100 * rank); np.random.seed(int(g["rng_seed"]) + 100 * rank)   # rank 0's streams rule
    if rank == 0:
        torch.manual_seed(int(g["rng_seed"])); np.random.seed(int(g["rng_seed"]))
    dyn.train(tuple(torch.from_numpy(x) for x in src), tuple(torch.from_numpy(x) for x in trg), max_epochs=2, batch_size=bs)
    got = []
    for h in dyn.history:
        got += [h["src_val"], h["trg_val"]]
    torch.save(dict(val=np.array(got), steps=dyn.total_steps, elites=[int(x) for x in m.elites.tolist()],
                    sd={k: v.cpu() for k, v in m.state_dict().items()}), os.path.join(tmp, f"dp_train_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_dynamics_train_vs_reference_golden(tmp_path):
    """The same g13 run on TWO ranks (each batch's rows split between them, one gradient all-reduce per step, rank 0's
    index streams broadcast, the explicit noise stream sliced per rank): replicas identical, and the run still
    reproduces the reference's single-process validation losses, step count and elites."""
    import os
    import torch.multiprocessing as mp
    port = 29800 + os.getpid() % 90
    mp.spawn(_dp_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"dp_train_r{r}.pt", weights_only=False) for r in (0, 1))
    g = gu.load("g13_dyn_train")
    assert r0["steps"] == r1["steps"] == int(g["total_steps"]) and r0["elites"] == r1["elites"]
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    close(r0["val"], g["validate"][:, 0], rtol=1e-4, atol=1e-8)
    assert sorted(r0["elites"]) == sorted(int(x) for x in g["elites"])
