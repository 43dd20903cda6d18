"""Deterministic synthetic inputs shared by the golden-vector generator and the tests.

Golden fixtures under tests/golden/*.npz hold only the *expected outputs* the
reference produced (plus small explicit inputs such as noise and indices); the
large inputs -- network weights -- are regenerated bit-identically from a seed by
the functions below (numpy PCG64, same numpy on the GPU box), and every fixture
stores a checksum of the weights it was generated with so drift is detected.

Weight names follow the reference's state_dict keys:
  dynamics  : `<layer>.weight [E,in,out]`, `<layer>.bias [E,1,out]`
              (algo/dynamics/mobody_module.py:383-389, layers :97-184)
  3-layer MLP: `network.{0,2,4}.{weight[out,in],bias[out]}`
              (algo/offline_offline/mobody.py:35-48)
"""
import numpy as np

E, H, L = 7, 256, 16


def _w(rng, shape, std):
    x = rng.standard_normal(shape).astype(np.float32)
    return np.clip(x, -2.0, 2.0) * np.float32(std)


def dyn_layer_dims(S, A):
    """(name, in, out) for every EnsembleLinear the hot path touches (mopo=0, latent_reward=0)."""
    return [
        ("zs1", S, H), ("zs2", H, H), ("zs3", H, 2 * L),
        ("za_src1", L + A, 32), ("za_src2", 32, 2 * L),
        ("za_trg1", L + A, 32), ("za_trg2", 32, 2 * L),
        ("transition1", L, H), ("transition2", H, H), ("transition3", H, S),
        ("reward_model1", 2 * S + A, H), ("reward_model2", H, H), ("reward_model3", H, 2),
    ]


def mopo_layer_dims(S, A):
    """The EnsembleLinears of the MOPO ablation (config['mopo'] = 1): the action-encoder slots hold a plain 3-layer MLP
    (mobody_module.py:114-118,133-137); the other layers exist but only the reward head is used."""
    return [(n, i, o) for n, i, o in dyn_layer_dims(S, A) if not n.startswith("za_")] + \
           [(pre + k, i, o) for pre in ("za_src", "za_trg") for k, i, o in (("1", S + A, H), ("2", H, H), ("3", H, S))]


def dyn_params(seed, S, A, scale=1.0, mopo=False):
    rng = np.random.default_rng(seed)
    p = {}
    for name, i, o in (mopo_layer_dims(S, A) if mopo else dyn_layer_dims(S, A)):
        p[name + ".weight"] = _w(rng, (E, i, o), scale / (2.0 * np.sqrt(i)))
        p[name + ".bias"] = _w(rng, (E, 1, o), 0.05)
    # spread the members apart so ensemble std / penalty are not tiny
    p["transition3.bias"] = p["transition3.bias"] + _w(rng, (E, 1, S), 0.05)
    if mopo:
        p["za_src3.bias"] = p["za_src3.bias"] + _w(rng, (E, 1, S), 0.05)
    return p


def mlp_params(seed, in_dim, out_dim, hidden=H):
    rng = np.random.default_rng(seed)
    p = {}
    dims = [(in_dim, hidden), (hidden, hidden), (hidden, out_dim)]
    for li, (i, o) in zip((0, 2, 4), dims):
        p[f"network.{li}.weight"] = _w(rng, (o, i), 1.0 / np.sqrt(i))
        p[f"network.{li}.bias"] = _w(rng, (o,), 0.05)
    return p


def walker_like_obs(rng, B, S):
    """States near the walker2d 'alive' box so the termination predicate is mostly false."""
    mu = np.zeros(S, np.float32)
    mu[0] = 1.25
    return (mu + 0.1 * rng.standard_normal((B, S))).astype(np.float32)


def batch(seed, N, S, A):
    rng = np.random.default_rng(seed)
    s = walker_like_obs(rng, N, S)
    a = rng.uniform(-1, 1, (N, A)).astype(np.float32)
    s2 = walker_like_obs(rng, N, S)
    r = rng.standard_normal((N, 1)).astype(np.float32)
    nd = (rng.uniform(0, 1, (N, 1)) > 0.05).astype(np.float32)
    return s, a, s2, r, nd


def checksum(params):
    """Order-independent float64 fingerprint of a parameter dict."""
    tot = 0.0
    for k in sorted(params):
        v = np.asarray(params[k], np.float64)
        tot += float(v.sum()) + 3.0 * float(np.abs(v).sum()) + 7.0 * float((v * v).sum())
    return tot


def pretrain_batch(seed, b, S, A):
    """Per-member bootstrap rows of one dynamics pre-training batch: obs[E,b,S], act[E,b,A], next_obs[E,b,S], rew[E,b,1]
    (mobody_dynamics.py:594-612 slices `train_obss[:, k*bs:(k+1)*bs]` of the bootstrapped arrays)."""
    rng = np.random.default_rng(seed)
    s = np.stack([walker_like_obs(rng, b, S) for _ in range(E)])
    a = rng.uniform(-1, 1, (E, b, A)).astype(np.float32)
    s2 = (s + 0.05 * rng.standard_normal((E, b, S))).astype(np.float32)
    r = rng.standard_normal((E, b, 1)).astype(np.float32)
    return s, a, s2, r


def noise_stream(seed):
    """The numpy generator make_golden.NoiseTap drew the reparameterisation / fake-next-state noise from:
    call `.standard_normal(shape).astype(float32)` in the reference's order."""
    return np.random.default_rng(seed)
