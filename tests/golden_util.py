"""Helpers shared by the oracle-vs-golden and HIP-vs-oracle tests."""
import os

import numpy as np

import gen_inputs as gi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def sub(x):
    """Same thinning rule as make_golden.sub."""
    f = np.asarray(x).reshape(-1)
    return f[::13].copy() if f.size > 4096 else f.copy()


def sub101(x):
    """Same thinning rule as make_golden.sub101 (pre-training fixtures)."""
    f = np.asarray(x).reshape(-1)
    return f[::101].copy() if f.size > 65536 else f[::17].copy() if f.size > 256 else f.copy()


def dyn_params_for(g):
    """Regenerate the dynamics weights a g234/g6 fixture was produced with and verify the checksum."""
    seed = int(g["dyn_seed"]) if "dyn_seed" in g else int(g["seed"])
    S, A = int(g["S"]), int(g["A"])
    p = gi.dyn_params(seed, S, A)
    ad = int(g["alive_dim"]) if "alive_dim" in g else 0
    p["transition3.bias"][:, 0, ad] += np.float32(float(g["alive_val"]))
    want = float(g["wsum"]) if "wsum" in g else float(g["wsum_dyn"])
    assert abs(gi.checksum(p) - want) <= 1e-9 * abs(want), "weight generator drifted from the fixture"
    return p


def policy_params(seed, S, A):
    pa = {"network." + k: v for k, v in gi.mlp_params(seed, S, A).items()}
    pq = {}
    for j, sd in ((1, seed + 1), (2, seed + 2)):
        pq.update({f"network{j}." + k: v for k, v in gi.mlp_params(sd, S + A, 1).items()})
    pv = {"network." + k: v for k, v in gi.mlp_params(seed + 3, S, 1).items()}
    return pa, pq, pv


def policy_cfg(S, A, **over):
    """MOBODY config of the reference's yaml + CLI defaults; lives in the package (mobody_amd.engine.default_config)."""
    from mobody_amd.engine import default_config
    return default_config(S, A, **over)


G7_VARIANTS = dict(default={}, noqw=dict(q_weighted=0), adv=dict(advantage=1), noscale=dict(scale_Q=0),
                   nofake=dict(fake_batch_scale=0), par=dict(penalty_type="par"), bc05=dict(bc_coef=0.05, trg_ratio=0.5))


def g7_batch(cfg, bs, S, A, src_reward_override=None):
    """Assemble the mixed batch the way mobody.py:399-400,516-529 does from the FixedRB presets."""
    src = gi.batch(501, 64, S, A); tar = gi.batch(502, 64, S, A); fake = gi.batch(503, 64, S, A)
    ns, nt = int(cfg["src_ratio"] * bs), int(cfg["trg_ratio"] * bs)
    nf = int(cfg["fake_batch_scale"] * bs)
    parts = [[x[:ns].copy() for x in src], [x[:nt].copy() for x in tar]]
    if src_reward_override is not None:
        parts[0][3] = src_reward_override
    if cfg["fake_batch_scale"] != 0:
        parts.append([x[:nf].copy() for x in fake])
    batch = tuple(np.concatenate([p[i] for p in parts], 0) for i in range(5))
    return batch, ns + nt


def g17_batch(k, cfg, bs, S, A):
    """Mixed batch of trajectory step k (0-based) as make_golden.py::RotatingRB serves it: window offset (k * 13) % (96 - n + 1)."""
    src = gi.batch(511, 96, S, A); tar = gi.batch(512, 96, S, A); fake = gi.batch(513, 96, S, A)
    ns, nt, nf = int(cfg["src_ratio"] * bs), int(cfg["trg_ratio"] * bs), int(cfg["fake_batch_scale"] * bs)
    win = lambda rows, n: [x[(k * 13) % (96 - n + 1):][:n].copy() for x in rows]
    parts = [win(src, ns), win(tar, nt), win(fake, nf)]
    return tuple(np.concatenate([p[i] for p in parts], 0) for i in range(5)), ns + nt


def mopo_params_for(g, tag):
    """The weights a g18 fixture was produced with (MOPO ablation), checksum verified."""
    S, A = int(g["S"]), int(g["A"])
    p = gi.dyn_params(int(g["seed"]), S, A, mopo=True)
    p["za_src3.bias"][:, 0, 0] += np.float32(-0.35 if tag == "walker" else -0.3)
    assert abs(gi.checksum(p) - float(g["wsum"])) <= 1e-9 * abs(float(g["wsum"])), "weight generator drifted from the fixture"
    return p


def dyn_kw(blob, S, A, mode):
    """planes= / precision= keyword arguments of ops.dyn_forward / dyn_step / rollout for MFMA mode `mode`."""
    from mobody_amd import ops
    if mode in (None, "f32", 0):
        return {}
    return dict(planes=ops.dyn_planes(blob, S, A, precision=mode), precision=mode)


def mlp_kw(blob, in_dim, out_dim, members, mode):
    """blob_T= / precision= keyword arguments of ops.mlp3_forward for MFMA mode `mode`."""
    from mobody_amd import ops
    if mode in (None, "f32", 0):
        return {}
    return dict(blob_T=ops.mlp_transpose(blob, in_dim, out_dim, members, precision=mode), precision=mode)
