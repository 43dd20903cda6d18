"""CPU oracle for the MOBODY hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this module; the product package never does (it fails loudly when the
HIP extension is missing instead of falling back to this).

What it is: a from-scratch fp32 restatement (PyTorch CPU ops + NumPy) of the
reference algorithm for the path named by BASELINE.json's north_star.  Every
function cites the reference file:line (relative to the reference repo) it
follows.  Parity of this oracle is pinned by `tests/golden/*.npz`, produced by
`tests/golden/make_golden.py` which imports the real reference in the build
container; `tests/test_oracle_golden.py` checks every
function here against those vectors.

Conventions: S=state_dim, A=action_dim, E=7 members, H=256, L=16 latent.
Dynamics parameters are a dict with the reference's state_dict key names
(`zs1.weight [E,in,out]`, `zs1.bias [E,1,out]`, ...); 3-layer MLPs use
`network.{0,2,4}.{weight,bias}` (nn.Linear layout [out,in]).
"""
import math

import numpy as np
import torch

E, H, L = 7, 256, 16

# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #


def T(x, dtype=torch.float32):
    if isinstance(x, torch.Tensor):
        return x.to(dtype)
    return torch.as_tensor(np.asarray(x), dtype=dtype)


def to_torch(params):
    return {k: T(v) if np.asarray(v).dtype.kind == "f" else torch.as_tensor(np.asarray(v)) for k, v in params.items()}


def swish(x):
    """algo/dynamics/mobody_module.py:9-15  x*sigmoid(x)."""
    return x * torch.sigmoid(x)


def ensemble_linear(x, W, b):
    """algo/dynamics/mobody_module.py:393-404.

    W:[E,in,out], b:[E,1,out]; x:[B,in] (shared by all members) or [E,B,in]
    -> [E,B,out].  (matmul broadcasting == the reference's two einsum forms.)
    """
    return torch.matmul(x, W) + b


def _el(p, name, x):
    return ensemble_linear(x, p[name + ".weight"], p[name + ".bias"])


def soft_clamp(x, lo, hi):
    """algo/dynamics/mobody_module.py:18-29."""
    x = hi - torch.nn.functional.softplus(hi - x)
    x = lo + torch.nn.functional.softplus(x - lo)
    return x


# --------------------------------------------------------------------------- #
# A2/A3: ensemble dynamics forward (inference mode: reparameterize -> mu)
# --------------------------------------------------------------------------- #


def dyn_encode_state(p, obs):
    """encode_state, mobody_module.py:217-225 (+ reparameterize :237-243 in inference)."""
    h = swish(_el(p, "zs1", obs))
    h = swish(_el(p, "zs2", h))
    z = _el(p, "zs3", h)
    return z[..., :L], z[..., L:]          # mu, logvar


def dyn_encode_action(p, zs, act, use_trg):
    """encode_trg_action / encode_src_action, mobody_module.py:245-271 (mopo=0)."""
    a = act.unsqueeze(0).expand(zs.shape[0], -1, -1)
    x = torch.cat([zs, a], -1)
    pre = "za_trg" if use_trg else "za_src"
    g = swish(_el(p, pre + "1", x))
    return _el(p, pre + "2", g)[..., :L]


def dyn_decode_transition(p, z):
    """encode_transition, mobody_module.py:287-293."""
    t = swish(_el(p, "transition1", z))
    t = swish(_el(p, "transition2", t))
    return _el(p, "transition3", t)


def dyn_forward_mopo(p, obs, act):
    """The MOPO ablation (config['mopo'] = 1): encode_state returns the state itself (:218-219), both action encoders are the
    3-layer MLP za_src1..3 on [s, a] (:245-256,264-266), encode_transition is the identity (:288-289), so
    forward_trg == forward_src == s + MLP_e([s, a])  (:315-330)."""
    x = torch.cat([obs, act], -1)
    g = swish(_el(p, "za_src1", x))
    g = swish(_el(p, "za_src2", g))
    return obs.unsqueeze(0) + _el(p, "za_src3", g), obs, obs


def is_mopo(p):
    return "za_src3.weight" in p


def dyn_forward(p, obs, act, use_trg=True):
    """forward_trg / forward_src, mobody_module.py:315-330 -> (mean[E,B,S], zs_mu, zs_logvar)."""
    if is_mopo(p):
        return dyn_forward_mopo(p, obs, act)
    zs, zs_logvar = dyn_encode_state(p, obs)
    za = dyn_encode_action(p, zs, act, use_trg)
    return dyn_decode_transition(p, zs + za), zs, zs_logvar


def dyn_reward(p, obs, act, next_obs):
    """encode_reward, mobody_module.py:295-302 -> (mu[E,B,1], logvar[E,B,1])."""
    x = torch.cat([obs, act, next_obs], -1)
    h = swish(_el(p, "reward_model1", x))
    h = swish(_el(p, "reward_model2", h))
    o = _el(p, "reward_model3", h)
    return o[..., :1], soft_clamp(o[..., 1:], -10.0, 0.5)


# --------------------------------------------------------------------------- #
# A6: termination predicates (host NumPy in the reference)
# --------------------------------------------------------------------------- #

TASK_IDS = {  # device-side enum used by the HIP kernels (include/mobody_hip.h)
    "never": 0, "halfcheetah": 1, "hopper": 2, "ant": 3, "walker2d": 4, "humanoid": 5, "pen": 6,
}


def resolve_task(task):
    """Substring dispatch in the reference's precedence order, terminal_funs.py:123-149."""
    order = [("halfcheetahvel", "never"), ("halfcheetah", "halfcheetah"), ("hopper", "hopper"),
             ("antangle", "ant"), ("ant", "ant"), ("walker2d", "walker2d"), ("point2denv", "never"),
             ("point2dwallenv", "never"), ("pendulum", "never"), ("humanoid", "humanoid"),
             ("pen", "pen"), ("door", "never")]
    for key, kind in order:
        if key in task:
            return kind
    raise TypeError("exceptions must derive from BaseException")  # `raise np.zeros` :149


def termination(task, obs, act, next_obs):
    """terminal_funs.py:10-121; returns bool [B,1]."""
    kind = resolve_task(task)
    n = np.asarray(next_obs)
    B = n.shape[0]
    if kind == "never":
        done = np.zeros(B, bool)
    elif kind == "halfcheetah":                                   # :10-16
        done = ~((n > -100).all(-1) & (n < 100).all(-1))
    elif kind == "hopper":                                        # :18-30 (abs of a bool == the bool)
        done = ~(np.isfinite(n).all(-1) & (n[:, 1:] < 100).all(-1) & (n[:, 0] > 0.7) & (np.abs(n[:, 1]) < 0.2))
    elif kind == "ant":                                           # :39-61
        done = ~(np.isfinite(n).all(-1) & (n[:, 0] >= 0.2) & (n[:, 0] <= 1.0))
    elif kind == "walker2d":                                      # :63-75
        done = ~((n > -100).all(-1) & (n < 100).all(-1) & (n[:, 0] > 0.8) & (n[:, 0] < 2.0)
                 & (n[:, 1] > -1.0) & (n[:, 1] < 1.0))
    elif kind == "humanoid":                                      # :98-104
        done = (n[:, 0] < 1.0) | (n[:, 0] > 2.0)
    elif kind == "pen":                                           # :106-113
        done = n[:, 26] < 0.075
    return done[:, None]


# --------------------------------------------------------------------------- #
# A4: one imagined transition
# --------------------------------------------------------------------------- #


def dyn_step(p, obs, act, eps, elite_idx, task, penalty_coef=0.0, use_penalty=True, use_trg=True):
    """MOBODYEnsembleDynamics.step, mobody_dynamics.py:193-265 (pairwise-diff uncertainty).

    eps:[E,B,S] unit normals (the reference draws torch.normal(0,std); explicit here),
    elite_idx:[B] member ids, or None to draw np.random.choice(elites,B) from the NumPy global
    stream exactly where the reference does (:224).
    Returns dict(next_obs[B,S], reward[B,1], terminal bool[B,1], penalty[B,1],
                 raw_reward[B,1], mean[E,B,S]).
    """
    obs, act, eps = T(obs), T(act), T(eps)
    if elite_idx is None:                     # random_elite_idxs, mobody_module.py:355-357 (NumPy global stream)
        elite_idx = np.random.choice(np.asarray(p["elites"]) if "elites" in p else np.arange(5), size=obs.shape[0])
    mean, _, _ = dyn_forward(p, obs, act, use_trg)                               # :211-214
    std = torch.std(mean, dim=0, keepdim=True)                                   # :218 unbiased
    samples = mean + eps * std                                                   # :220
    B = obs.shape[0]
    idx = torch.as_tensor(np.asarray(elite_idx), dtype=torch.long)
    next_obs = samples[idx, torch.arange(B)]                                     # :224-226
    r_mu, _ = dyn_reward(p, obs, act, next_obs)                                  # :235 (input shared by members)
    raw = r_mu.mean(0)                                                           # :236
    term = termination(task, obs.numpy(), act.numpy(), next_obs.numpy())        # :237
    m = mean[..., :-1]                                                           # :246 drops last state dim
    diff = m - m.mean(0)
    penalty = torch.amax(torch.norm(diff, dim=2), dim=0).reshape(B, 1)           # :247-256
    reward = raw
    if penalty_coef and use_penalty:                                             # :261-263
        reward = raw - penalty_coef * penalty
    return dict(next_obs=next_obs, reward=reward, terminal=term, penalty=penalty, raw_reward=raw, mean=mean)


# --------------------------------------------------------------------------- #
# A8: deterministic tanh actor, twin-Q, V   (nn.Linear layout)
# --------------------------------------------------------------------------- #


def mlp3(p, x, prefix=""):
    """MLPNetwork, mobody.py:35-48: Linear-ReLU-Linear-ReLU-Linear."""
    h = torch.relu(torch.nn.functional.linear(x, p[prefix + "network.0.weight"], p[prefix + "network.0.bias"]))
    h = torch.relu(torch.nn.functional.linear(h, p[prefix + "network.2.weight"], p[prefix + "network.2.bias"]))
    return torch.nn.functional.linear(h, p[prefix + "network.4.weight"], p[prefix + "network.4.bias"])


def actor(p, s, max_action=1.0):
    """Policy.forward, mobody.py:60-72; keys `network.network.{0,2,4}.*`."""
    return torch.tanh(mlp3(p, s, "network.")) * max_action


def twin_q(p, s, a):
    """DoubleQFunc.forward, mobody.py:74-83; keys `network{1,2}.network.{0,2,4}.*`."""
    x = torch.cat([s, a], 1)
    return mlp3(p, x, "network1."), mlp3(p, x, "network2.")


def value_fn(p, s):
    """ValueFunc.forward, mobody.py:50-57."""
    return mlp3(p, s, "network.")


# --------------------------------------------------------------------------- #
# A7: model rollout
# --------------------------------------------------------------------------- #


def rollout(actor_p, dyn_p, init_obs, H, eps_per_step, elite_per_step, task, cfg, use_trg=True,
            penalty_coef=0.0, draws=None):
    """MOBODY.rollout, mobody.py:596-657.

    eps_per_step[t]:[E,B_t,S], elite_per_step[t]:[B_t] follow the *compacted* batch
    of step t (terminated rows are dropped between steps, :635-639); alternatively `draws`
    is an iterator yielding (eps, elite_idx) once per executed step.
    Quirk Q1: `use_trg` lands in step()'s `use_penalty` slot (:614), the target
    model is always used.
    """
    if H == 0:
        return None, None
    obs = T(init_obs)
    out = {k: [] for k in ("obss", "next_obss", "actions", "rewards", "terminals", "penalty")}
    n_tr, rew_all = 0, []
    for t in range(H):
        act = actor(actor_p, obs, cfg["max_action"]).reshape(-1, cfg["action_dim"])          # :612
        eps_t, idx_t = next(draws) if draws is not None else (eps_per_step[t], elite_per_step[t])
        st = dyn_step(dyn_p, obs, act, eps_t, idx_t, task,
                      penalty_coef=penalty_coef, use_penalty=use_trg, use_trg=True)          # :614
        out["obss"].append(obs); out["next_obss"].append(st["next_obs"]); out["actions"].append(act)
        out["rewards"].append(st["reward"]); out["terminals"].append(T(st["terminal"].astype(np.float32)))
        out["penalty"].append(st["penalty"])
        n_tr += obs.shape[0]
        rew_all.append(st["reward"].numpy().ravel())
        alive = ~st["terminal"].ravel()
        if alive.sum() == 0:                                                                 # :636
            break
        obs = st["next_obs"][torch.as_tensor(alive)]                                         # :639
    res = {k: torch.cat(v, 0) for k, v in out.items()}
    if cfg["filter_bad_rollout"]:                                                            # :648-653
        keep = (res["penalty"] <= cfg["env_filter"]).squeeze(1)
        res = {k: v[keep] for k, v in res.items()}
    return res, dict(num_transitions=n_tr, reward_mean=float(np.concatenate(rew_all).mean()))


# --------------------------------------------------------------------------- #
# A10: ring-buffer append semantics
# --------------------------------------------------------------------------- #


def ring_append_plan(ptr, size, cap, M):
    """ReplayBuffer.add_batch, algo/utils.py:43-92.

    Returns ([(dst_start, src_start, length), ...], new_ptr, new_size) reproducing the
    single-wrap arithmetic (`used`, the `ptr == 0` re-entry at :82-91).
    """
    end = min(ptr + M, cap)
    used = end - ptr
    segs = [(ptr, 0, used)] if used > 0 else []
    new_ptr = end % cap
    new_size = min(size + used, cap)
    if new_ptr == 0:
        rest = M - used
        if rest > cap:
            raise RuntimeError("shape mismatch: batch overflows the ring twice")  # tensor shape error in the reference
        if rest > 0:
            segs.append((0, used, rest))
        new_ptr = rest
    return segs, new_ptr, new_size


class RingBuffer:
    """ReplayBuffer, algo/utils.py:13-148, reduced to what the path uses: SoA arrays, the single-wrap bulk append
    (`ring_append_plan`) and `sample` = rows at np.random.randint(0, size, n) (NumPy global stream, :128)."""

    FIELDS = ("state", "action", "next_state", "reward", "not_done")

    def __init__(self, S, A, cap):
        self.cap, self.ptr, self.size = int(cap), 0, 0
        self.state, self.action = np.zeros((cap, S), np.float32), np.zeros((cap, A), np.float32)
        self.next_state = np.zeros((cap, S), np.float32)
        self.reward, self.not_done = np.zeros((cap, 1), np.float32), np.zeros((cap, 1), np.float32)

    def add_batch(self, b):
        if b is None:                                                             # :44-45
            return
        rows = [np.asarray(b[k], np.float32).reshape(len(b["obss"]), -1) for k in ("obss", "actions", "next_obss", "rewards")]
        rows.append(1.0 - np.asarray(b["terminals"], np.float32).reshape(-1, 1))  # not_done = 1 - terminals (:73)
        segs, self.ptr, self.size = ring_append_plan(self.ptr, self.size, self.cap, len(rows[0]))
        for dst, src, n in segs:
            for f, r in zip(self.FIELDS, rows):
                getattr(self, f)[dst:dst + n] = r[src:src + n]

    def sample(self, n):
        ind = np.random.randint(0, self.size, size=n)
        return tuple(T(getattr(self, f)[ind]) for f in self.FIELDS)


def refresh(actor_p, dyn_p, src, tar, fake, cfg, task, draws, batch_size, sizes=(50000, 2000, 100), penalty_coef=0.0,
            classifier_update=None, cls_p=None):
    """The model-rollout refresh inside MOBODY.train, mobody.py:441-513, in the reference's order:
    sample(src, 50000) -> sample(tar, 2000) -> rollout(src states, src_rollout_length) -> add -> rollout(tar states,
    trg_rollout_length) -> add -> [use_src_sa_to_get_target_next_state: step(src s, src a), keep penalty < env_filter
    (STRICT, :466), add] -> [rollout_from_src: one classifier update unless dara (:480-481), sample(src, 50000),
    sample(tar, 100), rollout(cat, rollout_from_src_length, use_trg=False), rewards += penalty_coef * delta_r, add].
    `draws` yields (eps[E,B,S], elite_idx[B]) for every dynamics.step call in order."""
    draws = iter(draws)

    def roll(init, H, use_trg=True):
        return rollout(actor_p, dyn_p, init, H, None, None, task, cfg, use_trg=use_trg, penalty_coef=penalty_coef,
                       draws=draws)[0]

    s_init = src.sample(sizes[0])
    t_init = tar.sample(sizes[1])
    fake.add_batch(roll(s_init[0], cfg["src_rollout_length"]))
    fake.add_batch(roll(t_init[0], cfg["trg_rollout_length"]))
    if cfg["use_src_sa_to_get_target_next_state"]:
        eps, idx = next(draws)
        st = dyn_step(dyn_p, s_init[0], s_init[1], eps, idx, task, penalty_coef=penalty_coef)
        keep = (st["penalty"] < cfg["env_filter"]).squeeze(1)                      # strict '<', :466
        fake.add_batch(dict(obss=s_init[0][keep], next_obss=st["next_obs"][keep], actions=s_init[1][keep],
                            rewards=st["reward"][keep], terminals=T(st["terminal"].astype(np.float32))[keep]))
    if cfg["rollout_from_src"]:
        if cfg["penalty_type"] != "dara":
            classifier_update(src, tar, batch_size)
        s2 = src.sample(sizes[0]); t2 = tar.sample(sizes[2])
        tr = roll(torch.cat([s2[0], t2[0]], 0), cfg["rollout_from_src_length"], use_trg=False)
        with torch.no_grad():
            tr["rewards"] = tr["rewards"] + cfg["penalty_coef"] * dara_delta_r(cls_p, tr["obss"], tr["actions"], tr["next_obss"])
        fake.add_batch(tr)


# --------------------------------------------------------------------------- #
# Adam (torch.optim.Adam defaults), Polyak
# --------------------------------------------------------------------------- #


def adam_update(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor rule (mobody.py:127-131 uses defaults). In place; t is 1-based."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def polyak(target, online, tau):
    """update_target, mobody.py:183-187."""
    for k in target:
        target[k].copy_(tau * online[k] + (1.0 - tau) * target[k])


# --------------------------------------------------------------------------- #
# A11-A14: one MOBODY gradient step on a given mixed batch
# --------------------------------------------------------------------------- #


class TrainState:
    """Actor / twin-Q / target twin-Q / V parameters + Adam moments (all torch CPU fp32)."""

    def __init__(self, actor_p, q_p, v_p=None):
        self.actor = {k: T(v).clone() for k, v in actor_p.items()}
        self.q = {k: T(v).clone() for k, v in q_p.items()}
        self.q_targ = {k: v.clone() for k, v in self.q.items()}                    # deepcopy, mobody.py:116
        self.v = {k: T(v).clone() for k, v in (v_p or {}).items()}
        z = lambda d: {k: torch.zeros_like(v) for k, v in d.items()}
        self.m = dict(actor=z(self.actor), q=z(self.q), v=z(self.v))
        self.s = dict(actor=z(self.actor), q=z(self.q), v=z(self.v))
        self.t = dict(actor=0, q=0, v=0)


def _grads(loss, params):
    names = list(params)
    gs = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
    return {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(names, gs)}


def train_step(st, batch, n_true, cfg, apply=True):
    """MOBODY.train body after the minibatch is assembled, mobody.py:516-578.

    batch = (state[N,S], action[N,A], next_state[N,S], reward[N,1], not_done[N,1]) in the
    reference's concat order src|tar|fake (:525-529); the "true" BC batch is the first
    n_true rows (src|tar, :561-563).  Returns a dict of losses / grads / BC weights.
    cfg keys: gamma tau max_action critic_lr actor_lr weight bc_coef q_weighted advantage scale_Q.
    """
    s, a, s2, r, nd = [T(x) for x in batch]
    out = {}
    req = lambda d: {k: v.detach().clone().requires_grad_(True) for k, v in d.items()}

    if cfg.get("advantage", 0):                                   # :533-537, update_v_function :231-242
        vp = req(st.v)
        with torch.no_grad():
            qt1, qt2 = twin_q(st.q_targ, s, a)
            q_t = torch.min(qt1, qt2)
        adv = q_t - value_fn(vp, s)
        v_loss = torch.mean(torch.abs(0.7 - (adv < 0).float()) * adv ** 2)
        gv = _grads(v_loss, vp)
        out["v_loss"], out["v_grads"] = v_loss.detach(), gv
        if apply:
            st.t["v"] += 1
            for k in st.v:
                adam_update(st.v[k], gv[k], st.m["v"][k], st.s["v"][k], st.t["v"], cfg["critic_lr"])

    # ---- critic (update_q_functions :189-208 / _1 :210-229) ----
    qp = req(st.q)
    with torch.no_grad():
        if cfg.get("advantage", 0):
            q_next = value_fn(st.v, s2)
        else:
            a2 = actor(st.actor, s2, cfg["max_action"])
            t1, t2 = twin_q(st.q_targ, s2, a2)
            q_next = torch.min(t1, t2)
        y = r + nd * cfg["gamma"] * q_next
    q1, q2 = twin_q(qp, s, a)
    q_loss = torch.nn.functional.mse_loss(q1, y) + torch.nn.functional.mse_loss(q2, y)
    gq = _grads(q_loss, qp)
    out.update(q_loss=q_loss.detach(), q_grads=gq, td_target=y, q1=q1.detach(), q2=q2.detach())
    if apply:
        st.t["q"] += 1
        for k in st.q:
            adam_update(st.q[k], gq[k], st.m["q"][k], st.s["q"][k], st.t["q"], cfg["critic_lr"])
        polyak(st.q_targ, st.q, cfg["tau"])                                       # :552
    q_now = st.q if apply else {k: v.detach() for k, v in qp.items()}

    # ---- actor (update_policy :314-345 / update_policy_1 :278-310, bc_loss :246-276) ----
    ap = req(st.actor)
    pi = actor(ap, s, cfg["max_action"])
    b1, b2 = twin_q(q_now, s, pi)
    qv = torch.min(b1, b2)
    p_w = cfg["weight"] / qv.abs().mean().detach() if cfg.get("scale_Q", 1) else 1.0
    pi_loss = p_w * (-qv).mean()
    st_, at_ = s[:n_true], a[:n_true]
    pred = actor(ap, st_, cfg["max_action"])
    with torch.no_grad():
        c1, c2 = twin_q(q_now, st_, at_)
        qb = torch.min(c1, c2)
        if cfg.get("advantage", 0):
            adv = qb - value_fn(st.v, st_)
        else:
            adv = qb / qb.abs().mean()
        w = torch.exp(3 * adv).clamp(max=100.0)
    if not cfg.get("q_weighted", 1):
        w = torch.ones_like(w)
    bc = torch.mean(w * (pred - at_) ** 2)
    loss = pi_loss + cfg["bc_coef"] * bc
    ga = _grads(loss, ap)
    out.update(pi_loss=loss.detach(), bc_loss=bc.detach(), bc_w=w, pi=pi.detach(), q_pi=qv.detach(), actor_grads=ga)
    if apply:
        st.t["actor"] += 1
        for k in st.actor:
            adam_update(st.actor[k], ga[k], st.m["actor"][k], st.s["actor"][k], st.t["actor"], cfg["actor_lr"])
    return out


# --------------------------------------------------------------------------- #
# 8(f) row 1: dynamics pre-training (one optimizer step of MOBODYEnsembleDynamics.learn)
# --------------------------------------------------------------------------- #

TRAINED_LAYERS = ("zs1", "zs2", "zs3", "za_src1", "za_src2", "za_trg1", "za_trg2", "transition1", "transition2",
                  "transition3", "reward_model1", "reward_model2", "reward_model3")


def dyn_learn_losses(p, obs, act, next_obs, rew, noise, use_trg, encoder_loss_coef=1.0, transition_coef=1.0, reward_coef=1.0):
    """The loss of one `learn()` batch, mobody_dynamics.py:594-653 (no_vae=0, latent_reward=0, inverse_sep_reward_loss=0):

      encoder_loss (:300-330)   100 * [sum_e mean_{b,d}(dec(z1) - s)^2 + ... (dec(z2) - s')^2]
                                + 0.05 * KL(s) + 0.05 * KL(s')                       (get_kl_loss :332-335)
                                + sum_e mean_{b,16}((z3 + za(z3,a)) - z4)^2          (z4 = encode_state(s') under no_grad)
      transition_loss (:337-347) sum_e mean_{b,d}(forward(s,a) - s')^2               (state sample z5)
      reward_loss (:349-384)    [sum_e mean_b (r(s,a,fake) - r)^2 + sum_e mean_b (r(s,a,s') - r)^2] * (1 if trg else 0.01),
                                fake = mean6 + eps7 * std_e(mean6) with the gradient flowing through mean6 AND the std
      loss = transition + (5 if trg else 1) * encoder_loss_coef * encoder_loss + reward_loss          (:619-639)

    obs/next_obs [E,b,S], act [E,b,A], rew [E,b,1] are per-member bootstrap rows.  `noise` = the seven randn_like draws
    in the reference's order: z1(s), z2(s'), z3(s), z4(s'), z5(s), z6(s) as [E,b,16] and eps7 [E,b,S]
    (reparameterize, mobody_module.py:237-243: z = mu + eps * exp(0.5 * logvar) in training mode).
    Returns (loss, transition_loss, encoder_loss, recon_loss, kl_loss)."""
    s, a, s2, r = T(obs), T(act), T(next_obs), T(rew)
    n = [T(x) for x in noise]
    pre = "za_trg" if use_trg else "za_src"

    def enc(x, eps):                                                  # encode_state :217-225
        mu, lv = dyn_encode_state(p, x)
        return mu + eps * torch.exp(0.5 * lv), mu, lv

    def za(zs):                                                       # encode_*_action :245-271
        g = swish(_el(p, pre + "1", torch.cat([zs, a], -1)))
        return _el(p, pre + "2", g)[..., :L]

    kl = lambda mu, lv: 0.05 * (-0.5 * (1 + lv - mu.pow(2) - lv.exp()).mean(dim=(1, 2))).sum()
    z1, mu1, lv1 = enc(s, n[0])
    z2, mu2, lv2 = enc(s2, n[1])
    recon = ((dyn_decode_transition(p, z1) - s) ** 2).mean(dim=(1, 2)).sum() + \
            ((dyn_decode_transition(p, z2) - s2) ** 2).mean(dim=(1, 2)).sum()
    kl_loss = kl(mu1, lv1) + kl(mu2, lv2)
    z3, _, _ = enc(s, n[2])
    with torch.no_grad():
        z4, _, _ = enc(s2, n[3])
    enc_loss = 100 * recon + kl_loss + (((z3 + za(z3)) - z4) ** 2).mean(dim=(1, 2)).sum()
    z5, _, _ = enc(s, n[4])
    trans = ((dyn_decode_transition(p, z5 + za(z5)) - s2) ** 2).mean(dim=(1, 2)).sum()
    loss = transition_coef * trans + (5 if use_trg else 1) * encoder_loss_coef * enc_loss
    z6, _, _ = enc(s, n[5])
    mean6 = dyn_decode_transition(p, z6 + za(z6))
    fake = mean6 + n[6] * torch.std(mean6, dim=0, keepdim=True)
    rl = ((dyn_reward(p, s, a, fake)[0] - r) ** 2).mean(dim=(1, 2)).sum() + \
         ((dyn_reward(p, s, a, s2)[0] - r) ** 2).mean(dim=(1, 2)).sum()
    loss = loss + reward_coef * (rl if use_trg else 0.01 * rl)      # (transition_coef / reward_coef: 1, 1 in learn(); see dyn_learn_step_sep_reward)
    return loss, trans, enc_loss, recon, kl_loss


class DynTrainState:
    """Live dynamics parameters (the 13 trained EnsembleLinear layers) with torch.optim.Adam state PER PARAMETER:
    a parameter whose gradient is None in a step (za_trg* on source batches, za_src* on target ones) is skipped by
    Adam and keeps its own step count (train_mobody.py:801-804 hands every parameter to one Adam)."""

    def __init__(self, params, lr=1e-3):
        self.p = {k: T(v).clone() for k, v in params.items() if k.split(".")[0] in TRAINED_LAYERS and
                  k.split(".")[1] in ("weight", "bias")}
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.t = {k: 0 for k in self.p}
        self.lr = lr


def dyn_learn_step(st, obs, act, next_obs, rew, noise, use_trg, encoder_loss_coef=1.0, apply=True, with_reward=True):
    """zero_grad -> loss.backward -> Adam.step of one learn() batch (mobody_dynamics.py:641-643).
    with_reward=False: config inverse_sep_reward_loss = 1 (:637-641) -- reward_loss is neither evaluated nor added, so the reward
    head's parameters have no gradient (Adam skips them, their step counts stay).
    Returns dict(losses=(5 floats), grads={name: tensor or None})."""
    pr = {k: v.detach().clone().requires_grad_(True) for k, v in st.p.items()}
    losses = dyn_learn_losses(pr, obs, act, next_obs, rew, noise, use_trg, encoder_loss_coef, 1.0, 1.0 if with_reward else 0.0)
    names = list(pr)
    gs = torch.autograd.grad(losses[0], [pr[k] for k in names], allow_unused=True)
    grads = dict(zip(names, gs))
    if not with_reward:
        grads = {k: (None if k.startswith("reward_model") else g) for k, g in grads.items()}
    if apply:
        for k, g in grads.items():
            if g is None:
                continue
            st.t[k] += 1
            adam_update(st.p[k], g, st.m[k], st.v[k], st.t[k], st.lr)
    return dict(losses=tuple(float(x.detach()) for x in losses), grads=grads)


def dyn_learn_step_together(st, src_rows, trg_rows, noise_src, noise_trg, encoder_loss_coef=1.0, apply=True):
    """One optimizer step of learn_src_trg (config train_together = 1, mobody_dynamics.py:521-590): the loss of a SOURCE batch
    plus the loss of a TARGET batch whose encoder_loss is weighted 1 x encoder_loss_coef (not learn()'s 5 x, :571), one
    backward, one Adam step in which BOTH action encoders move.  Draw order: the source batch's seven, then the target's.
    Returns dict(losses=(total, trg transition, trg encoder, trg kl), grads)."""
    pr = {k: v.detach().clone().requires_grad_(True) for k, v in st.p.items()}
    ls = dyn_learn_losses(pr, *src_rows, noise_src, False, encoder_loss_coef)
    lt = dyn_learn_losses(pr, *trg_rows, noise_trg, True, encoder_loss_coef / 5.0)
    loss = ls[0] + lt[0]
    names = list(pr)
    gs = torch.autograd.grad(loss, [pr[k] for k in names], allow_unused=True)
    grads = dict(zip(names, gs))
    if apply:
        for k, g in grads.items():
            if g is None:
                continue
            st.t[k] += 1
            adam_update(st.p[k], g, st.m[k], st.v[k], st.t[k], st.lr)
    return dict(losses=(float(loss.detach()), float(lt[1].detach()), float(lt[2].detach()), float(lt[4].detach())), grads=grads)


def dyn_learn_step_sep_reward(st, src_rows, trg_rows, noise_src, noise_trg, apply=True):
    """One optimizer step of learn_sep_reward (config inverse_sep_reward_loss = 1, mobody_dynamics.py:482-519):
    loss = reward_loss(source batch) + reward_loss(target batch) -- no transition or encoder terms -- one backward, one Adam
    step; the gradient reaches the reward head and, through the fake next state, the encoder, both action encoders and the
    decoder.  Draws per domain: the forward's state sample (slot 5 of the seven) and the fake-next-state noise (slot 6)."""
    pr = {k: v.detach().clone().requires_grad_(True) for k, v in st.p.items()}
    ls = dyn_learn_losses(pr, *src_rows, noise_src, False, 0.0, 0.0, 1.0)
    lt = dyn_learn_losses(pr, *trg_rows, noise_trg, True, 0.0, 0.0, 1.0)
    loss = ls[0] + lt[0]
    names = list(pr)
    gs = torch.autograd.grad(loss, [pr[k] for k in names], allow_unused=True)
    grads = dict(zip(names, gs))
    if apply:
        for k, g in grads.items():
            if g is None:
                continue
            st.t[k] += 1
            adam_update(st.p[k], g, st.m[k], st.v[k], st.t[k], st.lr)
    return dict(losses=(float(loss.detach()),), grads=grads)


# --------------------------------------------------------------------------- #
# A15: DARA reward penalty with the reference's softmax quirks
# --------------------------------------------------------------------------- #


def classifier_probs(p, s, a, s2, noise_sas=None, noise_sa=None, std=1.0):
    """Classifier.forward, mobody.py:11-33; keys `{sas,sa}_classifier.network.*`.

    Heads output softmax *probabilities* (Q5/Q6); optional additive input noise.
    """
    sas = torch.cat([s, a, s2], -1)
    if noise_sas is not None:
        sas = sas + T(noise_sas) * std
    sa = torch.cat([s, a], -1)
    if noise_sa is not None:
        sa = sa + T(noise_sa) * std
    return (torch.softmax(mlp3(p, sas, "sas_classifier."), 1),
            torch.softmax(mlp3(p, sa, "sa_classifier."), 1))


def dara_delta_r(p, s, a, s2):
    """mobody.py:373-378: second softmax over the probabilities, log-ratio, clamp(-10,10)."""
    ps, pa = classifier_probs(p, T(s), T(a), T(s2))
    ls = torch.log(torch.softmax(ps, -1) + 1e-10)
    la = torch.log(torch.softmax(pa, -1) + 1e-10)
    d = ls[:, 1:] - la[:, 1:] - ls[:, :1] + la[:, :1]
    return d.clamp(-10, 10)


def classifier_loss(p, s, a, s2, label, noise_sas, noise_sa, std):
    """update_classifier, mobody.py:146-181: cross_entropy applied to probabilities (double softmax)."""
    ps, pa = classifier_probs(p, T(s), T(a), T(s2), noise_sas, noise_sa, std)
    lab = torch.as_tensor(np.asarray(label), dtype=torch.long)
    return torch.nn.functional.cross_entropy(ps, lab) + torch.nn.functional.cross_entropy(pa, lab)


# --------------------------------------------------------------------------- #
# Counter-based RNG twin of the device generator (throughput mode)
# --------------------------------------------------------------------------- #

_PH_M0, _PH_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PH_W0, _PH_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011), vectorised over uint32 arrays.  Device twin: csrc/rng.h."""
    c0, c1, c2, c3 = [np.asarray(c, np.uint32).copy() for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _PH_M0
            p1 = c2.astype(np.uint64) * _PH_M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_PH_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_PH_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _u01(x):
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24) + np.float32(2.0 ** -25)).astype(np.float32)


def rng_normal(seed, stream, call, n):
    """n unit normals: element i uses counter (i>>2, call, 0, 0), key (seed, stream), Box-Muller lane i&3."""
    i = np.arange(n, dtype=np.uint64)
    x = philox4x32((i >> np.uint64(2)).astype(np.uint32), np.uint32(call), np.uint32(0), np.uint32(0), seed, stream)
    lane = (i & np.uint64(3)).astype(np.int64)
    u1 = np.where(lane < 2, _u01(x[0]), _u01(x[2])).astype(np.float32)
    u2 = np.where(lane < 2, _u01(x[1]), _u01(x[3])).astype(np.float32)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    ang = (np.float32(2.0 * math.pi) * u2).astype(np.float32)
    return np.where(lane % 2 == 0, rad * np.cos(ang), rad * np.sin(ang)).astype(np.float32)


def rng_index(seed, stream, call, n, bound):
    """n integers in [0,bound): element i uses word i&3 of counter (i>>2, call, 0, 0); (x*bound)>>32."""
    i = np.arange(n, dtype=np.uint64)
    x = philox4x32((i >> np.uint64(2)).astype(np.uint32), np.uint32(call), np.uint32(0), np.uint32(0), seed, stream)
    lane = (i & np.uint64(3)).astype(np.int64)
    w = np.choose(lane, [x[0], x[1], x[2], x[3]]).astype(np.uint64)
    return ((w * np.uint64(bound)) >> np.uint64(32)).astype(np.int64)
