"""One-step imagined transitions -- host-side mirror of `algo/dynamics/mobody_dynamics.py`.

`MOBODYEnsembleDynamics(config, model, optim, scaler, terminal_fn, penalty_coef, uncertainty_mode)`
and `.step(obs, action, use_penalty=True, use_trg=True) -> (next_obs, reward, terminal, info)` keep the
reference's signature and return types (:162-172, :193-265): device tensors for next_obs / reward,
a NumPy bool [B,1] for `terminal`, `info = {'samples', 'raw_reward', 'penalty'}`.  Everything between
inputs and outputs is the fused HIP path (csrc/dynamics.hip); `step_device()` is the same call without
the NumPy conversion (no device sync), which is what the MOBODY mirror uses internally.

Randomness: elite ids come from `model.random_elite_idxs` (NumPy global RNG, host) when
`rng == 'numpy'` so the NumPy stream is consumed exactly as in the reference, or from the device
Philox generator when `rng == 'device'`; the Gaussian noise always comes from the device generator
(the reference's torch CUDA stream cannot be reproduced bit-for-bit anyway) unless `noise_fn` is set.
"""
import os

import numpy as np
import torch

from ... import dp, ops


class StandardScaler(object):
    """Identity scaler (the reference's fit() overwrites mu=0, std=1 and transform() returns its input,
    mobody_dynamics.py:83-160)."""

    def __init__(self, mu=None, std=None):
        self.mu, self.std = mu, std

    def fit(self, data):
        self.mu, self.std = 0, 1

    def transform(self, data):
        return data

    inverse_transform = transform_tensor = transform

    def save_scaler(self, save_path):
        np.save(os.path.join(save_path, "mu.npy"), np.zeros((1, 1), np.float32))
        np.save(os.path.join(save_path, "std.npy"), np.ones((1, 1), np.float32))

    def load_scaler(self, load_path):
        self.mu, self.std = 0, 1


class MOBODYEnsembleDynamics(object):
    def __init__(self, config, model, optim, scaler, terminal_fn, penalty_coef=0.0, uncertainty_mode="pairwise-diff",
                 rng="numpy", seed=0):
        if uncertainty_mode != "pairwise-diff":
            raise NotImplementedError("only the reference's default 'pairwise-diff' penalty is accelerated")
        self.model, self.optim = model, optim
        self.terminal_fn = terminal_fn
        self._penalty_coef = penalty_coef
        self._uncertainty_mode = uncertainty_mode
        self.obs_scaler, self.action_scaler = StandardScaler(), StandardScaler()
        self.config = config
        self.encoder_loss_coef = config["encoder_loss_coef"]
        self.domain_loss_coef = config["domain_loss_coef"]
        self.cycle_loss_coef = config["cycle_loss_coef"]
        self.encode_trg_diff = getattr(model, "encode_trg_diff", 0)
        self.rng, self.seed = rng, int(seed)
        self._calls = 0
        self.noise_fn = None          # optional hook: noise_fn((7, B, S)) -> unit normals (tests)
        self._ws = None
        task_id = getattr(terminal_fn, "task_id", None)
        if task_id is None:
            raise TypeError("terminal_fn must come from mobody_amd.algo.mb_utils.terminal_funs.get_termination_fn "
                            "(the predicate is evaluated inside the fused kernel)")
        self._task_id = task_id

    def step_device(self, obs, action, use_penalty=True, use_trg=True, alive=None, want_mean=False, elite_idx=None):
        m = self.model
        m.inference()
        obs = torch.as_tensor(obs, dtype=torch.float32).to(m.device).contiguous()
        action = torch.as_tensor(action, dtype=torch.float32).to(m.device).reshape(-1, m.action_dim).contiguous()
        B = obs.shape[0]
        self._calls += 1
        noise = self.noise_fn((7, B, m.obs_dim)) if self.noise_fn is not None else None
        if elite_idx is None and self.rng == "numpy":
            elite_idx = m.random_elite_idxs(B)
        need = 7 * B * (m.obs_dim + 1)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 1), dtype=torch.float32, device=m.device)
        return ops.dyn_step(m.packed(), m.obs_dim, m.action_dim, self._task_id, obs, action, noise=noise,
                            elite_idx=elite_idx, alive=alive, elites=[int(e) for e in m.elites.tolist()],
                            seed=(self.seed + dp.rank_salt()) & 0xFFFFFFFF, call=self._calls,
                            penalty_coef=float(self._penalty_coef or 0.0),
                            use_penalty=bool(use_penalty), use_trg=bool(use_trg), want_mean=want_mean,
                            workspace=self._ws)

    @torch.no_grad()
    def step(self, obs, action, use_penalty=True, use_trg=True):
        r = self.step_device(obs, action, use_penalty, use_trg, want_mean=True)
        info = {"samples": r["mean"], "raw_reward": r["raw_reward"], "penalty": r["penalty"]}
        terminal = r["terminal"].cpu().numpy().astype(bool)            # the reference returns host NumPy (:237)
        return r["next_obs"], r["reward"], terminal, info

    def model_error(self, obs, action, next_obs, reward):
        """Model error on real transitions as eval_policy_batch reports it (train_mobody.py:100-133):
        one `step(obs, action, False)` (no penalty, target model), then
        obs_mse = mean_rows ||next_obs_model - next_obs||_2 and reward_mse = mean (reward - reward_model)^2.
        Returns device scalars plus the per-row distances."""
        dev = self.model.device
        nxt = torch.as_tensor(next_obs, dtype=torch.float32).to(dev)
        rew = torch.as_tensor(reward, dtype=torch.float32).to(dev).reshape(-1)
        r = self.step_device(obs, action, False)
        dist = torch.sqrt(torch.sum((r["next_obs"] - nxt) ** 2, dim=1))
        return {"obs_mse": dist.mean(), "obs_mse_individual": dist,
                "reward_mse": torch.mean((rew - r["reward"].reshape(-1)) ** 2), "penalty": r["penalty"]}

    def train(self, *a, **k):
        raise NotImplementedError("dynamics pre-training (mobody_dynamics.py:731-978) is the first 'next' row of "
                                  "SURVEY 8(f); load a pretrained dynamics with .load(dir)")

    def save(self, save_path):
        torch.save(self.model.state_dict(), os.path.join(save_path, "dynamics.pth"))        # :1158-1161
        self.obs_scaler.save_scaler(save_path)

    def load(self, load_path):
        sd = torch.load(os.path.join(load_path, "dynamics.pth"), map_location=self.model.device, weights_only=True)
        self.model.load_state_dict(sd)                                                       # :1163-1166
        self.obs_scaler.load_scaler(load_path)
