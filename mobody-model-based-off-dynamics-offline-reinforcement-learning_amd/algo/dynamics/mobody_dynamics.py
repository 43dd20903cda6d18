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
        self.precision = ops.prec_id(str(config.get("mfma", ops.default_mfma())))     # MFMA mode of step(): 0 exact fp32 | bf16 / bf16x2 / bf16x3
        # pre-training follows the mode when it is the fp32-grade "f16x2", and stays on exact fp32 MFMA under the bf16 modes
        self.train_precision = 4 if self.precision == 4 else 0
        self._calls = 0
        self.noise_fn = None          # optional hook: noise_fn((7, B, S)) -> unit normals (tests)
        self.train_noise_fn = None    # optional hook: b -> (noise6[6,7,b,16], noise7[7,b,S]) device tensors (tests)
        self._ws = None
        self._pre_ws_by_b, self._pre_bufs_by_b, self._train_calls = {}, {}, 0
        # 1: replay full pre-training batches as one HIP graph (1 GPU, device noise).  Off by default since round 3: the step is
        # ~20 launches of 5-20 us, the host's eager launches stay ahead of the GPU, and a graph replay measured SLOWER (0.260 ms
        # against 0.237 per step at 256 rows x 7 members; with the side stream 0.266 against 0.228) -- for a host that cannot.
        self.train_graph = int(config.get("train_graph", 0))
        self._pre_graphs = {}
        self._pre_ctr = torch.zeros(4, dtype=torch.int64, device=model.device)
        self._pre_acc = torch.zeros(5, dtype=torch.float32, device=model.device)
        self._pre_loss = torch.zeros(5, dtype=torch.float32, device=model.device)
        task_id = getattr(terminal_fn, "task_id", None)
        if task_id is None:
            raise TypeError("terminal_fn must come from mobody_amd.algo.mb_utils.terminal_funs.get_termination_fn "
                            "(the predicate is evaluated inside the fused kernel)")
        self._task_id = task_id

    def step_device(self, obs, action, use_penalty=True, use_trg=True, alive=None, want_mean=False, elite_idx=None,
                    call=None, call_dev=None, seed_offset=0):
        """call / call_dev / seed_offset: explicit noise-stream position (call + call_dev[0], device word) for a captured
        HIP graph, whose replays cannot advance the host-side call counter."""
        m = self.model
        m.inference()
        obs = torch.as_tensor(obs, dtype=torch.float32).to(m.device).contiguous()
        action = torch.as_tensor(action, dtype=torch.float32).to(m.device).reshape(-1, m.action_dim).contiguous()
        B = obs.shape[0]
        if call is None:
            self._calls += 1
            call = self._calls
        noise = self.noise_fn((7, B, m.obs_dim)) if self.noise_fn is not None else None
        if elite_idx is None and self.rng == "numpy":
            elite_idx = m.random_elite_idxs(B)
        need = 7 * B * (m.obs_dim + 1)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 1), dtype=torch.float32, device=m.device)
        return ops.dyn_step(m.packed(), m.obs_dim, m.action_dim, self._task_id, obs, action, noise=noise,
                            elite_idx=elite_idx, alive=alive, elites=m.elites_host(),
                            seed=(self.seed + seed_offset + dp.rank_salt()) & 0xFFFFFFFF, call=call, call_dev=call_dev,
                            penalty_coef=float(self._penalty_coef or 0.0),
                            use_penalty=bool(use_penalty), use_trg=bool(use_trg), want_mean=want_mean,
                            workspace=self._ws, planes=m.planes() if self.precision else None, precision=self.precision,
                            mopo=m.packed_mopo() if getattr(m, "mopo", False) else None)

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

    # ------------------------------------------------------------------ pre-training (mobody_dynamics.py:594-653,731-978,1113-1156)
    def _check_pretrain_config(self):
        cfg = self.config
        if cfg.get("latent_reward"):
            raise NotImplementedError("latent_reward = 1 is outside the accelerated pre-training path (reference default 0; the "
                                      "reference's own learn() raises TypeError with it: reward_loss_with_latent calls "
                                      "encode_trg_action(action) without the state, mobody_dynamics.py:409 against mobody_module.py:258)")
        if (cfg.get("train_together") or cfg.get("inverse_sep_reward_loss")) and self._world()[0] > 1:
            raise NotImplementedError("train_together / inverse_sep_reward_loss are supported on one GPU (their joint steps are not sharded)")
        if cfg.get("train_together") and cfg.get("inverse_sep_reward_loss"):
            raise NotImplementedError("train_together together with inverse_sep_reward_loss is not mirrored")
        if cfg.get("train_with_src_threshold", 1) != 1 and self._world()[0] > 1:
            raise NotImplementedError("train_with_src_threshold != 1 (data_augmentation) is supported on one GPU")

    def _sep(self):
        return bool(self.config.get("inverse_sep_reward_loss"))

    def _enc_coef(self):
        """Weight of encoder_loss in the step's loss.  config['no_vae'] = 1 (mobody_dynamics.py:616-635): the reference neither
        evaluates nor adds encoder_loss -- the same gradients as weight 0 here (the reconstruction / KL / latent-consistency
        terms enter every gradient through this factor only) -- and reports 0 for its three numbers (_stats)."""
        return 0.0 if self.config.get("no_vae") else self.encoder_loss_coef

    def _stats(self, t):
        """(loss, transition, encoder, recon, kl) as learn() returns them.  Under no_vae the last three are 0 (:632-635) and
        the second equals the first: `loss = transition_loss` aliases the tensor, `loss += reward_loss` (:641) adds in place,
        so the reference's transition_loss.item() is the total (pinned by fixture g12_pretrain_walker_novae)."""
        t = tuple(float(x) for x in t)
        return (t[0], t[0], 0.0, 0.0, 0.0) if self.config.get("no_vae") else t

    def _lr(self):
        o = self.optim                                    # torch.optim.Adam(model.parameters(), lr=dynamics_lr) in the reference
        if o is not None and getattr(o, "param_groups", None):
            return float(o.param_groups[0]["lr"])
        return float(getattr(o, "lr", None) or self.config.get("dynamics_lr", 1e-3))

    def _world(self):
        d = torch.distributed
        return (d.get_world_size(), d.get_rank()) if d.is_available() and d.is_initialized() else (1, 0)

    def _learn_batch(self, use_trg, xenc, act, rew, b, b_global, lo_rel=0):
        """One optimizer step on the rows already laid out as the kernels want them (zero_grad, backward, Adam.step).
        Data parallel: this rank holds rows [lo_rel, lo_rel + b) of the batch's b_global rows."""
        m = self.model
        st = m.train_state(self.train_precision)
        S, A = m.obs_dim, m.action_dim
        ws = self._ws_for(max(b, 1))
        self._train_calls += 1
        n6 = n7 = None
        if self.train_noise_fn is not None:
            n6, n7 = self.train_noise_fn(b_global)        # the noise of the WHOLE batch on every rank (one stream), then this
            if b != b_global:                             # rank's rows of it: N ranks == one rank on the same batch
                n6, n7 = n6[:, :, lo_rel:lo_rel + b].contiguous(), n7[:, lo_rel:lo_rel + b].contiguous()
        world, _ = self._world()
        if b > 0:
            ops.pretrain_grads(S, A, b, use_trg, self._enc_coef(), st["blob"], st["blob_T"], xenc, act, rew, st["grad"],
                               self._pre_loss, ws, noise6=n6, noise7=n7,
                               seed=(self.seed + 77 + dp.rank_salt()) & 0xFFFFFFFF, call=self._train_calls, b_global=b_global,
                               precision=self.train_precision, reward_coef=0.0 if self._sep() else 1.0)
        else:                                             # data parallel: this rank has no row of a ragged last batch
            st["grad"].zero_(); self._pre_loss.zero_()
        if world > 1:
            torch.distributed.all_reduce(st["grad"])      # one 6.5 MB message per step
            torch.distributed.all_reduce(self._pre_loss)
        st["t_main"] += 1; st["t_za"][bool(use_trg)] += 1
        # inverse_sep_reward_loss: learn() leaves reward_loss out (:637-641) -- the reward head has no gradient, Adam skips it and
        # its step count (st["t_rw"], advanced by learn_sep_reward only) stays
        ops.pretrain_adam(S, A, use_trg, st["blob"], st["blob_T"], st["grad"], st["m"], st["v"], st["t_main"],
                          st["t_za"][bool(use_trg)], self._lr(), precision=self.train_precision,
                          net_mask=3 if self._sep() else 7, t_rw=None)
        m.mark_trained()
        return self._pre_loss

    def _gather_bufs(self, b):
        """The batch tensors of the bootstrap gather, kept per batch size (an eager pass would allocate three per step)."""
        if b not in self._pre_bufs_by_b:
            m, dev = self.model, self.model.device
            self._pre_bufs_by_b[b] = (torch.empty(7, 2 * b, m.obs_dim, dtype=torch.float32, device=dev),
                                      torch.empty(7, b, m.action_dim, dtype=torch.float32, device=dev),
                                      torch.empty(7, b, dtype=torch.float32, device=dev))
        return self._pre_bufs_by_b[b]

    def _ws_for(self, b):
        if b not in self._pre_ws_by_b:
            self._pre_ws_by_b[b] = ops.pretrain_workspace(self.model.obs_dim, self.model.action_dim, b, self.model.device)
        return self._pre_ws_by_b[b]

    def _learn_batch_fused(self, use_trg, xenc, act, rew, b, acc=None):
        """Single-GPU form of _learn_batch: the gradient reductions apply Adam themselves (mobody_pretrain_update); `acc`
        (device float[5]): the step's last launch adds the loss vector onto it (no launch of its own for learn()'s sums)."""
        m = self.model
        st = m.train_state(self.train_precision)
        S, A = m.obs_dim, m.action_dim
        ws = self._ws_for(b)
        self._train_calls += 1
        n6 = n7 = None
        if self.train_noise_fn is not None:
            n6, n7 = self.train_noise_fn(b)
        st["t_main"] += 1; st["t_za"][bool(use_trg)] += 1
        ops.pretrain_update(S, A, b, use_trg, self._enc_coef(), st["blob"], st["blob_T"], xenc, act, rew, st["m"], st["v"],
                            st["t_main"], st["t_za"][bool(use_trg)], self._lr(), self._pre_loss, ws, noise6=n6,
                            noise7=n7, seed=(self.seed + 77) & 0xFFFFFFFF, call=self._train_calls, precision=self.train_precision,
                            loss_acc=acc)
        m.mark_trained()
        return None if acc is not None else self._pre_loss

    def _learn_graph(self, use_trg, data, idx, batch_size, n_full):
        """`n_full` full batches of one pass as replays of ONE captured HIP graph (gather + forward + backward + fused Adam,
        ~22 launches): the batch offset into the bootstrap matrix, the noise call id and the Adam step counts live in
        device words that the graph advances itself.  Device-RNG noise only.  Returns the summed loss vector."""
        m = self.model
        st = m.train_state(self.train_precision)
        S, A, b, dev = m.obs_dim, m.action_dim, batch_size, m.device
        d = bool(use_trg)
        ws = self._ws_for(b)
        # every pointer and scalar the captured launches bake in: a graph replayed after any of them moved (a reloaded model,
        # a second train() call, a changed learning rate) would read freed memory or the old constant without any error
        key = (d, b, idx.shape[1], idx.data_ptr(), ws.data_ptr(), self._pre_ctr.data_ptr(), self._pre_acc.data_ptr(),
               self._pre_loss.data_ptr(), float(self._lr()), float(self._enc_coef()), int(self.seed), self.train_precision) \
            + tuple(t.data_ptr() for t in data) + tuple(st[k].data_ptr() for k in ("blob", "blob_T", "m", "v"))
        c = self._pre_ctr                                  # [batch index, call, t_main, t_za]: one launch advances all four
        c.copy_(torch.tensor([-1, self._train_calls, st["t_main"], st["t_za"][d]], dtype=torch.int64), non_blocking=False)
        self._pre_acc.zero_()
        if key not in self._pre_graphs:
            bufs = (torch.empty(7, 2 * b, S, dtype=torch.float32, device=dev), torch.empty(7, b, A, dtype=torch.float32, device=dev),
                    torch.empty(7, b, dtype=torch.float32, device=dev))

            def body():
                ops.counter_add(c, 1)
                ops.pretrain_gather(data[0], data[1], data[2], data[3], idx, 0, b, out=bufs, start_dev=c[0:1])
                ops.pretrain_update(S, A, b, d, self._enc_coef(), st["blob"], st["blob_T"], bufs[0], bufs[1], bufs[2],
                                    st["m"], st["v"], 1, 1, self._lr(), self._pre_loss, ws,
                                    seed=(self.seed + 77) & 0xFFFFFFFF, call=0, call_dev=c[1:2], t_dev=c[2:4],
                                    precision=self.train_precision, loss_acc=self._pre_acc)

            for k in [k for k in self._pre_graphs if k[:2] == key[:2]]:     # a stale graph of this domain / batch size
                del self._pre_graphs[k]
            body()                                        # warm-up (eager) step counts as batch 0
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            self._pre_graphs[key] = (g, bufs)
            first = 1                                     # (capturing executes nothing: the device counters only move when kernels run)
        else:
            first = 0
        g = self._pre_graphs[key][0]
        for _ in range(n_full - first):
            g.replay()
        self._train_calls += n_full
        st["t_main"] += n_full; st["t_za"][d] += n_full
        self.total_steps = getattr(self, "total_steps", 0) + n_full
        m.mark_trained()
        return self._pre_acc

    def _learn_loop(self, n, batch_size, step):
        """learn()'s batch loop (:604-650): `step(start, rows)` runs one batch and returns the device loss vector; the
        five reported numbers are the means over batches, fetched once per call."""
        self.model.training = True
        acc = torch.zeros(5, dtype=torch.float32, device=self.model.device)
        n_batch = int(np.ceil(n / batch_size))
        for k in range(n_batch):
            self.total_steps = getattr(self, "total_steps", 0) + 1
            r = step(k * batch_size, min(batch_size, n - k * batch_size), acc)
            if r is not None:                             # (None: the step's own last launch accumulated onto `acc`)
                acc += r
        return self._stats((acc / max(n_batch, 1)).tolist())

    def _shard(self, start, rows):
        """Rows of one batch owned by this rank (data parallel: contiguous slices, b_global = rows)."""
        world, rank = self._world()
        lo, hi = rows * rank // world, rows * (rank + 1) // world
        return start + lo, hi - lo

    # ---- config['train_with_src_threshold'] != 1: data_augmentation, mobody_dynamics.py:660-729 ---------------------------------
    def update_classifier(self, src_replay_buffer, tar_replay_buffer, batch_size, writer=None):
        """One step of the augmentation classifier (:660-683): batch_size source rows (label 0) + batch_size target rows
        (label 1), input noise, cross-entropy on the heads' probabilities, Adam -- the kernels of the DARA classifier
        (csrc/dara.hip; the reference's row permutation is dropped: the loss is a mean over rows, the noise iid)."""
        cls = self.classifier
        s0, a0, n0, _, _ = src_replay_buffer.sample(batch_size)
        s1, a1, n1, _, _ = tar_replay_buffer.sample(batch_size)
        s, a, s2 = (torch.cat(x, 0).contiguous() for x in ((s0, s1), (a0, a1), (n0, n1)))
        (z_sas, x_sas, h1s, h2s), (z_sa, x_sa, h1a, h2a) = cls.logits(s, a, s2, True, seed=(self.seed + 57) & 0xFFFFFFFF, save=True)
        dz_sas, dz_sa, loss = ops.dara_loss_grad(z_sas[0], z_sa[0], batch_size, None)
        for net, opt, dz, x, h1, h2 in ((cls.sas_classifier, cls.opt_sas, dz_sas, x_sas, h1s, h2s),
                                        (cls.sa_classifier, cls.opt_sa, dz_sa, x_sa, h1a, h2a)):
            self._cls_ws = ops.mlp3_backward(net.blob_T, net.in_dim, 2, 1, dz, x, h1, h2, opt.grad, getattr(self, "_cls_ws", None))
            opt.step()
        return loss[0], loss[1]

    def data_augmentation(self, buffer):
        """:685-729: train a domain classifier for 8 000 steps of 256 + 256 rows, then every SOURCE row whose sas head says
        "target" with probability above config['train_with_src_threshold'] -- after the reference's second softmax over the
        head's probabilities (F.softmax(sas_logits) where the "logits" already are softmax outputs, :701) -- is kept in
        `src_replay_buffer_sim_trg`; train() adds those rows to the target TRAINING set (:797-812)."""
        from ..offline_offline.mobody import _Classifier
        src_rb, tar_rb = buffer
        m = self.model
        cfg = self.config
        self.classifier = _Classifier(m.obs_dim, m.action_dim, m.device, cfg["gaussian_noise_std"], cfg["actor_lr"])
        for _ in range(8000):
            self.update_classifier(src_rb, tar_rb, 256, None)
        s, a, s2, r, nd = src_rb.sample_all()
        s, a, s2 = s.contiguous(), a.contiguous(), s2.contiguous()
        z_sas, _ = self.classifier.logits(s, a, s2, False)
        probs = torch.softmax(torch.softmax(z_sas[0], -1), -1)[:, 1]
        include = probs > float(cfg["train_with_src_threshold"])
        self.augment_probs, self.augment_include = probs, include
        sim = [x[include].contiguous() for x in (s, a, s2, r.reshape(-1, 1))]

        class _Sim(object):                               # what train() reads of the reference's second ReplayBuffer
            size = int(include.sum())

            @staticmethod
            def sample_all(cuda=True):
                return sim[0], sim[1], sim[2], sim[3], None

        self.src_replay_buffer_sim_trg = _Sim()
        print("number of added data", _Sim.size)

    # ---- config['inverse_sep_reward_loss'] = 1: learn_sep_reward, mobody_dynamics.py:482-519 -----------------------------------
    def _learn_sep_reward_batch(self, src, trg):
        """One optimizer step on reward_loss(source batch) + reward_loss(target batch) only; src / trg = (xenc, act, rew, b).
        Two gradient calls with encoder and transition weights 0, the blobs added, one Adam step: the reward head steps with its
        own count (learn() skipped it), everything the fake next state reaches -- encoder, decoder, both action encoders --
        with theirs.  Returns the device scalar of the total loss."""
        m = self.model
        st = m.train_state(self.train_precision)
        if "grad2" not in st:
            st["grad2"] = torch.zeros_like(st["grad"])
            self._pre_loss2 = torch.zeros_like(self._pre_loss)
        st.setdefault("t_rw", 0)
        S, A = m.obs_dim, m.action_dim
        st["grad"].zero_(); st["grad2"].zero_()
        for (xenc, act, rew, b), d, grad, out in ((src, False, st["grad"], self._pre_loss), (trg, True, st["grad2"], self._pre_loss2)):
            self._train_calls += 1
            n6 = n7 = None
            if self.train_noise_fn is not None:
                n6, n7 = self.train_noise_fn(b)
            ops.pretrain_grads(S, A, b, d, 0.0, st["blob"], st["blob_T"], xenc, act, rew, grad, out, self._ws_for(b), noise6=n6,
                               noise7=n7, seed=(self.seed + 77) & 0xFFFFFFFF, call=self._train_calls,
                               precision=self.train_precision, transition_coef=0.0, reward_coef=1.0)
        st["grad"] += st["grad2"]
        st["t_main"] += 1; st["t_rw"] += 1; st["t_za"][False] += 1; st["t_za"][True] += 1
        ops.pretrain_adam(S, A, True, st["blob"], st["blob_T"], st["grad"], st["m"], st["v"], st["t_main"], st["t_za"][True],
                          self._lr(), precision=self.train_precision, net_mask=7, t_rw=st["t_rw"])
        ops.pretrain_za_adam(S, A, False, st["blob"], st["grad"], st["m"], st["v"], st["t_za"][False], self._lr())
        m.mark_trained()
        return self._pre_loss[0] + self._pre_loss2[0]

    def _learn_sep_reward_loop(self, n_trg, batch_size, batch):
        self.model.training = True
        acc = torch.zeros((), dtype=torch.float32, device=self.model.device)
        n_batch = int(np.ceil(n_trg / batch_size))
        for k in range(n_batch):
            self.total_steps = getattr(self, "total_steps", 0) + 1
            acc += self._learn_sep_reward_batch(*batch(k))
        return float(acc / max(n_batch, 1))

    def learn_sep_reward(self, src_train_obss, src_train_actions, src_train_next_obss, src_train_rewards, trg_train_obss,
                         trg_train_actions, trg_train_next_obss, trg_train_rewards, batch_size):
        """mobody_dynamics.py:482-519 on per-member rows `[7, n, .]` of both domains -> mean total reward loss."""
        self._check_pretrain_config()
        dev = self.model.device
        f = lambda x: torch.as_tensor(x, dtype=torch.float32).to(dev)
        S_ = [f(src_train_obss), f(src_train_actions), f(src_train_next_obss), f(src_train_rewards).reshape(7, -1)]
        T_ = [f(trg_train_obss), f(trg_train_actions), f(trg_train_next_obss), f(trg_train_rewards).reshape(7, -1)]

        def rows(D, k):
            sl = slice(k * batch_size, (k + 1) * batch_size)
            s = D[0][:, sl]
            return torch.cat([s, D[2][:, sl]], 1).contiguous(), D[1][:, sl].contiguous(), D[3][:, sl].contiguous(), s.shape[1]

        return self._learn_sep_reward_loop(T_[0].shape[1], batch_size, lambda k: (rows(S_, k), rows(T_, k)))

    def _learn_sep_reward_indexed(self, src, src_idx, trg, trg_idx, batch_size):
        n_s, n_t = src_idx.shape[1], trg_idx.shape[1]

        def gather(data, idx, n, k):
            lo = k * batch_size
            b = min(batch_size, n - lo)
            assert b > 0, "learn_sep_reward walks the source rows in step with the target batches: not enough source rows"
            xenc, act, rew = ops.pretrain_gather(data[0], data[1], data[2], data[3], idx, lo, b)
            return xenc, act, rew, b

        return self._learn_sep_reward_loop(n_t, batch_size, lambda k: (gather(src, src_idx, n_s, k), gather(trg, trg_idx, n_t, k)))

    # ---- config['train_together'] = 1: learn_src_trg, mobody_dynamics.py:521-590 ------------------------------------------
    def _learn_src_trg_batch(self, src, trg):
        """One optimizer step on loss(source batch) + loss(target batch); src / trg = (xenc, act, rew, b).  The target
        batch's encoder_loss is weighted 1 x encoder_loss_coef here (:571), not learn()'s 5 x -- the kernels apply 5 x to a
        target batch, so it is called with a fifth of the coefficient; the two gradient blobs are summed (the shared nets
        get both contributions, each action encoder its own) and ONE Adam step moves everything, both encoders included.
        Returns the two device loss vectors (source, target)."""
        m = self.model
        st = m.train_state(self.train_precision)
        if "grad2" not in st:
            st["grad2"] = torch.zeros_like(st["grad"])
            self._pre_loss2 = torch.zeros_like(self._pre_loss)
        S, A = m.obs_dim, m.action_dim
        st["grad"].zero_(); st["grad2"].zero_()
        for (xenc, act, rew, b), d, grad, out, coef in ((src, False, st["grad"], self._pre_loss, self._enc_coef()),
                                                       (trg, True, st["grad2"], self._pre_loss2, self._enc_coef() / 5.0)):
            self._train_calls += 1
            n6 = n7 = None
            if self.train_noise_fn is not None:           # the reference draws the source batch's seven tensors, then the target's
                n6, n7 = self.train_noise_fn(b)
            ops.pretrain_grads(S, A, b, d, coef, st["blob"], st["blob_T"], xenc, act, rew, grad, out, self._ws_for(b),
                               noise6=n6, noise7=n7, seed=(self.seed + 77) & 0xFFFFFFFF, call=self._train_calls,
                               precision=self.train_precision)
        st["grad"] += st["grad2"]
        st["t_main"] += 1; st["t_za"][False] += 1; st["t_za"][True] += 1
        ops.pretrain_adam(S, A, True, st["blob"], st["blob_T"], st["grad"], st["m"], st["v"], st["t_main"], st["t_za"][True],
                          self._lr(), precision=self.train_precision)
        ops.pretrain_za_adam(S, A, False, st["blob"], st["grad"], st["m"], st["v"], st["t_za"][False], self._lr())
        m.mark_trained()
        return self._pre_loss, self._pre_loss2

    def _learn_src_trg_loop(self, n_trg, batch_size, batch):
        """learn_src_trg's loop over ceil(n_trg / batch_size) batches; batch(k) -> (src, trg) tuples.  Returns the reference's
        five numbers: mean total loss, mean TARGET transition / encoder loss, nan (it averages a list it never fills, :589),
        mean target KL."""
        self.model.training = True
        acc = torch.zeros(4, dtype=torch.float32, device=self.model.device)
        n_batch = int(np.ceil(n_trg / batch_size))
        for k in range(n_batch):
            self.total_steps = getattr(self, "total_steps", 0) + 1
            ls, lt = self._learn_src_trg_batch(*batch(k))
            acc += torch.stack([ls[0] + lt[0], lt[1], lt[2], lt[4]])
        a = (acc / max(n_batch, 1)).tolist()
        return (float(a[0]), float(a[1]), float(a[2]), float("nan"), float(a[3]))

    def learn_src_trg(self, use_trg_data, train_obss, train_actions, train_next_obss, train_rewards, train_obss_trg,
                      train_actions_trg, train_next_obss_trg, train_rewards_trg, batch_size, logvar_loss_coef, trg_transition=None):
        """mobody_dynamics.py:521-590 on per-member rows `[7, n, .]` of both domains (batch k = columns k bs .. (k + 1) bs of each)."""
        self._check_pretrain_config()
        dev = self.model.device
        f = lambda x: torch.as_tensor(x, dtype=torch.float32).to(dev)
        S_ = [f(train_obss), f(train_actions), f(train_next_obss), f(train_rewards).reshape(7, -1)]
        T_ = [f(train_obss_trg), f(train_actions_trg), f(train_next_obss_trg), f(train_rewards_trg).reshape(7, -1)]

        def rows(D, k):
            sl = slice(k * batch_size, (k + 1) * batch_size)
            s = D[0][:, sl]
            return torch.cat([s, D[2][:, sl]], 1).contiguous(), D[1][:, sl].contiguous(), D[3][:, sl].contiguous(), s.shape[1]

        return self._learn_src_trg_loop(T_[0].shape[1], batch_size, lambda k: (rows(S_, k), rows(T_, k)))

    def _learn_src_trg_indexed(self, src, src_idx, trg, trg_idx, batch_size):
        """learn_src_trg on device-resident data sets with [7, n] bootstrap index matrices (as _learn_indexed)."""
        n_s, n_t = src_idx.shape[1], trg_idx.shape[1]

        def gather(data, idx, n, k):
            lo = k * batch_size
            b = min(batch_size, n - lo)
            assert b > 0, "learn_src_trg walks the source rows in step with the target batches: not enough source rows"
            xenc, act, rew = ops.pretrain_gather(data[0], data[1], data[2], data[3], idx, lo, b)
            return xenc, act, rew, b

        return self._learn_src_trg_loop(n_t, batch_size, lambda k: (gather(src, src_idx, n_s, k), gather(trg, trg_idx, n_t, k)))

    def learn(self, use_trg_data, train_obss, train_actions, train_next_obss, train_rewards, batch_size, logvar_loss_coef,
              trg_transition=None):
        """mobody_dynamics.py:594-653: one pass over the per-member rows `[7, n, .]` in batches of `batch_size`.
        Returns (mean loss, mean transition_loss, mean encoder_loss, mean recon_loss, mean kl_loss)."""
        self._check_pretrain_config()
        dev = self.model.device
        f = lambda x: torch.as_tensor(x, dtype=torch.float32).to(dev)
        s, a, s2, r = f(train_obss), f(train_actions), f(train_next_obss), f(train_rewards).reshape(7, -1)

        def step(start, rows, acc=None):
            lo, b = self._shard(start, rows)
            sl = slice(lo, lo + b)
            xenc = torch.cat([s[:, sl], s2[:, sl]], 1).contiguous()
            return self._learn_batch(use_trg_data, xenc, a[:, sl].contiguous(), r[:, sl].contiguous(), b, rows, lo - start)

        return self._learn_loop(s.shape[1], batch_size, step)

    def _learn_indexed(self, use_trg, data, idx, batch_size):
        """learn() on a device-resident data set with a [7, n] bootstrap index matrix: the batch rows are gathered by a
        kernel straight into the layout the forward pass reads (the reference gathers [7, n, .] copies on the host every
        epoch and ships each batch over PCIe, :604-612)."""
        world, _ = self._world()

        def step(start, rows, acc=None):
            lo, b = self._shard(start, rows)
            if b == 0:
                return self._learn_batch(use_trg, None, None, None, 0, rows, lo - start)
            xenc, act, rew = ops.pretrain_gather(data[0], data[1], data[2], data[3], idx, lo, b, out=self._gather_bufs(b))
            if world == 1 and not self._sep():
                return self._learn_batch_fused(use_trg, xenc, act, rew, b, acc)
            return self._learn_batch(use_trg, xenc, act, rew, b, rows, lo - start)

        n = idx.shape[1]
        n_full = n // batch_size
        if world == 1 and self.train_graph and self.train_noise_fn is None and n_full >= 3 and not self._sep():
            self.model.training = True
            acc = self._learn_graph(use_trg, data, idx, batch_size, n_full).clone()
            n_batch = n_full
            if n % batch_size:                              # ragged last batch: eager
                self.total_steps += 1
                acc += step(n_full * batch_size, n - n_full * batch_size)
                n_batch += 1
            return self._stats((acc / n_batch).tolist())
        return self._learn_loop(n, batch_size, step)

    @torch.no_grad()
    def validate(self, use_trg_data, holdout_obss, holdout_actions, holdout_next_obss, holdout_rewards):
        """mobody_dynamics.py:1113-1140 -> (val_transition_loss, val_encode_loss), two lists of 7 floats."""
        m = self.model
        m.training = False
        m.inference()
        f = lambda x: torch.as_tensor(x, dtype=torch.float32).to(m.device).contiguous()
        out = ops.dyn_validate(m.packed(), m.obs_dim, m.action_dim, f(holdout_obss), f(holdout_actions), f(holdout_next_obss),
                               f(holdout_rewards), use_trg_data).cpu().numpy()
        m.uninference()
        return list(out[:7]), list(out[7:])

    def select_elites(self, metrics):
        """:1142-1146."""
        pairs = sorted([(metric, index) for metric, index in zip(metrics, range(len(metrics)))], key=lambda x: x[0])
        return [pairs[i][1] for i in range(self.model.num_elites)]

    def shuffle_rows(self, arr):
        """:656-658 (NumPy global stream): an independent permutation of every member's bootstrap indices."""
        if self.rng == "numpy":
            a = arr.cpu().numpy()
            idxes = np.argsort(np.random.uniform(size=a.shape), axis=-1)
            return torch.from_numpy(a[np.arange(a.shape[0])[:, None], idxes]).to(arr.device)
        return torch.gather(arr, 1, torch.argsort(torch.rand(arr.shape, device=arr.device), dim=1))

    def train(self, src_data, trg_data, max_epochs=None, max_epochs_since_update=5, batch_size=256, holdout_ratio=0.2,
              logvar_loss_coef=0.01, writer=None, buffer=None):
        """mobody_dynamics.py:731-978 (train_together=0): holdout split, bootstrap indices, per epoch one pass over the
        source rows and three over the target rows, validation, per-member early stopping on the TARGET holdout loss
        (saved copies refreshed on > 1 % improvement), finally elites = the num_elites best members and load_save().
        Index streams follow the reference (torch CPU generator for random_split / randint, NumPy for shuffle_rows)
        when rng == 'numpy'."""
        self._check_pretrain_config()
        m = self.model
        dev = m.device
        augment = self.config.get("train_with_src_threshold", 1) != 1
        if augment:                                                                         # :745-746
            self.data_augmentation(buffer)
        self.src_replay_buffer = src_data
        self.total_steps = 0
        self._pre_graphs.clear()                          # graphs of an earlier train() call captured that call's tensors
        f = lambda x, c: torch.as_tensor(x, dtype=torch.float32).reshape(len(x), c).to(dev)
        S, A = m.obs_dim, m.action_dim
        src = [f(src_data[0], S), f(src_data[1], A), f(src_data[2], S), f(src_data[3], 1)]
        trg = [f(trg_data[0], S), f(trg_data[1], A), f(trg_data[2], S), f(trg_data[3], 1)]
        n_src, n_trg = src[0].shape[0], trg[0].shape[0]
        src_hold = min(int(n_src * holdout_ratio), 1000)                                    # :762-763
        trg_hold = min(int(n_trg * holdout_ratio), 500)
        split = torch.utils.data.random_split
        s_tr, s_ho = split(range(n_src), (n_src - src_hold, src_hold))                      # :765-769
        t_tr, t_ho = split(range(n_trg), (n_trg - trg_hold, trg_hold))
        world, _ = self._world()
        if world > 1:                                     # data parallel: every rank trains rank 0's model ...
            self.model.broadcast_(torch.distributed)

        def bc(t):                                        # ... on rank 0's index streams
            if world > 1:
                torch.distributed.broadcast(t, 0)
            return t

        ix = lambda sp: bc(torch.as_tensor(sp.indices, dtype=torch.long, device=dev))
        src_tr = [x[ix(s_tr)].contiguous() for x in src]; src_ho = [x[ix(s_ho)].contiguous() for x in src]
        trg_tr = [x[ix(t_tr)].contiguous() for x in trg]; trg_ho = [x[ix(t_ho)].contiguous() for x in trg]
        self.obs_scaler.fit(None)                                                           # identity (Q4)
        n_s, n_t = n_src - src_hold, n_trg - trg_hold
        if augment:                                                                         # :797-812: the selected source rows join
            sim = self.src_replay_buffer_sim_trg.sample_all()                               # the target TRAINING set (not the holdout)
            trg_tr = [torch.cat([x, y.to(dev).reshape(len(y), -1)], 0).contiguous() for x, y in zip(trg_tr, sim[:4])]
            n_t = trg_tr[0].shape[0]
        E = m.num_ensemble
        trg_holdout_losses = [1e10 for _ in range(E)]
        src_idx = bc(torch.randint(n_s, size=[E, n_s]).to(device=dev, dtype=torch.int32)).contiguous()   # :826-827 (CPU generator)
        trg_idx = bc(torch.randint(n_t, size=[E, n_t]).to(device=dev, dtype=torch.int32)).contiguous()
        epoch, cnt = 0, 0
        self.history = []
        while True:
            epoch += 1
            self.epoch = epoch
            together = bool(self.config.get("train_together"))
            src_stats = self._learn_indexed(False, src_tr, src_idx, batch_size)              # :873-878 (:855-860 when together)
            if together:                                                                    # :853-880: one joint pass, no 3 x target
                trg_stats = self._learn_src_trg_indexed(src_tr, src_idx, trg_tr, trg_idx, batch_size)
            src_val, _ = self.validate(False, *src_ho)
            src_holdout_loss = float(np.sort(src_val)[:m.num_elites].mean())
            for _ in range(0 if together else 3):                                           # :897-907
                trg_stats = self._learn_indexed(True, trg_tr, trg_idx, batch_size)
            trg_val, trg_enc = self.validate(True, *trg_ho)
            trg_holdout_loss = float(np.sort(trg_val)[:m.num_elites].mean())
            self.history.append(dict(epoch=epoch, src=src_stats, trg=trg_stats, src_holdout=src_holdout_loss,
                                     trg_holdout=trg_holdout_loss, src_val=src_val, trg_val=trg_val, trg_reward_val=trg_enc))
            if writer is not None and not together:                                         # :890-894, 921-924 (the joint branch only prints)
                writer.add_scalar("src_loss/dynamics_train_loss", src_stats[1], global_step=epoch)
                writer.add_scalar("src_loss/dynamics_encoder_loss", src_stats[2], global_step=epoch)
                writer.add_scalar("src_loss/dynamics_domain_loss", src_stats[3], global_step=epoch)
                writer.add_scalar("src_loss/dynamics_holdout_loss", src_holdout_loss, global_step=epoch)
                writer.add_scalar("trg_loss/dynamics_train_loss", trg_stats[1], global_step=epoch)
                writer.add_scalar("trg_loss/dynamics_encoder_loss", trg_stats[2], global_step=epoch)
                writer.add_scalar("trg_loss/dynamics_holdout_loss", trg_holdout_loss, global_step=epoch)
            if not together and self._sep():                                                # :935-941 (after the target passes + validation)
                self._learn_sep_reward_indexed(src_tr, src_idx, trg_tr, trg_idx, batch_size)
            if not together:                                                                # (:943-944 sit inside the else branch)
                src_idx.copy_(bc(self.shuffle_rows(src_idx).contiguous()))                  # :934-935 (in place: the captured
                trg_idx.copy_(bc(self.shuffle_rows(trg_idx).contiguous()))                  #  graphs keep reading these tensors)
            indexes = []
            for i, new_loss, old_loss in zip(range(E), trg_val, trg_holdout_losses):        # :937-942
                if (old_loss - new_loss) / old_loss > 0.01:
                    indexes.append(i)
                    trg_holdout_losses[i] = new_loss
            if len(indexes) > 0:
                m.update_save(indexes)
                cnt = 0
            else:
                cnt += 1
            if (cnt >= max_epochs_since_update) or (max_epochs and (epoch >= max_epochs)):  # :951
                break
        indexes = self.select_elites(trg_holdout_losses)
        m.set_elites(indexes)
        m.load_save()
        m.training = False
        self.trg_holdout_losses = trg_holdout_losses
        print("elites:{} , holdout loss: {}".format(indexes, (np.sort(trg_holdout_losses)[:m.num_elites]).mean()))

    def save(self, save_path):
        torch.save(self.model.state_dict(), os.path.join(save_path, "dynamics.pth"))        # :1158-1161
        self.obs_scaler.save_scaler(save_path)

    def load(self, load_path):
        sd = torch.load(os.path.join(load_path, "dynamics.pth"), map_location=self.model.device, weights_only=True)
        self.model.load_state_dict(sd)                                                       # :1163-1166
        self.obs_scaler.load_scaler(load_path)
