"""MOBODY policy -- host-side mirror of `algo/offline_offline/mobody.py` (class MOBODY, :89-657).

Same plugin surface as the reference: `MOBODY(config, device)`, `.train(src_rb, tar_rb, batch_size,
writer, wandbrun)`, `.select_action(state, policy, cuda=False)`, `.rollout(init_obss, rollout_length,
use_trg)`, `.save/.load(prefix)`, attributes `.policy .q_funcs .target_q_funcs .dynamics .total_it
.fake_replay_buffer`.  What differs is where the arithmetic runs: every forward/backward/optimizer
op is a kernel of libmobody_hip.so working on packed weight blobs that stay resident in HBM; this
file only orders the calls (the order of mobody.py:347-578) and draws indices.

Data-parallel use (one process per GPU): when `torch.distributed` is initialised with world > 1 the
per-rank minibatch is `batch_size` rows per buffer, gradients are all-reduced (RCCL) as two flat
blobs per step and the two batch statistics of the actor loss as one 2-float message, so the
N-GPU update equals the 1-GPU update on the concatenated batch.

All MOBODY variants of the reference path are covered: penalty_type none/par/dara, scale_Q, q_weighted,
advantage (V-function), fake_batch_scale=0, rollout_from_src.
"""
import numpy as np
import torch

from ... import _lib, dp, ops, packing
from .. import utils

REFRESH_EVERY = 5000        # mobody.py:441
REFRESH_SRC, REFRESH_TAR = 50000, 2000    # :442-443
REFRESH_FROM_SRC_TAR = 100                # target init states of the rollout_from_src branch, :485
GRAPH_DYN_SEED = 0x51ED                    # seed offset of the ensemble-step noise stream of captured 'par' steps


class _PackedNet(object):
    """nn.Module-like handle on a packed 3-layer MLP (actor: 1 member, twin-Q: 2 members)."""

    def __init__(self, in_dim, out_dim, members, prefixes, device, out_mode=0, max_action=1.0, init=True, precision=0):
        self.in_dim, self.out_dim, self.members, self.prefixes = in_dim, out_dim, members, prefixes
        self.device, self.out_mode, self.max_action = device, out_mode, float(max_action)
        self.precision = int(precision)               # MFMA mode the net is evaluated in = format of the W2 planes in blob_T
        self.layout = _lib.mlp_layout(in_dim, out_dim, members)
        if init:                                      # nn.Linear default init (kaiming_uniform a=sqrt(5))
            sd = {}
            for p in prefixes:
                for li, (i, o) in zip((0, 2, 4), ((in_dim, 256), (256, 256), (256, out_dim))):
                    bound = 1.0 / np.sqrt(i)
                    sd[f"{p}network.{li}.weight"] = torch.empty(o, i).uniform_(-bound, bound)
                    sd[f"{p}network.{li}.bias"] = torch.empty(o).uniform_(-bound, bound)
            self.load_state_dict(sd)
        self.training = True

    def load_state_dict(self, sd):
        packed = packing.pack_mlp({k: v for k, v in sd.items()}, self.in_dim, self.out_dim, self.device,
                                  prefixes=list(self.prefixes))
        if getattr(self, "blob", None) is None:
            self.blob = packed
            self.blob_T = ops.mlp_transpose(self.blob, self.in_dim, self.out_dim, self.members, precision=self.precision)
        else:                                         # in place: captured graphs and Adam hold these pointers
            self.blob.copy_(packed)
            ops.mlp_transpose(self.blob, self.in_dim, self.out_dim, self.members, out=self.blob_T, precision=self.precision)
        self.version = getattr(self, "version", 0) + 1

    def state_dict(self):
        out = {}
        for p, m in zip(self.prefixes, packing.unpack_mlp(self.blob, self.in_dim, self.out_dim, self.members)):
            out.update({p + k: v for k, v in m.items()})
        return out

    def parameters(self):
        return list(self.state_dict().values())

    def clone(self):
        c = _PackedNet(self.in_dim, self.out_dim, self.members, self.prefixes, self.device, self.out_mode,
                       self.max_action, init=False, precision=self.precision)
        c.blob, c.blob_T = self.blob.clone(), self.blob_T.clone()
        return c

    def eval(self):
        self.training = False
        return self

    def to(self, device):
        return self

    def __call__(self, x, x2=None):
        x = torch.as_tensor(x, dtype=torch.float32).to(self.device)
        o = ops.mlp3_forward(self.blob, self.in_dim, self.out_dim, self.members, x, x2, self.out_mode, self.max_action,
                             blob_T=self.blob_T, precision=self.precision)
        return o[0] if self.members == 1 else tuple(o[m] for m in range(self.members))


class _Adam(object):
    """Flat Adam state for one packed blob; state_dict in torch.optim.Adam's format (mobody.py:584-594)."""

    def __init__(self, net, lr):
        self.net, self.lr, self.t = net, float(lr), 0
        self.m, self.v = torch.zeros_like(net.blob), torch.zeros_like(net.blob)
        self.grad = torch.zeros_like(net.blob)

    def step(self, target=None, tau=-1.0, grad_scale=1.0):
        self.t += 1
        n = self.net
        ops.adam_polyak(n.in_dim, n.out_dim, n.members, n.blob, n.blob_T, self.grad, self.m, self.v,
                        None if target is None else target.blob, self.t, self.lr, tau, grad_scale,
                        target_T=None if target is None else target.blob_T, precision=n.precision)

    def step_dev(self, t_dev, target=None, tau=-1.0):
        """Graph-capturable step: the 1-based step count is read from the device word `t_dev` (already advanced)."""
        n = self.net
        ops.adam_polyak_dev(n.in_dim, n.out_dim, n.members, n.blob, n.blob_T, self.grad, self.m, self.v,
                            None if target is None else target.blob, t_dev, self.lr, tau, 1.0,
                            target_T=None if target is None else target.blob_T, precision=n.precision)

    def _unpack(self, blob):
        n = self.net
        out = []
        for m in packing.unpack_mlp(blob, n.in_dim, n.out_dim, n.members):
            out += [m[f"network.{li}.{wb}"] for li in (0, 2, 4) for wb in ("weight", "bias")]
        return out

    def state_dict(self):
        ms, vs = self._unpack(self.m), self._unpack(self.v)
        state = {i: dict(step=torch.tensor(float(self.t)), exp_avg=ms[i], exp_avg_sq=vs[i]) for i in range(len(ms))}
        if self.t == 0:
            state = {}
        group = dict(lr=self.lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, maximize=False,
                     foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False,
                     params=list(range(len(ms))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        n = self.net
        st = sd["state"]
        self.lr = float(sd["param_groups"][0]["lr"])
        if not st:
            self.t = 0
            self.m.zero_(); self.v.zero_()
            return
        keys = [f"network.{li}.{wb}" for li in (0, 2, 4) for wb in ("weight", "bias")]
        per = len(keys)
        for name, dst in (("exp_avg", self.m), ("exp_avg_sq", self.v)):
            members = [{k: st[m * per + j][name] for j, k in enumerate(keys)} for m in range(n.members)]
            dst.copy_(packing.pack_mlp(members, n.in_dim, n.out_dim, n.device))        # in place (graph pointers)
        self.t = int(float(st[0]["step"]))


class _Classifier(object):
    """Domain classifier (mobody.py:11-33): two packed MLPs `sa_classifier` (S+A -> 2) and `sas_classifier`
    (2S+A -> 2) whose heads output softmax probabilities.  Training and the reward-penalty sweep run through the
    HIP kernels (csrc/dara.hip); `__call__` exists for API compatibility with `Classifier.forward`."""

    def __init__(self, S, A, device, gaussian_noise_std, lr):
        self.S, self.A, self.device = S, A, device
        self.action_dim, self.gaussian_noise_std = A, float(gaussian_noise_std)
        self.sa_classifier = _PackedNet(S + A, 2, 1, ("sa_classifier.",), device)
        self.sas_classifier = _PackedNet(2 * S + A, 2, 1, ("sas_classifier.",), device)
        self.opt_sa, self.opt_sas = _Adam(self.sa_classifier, lr), _Adam(self.sas_classifier, lr)
        self._calls = 0

    def logits(self, s, a, s2, with_noise, noise_sas=None, noise_sa=None, seed=0, save=False):
        self._calls += 1
        std = self.gaussian_noise_std if with_noise else 0.0
        x_sas, x_sa = ops.dara_inputs(s, a, s2, std, noise_sas, noise_sa, seed, self._calls)
        f = lambda net, x: ops.mlp3_forward(net.blob, net.in_dim, 2, 1, x, save=save)
        return f(self.sas_classifier, x_sas), f(self.sa_classifier, x_sa)

    def __call__(self, state_batch, action_batch, nextstate_batch, with_noise):
        zs, za = self.logits(state_batch, action_batch, nextstate_batch, with_noise)
        return torch.softmax(zs[0], 1), torch.softmax(za[0], 1)      # API-compat only; the training path never calls this

    def state_dict(self):
        return {**self.sa_classifier.state_dict(), **self.sas_classifier.state_dict()}

    def load_state_dict(self, sd):
        self.sa_classifier.load_state_dict({k: v for k, v in sd.items() if k.startswith("sa_classifier.")})
        self.sas_classifier.load_state_dict({k: v for k, v in sd.items() if k.startswith("sas_classifier.")})

    def parameters(self):
        return self.sa_classifier.parameters() + self.sas_classifier.parameters()


class MOBODY(object):
    def __init__(self, config, device, target_entropy=None):
        self.config = config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MOBODY (MI355X build) needs a GPU device; there is no CPU fallback")
        if int(config["hidden_sizes"]) != 256:
            raise ValueError("hidden_sizes must be 256 (the width the kernels are built for)")
        self.discount, self.tau = config["gamma"], config["tau"]
        self.update_interval = config["update_interval"]
        self.penalty_type = config["penalty_type"]
        S, A = int(config["state_dim"]), int(config["action_dim"])
        self.S, self.A = S, A
        # MFMA mode of the 256 x 256 layers: 'f32' exact | 'f16x2' | 'bf16x3' (both fp32-grade) | 'bf16x2' | 'bf16'
        self.mfma = str(config.get("mfma", ops.default_mfma()))
        self.precision = ops.prec_id(self.mfma)
        self.rng = config.get("rng", "numpy")               # 'numpy' = reference index/elite streams; 'device' = Philox
        self.seed = int(config.get("seed", 0))
        self.fake_replay_buffer = utils.ReplayBuffer(S, A, self.device, rng=self.rng, seed=self.seed + 17)
        self.total_it = 0
        self.q_funcs = _PackedNet(S + A, 1, 2, ("network1.", "network2."), self.device, precision=self.precision)
        self.target_q_funcs = self.q_funcs.clone().eval()                         # deepcopy, mobody.py:116
        self.policy = _PackedNet(S, A, 1, ("network.",), self.device, out_mode=1, max_action=config["max_action"],
                                 precision=self.precision)                         # select_action / rollouts at the configured precision
        self.v_func = _PackedNet(S, 1, 1, ("network.",), self.device)                # ValueFunc, mobody.py:50-57,121
        self.v_optimizer = _Adam(self.v_func, config["critic_lr"])
        self._v_ws = None
        self.q_optimizer = _Adam(self.q_funcs, config["critic_lr"])
        self.policy_optimizer = _Adam(self.policy, config["actor_lr"])
        self.classifier = _Classifier(S, A, self.device, config["gaussian_noise_std"], config["actor_lr"])
        self.dynamics = None
        self._ws, self._ws_key = None, None
        self._loss = torch.zeros(4, dtype=torch.float32, device=self.device)
        self._stats = torch.zeros(2, dtype=torch.float32, device=self.device)
        self._batch, self._batch_key = None, None
        self.last_losses = None
        # HIP-graph replay of the steady-state step (config['graph']=1, rng='device', single GPU, no PAR/DARA):
        # all per-step scalars (RNG call id, Adam step counts, buffer sizes) live in device words.
        # config['graph']: 0 never, 1 always, 2 auto (replay pays off for launch-bound steps, i.e. small minibatches)
        self.use_graph = int(config.get("graph", 0))
        self._graph, self._graph_key = None, None
        self._force_segments = False           # test hook: replay the data-parallel segments even with one rank
        # data-parallel graph mode: 'segments' = four graphs with eager all-reduces between them, 'captured' = ONE graph
        # holding the kernels and the three all-reduces (falls back to 'segments' if the capture is refused)
        # (measured with one RCCL rank at bs 4096: 0.385 ms/step captured vs 0.431 segments; tools/dp_capture_probe.py)
        import os
        self.dp_graph = str(config.get("dp_graph", os.environ.get("MOBODY_DP_GRAPH", "captured")))
        # one GPU: the gradient reduction applies Adam/Polyak itself (mobody_critic_update / mobody_actor_update);
        # config['fused_update']=0 keeps the separate gradient blobs + optimizer launches (what N > 1 ranks use)
        self.fused_update = int(config.get("fused_update", 1))
        self._ctr = torch.zeros(4, dtype=torch.int64, device=self.device)      # [rng call, critic t, actor t, V t]
        self._synced = False                   # data parallel: replicas broadcast from rank 0 before the first step
        self.classifier_noise_fn = None        # optional hook: n_rows -> (noise_sas[n,2S+A], noise_sa[n,S+A]) (tests)

    # ------------------------------------------------------------------ data-parallel replica hygiene
    def _seed_for(self, k):
        """Philox seed of the k-th index stream of this rank (src, tar, fake = 101, 102, 103): the rank is folded
        in here so that ranks never draw the same rows even when every rank was given the same config seed."""
        return (self.seed + k + dp.rank_salt()) & 0xFFFFFFFF

    def _opts(self):
        return (self.q_optimizer, self.policy_optimizer, self.v_optimizer, self.classifier.opt_sa, self.classifier.opt_sas)

    def sync_replicas(self):
        """world > 1: make this replica identical to rank 0's (weights, transposes, Adam moments and step counts,
        dynamics model).  The all-reduced gradients are only the gradient of the concatenated batch if every rank
        evaluates them at the same parameters; nothing else in the step exchanges parameters."""
        self._synced = True
        if self._world() == 1:
            return
        d = torch.distributed
        nets = (self.q_funcs, self.target_q_funcs, self.policy, self.v_func, self.classifier.sa_classifier,
                self.classifier.sas_classifier)
        steps = torch.tensor([o.t for o in self._opts()], dtype=torch.int64, device=self.device)
        for t in [x for n in nets for x in (n.blob, n.blob_T)] + [x for o in self._opts() for x in (o.m, o.v)] + [steps]:
            d.broadcast(t, 0)
        for o, t in zip(self._opts(), steps.tolist()):
            o.t = int(t)
        model = getattr(self.dynamics, "model", None)
        if model is not None and hasattr(model, "broadcast_"):
            model.broadcast_(d)
        self._graph = None

    # ------------------------------------------------------------------ acting
    def select_action(self, state, policy, cuda=False):
        """mobody.py:138-144."""
        x = state if isinstance(state, torch.Tensor) else torch.as_tensor(np.asarray(state), dtype=torch.float32)
        action = policy(x.reshape(-1, self.S).to(self.device))
        return action.squeeze() if cuda else action.squeeze().cpu().numpy()

    # ------------------------------------------------------------------ DARA classifier (mobody.py:146-181, 354-381)
    def update_classifier(self, src_replay_buffer, tar_replay_buffer, batch_size, writer=None, noise=None, labels=None,
                          rows=None):
        """One classifier step: rows = src(bs) | tar(bs) with labels 0 | 1 (the reference also permutes the rows; the
        loss is a mean over rows and the input noise is iid, so the permutation is dropped).  `rows`/`labels`/`noise`
        let tests supply the reference's exact permuted batch and noise."""
        cls = self.classifier
        if rows is None:
            # Draw order of the reference: src(bs), tar(bs), then with penalize_fake fake(bs), tar(2bs) (:147-154).
            # Its labels are always [0]*bs + [1]*bs and randperm runs over 2*bs entries (:161-165), so with
            # penalize_fake the rows that reach the classifier are src (label 0) and FAKE (label 1); both target draws
            # only advance the index stream.
            bufs, counts = [src_replay_buffer, tar_replay_buffer], [batch_size, batch_size]
            use_fake = bool(self.config["penalize_fake"]) and self.fake_replay_buffer.size > 0
            if use_fake:
                bufs += [self.fake_replay_buffer, tar_replay_buffer]; counts += [batch_size, 2 * batch_size]
            N = sum(counts)
            out = (torch.empty(N, self.S, device=self.device), torch.empty(N, self.A, device=self.device),
                   torch.empty(N, self.S, device=self.device), torch.empty(N, 1, device=self.device),
                   torch.empty(N, 1, device=self.device))
            o = 0
            for k in range(0, len(bufs), 2):          # the gather kernel takes up to three sources: two per launch
                n = sum(counts[k:k + 2])
                self._gather(bufs[k:k + 2], counts[k:k + 2], tuple(t[o:o + n] for t in out))
                o += n
            if use_fake:                              # rows [src | fake]
                pick = lambda t: torch.cat([t[:batch_size], t[2 * batch_size:3 * batch_size]], 0).contiguous()
            else:                                     # rows [src | tar]
                pick = lambda t: t[:2 * batch_size]
            s, a, s2 = pick(out[0]), pick(out[1]), pick(out[2])
            n_src = batch_size
        else:
            s, a, s2 = rows
            n_src = s.shape[0] // 2
        if noise is None and self.classifier_noise_fn is not None:
            noise = self.classifier_noise_fn(s.shape[0])
        nz = noise or (None, None)
        (z_sas, x_sas, h1s, h2s), (z_sa, x_sa, h1a, h2a) = cls.logits(s, a, s2, True, nz[0], nz[1], seed=self.seed + 31,
                                                                    save=True)
        dz_sas, dz_sa, loss = ops.dara_loss_grad(z_sas[0], z_sa[0], n_src, labels)
        for net, opt, dz, x, h1, h2 in ((cls.sas_classifier, cls.opt_sas, dz_sas, x_sas, h1s, h2s),
                                        (cls.sa_classifier, cls.opt_sa, dz_sa, x_sa, h1a, h2a)):
            self._cls_ws = ops.mlp3_backward(net.blob_T, net.in_dim, 2, 1, dz, x, h1, h2, opt.grad,
                                             getattr(self, "_cls_ws", None))
            world = self._world()
            if world > 1:         # the losses are means over the rank's rows -> global mean = sum / world
                torch.distributed.all_reduce(opt.grad)
            opt.step(grad_scale=1.0 / world)
        return loss[0], loss[1]                           # (loss_sa, loss_sas) as device scalars (the rank's share)

    def _dara_delta(self, s, a, s2, reward=None, coef=0.0):
        """Noise-free classifier pass + the DARC/DARA log-ratio penalty; adds coef*delta to `reward` in place."""
        z_sas, z_sa = self.classifier.logits(s, a, s2, False)
        return ops.dara_penalty(z_sas[0], z_sa[0], coef, reward, want_delta=reward is None)

    def _dara_warmup(self, src_replay_buffer, tar_replay_buffer, batch_size, writer):
        for _ in range(10 * 500):                                                        # :356
            self.update_classifier(src_replay_buffer, tar_replay_buffer, batch_size, writer)
        n, chunk = src_replay_buffer.size, 1 << 16                                       # reference sweeps 1000 rows at a time
        for i in range(0, n, chunk):
            j = min(n, i + chunk)
            rb = src_replay_buffer                         # the fields are column views of the row-interleaved store
            r = rb.reward[i:j].contiguous()
            self._dara_delta(rb.state[i:j].contiguous(), rb.action[i:j].contiguous(), rb.next_state[i:j].contiguous(), r,
                             self.config["penalty_coef"])
            rb.reward[i:j].copy_(r)

    # ------------------------------------------------------------------ rollouts
    def rollout(self, init_obss, rollout_length, use_trg=True):
        """mobody.py:596-657: returns (dict of tensors, info).  Tensors stay on the device.
        Quirk Q1 reproduced: `use_trg` reaches step()'s `use_penalty` slot; the target model is always used."""
        if rollout_length == 0:
            return None, None
        obs = torch.as_tensor(init_obss, dtype=torch.float32).to(self.device)
        keys = ("obss", "next_obss", "actions", "rewards", "terminals", "penalty")
        out = {k: [] for k in keys}
        n_tr = 0
        for _ in range(rollout_length):
            act = self.policy(obs).reshape(-1, self.A)
            r = self.dynamics.step_device(obs, act, use_trg)
            term = r["terminal"].clone()
            for k, v in zip(keys, (obs, r["next_obs"].clone(), act, r["reward"].clone(), term.to(torch.float32),
                                   r["penalty"].clone())):
                out[k].append(v)
            n_tr += obs.shape[0]
            alive = (term == 0).flatten()
            if int(alive.sum()) == 0:                       # host sync, as the reference's np check (:636)
                break
            obs = r["next_obs"][alive]
        res = {k: torch.cat(v, 0) for k, v in out.items()}
        rew_mean = res["rewards"].mean()
        if self.config["filter_bad_rollout"]:
            keep = (res["penalty"] <= self.config["env_filter"]).squeeze(1)
            res = {k: v[keep] for k, v in res.items()}
        return res, {"num_transitions": n_tr, "reward_mean": rew_mean}

    def _rollout_into_fake(self, init_obss, rollout_length, use_trg=True):
        """Same transitions as rollout()+add_batch, entirely on the device (`mobody_rollout`): rows keep their index, an
        alive mask replaces the shrinking batch, the penalty filter and the alive update are formed in the sample kernel and
        the kept rows are stream-compacted into the ring -- 7 launches per horizon step, no host work between steps."""
        if rollout_length == 0:
            return 0
        dyn, m, fb = self.dynamics, self.dynamics.model, self.fake_replay_buffer
        obs = init_obss.contiguous()
        B = obs.shape[0]
        m.inference()
        self._roll_ws = ops.rollout(m.packed(), self.policy.blob, self.S, self.A, dyn._task_id, self.policy.max_action, obs,
                                    rollout_length, list(m.elites_host()),
                                    (dyn.seed + dp.rank_salt()) & 0xFFFFFFFF, dyn._calls + 1, float(dyn._penalty_coef or 0.0),
                                    use_trg, True, self.config["env_filter"], self.config["filter_bad_rollout"],   # quirk Q1
                                    fb._fields(), fb.max_size, fb.ptr_size, getattr(self, "_roll_ws", None),
                                    dyn_planes=m.planes() if dyn.precision else None, actor_blob_T=self.policy.blob_T,
                                    precision=dyn.precision)
        dyn._calls += rollout_length
        fb._pull()
        return B * rollout_length

    def _refresh(self, src_rb, tar_rb, batch_size):
        """Model-rollout refresh of the fake buffer, mobody.py:441-513.

        Data parallel: rollouts have no cross-row coupling, so the 50 000 / 2 000 init states are SHARDED over the
        ranks -- every rank rolls ceil(n / world) rank-salted draws into its own fake-buffer shard and later samples its slice
        of the fake batch from that shard; no collective.  The union of the shards is the reference's buffer in distribution,
        and the refresh costs each rank 1 / world of the single-GPU refresh (`config['shard_refresh'] = 0` keeps every rank on
        the full counts)."""
        cfg = self.config
        world = self._world() if int(cfg.get("shard_refresh", 1)) else 1
        n_src, n_tar = -(-REFRESH_SRC // world), -(-REFRESH_TAR // world)
        s_idx = src_rb.draw_indices(n_src)
        t_idx = tar_rb.draw_indices(n_tar)
        src = ops.gather_batch([src_rb._fields()], [s_idx], self.S, self.A)
        tar = ops.gather_batch([tar_rb._fields()], [t_idx], self.S, self.A)
        if self.rng == "device" and not getattr(getattr(self.dynamics, "model", None), "mopo", False):
            self._rollout_into_fake(src[0], cfg["src_rollout_length"])
            self._rollout_into_fake(tar[0], cfg["trg_rollout_length"])
        else:                                  # NumPy-RNG parity mode, and the mopo ablation (host loop over mobody_mopo_step)
            tr, _ = self.rollout(src[0], cfg["src_rollout_length"])
            self.fake_replay_buffer.add_batch(tr)
            tr, _ = self.rollout(tar[0], cfg["trg_rollout_length"])
            self.fake_replay_buffer.add_batch(tr)
        if cfg["use_src_sa_to_get_target_next_state"]:                       # :460-475 (strict '<' filter)
            r = self.dynamics.step_device(src[0], src[1])
            keep = (r["penalty"] < cfg["env_filter"]).to(torch.uint8).squeeze(1)
            self.fake_replay_buffer.add_batch(dict(obss=src[0], next_obss=r["next_obs"], actions=src[1],
                                                   rewards=r["reward"], terminals=r["terminal"]), keep=keep)
        if cfg["rollout_from_src"]:                                           # :479-513
            if self.penalty_type != "dara":
                self.update_classifier(src_rb, tar_rb, batch_size)                 # train()'s batch_size, :481
            s_idx = src_rb.draw_indices(n_src)
            t_idx = tar_rb.draw_indices(-(-REFRESH_FROM_SRC_TAR // world))
            init = ops.gather_batch([src_rb._fields(), tar_rb._fields()], [s_idx, t_idx], self.S, self.A)[0]
            tr, _ = self.rollout(init, cfg["rollout_from_src_length"], use_trg=False)
            if tr is not None and tr["obss"].shape[0] > 0:
                self._dara_delta(tr["obss"].contiguous(), tr["actions"].contiguous(), tr["next_obss"].contiguous(),
                                 tr["rewards"], cfg["penalty_coef"])
                self.fake_replay_buffer.add_batch(tr)

    # ------------------------------------------------------------------ training
    # ------------------------------------------------------------------ HIP-graph fast path
    def _graph_ok(self, writer):
        want = self.use_graph == 1 or (self.use_graph == 2 and self._batch[0].shape[0] < 4096)
        # 'dara' only acts in the very first call (classifier warm-up + one-off reward rewrite).  'par' (the CLI's default,
        # mobody.py:428-434) relabels the source rows every step: its ensemble step and the reward shaping are captured with
        # the step, the noise call id read from the device counter.  The V phase of `advantage` is captured on one GPU (its
        # gradient all-reduce belongs to the eager exchange protocol).
        dyn = self.dynamics
        par_ok = self.penalty_type != "par" or (dyn is not None and getattr(dyn, "noise_fn", None) is None
                                                and dyn.rng == "device" and not getattr(dyn.model, "mopo", False))
        adv_ok = not self.config["advantage"] or (self._world() == 1 and self.fused_update and self._v_ws is not None)
        # logging steps run eagerly: every 5000th (losses / value scalars) and, under 'par', every 100th (mobody.py:432-433)
        logs = writer is not None and (self.total_it % 5000 == 0 or (self.penalty_type == "par" and self.total_it % 100 == 0))
        return (want and self.rng == "device" and par_ok and adv_ok and (self.total_it - 1) % REFRESH_EVERY != 0 and not logs)

    @staticmethod
    def _world():
        d = torch.distributed
        return d.get_world_size() if d.is_available() and d.is_initialized() else 1

    def _graph_segments(self, src, tar, batch_size, world, segmented):
        """The steady-state step as a list of closures that only enqueue kernels.  One rank: a single segment.  Data
        parallel: the four segments between the three collectives of mobody_amd/dp.py (the all-reduces themselves are
        issued eagerly between the replays, on the same stream)."""
        cfg, S, A = self.config, self.S, self.A
        ns, nt = int(cfg["src_ratio"] * batch_size), int(cfg["trg_ratio"] * batch_size)
        nf = int(cfg["fake_batch_scale"] * batch_size) if cfg["fake_batch_scale"] != 0 else 0
        N, Nt = ns + nt + nf, ns + nt
        Ng, Ntg = N * world, Nt * world
        c, b = self._ctr, self._batch
        bufs, cnts, seeds = [src, tar], [ns, nt], [self._seed_for(101), self._seed_for(102)]
        if nf > 0:
            bufs.append(self.fake_replay_buffer); cnts.append(nf); seeds.append(self._seed_for(103))

        par = self.penalty_type == "par"

        def par_relabel(call):
            # mobody.py:428-434: one ensemble step on the source rows, r -= coef * mean_d (s'_true - s'_model)^2; the noise
            # stream position is `call` + the device counter (a stream of its own: seed offset GRAPH_DYN_SEED)
            r = self.dynamics.step_device(b[0][:ns], b[1][:ns], call=call, call_dev=c[0:1], seed_offset=GRAPH_DYN_SEED)
            ops.par_penalty(b[2][:ns], r["next_obs"], b[3][:ns], cfg["penalty_coef"])

        def critic():
            ops.counter_add(c[0:3])
            ops.gather_batch_rng([rb._fields() for rb in bufs], cnts, seeds, [0] * len(bufs), c[0:1],
                                 [rb.ptr_size[1:2] for rb in bufs], S, A, b)
            if par:
                par_relabel(0)
            self.critic_grad(b, N, Nt, Ng, Ntg)

        def critic_apply_actor_stats():
            self.q_optimizer.step_dev(c[1:2], target=self.target_q_funcs, tau=self.tau)
            self.actor_stats(b, N, Nt, Ng, Ntg)

        def fused_step():
            # no launch of its own for the three counters: the gather draws with call id c[0] + 1 and advances the two Adam
            # step counts (it does not read them), the critic's optimizer launch advances c[0] (the gather is done with it)
            ops.gather_batch_rng([rb._fields() for rb in bufs], cnts, seeds, [1] * len(bufs), c[0:1],
                                 [rb.ptr_size[1:2] for rb in bufs], S, A, b, bump=(c[1:2], c[2:3], c[3:4]))
            if par and not cfg["advantage"] and int(cfg.get("par_overlap", 1)):
                # The ensemble step that relabels the source rewards (mobody.py:428-434) runs on a side stream NEXT TO the
                # critic's forwards -- none of them reads the rewards, the TD error in the backward's prologue is the first
                # reader -- and is joined before the backward: its launches (a few hundred workgroups each) and the forwards'
                # fill each other's idle issue slots.  In the captured graph the side stream is a parallel branch.
                main = torch.cuda.current_stream()
                if getattr(self, "_side_stream", None) is None:
                    self._side_stream = torch.cuda.Stream(device=self.device)
                side = self._side_stream
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    par_relabel(1)
                self.critic_update(b, N, Nt, t_dev=c[1:2], bump=c[0:1], phase=1)
                main.wait_stream(side)
                self.critic_update(b, N, Nt, t_dev=c[1:2], bump=c[0:1], phase=2)
                self.actor_stats(b, N, Nt, N, Nt)
                self.actor_update(b, N, Nt, t_dev=c[2:3])
                return
            if par:
                par_relabel(1)
            if cfg["advantage"]:                                             # V update first (mobody.py:533-537)
                self.value_grad(b, N, Nt, N, Nt)
                self.v_optimizer.step_dev(c[3:4])
            self.critic_update(b, N, Nt, t_dev=c[1:2], bump=c[0:1])
            self.actor_stats(b, N, Nt, N, Nt)
            self.actor_update(b, N, Nt, t_dev=c[2:3])

        def actor():
            self.actor_grad(b, N, Nt, Ng, Ntg)

        def actor_apply():
            self.policy_optimizer.step_dev(c[2:3])

        if not segmented:
            if self.fused_update:
                return [fused_step]
            return [lambda: (critic(), critic_apply_actor_stats(), actor(), actor_apply())]
        if self.dp_graph == "captured":
            # the three all-reduces are captured WITH the kernels: one graph launch per step, no host work between the
            # collectives (RCCL records its stream hand-offs into the capture like any cross-stream dependency)
            d = torch.distributed

            def whole_step():
                critic(); d.all_reduce(self.q_optimizer.grad)
                critic_apply_actor_stats(); d.all_reduce(self._stats)
                actor(); d.all_reduce(self.policy_optimizer.grad)
                actor_apply()
            return [whole_step]
        return [critic, critic_apply_actor_stats, actor, actor_apply]

    def _graph_step(self, src, tar, batch_size):
        world = self._world()
        segmented = world > 1 or (self._force_segments and torch.distributed.is_initialized())
        if segmented and self.dp_graph == "captured" and torch.distributed.get_backend() != "nccl":
            self.dp_graph = "segments"                  # only RCCL collectives can be recorded into a HIP graph (gloo stages
                                                        # through the host; a refused capture leaves the stream unusable)
        # every device pointer the captured kernels read or write: a reloaded checkpoint, a re-assigned fake buffer
        # or a resized minibatch must force a re-capture (replaying against freed tensors corrupts memory silently)
        fb = self.fake_replay_buffer
        nets = (self.q_funcs, self.target_q_funcs, self.policy, self.v_func)
        opts = (self.q_optimizer, self.policy_optimizer, self.v_optimizer)
        dyn_key = None
        if self.penalty_type == "par":                  # the captured ensemble step reads these
            m = self.dynamics.model
            dyn_key = (m.packed().data_ptr(), m.planes().data_ptr() if self.dynamics.precision else 0, self.dynamics.precision,
                       m.elites_host(), float(self.dynamics._penalty_coef or 0.0))
        key = (batch_size, id(src), id(tar), src.state.data_ptr(), tar.state.data_ptr(), world, segmented,
               id(fb), fb.state.data_ptr(), fb.ptr_size.data_ptr(), tuple(t.data_ptr() for t in self._batch),
               tuple((n.blob.data_ptr(), n.blob_T.data_ptr()) for n in nets), self.precision,
               tuple((o.m.data_ptr(), o.v.data_ptr(), o.grad.data_ptr()) for o in opts),
               None if self._ws is None else self._ws.data_ptr(), self.dp_graph, dyn_key,
               None if self._v_ws is None else self._v_ws.data_ptr())
        if self._graph is None or self._graph_key != key:
            torch.cuda.synchronize()
            self._ctr[1] = self.q_optimizer.t
            self._ctr[2] = self.policy_optimizer.t
            self._ctr[3] = self.v_optimizer.t
            if self.penalty_type == "par":               # size the ensemble step's workspace outside the capture
                self.dynamics.step_device(self._batch[0][:int(self.config["src_ratio"] * batch_size)],
                                          self._batch[1][:int(self.config["src_ratio"] * batch_size)], call=0)
            graphs = []
            try:
                for seg in self._graph_segments(src, tar, batch_size, world, segmented):
                    g = torch.cuda.CUDAGraph()
                    # thread-local capture: the process group's watchdog thread may touch the runtime meanwhile
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        seg()
                    graphs.append(g)
            except Exception as exc:
                import warnings
                torch.cuda.synchronize()
                if segmented and self.dp_graph == "captured":      # collectives not capturable here: segment graphs instead
                    warnings.warn(f"capturing the all-reduces failed ({exc!r}); falling back to segment graphs")
                    self.dp_graph, self._graph = "segments", None
                    return self._graph_step(src, tar, batch_size)
                warnings.warn(f"HIP-graph capture of the train() step failed ({exc!r}); continuing eagerly")
                self.use_graph, self._graph = 0, None          # capture refused: stay on the eager path for good
                return False
            self._graph, self._graph_key = graphs, key
        if len(self._graph) == 1:
            self._graph[0].replay()
        else:                                           # exchange protocol of dp.dp_update, segments replayed
            d = torch.distributed
            ga, gb, gc, gd = self._graph
            ga.replay()
            d.all_reduce(self.q_optimizer.grad)
            gb.replay()
            d.all_reduce(self._stats)
            gc.replay()
            d.all_reduce(self.policy_optimizer.grad)
            gd.replay()
        self.q_optimizer.t += 1
        self.policy_optimizer.t += 1
        if self.config["advantage"]:
            self.v_optimizer.t += 1
        return True

    def train(self, src_replay_buffer, tar_replay_buffer, batch_size=128, writer=None, wandbrun=None):
        """One gradient step, mobody.py:347-578."""
        cfg = self.config
        self.total_it += 1
        self.src_replay_buffer, self.tar_replay_buffer = src_replay_buffer, tar_replay_buffer
        if not self._synced:
            self.sync_replicas()
        S, A = self.S, self.A
        ns, nt = int(cfg["src_ratio"] * batch_size), int(cfg["trg_ratio"] * batch_size)
        nf = int(cfg["fake_batch_scale"] * batch_size) if cfg["fake_batch_scale"] != 0 else 0
        N, Nt = ns + nt + nf, ns + nt
        # graph replay only once the minibatch tensors of THIS batch size exist (an eager step allocates them)
        if (self._batch is not None and self._batch_key == (N,) and self._graph_ok(writer)
                and self._graph_step(src_replay_buffer, tar_replay_buffer, batch_size)):
            return
        if self._graph is not None:                       # an eager step (refresh/logging) moves the host-side counts
            self._graph = None
        if self.penalty_type == "dara" and self.total_it == 1:
            self._dara_warmup(src_replay_buffer, tar_replay_buffer, batch_size, writer)
        if self._batch_key != (N,):
            dev = self.device
            self._batch = (torch.empty(N, S, device=dev), torch.empty(N, A, device=dev), torch.empty(N, S, device=dev),
                           torch.empty(N, 1, device=dev), torch.empty(N, 1, device=dev))
            self._batch_key = (N,)
        b = self._batch
        # src | tar rows (mobody.py:399-400); index draws in the reference's order
        self._gather([src_replay_buffer, tar_replay_buffer], [ns, nt], tuple(t[:Nt] for t in b))
        if self.penalty_type == "par":                                        # :428-434
            r = self.dynamics.step_device(b[0][:ns], b[1][:ns])
            if writer is not None and self.total_it % 100 == 0:               # :432-433
                writer.add_scalar("train/reward_penalty_par", torch.mean((b[2][:ns] - r["next_obs"]) ** 2), global_step=self.total_it)
            ops.par_penalty(b[2][:ns], r["next_obs"], b[3][:ns], cfg["penalty_coef"])
        if (self.total_it - 1) % REFRESH_EVERY == 0:
            self._refresh(src_replay_buffer, tar_replay_buffer, batch_size)
        if nf > 0:                                                            # :523-529
            self._gather([self.fake_replay_buffer], [nf], tuple(t[Nt:] for t in b))
        log5k = writer is not None and self.total_it % 5000 == 0
        if log5k and cfg["advantage"]:                    # update_v_function's scalars, BEFORE this step's V update (:236-240)
            qt = ops.mlp3_forward(self.target_q_funcs.blob, S + A, 1, 2, b[0], b[1]).view(2, N)
            v0 = ops.mlp3_forward(self.v_func.blob, S, 1, 1, b[0]).view(N)
            writer.add_scalar("train/adv", (torch.minimum(qt[0], qt[1]) - v0).mean(), self.total_it)
            writer.add_scalar("train/value", v0.mean(), self.total_it)
        self._update(b, N, Nt)
        if log5k:
            if cfg["q_weighted"] and Nt > 0:              # bc_loss's exp_adv (:251-273): Q after this step's critic update, as there
                qb = ops.mlp3_forward(self.q_funcs.blob, S + A, 1, 2, b[0][:Nt], b[1][:Nt], blob_T=self.q_funcs.blob_T,
                                      precision=self.precision).view(2, Nt)
                qb = torch.minimum(qb[0], qb[1])
                if cfg["advantage"]:
                    adv = qb - ops.mlp3_forward(self.v_func.blob, S, 1, 1, b[0][:Nt]).view(Nt)
                else:
                    adv = qb / qb.abs().mean()
                writer.add_scalar("train/exp_adv", torch.exp(3.0 * adv).clamp(max=100.0).mean(), self.total_it)
            q_loss, pi_loss, bc_loss = [float(x) for x in self._loss[:3].tolist()]
            writer.add_scalar("train/q_loss", q_loss, self.total_it)
            writer.add_scalar("train/policy_loss", pi_loss, self.total_it)
            writer.add_scalar("train/bc_loss", bc_loss, self.total_it)
            # the reference's value scalars (:203-205, :332-338); read here after this step's updates, not between them
            prec = self.precision
            q12 = ops.mlp3_forward(self.q_funcs.blob, self.S + self.A, 1, 2, b[0], b[1], blob_T=self.q_funcs.blob_T, precision=prec)
            pi = ops.mlp3_forward(self.policy.blob, self.S, self.A, 1, b[0], out_mode=1, max_action=cfg["max_action"],
                                  blob_T=self.policy.blob_T, precision=prec)[0]
            qpi = ops.mlp3_forward(self.q_funcs.blob, self.S + self.A, 1, 2, b[0], pi, blob_T=self.q_funcs.blob_T, precision=prec)
            writer.add_scalar("train/q1", q12[0].mean(), self.total_it)
            writer.add_scalar("train/q_behavior", torch.minimum(q12[0], q12[1]).mean(), self.total_it)
            writer.add_scalar("train/q_policy", torch.minimum(qpi[0], qpi[1]).mean(), self.total_it)
            if wandbrun is not None:
                wandbrun.log({"train/policy_loss": pi_loss, "train/q_loss": q_loss}, step=self.total_it)

    def _gather(self, bufs, counts, out):
        """Minibatch rows of `bufs` into `out`: NumPy index stream + gather (rng='numpy'), or one kernel that draws
        the same indices `ReplayBuffer.draw_indices` would on the device (rng='device')."""
        if self.rng == "device":
            for rb in bufs:
                if rb.size <= 0:
                    raise ValueError("low >= high")               # np.random.randint(0, 0) in the reference (Q11)
                rb._draws += 1
            salt = dp.rank_salt()
            ops.gather_batch_rng([rb._fields() for rb in bufs], counts, [(rb.seed + salt) & 0xFFFFFFFF for rb in bufs],
                                 [rb._draws for rb in bufs], None, [rb.ptr_size[1:2] for rb in bufs], self.S, self.A, out)
        else:
            idx = [rb.draw_indices(n) for rb, n in zip(bufs, counts)]
            ops.gather_batch([rb._fields() for rb in bufs], idx, self.S, self.A, out=out)

    def _update(self, b, N, Nt):
        """critic step -> Adam+Polyak -> actor forward -> (stats all-reduce) -> actor backward -> Adam
        (exchange protocol: mobody_amd/dp.py)."""
        d = torch.distributed
        dist = d if d.is_available() and d.is_initialized() else None
        if self.fused_update and dp.world_size(dist) == 1:
            if self.config["advantage"]:                                     # V update first (mobody.py:533-537)
                self.value_grad(b, N, Nt, N, Nt)
                self.value_apply()
            self.critic_update(b, N, Nt)
            self.actor_stats(b, N, Nt, N, Nt)
            self.actor_update(b, N, Nt)
        else:
            dp.dp_update(self, b, N, Nt, dist)

    # ---- engine interface of dp.dp_update (every method only enqueues HIP kernels) ----
    def comm_device(self):
        return self.device

    def _dims(self, N, Nt, Ng, Ntg):
        if self._ws_key != (N, Nt):
            self._ws = ops.train_workspace(ops.train_dims(self.S, self.A, N, Nt, N, Nt), self.device)
            self._ws_key = (N, Nt)
        return ops.train_dims(self.S, self.A, N, Nt, Ng, Ntg), ops.hyper(self.config)

    # V-function phase of the advantage variant (update_v_function, mobody.py:231-242, 533-537)
    def has_value_phase(self):
        return bool(self.config["advantage"])

    def value_grad(self, b, N, Nt, Ng, Ntg):
        S, A = self.S, self.A
        qt = ops.mlp3_forward(self.target_q_funcs.blob, S + A, 1, 2, b[0], b[1])               # no grad, :233-234
        v, x, h1, h2 = ops.mlp3_forward(self.v_func.blob, S, 1, 1, b[0], save=True)
        dz3, loss = ops.value_loss_grad(qt.view(2, N), v.view(N), Ng)
        self._v_ws = ops.mlp3_backward(self.v_func.blob_T, S, 1, 1, dz3, x, h1, h2, self.v_optimizer.grad, self._v_ws)
        self._loss[3:4].copy_(loss)

    def value_grad_buffer(self):
        return self.v_optimizer.grad

    def value_apply(self):
        self.v_optimizer.step()

    def critic_grad(self, b, N, Nt, Ng, Ntg):
        dims, hyp = self._dims(N, Nt, Ng, Ntg)
        q_next = None
        if self.config["advantage"]:                                        # update_q_functions_1: y = r + nd*gamma*V(s')
            q_next = ops.mlp3_forward(self.v_func.blob, self.S, 1, 1, b[2]).view(N)
        ops.critic_step(dims, hyp, self.policy.blob, self.q_funcs.blob, self.q_funcs.blob_T, self.target_q_funcs.blob,
                        b, self.q_optimizer.grad, self._loss[0:1], self._ws, q_next=q_next,
                        policy_forward=self._policy_rides_along(), actor_blob_T=self.policy.blob_T,
                        qtarg_blob_T=self.target_q_funcs.blob_T)

    def _policy_rides_along(self):
        """pi(s) of the actor phase is evaluated in the critic phase's target-Q launch (the actor does not change in
        between); not in the advantage variant, whose critic call has no target-Q launch."""
        return not self.config["advantage"]

    def critic_update(self, b, N, Nt, t_dev=None, bump=None, phase=0):
        """critic_grad + critic_apply in the fused single-GPU form (same arithmetic, no gradient blob).  phase 1 / 2: only its
        forwards / only the backward + update (ops.critic_update)."""
        dims, hyp = self._dims(N, Nt, N, Nt)
        q_next = None
        if self.config["advantage"]:
            assert phase == 0, "the phased critic update is not used with the V-function target"
            q_next = ops.mlp3_forward(self.v_func.blob, self.S, 1, 1, b[2]).view(N)
        o = self.q_optimizer
        if t_dev is None and phase != 2:
            o.t += 1
        ops.critic_update(dims, hyp, self.policy.blob, self.q_funcs.blob, self.q_funcs.blob_T, self.target_q_funcs.blob, b,
                          o.m, o.v, o.t, o.lr, self._loss[0:1], self._ws, q_next=q_next, t_dev=t_dev,
                          policy_forward=self._policy_rides_along(), actor_blob_T=self.policy.blob_T,
                          qtarg_blob_T=self.target_q_funcs.blob_T, bump=bump, phase=phase)

    def actor_update(self, b, N, Nt, t_dev=None):
        dims, hyp = self._dims(N, Nt, N, Nt)
        v_true = None
        if self.config["advantage"] and Nt > 0:
            v_true = ops.mlp3_forward(self.v_func.blob, self.S, 1, 1, b[0][:Nt]).view(Nt)
        o = self.policy_optimizer
        if t_dev is None:
            o.t += 1
        ops.actor_update(dims, hyp, self.policy.blob, self.policy.blob_T, self.q_funcs.blob, self.q_funcs.blob_T, b[0], b[1],
                         self._stats, o.m, o.v, o.t, o.lr, self._loss[1:3], self._ws, v_true=v_true, t_dev=t_dev)

    def critic_grad_buffer(self):
        return self.q_optimizer.grad

    def critic_apply(self):
        self.q_optimizer.step(target=self.target_q_funcs, tau=self.tau)        # Adam then update_target (:546-552)

    def actor_stats(self, b, N, Nt, Ng, Ntg):
        dims, hyp = self._dims(N, Nt, Ng, Ntg)
        ops.actor_forward(dims, hyp, self.policy.blob, self.q_funcs.blob, b[0], b[1], self._stats, self._ws,
                          policy_ready=self._policy_rides_along(), actor_blob_T=self.policy.blob_T, q_blob_T=self.q_funcs.blob_T)

    def stats_buffer(self):
        return self._stats

    def actor_grad(self, b, N, Nt, Ng, Ntg):
        dims, hyp = self._dims(N, Nt, Ng, Ntg)
        v_true = None
        if self.config["advantage"] and Nt > 0:                              # adv = q_b - V(s_true), :254-256
            v_true = ops.mlp3_forward(self.v_func.blob, self.S, 1, 1, b[0][:Nt]).view(Nt)
        ops.actor_backward(dims, hyp, self.policy.blob, self.policy.blob_T, self.q_funcs.blob, self.q_funcs.blob_T,
                           b[0], b[1], self._stats, self.policy_optimizer.grad, self._loss[1:3], self._ws, v_true=v_true)

    def actor_grad_buffer(self):
        return self.policy_optimizer.grad

    def actor_apply(self):
        self.policy_optimizer.step()

    def losses(self):
        """(q_loss, pi_loss, bc_loss) of the last step (forces a device sync)."""
        return tuple(float(x) for x in self._loss[:3].tolist())

    # ------------------------------------------------------------------ checkpoints (mobody.py:584-594)
    def save(self, filename):
        torch.save(self.q_funcs.state_dict(), filename + "_critic")
        torch.save(self.q_optimizer.state_dict(), filename + "_critic_optimizer")
        torch.save(self.policy.state_dict(), filename + "_actor")
        torch.save(self.policy_optimizer.state_dict(), filename + "_actor_optimizer")

    def load(self, filename):
        self._graph = None                              # re-capture: the device-side Adam step counts change
        self._synced = False                            # (ranks loading different files are re-aligned to rank 0)
        ld = lambda s: torch.load(filename + s, map_location="cpu", weights_only=True)
        self.q_funcs.load_state_dict(ld("_critic"))
        self.q_optimizer.load_state_dict(ld("_critic_optimizer"))
        self.policy.load_state_dict(ld("_actor"))
        self.policy_optimizer.load_state_dict(ld("_actor_optimizer"))
