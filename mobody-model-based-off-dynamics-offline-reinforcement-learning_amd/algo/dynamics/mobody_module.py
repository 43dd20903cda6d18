"""Ensemble latent dynamics network -- host-side mirror of `algo/dynamics/mobody_module.py`.

Keeps the reference constructor signature (:50-64), the parameter names/shapes of its state_dict
(Appendix B of SURVEY.md: `<layer>.weight [7,in,out]`, `.bias [7,1,out]`, `.saved_weight/.saved_bias`,
`max/min_logvar(_latent)`, `elites`), the elite bookkeeping (`set_elites` :351-353,
`random_elite_idxs` :355-357) and `inference()/uninference()`.  The forward passes run in the HIP
library on a packed copy of the weights (`packed()`, refreshed whenever the tensors change).
Pre-training (`MOBODYEnsembleDynamics.train/learn`) works on a second packed copy, the TRAINING blob
(`train_state()`: csrc/pretrain.hip's MobodyPretrainLayout with its transposes, gradient and Adam moments);
while it is ahead of the reference-layout tensors `_p` it is the master, and every reader of `_p`
(`state_dict`, `packed`, `update_save`) first pulls it back (`_sync_from_train`).  The action decoders
`za_de_*` take no part in any loss (their .grad is None in the reference) and are only carried in the state_dict.
"""
import numpy as np
import torch

from ... import _lib, ops, packing

LATENT = 16


class MOBODYModule(object):
    def __init__(self, obs_dim, action_dim, hidden_dims, num_ensemble=7, num_elites=5, activation=None,
                 weight_decays=None, with_reward=True, device="cpu", reward_relu=False, config=None):
        if isinstance(hidden_dims, (list, tuple)):
            hidden_dims = hidden_dims[0]
        if int(hidden_dims) != 256 or int(num_ensemble) != 7:
            raise ValueError("the MI355X kernels are built for hidden_dims=256, num_ensemble=7 (reference defaults)")
        self.config = config or {}
        if self.config.get("latent_reward"):
            raise NotImplementedError("the latent_reward ablation is outside the accelerated path")
        self.mopo = bool(self.config.get("mopo"))        # MOPO ablation: a plain ensemble MLP s + f(s, a) (:114-118,133-137)
        self.obs_dim, self.action_dim = int(obs_dim), int(action_dim)
        self.num_ensemble, self.num_elites = 7, int(num_elites)
        self.device = torch.device("cuda" if str(device) == "cpu" and torch.cuda.is_available() else device)
        self.training = True
        self._with_reward = 0
        self.encode_trg_diff = 0
        self._layout = _lib.dyn_layout(self.obs_dim, self.action_dim)
        S, A, H = self.obs_dim, self.action_dim, 256
        dims = dict(zs1=(S, H), zs2=(H, H), zs3=(H, 2 * LATENT), za_src1=(LATENT + A, 32), za_src2=(32, 2 * LATENT),
                    za_de_src1=(LATENT, 8), za_de_src2=(8, A), za_trg1=(LATENT + A, 32), za_trg2=(32, 2 * LATENT),
                    za_de_trg1=(LATENT, 8), za_de_trg2=(8, A), transition1=(LATENT, H), transition2=(H, H),
                    transition3=(H, S), reward_model1=(2 * S + A, H), reward_model2=(H, H), reward_model3=(H, 2))
        if self.mopo:                                             # same module_list order as the reference (:114-118,133-137)
            order = []
            for name, io in dims.items():
                order.append((name, io))
                if name in ("za_src2", "za_trg2"):
                    order.append((name[:-1] + "3", (H, S)))
            dims = dict(order)
            for pre in ("za_src", "za_trg"):
                dims[pre + "1"], dims[pre + "2"] = (S + A, H), (H, H)
        self._p = {}
        for name, (i, o) in dims.items():                         # EnsembleLinear.__init__ :371-389
            w = torch.empty(7, i, o, device=self.device)
            torch.nn.init.trunc_normal_(w, std=1 / (2 * i ** 0.5))
            self._p[name + ".weight"] = w
            self._p[name + ".bias"] = torch.zeros(7, 1, o, device=self.device)
            self._p[name + ".saved_weight"] = w.clone()
            self._p[name + ".saved_bias"] = torch.zeros(7, 1, o, device=self.device)
        self._p["max_logvar"] = torch.ones(S, device=self.device) * 0.5
        self._p["min_logvar"] = torch.ones(S, device=self.device) * -10
        self._p["max_logvar_latent"] = torch.ones(LATENT, device=self.device) * 20
        self._p["min_logvar_latent"] = torch.ones(LATENT, device=self.device) * -20
        self._p["elites"] = torch.arange(self.num_elites, device=self.device)
        self._blob, self._planes, self._mopo = None, None, None
        self._train, self._train_ahead = None, False
        self.layer_names = list(dims)                              # module_list order, mobody_module.py:97-150

    # ---- nn.Module-like surface ----
    @property
    def elites(self):
        return self._p["elites"]

    def elites_host(self):
        """The elite member ids as a host tuple, cached per elites tensor (a captured HIP graph may not read device memory
        from the host; the kernels take the ids as launch arguments)."""
        t = self._p["elites"]
        c = getattr(self, "_elites_host", None)
        if c is None or c[0] is not t:
            self._elites_host = c = (t, tuple(int(e) for e in t.tolist()))
        return c[1]

    def state_dict(self):
        self._sync_from_train()
        return {k: v.detach().clone() for k, v in self._p.items()}

    def load_state_dict(self, sd, strict=True):
        self._sync_from_train()
        self._train = None                                          # Adam moments belong to the old weights
        for k, v in sd.items():
            if k not in self._p:
                if strict:
                    raise KeyError(f"unexpected key {k} in dynamics state_dict")
                continue
            t = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(self.device)
            if k != "elites" and tuple(t.shape) != tuple(self._p[k].shape):
                raise RuntimeError(f"size mismatch for {k}: {tuple(t.shape)} vs {tuple(self._p[k].shape)}")
            self._p[k] = t.to(torch.int64 if k == "elites" else torch.float32).contiguous()
        if strict:
            missing = [k for k in self._p if k not in sd]
            if missing:
                raise KeyError(f"missing keys in dynamics state_dict: {missing[:4]}...")
        self._blob = None

    def parameters(self):
        return [v for k, v in self._p.items() if k != "elites"]

    def broadcast_(self, dist, src=0):
        """Data parallel: adopt rank `src`'s tensors (every rank rolls out with the same model)."""
        self._sync_from_train()
        self._train = None
        for k in sorted(self._p):
            dist.broadcast(self._p[k], src)
        self._blob = None

    def to(self, device):
        return self

    def inference(self):
        self.training = False

    def uninference(self):
        self.training = True

    def set_elites(self, indexes):
        indexes = [int(i) for i in indexes]
        assert len(indexes) <= self.num_ensemble and max(indexes) < self.num_ensemble
        self._p["elites"] = torch.as_tensor(list(indexes), dtype=torch.int64, device=self.device)

    def random_elite_idxs(self, batch_size):
        return np.random.choice(self.elites.cpu().numpy(), size=batch_size)      # NumPy global RNG, :355-357

    def load_save(self):
        """EnsembleLinear.load_save for every layer (mobody_module.py:337-339,407-409): weight <- saved_weight."""
        self._sync_from_train()
        for k in list(self._p):
            if k.endswith(".saved_weight"):
                self._p[k[:-13] + ".weight"] = self._p[k].clone()
            if k.endswith(".saved_bias"):
                self._p[k[:-11] + ".bias"] = self._p[k].clone()
        self._blob = None
        self._push_to_train()

    def update_save(self, indexes):
        """EnsembleLinear.update_save for every layer (:341-343,411-413): saved[idx] <- weight[idx] for the members that
        improved on the holdout set."""
        self._sync_from_train()
        idx = torch.as_tensor(list(indexes), dtype=torch.long, device=self.device)
        for name in self.layer_names:
            self._p[name + ".saved_weight"][idx] = self._p[name + ".weight"][idx]
            self._p[name + ".saved_bias"][idx] = self._p[name + ".bias"][idx]

    # ---- training blob (dynamics pre-training) ----
    def train_state(self, precision=None):
        """dict(blob, blob_T, grad, m, v, t_main, t_za={False: .., True: ..}, prec) of the packed training copy; `precision`
        (0 exact fp32 | 4 f16x2; None = leave as is) = the mode of the coming optimizer steps: the T blob carries that mode's W2 planes."""
        if self.mopo:
            raise NotImplementedError("pre-training with config['mopo'] = 1 is outside the accelerated path "
                                      "(inference / rollouts / checkpoints of such a model are supported)")
        if self._train is None:
            blob = packing.pack_pretrain(self._p, self.obs_dim, self.action_dim, self.device)
            z = lambda: torch.zeros_like(blob)
            precision = precision or 0
            self._train = dict(blob=blob, blob_T=ops.pretrain_transpose(blob, self.obs_dim, self.action_dim, precision=precision),
                               grad=z(), m=z(), v=z(), t_main=0, t_za={False: 0, True: 0}, prec=precision)
        elif precision is not None and self._train["prec"] != precision:          # the mode changed under a live training copy: rebuild the planes
            ops.pretrain_transpose(self._train["blob"], self.obs_dim, self.action_dim, out=self._train["blob_T"], precision=precision)
            self._train["prec"] = precision
        return self._train

    def mark_trained(self):
        """The training blob moved (an optimizer step ran): `_p` and the inference blob are stale until pulled."""
        self._train_ahead = True
        self._blob = None

    def _sync_from_train(self):
        if self._train is not None and self._train_ahead:
            packing.unpack_pretrain(self._train["blob"], self.obs_dim, self.action_dim, into=self._p)
            self._train_ahead = False
            self._blob = None

    def _push_to_train(self):
        """`_p` changed under a live training blob (load_save): re-pack the weights, keep the Adam state."""
        if self._train is not None:
            self._train["blob"].copy_(packing.pack_pretrain(self._p, self.obs_dim, self.action_dim, self.device))
            ops.pretrain_transpose(self._train["blob"], self.obs_dim, self.action_dim, out=self._train["blob_T"],
                                   precision=self._train["prec"])
            self._train_ahead = False

    # ---- HIP side ----
    def packed(self):
        self._sync_from_train()
        if self._blob is None:
            p = self._p
            if self.mopo:                               # the encoder slots of the layout are unused: zeros of their latent shapes
                p = dict(p)
                A = self.action_dim
                for pre in ("za_src", "za_trg"):
                    for n_, (i, o) in ((pre + "1", (LATENT + A, 32)), (pre + "2", (32, 2 * LATENT))):
                        p[n_ + ".weight"] = torch.zeros(7, i, o, device=self.device)
                        p[n_ + ".bias"] = torch.zeros(7, 1, o, device=self.device)
            self._blob = packing.pack_dynamics(p, self.obs_dim, self.action_dim, self.device)
            self._planes, self._mopo = None, None
        return self._blob

    def packed_mopo(self):
        """(blob, blob_T) of the MOPO ablation's 7-member MLP za_src1..3 in mobody_mlp_layout(S + A, S, 7) (EnsembleLinear
        weights are [in, out]; the generic packer takes nn.Linear's [out, in])."""
        self.packed()
        if self._mopo is None:
            S, A = self.obs_dim, self.action_dim
            members = [{f"network.{li}.weight": self._p[f"za_src{k}.weight"][e].t().contiguous()
                        for li, k in ((0, 1), (2, 2), (4, 3))} | {f"network.{li}.bias": self._p[f"za_src{k}.bias"][e, 0]
                                                                   for li, k in ((0, 1), (2, 2), (4, 3))} for e in range(7)]
            blob = packing.pack_mlp(members, S + A, S, self.device)
            self._mopo = (blob, ops.mlp_transpose(blob, S + A, S, 7, precision=self._prec()))
        return self._mopo

    def _prec(self):
        return ops.prec_id(self.config.get("mfma", ops.default_mfma()))

    def planes(self):
        """16-bit planes of the three 256 x 256 layers in the configured split-precision mode's format, rebuilt with the
        packed blob."""
        blob = self.packed()
        if self._planes is None:
            self._planes = ops.dyn_planes(blob, self.obs_dim, self.action_dim, precision=self._prec() or 3)
        return self._planes

    def _fwd(self, state, action, use_trg):
        prec = self._prec()
        if self.mopo:                                   # s + f(s, a): forward_trg == forward_src (:264-266)
            s = torch.as_tensor(state, dtype=torch.float32).to(self.device).contiguous()
            a = torch.as_tensor(action, dtype=torch.float32).to(self.device).reshape(-1, self.action_dim).contiguous()
            B = s.shape[0]                              # the means of one mopo step (zero noise, member 0 picked: both unused here)
            r = ops.dyn_step(self.packed(), self.obs_dim, self.action_dim, 0, s, a,
                             noise=torch.zeros(7, B, self.obs_dim, device=self.device),
                             elite_idx=torch.zeros(B, dtype=torch.int32, device=self.device), want_mean=True,
                             planes=self.planes() if prec else None, precision=prec, mopo=self.packed_mopo())
            return r["mean"], None, None
        mean = ops.dyn_forward(self.packed(), self.obs_dim, self.action_dim, state, action, use_trg,
                               planes=self.planes() if prec else None, precision=prec)
        return mean, None, None        # (mean, zs_mu, zs_logvar): the latent stats are unused by the hot path

    def forward_trg(self, state, action):
        return self._fwd(state, action, True)

    def forward_src(self, state, action):
        return self._fwd(state, action, False)
