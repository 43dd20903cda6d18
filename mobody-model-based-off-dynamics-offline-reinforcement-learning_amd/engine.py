"""Minimal driver of the C ABI for one (S, A): packed parameters, Adam moments and workspace, one MOBODY gradient step
per `step()` call (critic -> Adam + Polyak -> actor forward -> actor backward -> Adam).  Used by `__graft_entry__.smoke()`
and the parity tests; the full plugin surface is `algo/offline_offline/mobody.py`.

`default_config(S, A, **over)` is the MOBODY config of the reference's walker2d yaml + CLI defaults
(config/mobody/*.yaml, train_mobody.py:470-531) that every caller of the mirror needs."""
import torch

from . import ops, packing


def default_config(S, A, **over):
    cfg = dict(gamma=0.99, tau=0.005, update_interval=2, state_dim=S, action_dim=A, penalty_type="none",
               hidden_sizes=256, max_action=1.0, critic_lr=3e-4, actor_lr=3e-4, gaussian_noise_std=1.0,
               penalize_fake=0, src_ratio=1, trg_ratio=1, src_rollout_length=1, trg_rollout_length=1,
               use_src_sa_to_get_target_next_state=1, env_filter=10.0, rollout_from_src=0, fake_batch_scale=0.5,
               advantage=0, scale_Q=1, weight=2.5, bc_coef=1.0, q_weighted=1, filter_bad_rollout=1,
               penalty_coef=0.1, mopo=0, latent_reward=0, encoder_loss_coef=1, domain_loss_coef=0.0,
               cycle_loss_coef=0.3)
    cfg.update(over)
    return cfg


class Engine:
    """Minimal driver of the C ABI for one (S, A): packed params, Adam moments, workspace."""

    def __init__(self, S, A, pa, pq, dev):
        self.ops, self.packing, self.S, self.A, self.dev = ops, packing, S, A, dev
        self.actor = packing.pack_mlp([{k[len("network."):]: v for k, v in pa.items()}], S, A, dev)
        self.q = packing.pack_mlp(pq, S + A, 1, dev, prefixes=["network1.", "network2."])
        self.qt = self.q.clone()
        self.prec = ops.prec_id(ops.default_mfma())    # plane format of the T blobs (re-built when a step asks for another)
        self.actor_T = ops.mlp_transpose(self.actor, S, A, 1, precision=self.prec)
        self.q_T = ops.mlp_transpose(self.q, S + A, 1, 2, precision=self.prec)
        self.qt_T = self.q_T.clone()                   # the target's T blob (its W2 planes follow the Polyak update)
        z = torch.zeros_like
        self.ma, self.va, self.mq, self.vq = z(self.actor), z(self.actor), z(self.q), z(self.q)
        self.ga, self.gq = z(self.actor), z(self.q)
        self.t = 0
        self.loss = torch.zeros(4, device=dev)
        self.stats = torch.zeros(2, device=dev)

    def step(self, batch, n_true, cfg, apply=True, dims=None):
        ops = self.ops
        S, A = self.S, self.A
        b = [torch.as_tensor(x, dtype=torch.float32).to(self.dev).contiguous() for x in batch]
        N = b[0].shape[0]
        dims = dims or ops.train_dims(S, A, N, n_true)
        hyp = ops.hyper(cfg)
        if (hyp.precision == 4) != (self.prec == 4):   # 'f16x2' keeps its own W2 planes; the other modes share theirs
            ops.mlp_transpose(self.actor, S, A, 1, out=self.actor_T, precision=hyp.precision)
            ops.mlp_transpose(self.q, S + A, 1, 2, out=self.q_T, precision=hyp.precision)
            ops.mlp_transpose(self.qt, S + A, 1, 2, out=self.qt_T, precision=hyp.precision)
        self.prec = hyp.precision
        ws = ops.train_workspace(dims, self.dev)
        ops.critic_step(dims, hyp, self.actor, self.q, self.q_T, self.qt, b, self.gq, self.loss[0:1], ws,
                        actor_blob_T=self.actor_T, qtarg_blob_T=self.qt_T)
        if apply:
            self.t += 1
            ops.adam_polyak(S + A, 1, 2, self.q, self.q_T, self.gq, self.mq, self.vq, self.qt, self.t, cfg["critic_lr"], cfg["tau"],
                            target_T=self.qt_T, precision=self.prec)
        ops.actor_forward(dims, hyp, self.actor, self.q, b[0], b[1], self.stats, ws, actor_blob_T=self.actor_T, q_blob_T=self.q_T)
        ops.actor_backward(dims, hyp, self.actor, self.actor_T, self.q, self.q_T, b[0], b[1], self.stats, self.ga,
                           self.loss[1:3], ws)
        if apply:
            ops.adam_polyak(S, A, 1, self.actor, self.actor_T, self.ga, self.ma, self.va, None, self.t, cfg["actor_lr"],
                            precision=self.prec)
        torch.cuda.synchronize()
        return dict(q_loss=float(self.loss[0]), pi_loss=float(self.loss[1]), bc_loss=float(self.loss[2]))

    def unpack(self, blob, which):
        if which == "actor":
            return {"network." + k: v for k, v in self.packing.unpack_mlp(blob, self.S, self.A, 1)[0].items()}
        ms = self.packing.unpack_mlp(blob, self.S + self.A, 1, 2)
        return {f"network{j + 1}." + k: v for j in range(2) for k, v in ms[j].items()}
