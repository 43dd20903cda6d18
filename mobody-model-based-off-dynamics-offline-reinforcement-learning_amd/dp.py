"""Data-parallel exchange protocol of one MOBODY gradient step (host logic, device agnostic).

One process per GPU; every rank holds replicated weights and draws its own rows.  The engine
computes LOCAL shares of the GLOBAL means (losses and gradients are scaled by 1/N_global inside the
kernels), so plain SUM all-reduces make the N-rank update equal the 1-rank update on the
concatenated batch:

    critic gradients (one flat blob)          all_reduce(SUM)   -> Adam + Polyak on every rank
    stats = [sum|min Q(s,pi(s))|, sum|min Q(s_t,a_t)|]  all_reduce(SUM)   (needed BEFORE the actor backward:
                                              p_w = w / mean|q| and adv = q_b / mean|q_b| are global normalisers,
                                              mobody.py:318,259)
    actor gradients (one flat blob)           all_reduce(SUM)   -> Adam on every rank

`engine` is any object with the six methods used below; the product engine is
`MOBODY` (HIP kernels); `tests/test_dp_protocol.py` drives the same function with a CPU engine
built on the oracle under gloo, world size 2.
"""


def rank_salt(dist=None):
    """Per-rank offset folded into every device-RNG seed (index draws, rollout noise, elite picks): 0 on one process."""
    if dist is None:
        import torch
        dist = torch.distributed
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return 1000003 * dist.get_rank()
    return 0


def world_size(dist):
    return dist.get_world_size() if dist is not None and dist.is_available() and dist.is_initialized() else 1


def dp_update(engine, batch, n_rows, n_true, dist=None, equal_shards=True):
    world = world_size(dist)
    n_glob, nt_glob = n_rows * world, n_true * world          # every rank draws the same number of rows
    if world > 1 and not equal_shards:              # ragged shards: agree on the global counts first (costs a sync)
        import torch
        cnt = torch.tensor([n_rows, n_true], dtype=torch.int64, device=engine.comm_device())
        dist.all_reduce(cnt)
        n_glob, nt_glob = int(cnt[0]), int(cnt[1])
    if getattr(engine, "has_value_phase", lambda: False)():      # config['advantage']: V update first (mobody.py:533-537)
        engine.value_grad(batch, n_rows, n_true, n_glob, nt_glob)
        if world > 1:
            dist.all_reduce(engine.value_grad_buffer())
        engine.value_apply()
    engine.critic_grad(batch, n_rows, n_true, n_glob, nt_glob)
    if world > 1:
        dist.all_reduce(engine.critic_grad_buffer())
    engine.critic_apply()                           # Adam, then Polyak (mobody.py:546-552)
    engine.actor_stats(batch, n_rows, n_true, n_glob, nt_glob)
    if world > 1:
        dist.all_reduce(engine.stats_buffer())
    engine.actor_grad(batch, n_rows, n_true, n_glob, nt_glob)
    if world > 1:
        dist.all_reduce(engine.actor_grad_buffer())
    engine.actor_apply()
    return n_glob, nt_glob
