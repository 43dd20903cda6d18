import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch, numpy as np
import golden_util as gu
from mobody_amd import synthetic, ops
from mobody_amd.algo import utils
from mobody_amd.algo.call_algo import call_algo
dev = torch.device("cuda:0")
S, A, task, bs = 17, 6, "walker2d-medium-v2", 64
src = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=4000, rng="device", seed=1), 4000, task, 0)
tar = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=500, rng="device", seed=2), 500, task, 1)
fake_rows = gu.gi.batch(9, 300, S, A)
def make(graph):
    torch.manual_seed(3)
    cfg = gu.policy_cfg(S, A, rng="device", seed=7, graph=graph, src_rollout_length=0, trg_rollout_length=0, use_src_sa_to_get_target_next_state=0)
    pol = call_algo("mobody", cfg, 3, dev)
    pol.fake_replay_buffer.add_batch(dict(obss=fake_rows[0], actions=fake_rows[1], next_obss=fake_rows[2], rewards=fake_rows[3], terminals=1.0 - fake_rows[4]))
    return pol
g = make(1); e = make(0)
print("init equal", torch.equal(g.q_funcs.blob, e.q_funcs.blob))
g.train(src, tar, bs, None, None); e.train(src, tar, bs, None, None)
torch.cuda.synchronize()
print("after eager step1 equal", torch.equal(g.q_funcs.blob, e.q_funcs.blob), torch.equal(g.policy.blob, e.policy.blob))
g.train(src, tar, bs, None, None)
torch.cuda.synchronize()
print("ctr", g._ctr.tolist(), "losses", g.losses())
gb = [t.clone() for t in g._batch]
e.total_it += 1
c = torch.tensor([1], dtype=torch.int64, device=dev)
idx = [ops.sample_indices(7 + 101, 3, c, 0, bs, src.ptr_size[1:2]), ops.sample_indices(7 + 102, 3, c, 0, bs, tar.ptr_size[1:2]), ops.sample_indices(7 + 103, 3, c, 0, bs // 2, e.fake_replay_buffer.ptr_size[1:2])]
ops.gather_batch([src._fields(), tar._fields(), e.fake_replay_buffer._fields()], idx, S, A, out=e._batch)
print("batch equal", [torch.equal(a, b) for a, b in zip(gb, e._batch)])
e._update(e._batch, int(2.5 * bs), 2 * bs)
torch.cuda.synchronize()
print("e losses", e.losses())
print("q grad equal", torch.equal(g.q_optimizer.grad, e.q_optimizer.grad), (g.q_optimizer.grad - e.q_optimizer.grad).abs().max().item())
print("q m equal", torch.equal(g.q_optimizer.m, e.q_optimizer.m), "v", torch.equal(g.q_optimizer.v, e.q_optimizer.v))
print("q blob maxdiff", (g.q_funcs.blob - e.q_funcs.blob).abs().max().item(), "qt", (g.target_q_funcs.blob - e.target_q_funcs.blob).abs().max().item())
print("actor blob maxdiff", (g.policy.blob - e.policy.blob).abs().max().item())
