"""Step time of the train() kernels (critic + actor phases through the C ABI, eager, no Adam cost excluded) and of one
ensemble step for the other BASELINE shapes: python tools/shape_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import golden_util as gu
from mobody_amd import ops, packing
from mobody_amd.engine import Engine

dev = torch.device("cuda:0")
for name, S, A, N, Nt, task in (("C2 walker", 17, 6, 10240, 8192, 4), ("C3 halfcheetah", 17, 6, 40960, 32768, 1),
                                ("C4 ant /GPU", 111, 8, 20480, 16384, 3), ("C5 pen", 45, 24, 10240, 8192, 6)):
    cfg = gu.policy_cfg(S, A)
    pa, pq, _ = gu.policy_params(1, S, A)
    eng = Engine(S, A, pa, pq, dev)
    batch = gu.gi.batch(3, N, S, A)
    for _ in range(3):
        eng.step(batch, Nt, cfg)
    torch.cuda.synchronize()
    b = [torch.as_tensor(x, dtype=torch.float32).to(dev).contiguous() for x in batch]
    dims, hyp = ops.train_dims(S, A, N, Nt), ops.hyper(cfg)
    ws = ops.train_workspace(dims, dev)
    g = torch.cuda.CUDAGraph()

    def body():
        ops.critic_step(dims, hyp, eng.actor, eng.q, eng.q_T, eng.qt, b, eng.gq, eng.loss[0:1], ws)
        ops.adam_polyak(S + A, 1, 2, eng.q, eng.q_T, eng.gq, eng.mq, eng.vq, eng.qt, 5, 3e-4, 0.005)
        ops.actor_forward(dims, hyp, eng.actor, eng.q, b[0], b[1], eng.stats, ws)
        ops.actor_backward(dims, hyp, eng.actor, eng.actor_T, eng.q, eng.q_T, b[0], b[1], eng.stats, eng.ga, eng.loss[1:3], ws)
        ops.adam_polyak(S, A, 1, eng.actor, eng.actor_T, eng.ga, eng.ma, eng.va, None, 5, 3e-4)
    body(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        body()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
    q = (S + A) * 256 + 65536 + 256; ac = S * 256 + 65536 + 256 * A
    flops = 2.0 * (N * (2 * q + ac + 2 * q) + Nt * 2 * q + N * (ac + 2 * q)) \
        + 2.0 * N * (2 * (256 + 65536) + 2 * (256 + 65536 + 256 * A) + (256 * A + 65536)) + 2.0 * N * (2 * q + ac)
    # ensemble step over 50 000 rows
    p = gu.gi.dyn_params(7, S, A); blob = packing.pack_dynamics(p, S, A, dev)
    rng = np.random.default_rng(0); B = 50000
    obs = torch.from_numpy(gu.gi.walker_like_obs(rng, B, S)).to(dev); act = torch.from_numpy(rng.uniform(-1, 1, (B, A)).astype(np.float32)).to(dev)
    for _ in range(2):
        ops.dyn_step(blob, S, A, task, obs, act, seed=1, call=1, penalty_coef=0.1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ops.dyn_step(blob, S, A, task, obs, act, seed=1, call=1, penalty_coef=0.1)
    torch.cuda.synchronize(); dms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name:16s} S={S:3d} A={A:2d} N={N:6d}: train kernels {ms:.3f} ms/step ({N / ms / 1e3:.1f} M rows/s, {flops / ms / 1e9:.1f} TFLOP/s useful) | dyn_step 50k rows {dms:.3f} ms ({B / dms / 1e3:.1f} M transitions/s)")
