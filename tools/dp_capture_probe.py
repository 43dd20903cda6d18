"""Probe (GPU box, one rank, RCCL): (a) can torch.distributed.all_reduce be captured inside a HIP graph next to the
library's kernels; (b) per-step cost of the three data-parallel modes at world 1: single graph (no collectives),
four segment graphs + eager all-reduces, one graph with the all-reduces captured."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

from mobody_amd import engine, synthetic
from mobody_amd.algo import utils
from mobody_amd.algo.call_algo import call_algo

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
S, A, bs, task = 17, 6, int(os.environ.get("BS", 4096)), "walker2d-medium-v2"
res = {}
for mode in ("single", "segments", "captured"):
    cfg = engine.default_config(S, A, rng="device", seed=3, penalty_type="none", batch_size=bs, graph=1)
    cfg["dp_graph"] = "captured" if mode == "captured" else "segments"
    torch.manual_seed(0); np.random.seed(0)
    pol = call_algo("mobody", cfg, 3, dev)
    pol._force_segments = mode != "single"
    src = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=100000, rng="device", seed=100), 100000, task, 0)
    tar = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=5000, rng="device", seed=200), 5000, task, 50)
    pol.fake_replay_buffer = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=30000, rng="device", seed=300), 30000, task, 90)
    pol.total_it = 1
    try:
        for _ in range(20):
            pol.train(src, tar, bs, None, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            pol.train(src, tar, bs, None, None)
        torch.cuda.synchronize()
        res[mode] = dict(ms_per_step=(time.perf_counter() - t0) / 300 * 1e3, graphs=len(pol._graph) if pol._graph else 0,
                         q0=float(pol.q_funcs.blob.double().sum()))
    except Exception as exc:
        res[mode] = dict(error=repr(exc))
print(json.dumps(res))
dist.destroy_process_group()
