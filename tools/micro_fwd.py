"""Micro-benchmark: latency of single mobody_mlp3_forward launches (events on torch's stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import golden_util as gu
from mobody_amd import ops, packing

dev = torch.device("cuda:0")
S, A = 17, 6
pa, pq, _ = gu.policy_params(1, S, A)
ab = packing.pack_mlp([{k[len("network."):]: v for k, v in pa.items()}], S, A, dev)
qb = packing.pack_mlp(pq, S + A, 1, dev, prefixes=["network1.", "network2."])


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rows in [int(x) for x in sys.argv[1:]] or (32, 64, 640, 2560, 10240, 40960):
    s = torch.randn(rows, S, device=dev); a = torch.rand(rows, A, device=dev)
    t_actor = timeit(lambda: ops.mlp3_forward(ab, S, A, 1, s, out_mode=1))
    t_q = timeit(lambda: ops.mlp3_forward(qb, S + A, 1, 2, s, a))
    t_qs = timeit(lambda: ops.mlp3_forward(qb, S + A, 1, 2, s, a, save=True))
    print(f"rows {rows:6d}: actor {t_actor:7.1f} us   twinQ {t_q:7.1f} us   twinQ+save {t_qs:7.1f} us")
