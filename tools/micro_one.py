"""One launch shape repeated (for PMC collection): python tools/micro_one.py ROWS [save]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import golden_util as gu
from mobody_amd import ops, packing
dev = torch.device("cuda:0")
S, A = 17, 6
rows = int(sys.argv[1]); save = len(sys.argv) > 2
_, pq, _ = gu.policy_params(1, S, A)
qb = packing.pack_mlp(pq, S + A, 1, dev, prefixes=["network1.", "network2."])
s = torch.randn(rows, S, device=dev); a = torch.rand(rows, A, device=dev)
for _ in range(20):
    ops.mlp3_forward(qb, S + A, 1, 2, s, a, save=save)
torch.cuda.synchronize()
