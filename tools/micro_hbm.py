"""HBM-side kernels of the rollout / minibatch path at sizes where launch latency no longer dominates: achieved GB/s of
ALGORITHMIC bytes (rows x (2S+A+2) x 4 B each way for gather / append; (7S + S + 2) x 4 B per row for the sample kernel)
against the 8 TB/s HBM3E peak.  One JSON line.   python tools/micro_hbm.py [ring|arrays|both]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from mobody_amd import ops

dev = torch.device("cuda:0")
S, A = 17, 6
PEAK = 8000.0


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


out = {}
cap = 4_000_000
fields = (torch.randn(cap, S, device=dev), torch.randn(cap, A, device=dev), torch.randn(cap, S, device=dev),
          torch.randn(cap, 1, device=dev), torch.ones(cap, 1, device=dev))
ring = ops.RingView(torch.zeros(cap, ops.ring_pitch(S, A), device=dev), S, A)
for v, t in zip(ring, fields):
    v.copy_(t)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for layout, buf in (("ring", ring), ("arrays", fields)):          # row-interleaved ring (ReplayBuffer) / five separate arrays
    if which not in ("both", layout):
        continue
    for rows in (10240, 52000, 1_000_000):
        idx = torch.randint(cap, (rows,), device=dev, dtype=torch.int32)
        dst = tuple(torch.empty(rows, t.shape[1], device=dev) for t in fields)
        t = timeit(lambda: ops.gather_batch([buf], [idx], S, A, out=dst))
        byt = rows * (2 * S + A + 2) * 4 * 2
        out[f"k_gather_{layout}_{rows}"] = dict(us=t * 1e6, algorithmic_GBps=byt / t / 1e9, frac_of_hbm_peak=byt / t / 1e9 / PEAK)
        ps = torch.zeros(2, dtype=torch.int64, device=dev)
        term = torch.zeros(rows, 1, dtype=torch.uint8, device=dev)
        keep = (torch.rand(rows, device=dev) < 0.9).to(torch.uint8)
        t = timeit(lambda: ops.ring_append(buf, cap, ps, S, A, dst[0], dst[1], dst[2], dst[3], term, keep))
        byt = int(rows * 0.9) * (2 * S + A + 2) * 4 * 2
        out[f"ring_append_{layout}_{rows}"] = dict(us=t * 1e6, algorithmic_GBps=byt / t / 1e9, frac_of_hbm_peak=byt / t / 1e9 / PEAK)
print(json.dumps(out))
