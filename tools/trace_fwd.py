"""Phase timeline of k_mlp3_fwd workgroups (needs a -DMOBODY_TRACE build of mlp_fwd.hip; diagnostic only).
python tools/trace_fwd.py ROWS MEMBERS [save]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import golden_util as gu
from mobody_amd import ops, packing, _lib
dev = torch.device("cuda:0")
S, A = 17, 6
rows = int(sys.argv[1]); members = int(sys.argv[2]); save = len(sys.argv) > 3
pa, pq, _ = gu.policy_params(1, S, A)
if members == 2:
    blob = packing.pack_mlp(pq, S + A, 1, dev, prefixes=["network1.", "network2."]); ind, outd = S + A, 1
else:
    blob = packing.pack_mlp([{k[len("network."):]: v for k, v in pa.items()}], S, A, dev); ind, outd = S, A
s = torch.randn(rows, S, device=dev); a = torch.rand(rows, A, device=dev)
run = (lambda: ops.mlp3_forward(blob, ind, outd, 2, s, a, save=save)) if members == 2 else (lambda: ops.mlp3_forward(blob, ind, outd, 1, s, save=save))
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
nb = (rows + 31) // 32 * members
buf = (C.c_ulonglong * (nb * 8))()
lib = _lib.load()
assert lib.mobody_debug_trace(buf, nb * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8)[:, :6].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0            # wall_clock64: 100 MHz
print(f"rows {rows} members {members} blocks {nb}: launch {e0.elapsed_time(e1)*1e3:.1f} us (event)")
names = ["start", "input in LDS", "layer1 done", "layer2 gemm done", "layer2 done", "end"]
for k, n in enumerate(names):
    print(f"  {n:18s} mean {us[:,k].mean():6.2f}  min {us[:,k].min():6.2f}  max {us[:,k].max():6.2f}")
d = np.diff(us, axis=1)
print("  phase durations (mean/max):", ", ".join(f"{n}: {d[:,k].mean():.2f}/{d[:,k].max():.2f}" for k, n in enumerate(["load", "L1", "L2 gemm", "L2 epi", "L3"])))
late = us[:, 0] > 3.0
print(f"  blocks starting later than 3 us: {late.sum()} (mean start {us[late,0].mean() if late.any() else 0:.1f})")
