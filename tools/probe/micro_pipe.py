"""A/B of the f16x2 twin-Q forward (no saves) under graph replay: run after tools/ab_build.sh mlp_fwd_pipe.hip (file copied into csrc/) "-DFWD_PIPE_MT=.."
    python tools/probe/micro_pipe.py [ROWS ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import golden_util as gu
from mobody_amd import _lib, ops, packing
from mobody_amd._lib import ptr

dev = torch.device("cuda:0")
lib = _lib.load()
S, A = 17, 6
pa, pq, _ = gu.policy_params(1, S, A)
qb = packing.pack_mlp(pq, S + A, 1, dev, prefixes=["network1.", "network2."])
qT = ops.mlp_transpose(qb, S + A, 1, 2, precision=4)


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


line = os.environ.get("TAG", "") + ":"
for rows in [int(x) for x in sys.argv[1:]] or (10240, 15360, 40960):
    s = torch.randn(rows, S, device=dev); a = torch.rand(rows, A, device=dev)
    out = torch.empty(2, rows, 1, device=dev)

    def fwd():
        rc = lib.mobody_mlp3_forward(ptr(qb), ptr(qT), 4, S + A, 1, 2, ptr(s), S, ptr(a), A, rows, 0, 1.0, ptr(out), None, None, None, _lib.cur_stream())
        assert rc == 0, lib.mobody_last_error()

    line += f"  rows {rows}: {timeit(fwd):6.1f} us"
print(line)
