"""Micro-benchmark of the fused MLP forward (mobody_mlp3_forward) in every MFMA mode: twin-Q at a few row counts, with and
without the saved activations (x, h1, h2) the weight-gradient kernel reads, accuracy against the fp32 kernel's output.
    python tools/micro_fwd_bf.py [ROWS ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
qT = {p: ops.mlp_transpose(qb, S + A, 1, 2, precision=p) for p in (3, 4)}
L = _lib.mlp_layout(S + A, 1, 2)


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()                     # graph replay: the kernels back to back, no host launch gaps
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rows in [int(x) for x in sys.argv[1:]] or (2560, 10240, 40960):
    s = torch.randn(rows, S, device=dev); a = torch.rand(rows, A, device=dev)
    out = torch.empty(2, rows, 1, device=dev)
    sx = torch.empty(rows, L.Kp1, device=dev); sh1 = torch.empty(2, rows, 256, device=dev); sh2 = torch.empty(2, rows, 256, device=dev)

    def fwd(prec, save):
        rc = lib.mobody_mlp3_forward(ptr(qb), ptr(qT[4 if prec == 4 else 3]), prec, S + A, 1, 2, ptr(s), S, ptr(a), A, rows, 0, 1.0, ptr(out),
                                     ptr(sx if save else None), ptr(sh1 if save else None), ptr(sh2 if save else None), _lib.cur_stream())
        assert rc == 0, lib.mobody_last_error()

    fwd(0, False); torch.cuda.synchronize(); ref = out.clone()
    line = f"rows {rows:6d}:"
    for prec, name in enumerate(("f32", "bf16", "bf16x2", "bf16x3", "f16x2")):
        fwd(prec, False); torch.cuda.synchronize()
        err = float((out - ref).abs().max() / ref.abs().max())
        line += f" | {name} {timeit(lambda: fwd(prec, False)):6.1f} us, saving {timeit(lambda: fwd(prec, True)):6.1f} us, err {err:.1e}"
    print(line)
