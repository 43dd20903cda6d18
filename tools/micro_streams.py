"""Experiment: does splitting a launch chain into two half-batch chains on two streams (captured as parallel graph branches)
hide the fill/drain of launches that hold only ~2.5 tiles per CU?  Chain = K dependent twin-Q forwards (stream order).
    python tools/micro_streams.py [ROWS] [K]"""
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
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pa, pq, _ = gu.policy_params(1, S, A)
qb = packing.pack_mlp(pq, S + A, 1, dev, prefixes=["network1.", "network2."])
qT = ops.mlp_transpose(qb, S + A, 1, 2)
s = torch.randn(rows, S, device=dev); a = torch.rand(rows, A, device=dev)
out = torch.empty(2, rows, 1, device=dev)


def fwd(prec, r0, n, o):
    rc = lib.mobody_mlp3_forward(ptr(qb), ptr(qT), prec, S + A, 1, 2, s[r0:r0 + n].data_ptr(), S, a[r0:r0 + n].data_ptr(), A, n, 0, 1.0,
                                 o.data_ptr(), None, None, None, _lib.cur_stream())
    assert rc == 0, lib.mobody_last_error()


def timed(build, reps=20):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            build()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


side = torch.cuda.Stream()
o0, o1 = torch.empty(2, rows // 2, 1, device=dev), torch.empty(2, rows // 2, 1, device=dev)
od = torch.empty(2, rows, 1, device=dev)
for prec, name in ((0, "f32"), (3, "bf16x3")):
    fwd(prec, 0, rows, out); fwd(prec, 0, rows // 2, o0); torch.cuda.synchronize()

    def one():
        for _ in range(K):
            fwd(prec, 0, rows, out)

    def two():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(K):
                fwd(prec, rows // 2, rows // 2, o1)
        for _ in range(K):
            fwd(prec, 0, rows // 2, o0)
        cur.wait_stream(side)

    def two_offset(delay_rows):
        def build():
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                fwd(prec, 0, delay_rows, od)                  # phase offset: a short forward first
                for _ in range(K):
                    fwd(prec, rows // 2, rows // 2, o1)
            for _ in range(K):
                fwd(prec, 0, rows // 2, o0)
            cur.wait_stream(side)
        return build

    def two_serial():
        for _ in range(K):
            fwd(prec, 0, rows // 2, o0)
        for _ in range(K):
            fwd(prec, rows // 2, rows // 2, o1)

    print(f"{name}: chain of {K} forwards over {rows} rows: one stream {timed(one):7.1f} us | two half-batch chains on two streams "
          f"{timed(two):7.1f} us | the two half chains back to back {timed(two_serial):7.1f} us | two streams, the second one "
          f"offset by a forward over 1024 / 2560 / 4096 rows: {timed(two_offset(1024)):7.1f} / {timed(two_offset(2560)):7.1f} / {timed(two_offset(4096)):7.1f} us")
