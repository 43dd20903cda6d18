"""Micro-benchmark of the split-precision forward (mobody_mlp3_forward_bf) against the fp32 forward: twin-Q at a few row
counts, every mode, accuracy against the fp32 kernel's output."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import golden_util as gu
from mobody_amd import _lib, ops, packing

dev = torch.device("cuda:0")
lib = _lib.load()
S, A = 17, 6
pa, pq, _ = gu.policy_params(1, S, A)
qb = packing.pack_mlp(pq, S + A, 1, dev, prefixes=["network1.", "network2."])
planes = torch.zeros(2 * 3 * 256 * 256, dtype=torch.bfloat16, device=dev)
vp = C.c_void_p
lib.mobody_mlp_w2_planes.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp]
lib.mobody_mlp3_forward_bf.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int64, C.c_int, C.c_float, vp, vp]
assert lib.mobody_mlp_w2_planes(S + A, 1, 2, qb.data_ptr(), planes.data_ptr(), _lib.cur_stream()) == 0


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


for rows in [int(x) for x in sys.argv[1:]] or (2560, 10240, 40960):
    s = torch.randn(rows, S, device=dev); a = torch.rand(rows, A, device=dev)
    ref = ops.mlp3_forward(qb, S + A, 1, 2, s, a)
    t32 = timeit(lambda: ops.mlp3_forward(qb, S + A, 1, 2, s, a))
    line = f"rows {rows:6d}: f32 {t32:7.1f} us"
    for prec in (1, 2, 3):
        out = torch.empty(2, rows, 1, device=dev)
        f = lambda: lib.mobody_mlp3_forward_bf(qb.data_ptr(), planes.data_ptr(), prec, S + A, 1, 2, s.data_ptr(), S, a.data_ptr(), A,
                                              rows, 0, 1.0, out.data_ptr(), _lib.cur_stream())
        assert f() == 0, lib.mobody_last_error()
        torch.cuda.synchronize()
        err = float((out - ref).abs().max() / ref.abs().max())
        line += f" | prec{prec} {timeit(f):7.1f} us err {err:.1e}"
    print(line, " RG", os.environ.get("MOBODY_BF_RG", "2"))
