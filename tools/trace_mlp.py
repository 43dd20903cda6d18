"""Phase timeline of the fused MLP kernels' workgroups (diagnostic).  Needs a trace build of the library:
    MOBODY_TRACE=1 python mobody_amd/csrc/build.py --force      (rebuild without the variable afterwards)
    python tools/trace_mlp.py fwd|critic|actor [ROWS]
fwd: twin-Q forward (2 members); critic: mobody_critic_step (the trace left behind is its backward);
actor: mobody_actor_forward + mobody_actor_backward (workgroups >= tiles: frozen-Q dX backward, < tiles: actor backward)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import golden_util as gu
from mobody_amd import ops, packing, _lib
from mobody_amd.engine import Engine

dev = torch.device("cuda:0")
S, A = 17, 6
mode = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10240
Nt = N * 4 // 5
pa, pq, _ = gu.policy_params(1, S, A)
eng = Engine(S, A, pa, pq, dev)
cfg = gu.policy_cfg(S, A, mfma=os.environ.get('MOBODY_MFMA', 'f32'))
b = [torch.as_tensor(x, dtype=torch.float32).to(dev).contiguous() for x in gu.gi.batch(3, N, S, A)]
dims, hyp = ops.train_dims(S, A, N, Nt), ops.hyper(cfg)
ws = ops.train_workspace(dims, dev)


def run():
    if mode == "fwd":
        # (the pipelined f16x2 kernel carries no fp32 copy of layer 1: traced without the saves)
        ops.mlp3_forward(eng.q, S + A, 1, 2, b[0], b[1], save=not int(os.environ.get('PIPE_MT', '0')), blob_T=eng.q_T, precision=cfg['mfma'])
    elif mode == "critic":
        ops.critic_step(dims, hyp, eng.actor, eng.q, eng.q_T, eng.qt, b, eng.gq, eng.loss[0:1], ws, actor_blob_T=eng.actor_T, qtarg_blob_T=eng.qt_T)
    else:
        ops.actor_forward(dims, hyp, eng.actor, eng.q, b[0], b[1], eng.stats, ws, actor_blob_T=eng.actor_T, q_blob_T=eng.q_T)
        ops.actor_backward(dims, hyp, eng.actor, eng.actor_T, eng.q, eng.q_T, b[0], b[1], eng.stats, eng.ga, eng.loss[1:3], ws)


for _ in range(5):
    run()
torch.cuda.synchronize()
run()
torch.cuda.synchronize()
tiles = (N + 31) // 32
nb = tiles * 2
buf = (C.c_ulonglong * (nb * 8))()
lib = _lib.load()
assert lib.mobody_debug_trace(buf, nb * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
FWD = ["start", "input in LDS", "layer1 done", "layer2 gemm done", "layer2 done", "end"]
PIPE = int(os.environ.get("PIPE_MT", "0"))        # trace of k_mlp3_fwd_pipe built with -DFWD_PIPE_MT=PIPE (mode fwd)
if PIPE and mode == "fwd":
    nb = ((N + 64 * PIPE - 1) // (64 * PIPE)) * 2
    t = t[:nb]
    FWD = ["start", "inputs in LDS", "layer1 P,Q + maxima", "epi1(P)", "gemm2(P) | epi1(Q)", "gemm2(Q) | epi2(P)", "epi2(Q)", "end"]
BWD = ["start", "seed in LDS", "W3T gemm done", "mask epi 1 done", "W2T gemm done", "mask epi 2 done", "end (dX)"]


def report(title, tt, names):
    us = (tt[:, :len(names)] - tt[:, 0].min()) / 100.0            # wall_clock64: 100 MHz
    print(f"{title}: {len(tt)} workgroups, last end {us[:, len(names) - 1].max():.1f} us")
    for k, n in enumerate(names):
        print(f"  {n:18s} mean {us[:, k].mean():6.2f}  min {us[:, k].min():6.2f}  max {us[:, k].max():6.2f}")
    d = np.diff(us, axis=1)
    print("  phase durations mean/max:", ", ".join(f"{names[k + 1]}: {d[:, k].mean():.2f}/{d[:, k].max():.2f}" for k in range(len(names) - 1)))


if mode == "fwd":
    report("twin-Q forward", t, FWD)
    if cfg["mfma"] != "f32" and not PIPE:         # slots 6, 7: layer-1 GEMM done / layer-1 epilogue (plane split + h1 save) done
        us = (t - t[:, 0].min()) / 100.0
        print(f"  layer 1 (split-precision tile): gemm done {us[:, 6].mean():.2f}, epilogue done {us[:, 7].mean():.2f}, "
              f"barrier passed {us[:, 2].mean():.2f}")
elif mode == "critic":
    report("critic backward", t, BWD)
else:
    report("actor backward (1 member)", t[:tiles], BWD[:6])
    report("frozen-Q dX backward (member 1)", t[tiles:], BWD)

st = (t[:, 0] - t[:, 0].min()) / 100.0
print("start-time histogram (us: workgroups):", {f"<{e}": int(((st >= b0) & (st < e)).sum()) for b0, e in ((0, 2), (2, 10), (10, 20), (20, 30), (30, 100))})
