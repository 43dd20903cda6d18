"""Debug aid: decode the h1 / dz2 fp16 planes a critic step leaves in the workspace and compare with torch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import golden_util as gu
from mobody_amd import ops, _lib
from mobody_amd.engine import Engine
from oracle import mobody_oracle as O
dev = torch.device("cuda:0")
S, A, N = 17, 6, 640
Nt = 512
pa, pq, _ = gu.policy_params(77, S, A)
batch = gu.gi.batch(5, N, S, A)
cfg = gu.policy_cfg(S, A, mfma="f16x2")
eng = Engine(S, A, pa, pq, dev)
b = [torch.as_tensor(x, dtype=torch.float32).to(dev).contiguous() for x in batch]
dims, hyp = ops.train_dims(S, A, N, Nt), ops.hyper(cfg)
eng.step(batch, Nt, cfg, apply=False)   # rebuilds T blobs for f16
ws = ops.train_workspace(dims, dev); ws.zero_()
ops.critic_step(dims, hyp, eng.actor, eng.q, eng.q_T, eng.qt, b, eng.gq, eng.loss[0:1], ws, actor_blob_T=eng.actor_T, qtarg_blob_T=eng.qt_T)
torch.cuda.synchronize()
Lq, La = _lib.mlp_layout(S + A, 1, 2), _lib.mlp_layout(S, A, 1)
off = 0
def take(n):
    global off
    o = off; off += (n + 3) & ~3
    return o
N32 = (N + 31) & ~31
o = {}
for name, n in (("pi", N * A), ("pin", N * A), ("qt", 2 * N), ("q", 2 * N), ("qb", 2 * Nt), ("xq", N * Lq.Kp1), ("h1q", 2 * N32 * 256),
                ("h2q", 2 * N * 256), ("xa", N * La.Kp1), ("h1a", N32 * 256), ("h2a", N * 256)):
    o[name] = take(n)
mw = ((N + 31) // 32) * 256
for name, n in (("mq1", 2 * mw), ("mq2", 2 * mw), ("ma1", mw), ("ma2", mw), ("eh1q", 2 * (N32 // 32)), ("eh1a", N32 // 32), ("edz2", 2 * (N32 // 32)),
                ("dz3q", 2 * N * Lq.Np3), ("dz2", 2 * N32 * 256)):
    o[name] = take(n)
w = ws.cpu()
def planes(name, members):
    raw = w[o[name]:o[name] + members * N32 * 256].view(torch.int16).view(members, 2, N32 // 8, 256, 8).view(torch.float16).float()
    return raw.permute(0, 1, 2, 4, 3).reshape(members, 2, N32, 256)
h1p = planes("h1q", 2)
e = w[o["eh1q"]:o["eh1q"] + 2 * (N32 // 32)].view(torch.int32).view(2, N32 // 32)
print("eh1q", e[0, :8].tolist(), e[1, :8].tolist())
h1 = (h1p[:, 0] + h1p[:, 1]) * torch.exp2(-e.float()).repeat_interleave(32, dim=1)[:, :, None]
P = O.to_torch(pq)
x = torch.cat([O.T(batch[0]), O.T(batch[1])], 1)
for m in (0, 1):
    want = torch.relu(torch.nn.functional.linear(x, P[f"network{m+1}.network.0.weight"], P[f"network{m+1}.network.0.bias"]))
    d = (h1[m, :N] - want).abs().max() / want.abs().max()
    print("h1 member", m, "rel err", float(d))
dzp = planes("dz2", 2)
e2 = w[o["edz2"]:o["edz2"] + 2 * (N32 // 32)].view(torch.int32).view(2, N32 // 32)
print("edz2", e2[0, :8].tolist(), e2[1, :8].tolist())
dz2 = (dzp[:, 0] + dzp[:, 1]) * torch.exp2(-e2.float()).repeat_interleave(32, dim=1)[:, :, None]
# reference dW2 from decoded planes vs engine grad
g = eng.unpack(eng.gq, "q")
for m in (0, 1):
    dW2 = h1[m, :N].double().T @ dz2[m, :N].double()          # [k][n] ; nn.Linear weight is [n][k]
    got = g[f"network{m+1}.network.2.weight"].cpu().double().T
    print("member", m, "dW2 from planes vs kernel: rel", float((dW2 - got).abs().max() / dW2.abs().max()), "max", float(dW2.abs().max()))
m = 0
dW2 = h1[m, :N].double().T @ dz2[m, :N].double()
got = g["network1.network.2.weight"].cpu().double().T
print("got[:3,:6]", got[:3, :6].numpy()); print("want[:3,:6]", dW2[:3, :6].numpy())
# candidates
A0, A1 = h1p[m, 0, :N].double(), h1p[m, 1, :N].double()
B0, B1 = dzp[m, 0, :N].double(), dzp[m, 1, :N].double()
sc = torch.exp2(-(e[m].float() + e2[m].float())).repeat_interleave(32)[:N].double()
def rel(c): return float((c - got).abs().max() / got.abs().max())
print("a0b0 only", rel((A0 * sc[:, None]).T @ B0))
print("no a1b0", rel((A0 * sc[:, None]).T @ (B0 + B1)))
print("full", rel(((A0 + A1) * sc[:, None]).T @ (B0 + B1)))
for lo, hi in ((0, 16), (16, 32)):
    mask = torch.zeros(N, dtype=torch.bool)
    for t in range(0, N, 32): mask[t + lo:t + hi] = True
    print("rows", lo, hi, "of each tile only", rel(((A0 + A1) * sc[:, None] * mask[:, None]).T @ (B0 + B1)))
for w in range(8):
    mask = torch.zeros(N, dtype=torch.bool); mask[w * 80:(w + 1) * 80] = True
    print("wave", w, rel(((A0 + A1) * sc[:, None] * mask[:, None]).T @ (B0 + B1)))
