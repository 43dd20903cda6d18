"""Debug aid: gradients of one train step, f16x2 vs f32, per tensor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import golden_util as gu
from mobody_amd.engine import Engine
dev = torch.device("cuda:0")
S, A = 17, 6
N = int(sys.argv[1]) if len(sys.argv) > 1 else 640
Nt = N * 4 // 5
pa, pq, _ = gu.policy_params(77, S, A)
batch = gu.gi.batch(5, N, S, A)
res = {}
for mode in ("f32", "f16x2", "bf16x3"):
    cfg = gu.policy_cfg(S, A, mfma=mode)
    eng = Engine(S, A, pa, pq, dev)
    out = eng.step(batch, Nt, cfg, apply=False)
    res[mode] = (out, {("q", k): v.cpu().numpy() for k, v in eng.unpack(eng.gq, "q").items()} | {("a", k): v.cpu().numpy() for k, v in eng.unpack(eng.ga, "actor").items()})
for mode in ("f16x2", "bf16x3"):
    print(mode, res[mode][0], res["f32"][0])
    for k, v in res["f32"][1].items():
        d = np.abs(res[mode][1][k] - v).max() / max(np.abs(v).max(), 1e-30)
        print(f"  {k}: rel diff {d:.2e}  (max {np.abs(v).max():.3e})")
