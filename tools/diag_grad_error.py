"""Diagnostic: error of (a) the fp32 oracle and (b) the HIP path against an fp64 evaluation of the same step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import golden_util as gu
from oracle import mobody_oracle as O
from mobody_amd.engine import Engine

# usage: diag_grad_error.py [S A N Nt]   (default: the golden g7 batch, bs=32)
if len(sys.argv) == 5:
    S, A, N, n_true = (int(x) for x in sys.argv[1:5])
    cfg = gu.policy_cfg(S, A)
    pa, pq, pv = gu.policy_params(91, S, A)
    batch = gu.gi.batch(17, N, S, A)
else:
    S, A, bs = 17, 6, 32
    cfg = gu.policy_cfg(S, A)
    pa, pq, pv = gu.policy_params(401, S, A)
    batch, n_true = gu.g7_batch(cfg, bs, S, A)
st32 = O.TrainState(pa, pq, pv); o32 = O.train_step(st32, batch, n_true, cfg, apply=False)
T32 = O.T
O.T = lambda x, dtype=torch.float64: (x.to(torch.float64) if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x), dtype=torch.float64))
st64 = O.TrainState(pa, pq, pv); o64 = O.train_step(st64, batch, n_true, cfg, apply=False)
O.T = T32
eng = Engine(S, A, pa, pq, torch.device("cuda:0")); got = eng.step(batch, n_true, cfg, apply=False)
for k in ("q_loss", "pi_loss", "bc_loss"):
    t = float(o64[k]); print(f"{k}: fp64 {t:.9f}  ref32 rel err {abs(float(o32[k])-t)/abs(t):.2e}  hip rel err {abs(got[k]-t)/abs(t):.2e}")
w64 = o64["bc_w"].numpy().ravel(); w32 = o32["bc_w"].numpy().ravel().astype(np.float64)
print("bc weights: ref32 max rel err", np.abs(w32-w64).max()/1, "w range", w64.min(), w64.max())
for nm, blob, key in (("q", eng.gq, "q_grads"), ("actor", eng.ga, "actor_grads")):
    for k, v in eng.unpack(blob, nm).items():
        t = o64[key][k].numpy(); r = o32[key][k].numpy().astype(np.float64); h = v.cpu().numpy().astype(np.float64)
        mx = np.abs(t).max()
        print(f"{nm:5s} {k:28s} max|g|={mx:.2e}  ref32 err: max {np.abs(r-t).max()/mx:.2e} | hip err: max {np.abs(h-t).max()/mx:.2e}   (relative to max|g|)")
