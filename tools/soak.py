"""Long run of the product path: N gradient steps through the CLI mirror in graph mode on synthetic buffers, crossing the
5000-step model-rollout refreshes until the 1 000 000-row fake buffer wraps.  python tools/soak.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
from mobody_amd import train_mobody as tm

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 56000
t0 = time.time()
pol = tm.main(["--policy", "MOBODY", "--env", "walker2d-friction", "--shift_level", "2.0", "--seed", "3", "--synthetic", "1",
               "--rng", "device", "--penalty_type", "none", "--src_rows", "200000", "--tar_rows", "5000",
               "--params", '{"batch_size": 256, "max_step": %d, "graph": 1}' % steps, "--max_step", str(steps),
               "--log_every", "6000"])
torch.cuda.synchronize()
fb = pol.fake_replay_buffer
print(f"{steps} steps in {time.time() - t0:.1f} s; fake buffer size {fb.size} ptr {fb.ptr}; optimizer steps {pol.q_optimizer.t}/{pol.policy_optimizer.t};"
      f" graph {'on' if pol._graph is not None else 'off'}; losses {pol.losses()}")
assert pol.q_optimizer.t == steps and fb.size == min(1000000, 102000 * ((steps - 1) // 5000 + 1)) and all(x == x for x in pol.losses())
assert torch.isfinite(pol.policy.blob).all() and torch.isfinite(pol.q_funcs.blob).all() and torch.isfinite(fb.state).all()
print("soak ok")
