"""Pre-training step time, HIP-graph replay against eager launches (the side stream is a build flag of csrc/pretrain.hip:
bash tools/ab_build.sh pretrain.hip "-DPRE_SIDE_STREAM=0").  GPU box:  TAG=label python tools/pre_ab.py"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mobody_amd import engine, synthetic
from mobody_amd.algo.dynamics.mobody_dynamics import MOBODYEnsembleDynamics
from mobody_amd.algo.dynamics.mobody_module import MOBODYModule
from mobody_amd.algo.mb_utils.terminal_funs import get_termination_fn
S, A, b, steps = 17, 6, 256, 300
dev = torch.device("cuda:0")
for graph in (1, 0):
    cfg = engine.default_config(S, A, no_vae=0, inverse_sep_reward_loss=0, train_together=0, train_with_src_threshold=1, dynamics_lr=1e-3, mfma="f16x2", train_graph=graph)
    m = MOBODYModule(S, A, 256, 7, 5, device=dev, config=cfg)
    dyn = MOBODYEnsembleDynamics(cfg, m, None, None, get_termination_fn("walker2d-medium-v2"), penalty_coef=0.1, rng="device", seed=1)
    g = torch.Generator().manual_seed(0)
    n = 200000
    data = [torch.randn(n, S, generator=g).to(dev), (torch.rand(n, A, generator=g) * 2 - 1).to(dev), torch.randn(n, S, generator=g).to(dev), torch.randn(n, 1, generator=g).to(dev)]
    idx = torch.randint(n, (7, steps * b), generator=g).to(device=dev, dtype=torch.int32).contiguous()
    dyn._learn_indexed(True, data, idx[:, :5 * b].contiguous(), b)
    dyn._learn_indexed(True, data, idx, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dyn._learn_indexed(True, data, idx, b)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(os.environ.get("TAG", ""), "graph" if graph else "eager", f"{dt / steps * 1e3:.4f} ms/step", flush=True)
