"""Dynamics pre-training throughput on the GPU box: optimizer steps per second of
MOBODYEnsembleDynamics.learn at the reference's batch size (256 rows per member, 7 members), walker2d shapes by default,
through the mirror (`_learn_indexed`: bootstrap gather + mobody_pretrain_grads + mobody_pretrain_adam), next to the CPU
oracle (`oracle.dyn_learn_step`, torch CPU fp32 autograd).  One JSON line.

FLOPs per step (useful): per member and row, forward MACs of the three big nets are enc 2x, dec 4x, reward 2x
(S*256+65536+8192 | 4096+65536+256*S | (2S+A)*256+65536+512); forward + backward = 3x  ->  x 7 members x b rows x 2."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def flops_per_step(S, A, b):
    enc = S * 256 + 65536 + 256 * 32
    dec = 16 * 256 + 65536 + 256 * S
    rw = (2 * S + A) * 256 + 65536 + 512
    return 2.0 * 3.0 * 7 * b * (2 * enc + 4 * dec + 2 * rw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=17); ap.add_argument("--A", type=int, default=6)
    ap.add_argument("--b", type=int, default=256); ap.add_argument("--rows", type=int, default=200000)
    ap.add_argument("--steps", type=int, default=300); ap.add_argument("--no_cpu", action="store_true")
    ap.add_argument("--mfma", default="f16x2", choices=["f32", "f16x2"])
    args = ap.parse_args()
    from mobody_amd import engine, synthetic
    from mobody_amd.algo.dynamics.mobody_dynamics import MOBODYEnsembleDynamics
    from mobody_amd.algo.dynamics.mobody_module import MOBODYModule
    from mobody_amd.algo.mb_utils.terminal_funs import get_termination_fn
    S, A, b = args.S, args.A, args.b
    dev = torch.device("cuda:0")
    task = "walker2d-medium-v2" if S == 17 else "ant-medium-v2" if S == 111 else "pen-human-v1"
    cfg = engine.default_config(S, A, no_vae=0, inverse_sep_reward_loss=0, train_together=0, train_with_src_threshold=1, dynamics_lr=1e-3,
                                mfma=args.mfma)
    m = MOBODYModule(S, A, 256, 7, 5, device=dev, config=cfg)
    dyn = MOBODYEnsembleDynamics(cfg, m, None, None, get_termination_fn(task), penalty_coef=0.1, rng="device", seed=1)
    g = torch.Generator().manual_seed(0)
    mu = torch.from_numpy(synthetic.alive_mean(task, S))
    n = args.rows
    data = [(mu + 0.1 * torch.randn(n, S, generator=g)).to(dev), (torch.rand(n, A, generator=g) * 2 - 1).to(dev),
            (mu + 0.1 * torch.randn(n, S, generator=g)).to(dev), torch.randn(n, 1, generator=g).to(dev)]
    idx = torch.randint(n, (7, args.steps * b), generator=g).to(device=dev, dtype=torch.int32).contiguous()
    warm = idx[:, :20 * b].contiguous()
    dyn._learn_indexed(True, data, warm, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = dyn._learn_indexed(True, data, idx, b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = dict(metric="dynamics pre-training optimizer steps/sec", S=S, A=A, rows_per_member=b, steps=args.steps,
               steps_per_sec=args.steps / dt, ms_per_step=dt / args.steps * 1e3, samples_per_sec=args.steps * b / dt,
               useful_tflops=flops_per_step(S, A, b) * args.steps / dt / 1e12, frac_f32_mfma_peak=flops_per_step(S, A, b) * args.steps / dt / 157.3e12,
               mean_losses=stats)
    if not args.no_cpu:
        from oracle import mobody_oracle as O
        threads = min(16, len(os.sched_getaffinity(0)))
        torch.set_num_threads(threads)
        rng = np.random.default_rng(0)
        p = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
        st = O.DynTrainState(p)
        rows = [x[:7 * b].reshape(7, b, -1).cpu().numpy() for x in data]
        nz = [rng.standard_normal((7, b, 16)).astype(np.float32) for _ in range(6)] + [rng.standard_normal((7, b, S)).astype(np.float32)]
        O.dyn_learn_step(st, *rows, nz, True)
        t0 = time.time(); k = 0
        while time.time() - t0 < 8.0 or k < 3:
            O.dyn_learn_step(st, *rows, nz, True); k += 1
        out["cpu_baseline"] = dict(steps_per_sec=k / (time.time() - t0), cores=threads, kind="port",
                                   sample=f"{k} oracle learn steps (torch CPU fp32 autograd, {threads} threads)")
        out["gpu_over_cpu"] = out["steps_per_sec"] / out["cpu_baseline"]["steps_per_sec"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
