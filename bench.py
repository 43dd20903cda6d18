#!/usr/bin/env python
"""MOBODY hot-path benchmark (contract: one JSON line on rank 0).

Workload (BASELINE.json configs[1]): walker2d-friction shapes S=17, A=6, ensemble 7, rollout_len 1,
batch_size 4096 per GPU -> every `MOBODY.train()` step consumes N = 2.5*4096 = 10 240 synthetic
transitions (4096 source + 4096 target + 2048 model-generated rows) already resident in HBM:
replay gather -> twin-Q TD update (+Adam, Polyak) -> Q-scaled actor + Q-weighted-BC update (+Adam).
The model-rollout refresh (50 000 + 2 000 + 50 000 imagined transitions every 5000 steps,
mobody.py:441-475) runs in the warm-up (step 1) and is measured separately below.

  value  = minibatch transitions consumed per second by the timed K train() steps, summed over ranks
           (weak scaling: every rank draws its own 4096-row minibatch; gradients and the two actor
           statistics are all-reduced over RCCL each step)
  extras = grad_steps_per_sec (train() calls/s), rollout_transitions_per_sec (actor + ensemble
           step on 50 000 rows), per-kernel-family roofline, CPU baseline (oracle on host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

S, A, BS, TASK = 17, 6, 4096, "walker2d-medium-v2"
PEAK_F32_TFLOPS = 157.3          # MI355X dense fp32 MFMA peak (MI355X_MICROARCH.md)
FAMILIES = ["k_mlp3_fwd", "k_mlp3_bwd", "k_wgrad", "k_dyn_fwd"]


def macs():
    actor = S * 256 + 65536 + 256 * A
    q = (S + A) * 256 + 65536 + 256
    dyn = 7 * (S * 256 + 65536 + 256 * 32 + (16 + A) * 32 + 32 * 32 + 16 * 256 + 65536 + 256 * S
               + (2 * S + A) * 256 + 65536 + 512)
    return actor, q, dyn


def train_flops(N, Nt):
    """Useful (executed) FLOPs of one train() step per kernel family; SURVEY 8(d) counts two more
    forwards that the reference executes but never uses (Q7), which this build skips."""
    actor, q, _ = macs()
    fwd = N * (2 * actor + 6 * q) + Nt * 2 * q
    bwd = N * (2 * (256 + 65536) + 2 * (256 + 65536 + 256 * A) + (256 * A + 65536))
    wg = N * (2 * q + actor)
    return {"k_mlp3_fwd": 2.0 * fwd, "k_mlp3_bwd": 2.0 * bwd, "k_wgrad": 2.0 * wg}


def build(dev, rank, bs, graph):
    from mobody_amd import synthetic
    from mobody_amd.algo import utils
    from mobody_amd.algo.call_algo import call_algo
    from mobody_amd.algo.dynamics.mobody_dynamics import MOBODYEnsembleDynamics
    from mobody_amd.algo.dynamics.mobody_module import MOBODYModule
    from mobody_amd.algo.mb_utils.terminal_funs import get_termination_fn
    import golden_util as gu
    cfg = gu.policy_cfg(S, A, rng="device", seed=rank, penalty_type="none", batch_size=bs, graph=graph)
    torch.manual_seed(rank); np.random.seed(rank)
    pol = call_algo("mobody", cfg, 3, dev)
    src = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=1000000, rng="device", seed=100 + rank), 1000000, TASK, rank)
    tar = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=5000, rng="device", seed=200 + rank), 5000, TASK, 1000 + rank)
    model = synthetic.alive_dynamics(MOBODYModule(S, A, 256, 7, 5, device=dev, config=cfg), TASK)
    pol.dynamics = MOBODYEnsembleDynamics(cfg, model, None, None, get_termination_fn(TASK), penalty_coef=0.1, rng="device", seed=300 + rank)
    return pol, src, tar, cfg


def prof_pass(pol, src, tar, bs, steps):
    """Instrumented pass: HIP event pairs around every launch of the heavy kernel families (library hook).
    Runs eagerly (event records are not part of a captured graph); the kernels are the same."""
    from mobody_amd import _lib
    lib = _lib.load()
    pol.use_graph = 0
    _lib.check(lib.mobody_prof_begin(steps * 64), "prof_begin")
    for _ in range(steps):
        pol.train(src, tar, bs, None, None)
    ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
    _lib.check(lib.mobody_prof_end(ms, cnt, 8), "prof_end")
    return {FAMILIES[i]: (ms[i] / steps, cnt[i] / steps) for i in range(3)}


def rollout_rate(pol, src, reps=5, B=50000):
    """Imagined transitions per second: actor forward + fused ensemble step on B rows (events on torch's stream,
    which is the stream every kernel of the library is launched on)."""
    from mobody_amd import _lib, ops
    lib = _lib.load()
    obs = src.state[:B].contiguous()
    for _ in range(2):
        pol.dynamics.step_device(obs, pol.policy(obs))
    torch.cuda.synchronize()
    _lib.check(lib.mobody_prof_begin(reps * 8), "prof_begin")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pol.dynamics.step_device(obs, pol.policy(obs))
    e1.record(); torch.cuda.synchronize()
    ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
    _lib.check(lib.mobody_prof_end(ms, cnt, 8), "prof_end")
    total_ms = e0.elapsed_time(e1) / reps
    return B / (total_ms * 1e-3), total_ms, ms[3] / reps


def cpu_baseline(cfg, bs, budget_s=20.0):
    """The CPU oracle (PyTorch CPU ops, the reference's algorithm) on this host's cores, same shapes."""
    from oracle import mobody_oracle as O
    import golden_util as gu
    # the GPU box gives one GPU a 16-core CPU share (sched_getaffinity still lists every core of the host)
    threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(threads)
    N, Nt = int(2.5 * bs), 2 * bs
    pa, pq, pv = gu.policy_params(1, S, A)
    st = O.TrainState(pa, pq, pv)
    batch = gu.gi.batch(3, N, S, A)
    O.train_step(st, batch, Nt, cfg)                       # warm-up
    t0 = time.time(); n = 0
    while time.time() - t0 < budget_s * 0.6 or n < 2:
        O.train_step(st, batch, Nt, cfg); n += 1
    dt_train = (time.time() - t0) / n
    p = O.to_torch(gu.gi.dyn_params(2, S, A))
    rng = np.random.default_rng(0)
    B = 4096
    obs = gu.gi.walker_like_obs(rng, B, S); act = rng.uniform(-1, 1, (B, A)).astype(np.float32)
    eps = rng.standard_normal((7, B, S)).astype(np.float32); idx = rng.integers(0, 5, B)
    with torch.no_grad():
        O.dyn_step(p, obs, act, eps, idx, TASK, 0.1)
        t1 = time.time(); m = 0
        while time.time() - t1 < budget_s * 0.3 or m < 2:
            O.dyn_step(p, obs, act, eps, idx, TASK, 0.1); m += 1
    dt_step = (time.time() - t1) / m
    return dict(value=N / dt_train, unit="transitions/s", cores=threads, kind="port",
                sample=f"{n} oracle train steps at N={N} rows (S={S},A={A}) + {m} oracle dyn_step calls at B={B}; "
                       f"torch CPU fp32, {threads} threads",
                grad_steps_per_sec=1.0 / dt_train, rollout_transitions_per_sec=B / dt_step)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch_size", type=int, default=BS)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=1, help="HIP-graph replay of the steady-state step: 0 never, 1 always, 2 auto (minibatches under 4096 rows); with N > 1 ranks the segments between the three all-reduces are replayed")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        # RCCL ("nccl") over xGMI in production; MOBODY_BENCH_BACKEND=gloo only to rehearse the multi-process
        # path with several ranks sharing one GPU (RCCL refuses duplicate devices)
        backend = os.environ.get("MOBODY_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
    bs = args.batch_size
    N, Nt = int(2.5 * bs), 2 * bs
    pol, src, tar, cfg = build(dev, rank, bs, args.graph)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):                  # step 1 includes the model-rollout refresh
        pol.train(src, tar, bs, None, None)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pol.train(src, tar, bs, None, None)
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    losses = pol.losses()

    # the instrumented passes call train() (collectives when world > 1): every rank runs them, rank 0 reports
    fam = prof_pass(pol, src, tar, bs, 20)
    roll_rate, roll_ms, dynfwd_ms = rollout_rate(pol, src)
    out = None
    if rank == 0:
        fl = train_flops(N, Nt)
        kern = {}
        for k in ("k_mlp3_fwd", "k_mlp3_bwd", "k_wgrad"):
            ms, cnt = fam[k]
            kern[k] = dict(launches_per_step=cnt, ms_per_step=ms, tflops=fl[k] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
        _, _, dyn = macs()
        kern["k_dyn_fwd"] = dict(launches_per_step=1, ms_per_step=dynfwd_ms,
                                 tflops=2.0 * (dyn - 7 * ((2 * S + A) * 256 + 65536 + 512)) * 50000 / (dynfwd_ms * 1e-3) / 1e12)
        dom = max(("k_mlp3_fwd", "k_mlp3_bwd", "k_wgrad"), key=lambda k: kern[k]["ms_per_step"])
        d = kern[dom]
        per_launch_flops = fl[dom] / d["launches_per_step"]
        avg_ms = d["ms_per_step"] / d["launches_per_step"]
        roofline = dict(kernel=dom, bound="mfma", achieved=per_launch_flops / (avg_ms * 1e-3) / 1e12, peak=PEAK_F32_TFLOPS,
                        unit="TFLOP/s", traffic=None, avg_launch_ms=avg_ms, launches_per_step=d["launches_per_step"],
                        flops_per_launch=per_launch_flops)
        roofline["frac"] = roofline["achieved"] / roofline["peak"]
        # HBM bytes of one launch of the dominant kernel from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE passes, committed under profiles/; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note)
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_f_pmc_traffic.json")))
            fam = [e for e in pm if dom in e["kernel"] and e["fetch_kb_raw"] and e["write_kb"]]
            if fam and bs == BS:
                top = max(e["launches"] for e in fam)
                fam = [e for e in fam if 2 * e["launches"] >= top]          # the train() step's launches of this family
                n = sum(e["launches"] for e in fam)
                roofline["traffic"] = sum((2.0 * e["fetch_kb_raw"] + e["write_kb"]) * 1024.0 * e["launches"] for e in fam) / n
                roofline["traffic_note"] = "PMC launch-weighted mean over " + "; ".join(
                    f"{e['launches']} x {e['kernel'].strip()} grid {e['grid']}" for e in fam)
        except Exception:
            pass
        out = {
            "metric": "transitions/sec", "value": N * world * args.steps / dt, "unit": "transitions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"walker2d-friction shapes S={S} A={A}, ensemble 7, rollout_len 1, batch_size {bs}/GPU "
                                   f"(N={N} rows per train() step: src|tar|fake = {bs}|{bs}|{bs // 2}), fp32 MFMA",
                       "rows_per_step_per_gpu": N, "parallelism": f"dp{world}",
                       "hip_graph": bool(args.graph == 1 or (args.graph == 2 and N < 4096))},
            "grad_steps_per_sec": args.steps / dt,
            "rollout_transitions_per_sec": roll_rate, "rollout_ms_per_50000": roll_ms,
            "rollout_refresh_amortised_ms_per_step": (152000.0 / roll_rate) * 1e3 / 5000.0,
            "roofline": roofline, "kernels": kern, "final_losses": losses,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, bs)
            out["gpu_over_cpu_grad_steps"] = out["grad_steps_per_sec"] / out["cpu_baseline"]["grad_steps_per_sec"]
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
