#!/usr/bin/env python
"""MOBODY hot-path benchmark (contract: one JSON line on rank 0).

Workload (default `--config c2` = BASELINE.json configs[1]): walker2d-friction shapes S=17, A=6, ensemble 7,
rollout_len 1, batch_size 4096 per GPU -> every `MOBODY.train()` step consumes N = 2.5*4096 = 10 240 synthetic
transitions (4096 source + 4096 target + 2048 model-generated rows) already resident in HBM:
replay gather -> twin-Q TD update (+Adam, Polyak) -> Q-scaled actor + Q-weighted-BC update (+Adam).
The model-rollout refresh of the reference (50 000 + 2 000 init states x rollout_len + 50 000 relabels every 5000
steps, mobody.py:441-475) is inside the timed region at the reference's cadence: K timed steps are followed by a
refresh of K/5000 of that size (same code path, scaled row counts), so `ms_per_step` is the amortised cost of a step.

  value  = minibatch transitions consumed per second by the K timed train() steps, summed over ranks (weak scaling:
           every rank draws its own minibatch; gradients and the two actor statistics are all-reduced over RCCL)
  also   = grad_steps_per_sec (train() calls/s, the north_star's target quantity), rollout_transitions_per_sec
           (actor + fused ensemble step, rows produced per second), per-kernel-family roofline, CPU baseline.

`--gpus N` without a torch.distributed launcher starts its N ranks itself (children are spawned before the parent
touches the GPU); under `python -m torch.distributed.run` the ranks come from the environment.
Other configs (`--config c1|c3|c4|c5`) are the shapes of BASELINE.json configs[0], [2], [3], [4].
"""
import argparse
import ctypes as C
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3          # MI355X dense fp32 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak
FAMILIES = ["k_mlp3_fwd", "k_mlp3_bwd", "k_wgrad", "k_dyn_fwd"]
CONFIGS = {   # BASELINE.json configs[i] -> shapes (per GPU)
    "c1": dict(S=17, A=6, bs=256, H=1, task="walker2d-medium-v2", penalty_type="none",
               label="walker2d-friction shapes, batch 256 (the reference's CPU-runnable case)"),
    "c2": dict(S=17, A=6, bs=4096, H=1, task="walker2d-medium-v2", penalty_type="none",
               label="walker2d-friction shapes, batch 4096 per GPU"),
    "c3": dict(S=17, A=6, bs=16384, H=5, task="halfcheetah-medium-v2", penalty_type="none",
               label="halfcheetah-kinematic shapes, rollout_len 5, batch 16384"),
    "c4": dict(S=111, A=8, bs=8192, H=1, task="ant-medium-v2", penalty_type="none",
               label="ant-friction shapes, batch 65536 over 8 GPUs = 8192 per GPU"),
    "c5": dict(S=45, A=24, bs=4096, H=1, task="pen-human-v1", penalty_type="dara",
               label="adroit pen shapes, DARA penalty + mixed src/trg/rollout batch, batch 4096 per GPU"),
}


def macs(S, A):
    actor = S * 256 + 65536 + 256 * A
    q = (S + A) * 256 + 65536 + 256
    rw = 7 * ((2 * S + A) * 256 + 65536 + 512)
    dyn = 7 * (S * 256 + 65536 + 256 * 32 + (16 + A) * 32 + 32 * 32 + 16 * 256 + 65536 + 256 * S) + rw
    return actor, q, dyn, rw


def train_flops(S, A, N, Nt):
    """Useful (executed) FLOPs of one train() step per kernel family; SURVEY 8(d) counts two more
    forwards that the reference executes but never uses (Q7), which this build skips."""
    actor, q, _, _ = macs(S, A)
    fwd = N * (2 * actor + 6 * q) + Nt * 2 * q
    bwd = N * (2 * (256 + 65536) + 2 * (256 + 65536 + 256 * A) + (256 * A + 65536))
    wg = N * (2 * q + actor)
    return {"k_mlp3_fwd": 2.0 * fwd, "k_mlp3_bwd": 2.0 * bwd, "k_wgrad": 2.0 * wg}


def wide_flops(S, A, N, Nt):
    """The part of each family's FLOPs that sits in 256 x 256 GEMMs -- what the split-precision modes move to the bf16 core
    (forward: one per network pass; backward: dz2 W2^T, one per network pass; weight gradients: none, they stay fp32)."""
    return {"k_mlp3_fwd": 2.0 * 65536 * (8 * N + 2 * Nt), "k_mlp3_bwd": 2.0 * 65536 * 5 * N, "k_wgrad": 0.0}


NPROD = {"f32": 0, "bf16": 1, "bf16x2": 3, "bf16x3": 6, "f16x2": 3}


def effective_peak(total_flops, wide, mfma):
    """MFMA roofline of a kernel whose instruction mix is part exact fp32 MFMA, part bf16 MFMA with NPROD products per
    fp32 product: the time both pipes need at their dense peaks (157.3 TF fp32-input, 2.5 PF bf16; MI355X_MICROARCH.md),
    expressed as fp32-equivalent TFLOP/s of the kernel's ALGORITHMIC flops."""
    if NPROD[mfma] == 0:
        return PEAK_F32_TFLOPS
    t_min = (total_flops - wide) / (PEAK_F32_TFLOPS * 1e12) + wide * NPROD[mfma] / (PEAK_BF16_TFLOPS * 1e12)
    return total_flops / t_min / 1e12


# ------------------------------------------------------------------------------------------------ launcher
def spawn_ranks(args):
    """Parent of `--gpus N` (no launcher in the environment): start N children, one per GPU, BEFORE anything here
    touches the GPU (a process that has initialised HIP must not be replaced or forked from), wait, and pass rank 0's
    JSON line through."""
    port = int(os.environ.get("MASTER_PORT", 29000 + os.getpid() % 2000))
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # Poll every child: a rank that dies early (bad device, RCCL init failure) would leave rank 0 waiting in a collective
    # until the RCCL watchdog fires, so on the first non-zero exit the remaining children (ours, by handle) are stopped.
    import threading
    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() is not None and p.returncode != 0:
                failed = r
                break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        print(f"bench.py: rank {failed} exited with code {procs[failed].returncode}; stopped the other ranks", file=sys.stderr)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out = b"".join(out_chunks)
    for line in out.decode().splitlines():              # only the JSON line (gloo / RCCL may chat on stdout)
        if line.startswith("{"):
            print(line)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs) if failed is None else (abs(procs[failed].returncode) or 1)


# ------------------------------------------------------------------------------------------------ GPU side
def build(dev, c, graph, mfma="f32", buffers=None):
    import numpy as np
    import torch
    from mobody_amd import engine, synthetic
    from mobody_amd.algo import utils
    from mobody_amd.algo.call_algo import call_algo
    from mobody_amd.algo.dynamics.mobody_dynamics import MOBODYEnsembleDynamics
    from mobody_amd.algo.dynamics.mobody_module import MOBODYModule
    from mobody_amd.algo.mb_utils.terminal_funs import get_termination_fn
    S, A, bs, task = c["S"], c["A"], c["bs"], c["task"]
    # the SAME seeds on every rank: the mirror folds the rank into its index / noise streams and broadcasts rank 0's
    # replica before the first step, so a caller cannot get either wrong
    cfg = engine.default_config(S, A, rng="device", seed=0, penalty_type=c["penalty_type"], batch_size=bs, graph=graph,
                                src_rollout_length=c["H"], trg_rollout_length=c["H"], mfma=mfma)
    torch.manual_seed(0); np.random.seed(0)
    pol = call_algo("mobody", cfg, 3, dev)
    if buffers is not None:
        src, tar = buffers
    else:
        src = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=1000000, rng="device", seed=100), 1000000, task, 0)
        tar = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=5000, rng="device", seed=200), 5000, task, 1000)
    model = synthetic.alive_dynamics(MOBODYModule(S, A, 256, 7, 5, device=dev, config=cfg), task)
    pol.dynamics = MOBODYEnsembleDynamics(cfg, model, None, None, get_termination_fn(task), penalty_coef=0.1, rng="device", seed=300)
    return pol, src, tar, cfg


def scaled_refresh(pol, src, tar, bs, steps):
    """The fake-buffer refresh at the reference's cadence: `steps`/5000 of its 50 000 / 2 000 init states (and relabels),
    through the product's own `_refresh`."""
    from mobody_amd.algo.offline_offline import mobody as M
    n_s, n_t = max(1, math.ceil(50000 * steps / 5000)), max(1, math.ceil(2000 * steps / 5000))
    old = (M.REFRESH_SRC, M.REFRESH_TAR)
    M.REFRESH_SRC, M.REFRESH_TAR = n_s, n_t
    try:
        pol._refresh(src, tar, bs)
    finally:
        M.REFRESH_SRC, M.REFRESH_TAR = old
    H = pol.config["src_rollout_length"]
    return n_s * H + n_t * pol.config["trg_rollout_length"] + n_s


def prof_pass(pol, src, tar, bs, steps):
    """Instrumented pass: HIP event pairs around every launch of the heavy kernel families (library hook).
    Runs eagerly (event records are not part of a captured graph); the kernels are the same."""
    from mobody_amd import _lib
    lib = _lib.load()
    graph, pol.use_graph = pol.use_graph, 0
    _lib.check(lib.mobody_prof_begin(steps * 64), "prof_begin")
    for _ in range(steps):
        pol.train(src, tar, bs, None, None)
    ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
    _lib.check(lib.mobody_prof_end(ms, cnt, 8), "prof_end")
    pol.use_graph = graph
    return {FAMILIES[i]: (ms[i] / steps, cnt[i] / steps) for i in range(3)}


def rollout_rate(pol, src, H, reps=5, B=50000):
    """Imagined transitions per second: H steps of (actor forward + fused ensemble step + mask + ring append) on B
    init states -- the refresh's rollout as the product runs it (events on torch's stream, which is the stream every
    kernel of the library is launched on)."""
    import torch
    from mobody_amd import _lib
    lib = _lib.load()
    obs = src.state[:B].contiguous()
    for _ in range(2):
        pol._rollout_into_fake(obs, H)
    torch.cuda.synchronize()
    _lib.check(lib.mobody_prof_begin(reps * 8 * H), "prof_begin")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pol._rollout_into_fake(obs, H)
    e1.record(); torch.cuda.synchronize()
    ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
    _lib.check(lib.mobody_prof_end(ms, cnt, 8), "prof_end")
    total_ms = e0.elapsed_time(e1) / reps
    return B * H / (total_ms * 1e-3), total_ms, ms[3] / (reps * H)


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _mlp_params(rng, i, o, prefix):
    import numpy as np
    p = {}
    for li, (a, b) in zip((0, 2, 4), ((i, 256), (256, 256), (256, o))):
        bound = 1.0 / np.sqrt(a)
        p[f"{prefix}network.{li}.weight"] = rng.uniform(-bound, bound, (b, a)).astype(np.float32)
        p[f"{prefix}network.{li}.bias"] = rng.uniform(-bound, bound, (b,)).astype(np.float32)
    return p


def _dyn_params(rng, S, A):
    import numpy as np
    dims = dict(zs1=(S, 256), zs2=(256, 256), zs3=(256, 32), za_src1=(16 + A, 32), za_src2=(32, 32), za_trg1=(16 + A, 32),
                za_trg2=(32, 32), transition1=(16, 256), transition2=(256, 256), transition3=(256, S),
                reward_model1=(2 * S + A, 256), reward_model2=(256, 256), reward_model3=(256, 2))
    p = {}
    for k, (i, o) in dims.items():
        p[k + ".weight"] = (np.clip(rng.standard_normal((7, i, o)), -2, 2) / (2 * np.sqrt(i))).astype(np.float32)
        p[k + ".bias"] = np.zeros((7, 1, o), np.float32)
    return p


def cpu_baseline(c, cfg):
    """The CPU oracle (PyTorch CPU ops, the reference's algorithm, pinned to the reference by tests/golden) on this host:
    protocol of BASELINE.md section 3 -- grad-steps/s at C1 shapes (bs=256; 20 warm-up + 200 timed steps at all cores,
    a bounded count at 1 thread), the bench config's own step, and transitions/s of `step` at B=4096 plus one 52 000-row
    rollout; each at 1 thread (the reference's shipped setting, train_mobody.py:3-5,50-51) and at all cores of the GPU
    box's CPU share.  About 30 s of CPU work in total."""
    import numpy as np
    import torch
    from mobody_amd import engine, synthetic
    from oracle import mobody_oracle as O
    ncores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    S, A, bs, task = c["S"], c["A"], c["bs"], c["task"]
    rng = np.random.default_rng(0)

    def batch(N, S_, A_):
        mu = synthetic.alive_mean(task, S_)
        return ((mu + 0.1 * rng.standard_normal((N, S_))).astype(np.float32), rng.uniform(-1, 1, (N, A_)).astype(np.float32),
                (mu + 0.1 * rng.standard_normal((N, S_))).astype(np.float32), rng.standard_normal((N, 1)).astype(np.float32),
                np.ones((N, 1), np.float32))

    def train_rate(S_, A_, bs_, warm, steps, budget):
        ocfg = engine.default_config(S_, A_)
        pa = _mlp_params(rng, S_, A_, "network.")
        pq = {**_mlp_params(rng, S_ + A_, 1, "network1."), **_mlp_params(rng, S_ + A_, 1, "network2.")}
        st = O.TrainState(pa, pq, None)
        b = batch(int(2.5 * bs_), S_, A_)
        for _ in range(warm):
            O.train_step(st, b, 2 * bs_, ocfg)
        t0 = time.time(); n = 0
        while n < steps and (time.time() - t0 < budget or n < 2):
            O.train_step(st, b, 2 * bs_, ocfg); n += 1
        return n / (time.time() - t0), n

    def step_rate(S_, A_, B, calls, budget):
        p = O.to_torch(_dyn_params(rng, S_, A_))
        p["transition3.bias"] = p["transition3.bias"] + torch.from_numpy(synthetic.alive_mean(task, S_)).view(1, 1, -1)
        obs, act = batch(B, S_, A_)[:2]
        eps = rng.standard_normal((7, B, S_)).astype(np.float32); idx = rng.integers(0, 5, B)
        with torch.no_grad():
            O.dyn_step(p, obs, act, eps, idx, task, 0.1)
            t0 = time.time(); n = 0
            while n < calls and (time.time() - t0 < budget or n < 1):
                O.dyn_step(p, obs, act, eps, idx, task, 0.1); n += 1
        return B * n / (time.time() - t0), n

    legs = {}
    for threads in (ncores, 1):
        torch.set_num_threads(threads)
        tag = "all_cores" if threads == ncores and ncores > 1 else "1_thread"
        c1_rate, c1_n = train_rate(17, 6, 256, 20 if threads > 1 else 3, 200 if threads > 1 else 60, 4.0)
        cfg_rate, cfg_n = (c1_rate, c1_n) if (S, A, bs) == (17, 6, 256) else train_rate(S, A, bs, 1, 20, 4.0)
        st_rate, st_n = step_rate(S, A, 4096, 10, 3.0)
        ro_rate, _ = step_rate(S, A, 52000, 1, 0.0) if threads > 1 else (None, 0)
        legs[tag] = dict(threads=threads, c1_grad_steps_per_sec=c1_rate, c1_steps_timed=c1_n,
                         config_grad_steps_per_sec=cfg_rate, config_steps_timed=cfg_n,
                         step_transitions_per_sec_B4096=st_rate, step_calls_timed=st_n,
                         rollout_transitions_per_sec_52000=ro_rate)
        if ncores == 1:
            break
    torch.set_num_threads(ncores)
    main = legs.get("all_cores", legs["1_thread"])
    N = int(2.5 * bs)
    return dict(value=N * main["config_grad_steps_per_sec"], unit="transitions/s", cores=main["threads"], kind="port",
                cpu_model=cpu_model(), host_cores_visible=os.cpu_count(),
                sample=f"oracle (torch CPU fp32) on {main['threads']} threads: {main['config_steps_timed']} train steps at N={N} rows "
                       f"(S={S},A={A}), {main['c1_steps_timed']} train steps at C1 (bs=256, N=640), {main['step_calls_timed']} dyn_step calls at "
                       f"B=4096, one 52000-row step; the same legs at 1 thread (reference's shipped setting) under `legs`",
                grad_steps_per_sec=main["config_grad_steps_per_sec"], c1_grad_steps_per_sec=main["c1_grad_steps_per_sec"],
                rollout_transitions_per_sec=main["rollout_transitions_per_sec_52000"] or main["step_transitions_per_sec_B4096"],
                legs=legs)


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch_size", type=int, default=None, help="override the config's per-GPU batch size")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--mfma", default="bf16x3", choices=["f32", "f16x2", "bf16x3", "bf16x2", "bf16"],
                    help="MFMA mode of the 256 x 256 forward / backward GEMMs: bf16x3 (default: three-term split, six products, holds the "
                         "fp32 parity tolerances), exact fp32 (the parity-test mode), bf16x2 (~6e-6) or plain bf16 (~3e-3)")
    ap.add_argument("--no_mode_sweep", action="store_true", help="skip the short runs of the other MFMA modes")
    ap.add_argument("--graph", type=int, default=1, help="HIP-graph replay of the steady-state step: 0 never, 1 always, 2 auto (minibatches under 4096 rows); with N > 1 ranks the segments between the three all-reduces are replayed")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np  # noqa: F401
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
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
        world = torch.distributed.get_world_size()          # what the process group actually sees
    c = dict(CONFIGS[args.config])
    if args.batch_size:
        c["bs"] = args.batch_size
    S, A, bs = c["S"], c["A"], c["bs"]
    N, Nt = int(2.5 * bs), 2 * bs
    pol, src, tar, cfg = build(dev, c, args.graph, args.mfma)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):                  # step 1 includes the full model-rollout refresh (and the DARA warm-up)
        pol.train(src, tar, bs, None, None)
    scaled_refresh(pol, src, tar, bs, args.steps)         # warm the scaled refresh's shapes too
    barrier()
    import gc
    gc.collect(); gc.disable()                            # a generation-2 collection inside the timed loop is a 40 ms host stall
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    ev[0].record()
    for _ in range(args.steps):
        pol.train(src, tar, bs, None, None)
    ev[1].record()
    rolled = scaled_refresh(pol, src, tar, bs, args.steps)
    barrier()
    t1 = time.perf_counter()
    gc.enable()
    t = torch.tensor([t1 - t0, ev[0].elapsed_time(ev[1]) * 1e-3], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt, dt_steps = float(t[0]), float(t[1])
    losses = pol.losses()
    replicas_identical = None
    if world > 1:                                         # data-parallel replicas must still be bit-identical
        h = torch.stack([b.double().sum() for b in (pol.q_funcs.blob, pol.target_q_funcs.blob, pol.policy.blob)])
        hs = [torch.zeros_like(h) for _ in range(world)]
        torch.distributed.all_gather(hs, h)
        replicas_identical = all(torch.equal(hs[0], x) for x in hs)

    # the other MFMA modes, 100 steps each on the same buffers (extra information; the headline is args.mfma)
    sweep = {}
    if not args.no_mode_sweep and world == 1:             # single GPU only: the scaling runs time the headline mode alone
        for mode in ("f32", "f16x2", "bf16x3", "bf16x2", "bf16"):
            if mode == args.mfma:
                continue
            p2 = build(dev, c, args.graph, mode, buffers=(src, tar))[0]
            for _ in range(12):
                p2.train(src, tar, bs, None, None)
            barrier()
            ts = time.perf_counter()
            for _ in range(100):
                p2.train(src, tar, bs, None, None)
            barrier()
            tq = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=dev)
            if world > 1:
                torch.distributed.all_reduce(tq, op=torch.distributed.ReduceOp.MAX)
            sweep[mode] = dict(ms_per_step_refresh_excluded=float(tq[0]) / 100 * 1e3, grad_steps_per_sec=100 / float(tq[0]))
            del p2
    # the instrumented passes call train() (collectives when world > 1): every rank runs them, rank 0 reports
    fam = prof_pass(pol, src, tar, bs, 20)
    roll_rate, roll_ms, dynfwd_ms = rollout_rate(pol, src, c["H"])
    out = None
    if rank == 0:
        fl = train_flops(S, A, N, Nt)
        kern = {}
        for k in ("k_mlp3_fwd", "k_mlp3_bwd", "k_wgrad"):
            ms, cnt = fam[k]
            kern[k] = dict(launches_per_step=cnt, ms_per_step=ms, tflops=fl[k] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
        _, _, dyn, rw = macs(S, A)
        kern["k_dyn_fwd"] = dict(launches_per_step=1, ms_per_step=dynfwd_ms,
                                 tflops=2.0 * (dyn - rw) * 50000 / (dynfwd_ms * 1e-3) / 1e12 if dynfwd_ms > 0 else 0.0)
        dom = max(("k_mlp3_fwd", "k_mlp3_bwd", "k_wgrad"), key=lambda k: kern[k]["ms_per_step"])
        d = kern[dom]
        per_launch_flops = fl[dom] / d["launches_per_step"]
        avg_ms = d["ms_per_step"] / d["launches_per_step"]
        peak = effective_peak(fl[dom], wide_flops(S, A, N, Nt)[dom], args.mfma)
        roofline = dict(kernel=dom, bound="mfma", achieved=per_launch_flops / (avg_ms * 1e-3) / 1e12, peak=peak,
                        unit="TFLOP/s", traffic=None, avg_launch_ms=avg_ms, launches_per_step=d["launches_per_step"],
                        flops_per_launch=per_launch_flops,
                        peak_note="fp32-equivalent TFLOP/s of the kernel's algorithmic flops if its fp32-MFMA part ran at 157.3 TF and "
                                  f"its 256x256 GEMMs ({NPROD[args.mfma]} bf16 MFMAs per fp32 product) at 2.5 PF" if args.mfma != "f32"
                                  else "dense fp32-input MFMA peak")
        roofline["frac"] = roofline["achieved"] / roofline["peak"]
        # HBM bytes of one launch of the dominant kernel: PMC counters need their own rocprofv3 passes (FETCH_SIZE and
        # WRITE_SIZE cannot share one, and not with a timing run), so the figure comes from the committed summary of
        # those passes over this same command (profiles/, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note)
        for tag in ("r02", "r01_f"):
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")))
            except OSError:
                continue
            sel = [e for e in pm if dom in e["kernel"] and e["fetch_kb_raw"] and e["write_kb"]]
            if sel and args.config == "c2" and bs == CONFIGS["c2"]["bs"]:
                top = max(e["launches"] for e in sel)
                sel = [e for e in sel if 2 * e["launches"] >= top]          # the train() step's launches of this family
                n = sum(e["launches"] for e in sel)
                roofline["traffic"] = sum((2.0 * e["fetch_kb_raw"] + e["write_kb"]) * 1024.0 * e["launches"] for e in sel) / n
                roofline["traffic_note"] = f"profiles/{tag}_pmc_traffic.json, launch-weighted mean over " + "; ".join(
                    f"{e['launches']} x {e['kernel'].strip()} grid {e['grid']}" for e in sel)
            break
        graph_on = bool(args.graph == 1 or (args.graph == 2 and N < 4096))
        out = {
            "metric": "transitions/sec (minibatch rows through train(), refresh amortised at 1/5000) + grad-steps/sec",
            "value": N * world * args.steps / dt, "unit": "transitions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.mfma, "data": "synthetic",
            "dtype_note": {"f32": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32): the reference's arithmetic",
                           "bf16x3": "fp32 operands as three bf16 terms, six bf16 MFMAs per fp32 product, fp32 accumulate: fp32-grade "
                                     "(~5e-7 of max|out| from the fp32 kernels); holds the fp32 parity tolerances against the reference's "
                                     "golden vectors (tests/test_hip_precision.py; the whole -m gpu suite passes with MOBODY_MFMA=bf16x3); "
                                     "the exact-fp32 step time of the same run is under other_mfma_modes.f32",
                           "f16x2": "fp32 operands as two fp16 terms (22 significand bits) with exact power-of-two tile scales, three "
                                    "fp16 MFMAs per fp32 product, fp32 accumulate: fp32-grade; holds the fp32 parity tolerances "
                                    "against the reference's golden vectors",
                           "bf16x2": "two bf16 terms, three products (~6e-6): throughput mode, not parity grade",
                           "bf16": "plain bf16 MFMA inputs (~3e-3): throughput mode, not parity grade"}[args.mfma],
            "config": {"workload": f"{args.config}: {c['label']} (S={S} A={A}, ensemble 7, rollout_len {c['H']}, N={N} rows per "
                                   f"train() step: src|tar|fake = {bs}|{bs}|{bs // 2}), "
                                   + ("exact fp32 MFMA" if args.mfma == "f32" else f"256x256 GEMMs (forward, backward, weight gradient) on the {args.mfma} split-precision MFMA "
                                      "core, everything else exact fp32 MFMA"),
                       "name": args.config, "rows_per_step_per_gpu": N, "parallelism": f"dp{world}", "hip_graph": graph_on},
            "grad_steps_per_sec": args.steps / dt,
            "grad_steps_per_sec_refresh_excluded": args.steps / dt_steps,
            "rollout_transitions_per_sec": roll_rate, "rollout_ms_per_call": roll_ms,
            "rollout_rows_in_timed_region": rolled,
            "roofline": roofline, "kernels": kern, "final_losses": losses, "replicas_identical": replicas_identical,
            "other_mfma_modes": sweep,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(c, cfg)
            out["gpu_over_cpu_grad_steps"] = out["grad_steps_per_sec"] / out["cpu_baseline"]["grad_steps_per_sec"]
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if replicas_identical is False:
        sys.exit(3)


if __name__ == "__main__":
    main()
