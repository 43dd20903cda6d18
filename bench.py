#!/usr/bin/env python
"""MOBODY hot-path benchmark (contract: one JSON line on rank 0).

Workload (default `--config c2` = BASELINE.json configs[1]): walker2d-friction shapes S=17, A=6, ensemble 7,
rollout_len 1, batch_size 4096 per GPU -> every `MOBODY.train()` step consumes N = 2.5*4096 = 10 240 synthetic
transitions (4096 source + 4096 target + 2048 model-generated rows) already resident in HBM:
replay gather -> twin-Q TD update (+Adam, Polyak) -> Q-scaled actor + Q-weighted-BC update (+Adam).
The model-rollout refresh of the reference (50 000 + 2 000 init states x rollout_len + 50 000 relabels every 5000
steps, mobody.py:441-475) is timed once at FULL size right after the K timed steps (between barriers of its own) and
charged at the reference's cadence: `ms_per_step` = K-step time / K + refresh time / 5000.

  value  = minibatch transitions consumed per second by the K timed train() steps, summed over ranks (weak scaling:
           every rank draws its own minibatch; gradients and the two actor statistics are all-reduced over RCCL)
  also   = grad_steps_per_sec (train() calls/s, the north_star's target quantity), rollout_transitions_per_sec
           (actor + fused ensemble step, rows produced per second), per-kernel-family roofline, CPU baseline.

`--gpus N` without a torch.distributed launcher starts its N ranks itself (children are spawned before the parent
touches the GPU); under `python -m torch.distributed.run` the ranks come from the environment.
Other configs (`--config c1|c3|c4|c5`) are the shapes of BASELINE.json configs[0], [2], [3], [4]; `--penalty par` runs the
CLI's default reward shaping (one ensemble step on the source rows inside every train() step, mobody.py:428-434);
`--config pretrain` measures the step BEFORE the hot path, dynamics pre-training (MOBODYEnsembleDynamics.learn, 256 rows x 7
members per optimizer step, mobody_dynamics.py:594-653), with its own JSON line.
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
PEAK_BF16_TFLOPS = 2500.0        # dense bf16 / fp16 MFMA peak
REFRESH_EVERY = 5000             # steps between two fake-buffer refreshes (mobody.py:441)
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
    """The part of each family's FLOPs that sits in 256 x 256 GEMMs -- what the split-precision modes move to the 16-bit MFMA
    core (forward: one per network pass; backward: dz2 W2^T, one per network pass; weight gradients: the dW2 = h1^T dz2
    job of the three nets with gradients)."""
    return {"k_mlp3_fwd": 2.0 * 65536 * (8 * N + 2 * Nt), "k_mlp3_bwd": 2.0 * 65536 * 5 * N, "k_wgrad": 2.0 * 65536 * 3 * N}


NPROD = {"f32": 0, "bf16": 1, "bf16x2": 3, "bf16x3": 6, "f16x2": 3}


def effective_peak(total_flops, wide, mfma):
    """MFMA roofline of a kernel whose instruction mix is part exact fp32 MFMA, part 16-bit (bf16 / fp16) MFMA with NPROD
    products per fp32 product: the time both pipes need at their dense peaks (157.3 TF fp32-input, 2.5 PF bf16 = fp16; MI355X_MICROARCH.md),
    expressed as fp32-equivalent TFLOP/s of the kernel's ALGORITHMIC flops."""
    if NPROD[mfma] == 0:
        return PEAK_F32_TFLOPS
    t_min = (total_flops - wide) / (PEAK_F32_TFLOPS * 1e12) + wide * NPROD[mfma] / (PEAK_BF16_TFLOPS * 1e12)
    return total_flops / t_min / 1e12


def src_fingerprint():
    """sha1 over the kernel sources and this file: ties a committed PMC summary to the build it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha1()
    pk = os.path.join(ROOT, "mobody-model-based-off-dynamics-offline-reinforcement-learning_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(pk, "*.hip")) + glob.glob(os.path.join(pk, "*.h"))) + [os.path.join(ROOT, "include", "mobody_hip.h")]:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


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


def full_refresh(pol, src, tar, bs):
    """One fake-buffer refresh at the reference's FULL size (50 000 source + 2 000 target init states x rollout_len, then
    the 50 000 (s, a) relabels; mobody.py:441-475) through the product's own `_refresh`.  Returns the rows it rolled."""
    from mobody_amd.algo.offline_offline import mobody as M
    pol._refresh(src, tar, bs)
    w = pol._world()                                     # data parallel: the init states are sharded over the ranks
    n_s, n_t = -(-M.REFRESH_SRC // w), -(-M.REFRESH_TAR // w)
    return n_s * pol.config["src_rollout_length"] + n_t * pol.config["trg_rollout_length"] + n_s


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


# ------------------------------------------------------------------------------------------------ dynamics pre-training
def pretrain_flops(S, A, b):
    """Useful FLOPs of one learn() optimizer step (b rows x 7 members): forward MACs of the three big nets are state encoder
    2x, transition decoder 4x, reward head 2x per row; forward + backward = 3x the forward."""
    enc = S * 256 + 65536 + 256 * 32
    dec = 16 * 256 + 65536 + 256 * S
    rw = (2 * S + A) * 256 + 65536 + 512
    return 2.0 * 3.0 * 7 * b * (2 * enc + 4 * dec + 2 * rw)


def bench_pretrain(args, dev, world, rank):
    """`--config pretrain`: optimizer steps per second of MOBODYEnsembleDynamics.learn (mobody_dynamics.py:594-653) at the
    reference's batch (256 rows x 7 members), walker2d shapes, device-Philox noise, the mirror's eager two-stream step
    (config train_graph = 1: its HIP-graph replay); K timed steps between barriers.  Single GPU (the data-parallel form shards the batch rows)."""
    import numpy as np
    import torch
    from mobody_amd import _lib, engine, synthetic
    from mobody_amd.algo.dynamics.mobody_dynamics import MOBODYEnsembleDynamics
    from mobody_amd.algo.dynamics.mobody_module import MOBODYModule
    from mobody_amd.algo.mb_utils.terminal_funs import get_termination_fn
    assert world == 1, "--config pretrain is a single-GPU measurement"
    S, A, b, task = 17, 6, args.batch_size or 256, "walker2d-medium-v2"
    steps, warm = args.steps, max(args.warmup, 3)
    cfg = engine.default_config(S, A, no_vae=0, inverse_sep_reward_loss=0, train_together=0, train_with_src_threshold=1, dynamics_lr=1e-3,
                                mfma=args.mfma)
    m = MOBODYModule(S, A, 256, 7, 5, device=dev, config=cfg)
    dyn = MOBODYEnsembleDynamics(cfg, m, None, None, get_termination_fn(task), penalty_coef=0.1, rng="device", seed=1)
    f16 = dyn.train_precision == 4                      # pre-training follows --mfma when it is f16x2, else exact fp32 MFMA
    graph = bool(dyn.train_graph)                       # (off by default: the eager step measured faster than its graph replay)
    g = torch.Generator().manual_seed(0)
    mu = torch.from_numpy(synthetic.alive_mean(task, S))
    n = 200000
    data = [(mu + 0.1 * torch.randn(n, S, generator=g)).to(dev), (torch.rand(n, A, generator=g) * 2 - 1).to(dev),
            (mu + 0.1 * torch.randn(n, S, generator=g)).to(dev), torch.randn(n, 1, generator=g).to(dev)]
    idx = torch.randint(n, (7, steps * b), generator=g).to(device=dev, dtype=torch.int32).contiguous()
    dyn._learn_indexed(True, data, idx[:, :warm * b].contiguous(), b)
    dyn._learn_indexed(True, data, idx, b)                # (train_graph = 1: captures the graph of this index matrix)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = dyn._learn_indexed(True, data, idx, b)        # K optimizer steps
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # per-family times of the same step, eager (event pairs are not part of a captured graph)
    lib = _lib.load()
    dyn.train_graph = 0
    k = min(steps, 50)
    _lib.check(lib.mobody_prof_begin(k * 64), "prof_begin")
    dyn._learn_indexed(True, data, idx[:, :k * b].contiguous(), b)
    ms = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
    _lib.check(lib.mobody_prof_end(ms, cnt, 8), "prof_end")
    fam = {FAMILIES[i]: dict(ms_per_step=ms[i] / k, launches_per_step=cnt[i] / k) for i in range(3)}
    fl = pretrain_flops(S, A, b)
    dom = max(fam, key=lambda f: fam[f]["ms_per_step"])
    share = {"k_mlp3_fwd": 1.0 / 3.0, "k_mlp3_bwd": 1.0 / 3.0, "k_wgrad": 1.0 / 3.0}[dom]     # fwd : dX : dW = 1 : 1 : 1
    d = fam[dom]
    per_launch = fl * share / max(d["launches_per_step"], 1)
    avg_ms = d["ms_per_step"] / max(d["launches_per_step"], 1)
    roofline = dict(kernel=dom, bound="mfma", achieved=per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
                    peak=PEAK_F32_TFLOPS, unit="TFLOP/s", traffic=None, avg_launch_ms=avg_ms,
                    launches_per_step=d["launches_per_step"], flops_per_launch=per_launch,
                    peak_note="dense fp32-input MFMA peak" + (" (f16x2: the 256 x 256 layers run as three fp16 MFMAs per fp32 product; "
                                                              "priced against the fp32 peak the first / last layers still use)" if f16 else
                                                              " (the pre-training kernels run exact fp32 MFMA)"),
                    traffic_note="null: no PMC pass for this configuration")
    roofline["frac"] = roofline["achieved"] / roofline["peak"]
    out = {"metric": "dynamics pre-training optimizer steps/sec (MOBODYEnsembleDynamics.learn, the step before the hot path)",
           "value": steps / dt, "unit": "optimizer steps/s", "n_gpus": 1, "steps": steps, "warmup": warm,
           "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f16x2 split (fp32-grade), fp32 accumulate" if f16 else "f32", "mfma": "f16x2" if f16 else "f32", "data": "synthetic",
           "config": {"workload": f"pretrain: walker2d shapes (S={S} A={A}), {b} rows x 7 members per optimizer step, target-domain batches, "
                                  "device-Philox reparameterisation noise, " + ("HIP-graph replay" if graph else "eager launches (two streams)"),
                      "name": "pretrain", "rows_per_step": 7 * b, "parallelism": "dp1", "hip_graph": graph},
           "samples_per_sec": steps * b * 7 / dt, "useful_tflops": fl * steps / dt / 1e12,
           "frac_f32_mfma_peak_whole_step": fl * steps / dt / (PEAK_F32_TFLOPS * 1e12),
           "roofline": roofline, "kernels": fam, "mean_losses": list(stats)}
    if not args.no_cpu_baseline:
        from oracle import mobody_oracle as O
        threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        torch.set_num_threads(threads)
        rng = np.random.default_rng(0)
        st = O.DynTrainState({k_: v.cpu().numpy() for k_, v in m.state_dict().items()})
        rows = [x[:7 * b].reshape(7, b, -1).cpu().numpy() for x in data]
        nz = [rng.standard_normal((7, b, 16)).astype(np.float32) for _ in range(6)] + [rng.standard_normal((7, b, S)).astype(np.float32)]
        O.dyn_learn_step(st, *rows, nz, True)
        tc = time.time(); kk = 0
        while time.time() - tc < 10.0 or kk < 3:
            O.dyn_learn_step(st, *rows, nz, True); kk += 1
        rate = kk / (time.time() - tc)
        out["cpu_baseline"] = dict(value=rate, unit="optimizer steps/s", cores=threads, kind="port", cpu_model=cpu_model(),
                                   sample=f"{kk} oracle learn steps (torch CPU fp32 autograd, {threads} threads) at the same shapes")
        out["gpu_over_cpu"] = out["value"] / rate
    print(json.dumps(out))
    sys.stdout.flush()


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS) + ["pretrain"])
    ap.add_argument("--penalty", default=None, choices=["none", "par", "dara"], help="override the config's penalty_type ('par' = the CLI default)")
    ap.add_argument("--batch_size", type=int, default=None, help="override the config's per-GPU batch size")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--mfma", default="f16x2", choices=["f32", "f16x2", "bf16x3", "bf16x2", "bf16"],
                    help="MFMA mode of the 256 x 256 GEMMs: f16x2 (default: two fp16 terms, three products, holds the fp32 parity "
                         "tolerances), bf16x3 (three bf16 terms, six products, same grade), exact fp32, bf16x2 (~6e-6) or plain bf16 (~3e-3)")
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
    if args.config == "pretrain":
        return bench_pretrain(args, dev, world, rank)
    c = dict(CONFIGS[args.config])
    if args.batch_size:
        c["bs"] = args.batch_size
    if args.penalty:
        c["penalty_type"] = args.penalty
        c["label"] += f", penalty_type {args.penalty}"
    S, A, bs = c["S"], c["A"], c["bs"]
    N, Nt = int(2.5 * bs), 2 * bs
    pol, src, tar, cfg = build(dev, c, args.graph, args.mfma)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):                  # step 1 includes the full model-rollout refresh (and the DARA warm-up)
        pol.train(src, tar, bs, None, None)
    barrier()
    import gc
    gc.collect(); gc.disable()                            # a generation-2 collection inside the timed loop is a 40 ms host stall
    t0 = time.perf_counter()
    for _ in range(args.steps):                           # EXACTLY K steps between two barrier + synchronize pairs
        pol.train(src, tar, bs, None, None)
    barrier()
    t1 = time.perf_counter()
    # one refresh at the reference's full size, timed on its own and charged at its cadence (once per 5000 steps)
    rolled = full_refresh(pol, src, tar, bs)
    barrier()
    t2 = time.perf_counter()
    gc.enable()
    t = torch.tensor([t1 - t0, t2 - t1], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt_steps, dt_refresh = float(t[0]), float(t[1])
    ms_step = dt_steps / args.steps * 1e3 + dt_refresh * 1e3 / REFRESH_EVERY
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
                                  f"its 256x256 GEMMs ({NPROD[args.mfma]} 16-bit MFMAs per fp32 product) at 2.5 PF" if args.mfma != "f32"
                                  else "dense fp32-input MFMA peak")
        roofline["frac"] = roofline["achieved"] / roofline["peak"]
        # HBM bytes of one launch of the dominant kernel: PMC counters need their own rocprofv3 passes (FETCH_SIZE and
        # WRITE_SIZE cannot share one, and not with a timing run), so the figure can only come from a summary of those
        # passes over this same command.  It is used ONLY when that summary was measured on exactly this build (source
        # fingerprint recorded by tools/pmc_traffic.py); anything else is reported as null with the reason.
        fp = src_fingerprint()
        roofline["traffic_note"] = f"null: no PMC summary for this build (source fingerprint {fp}) under profiles/"
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
            try:
                pm = json.load(open(path))
            except (OSError, ValueError):
                continue
            if not isinstance(pm, dict) or pm.get("src_fingerprint") != fp or pm.get("config") != args.config or pm.get("mfma") != args.mfma:
                continue
            sel = [e for e in pm["kernels"] if dom in e["kernel"] and e["fetch_kb_raw"] and e["write_kb"]]
            if sel and bs == CONFIGS[args.config]["bs"]:
                top = max(e["launches"] for e in sel)
                sel = [e for e in sel if 2 * e["launches"] >= top]          # the train() step's launches of this family
                n = sum(e["launches"] for e in sel)
                roofline["traffic"] = sum((2.0 * e["fetch_kb_raw"] + e["write_kb"]) * 1024.0 * e["launches"] for e in sel) / n
                roofline["traffic_note"] = (f"{os.path.relpath(path, ROOT)} (same source fingerprint; FETCH_SIZE doubled per "
                                            "MI355X_MICROARCH.md), launch-weighted mean over " + "; ".join(
                                                f"{e['launches']} x {e['kernel'].strip()} grid {e['grid']}" for e in sel))
                break
        graph_on = bool(args.graph == 1 or (args.graph == 2 and N < 4096))
        out = {
            "metric": "transitions/sec (minibatch rows through train(), refresh amortised at 1/5000) + grad-steps/sec",
            "value": N * world / (ms_step * 1e-3), "unit": "transitions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.mfma, "data": "synthetic",
            "dtype_note": {"f32": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32): the reference's arithmetic",
                           "bf16x3": "fp32 operands as three bf16 terms, six bf16 MFMAs per fp32 product, fp32 accumulate: fp32-grade "
                                     "(~5e-7 of max|out| from the fp32 kernels); holds the fp32 parity tolerances against the reference's "
                                     "golden vectors (tests/test_hip_precision.py)",
                           "f16x2": "fp32 operands as two fp16 terms (22 significand bits) with exact power-of-two tile scales, three "
                                    "fp16 MFMAs per fp32 product, fp32 accumulate: fp32-grade (~3e-7 of max|out| from the fp32 kernels); "
                                    "every golden-vector suite of `pytest -m gpu` runs in this mode AND in exact fp32 (tests/conftest.py "
                                    "`mfma` fixture) at the same tolerances; the exact-fp32 step time of the same run is under "
                                    "other_mfma_modes.f32",
                           "bf16x2": "two bf16 terms, three products (~6e-6): throughput mode, not parity grade",
                           "bf16": "plain bf16 MFMA inputs (~3e-3): throughput mode, not parity grade"}[args.mfma],
            "config": {"workload": f"{args.config}: {c['label']} (S={S} A={A}, ensemble 7, rollout_len {c['H']}, N={N} rows per "
                                   f"train() step: src|tar|fake = {bs}|{bs}|{bs // 2}), "
                                   + ("exact fp32 MFMA" if args.mfma == "f32" else f"256x256 GEMMs (forward, backward, weight gradient) on the {args.mfma} split-precision MFMA "
                                      "core, everything else exact fp32 MFMA") + f", penalty_type {c['penalty_type']}",
                       "name": args.config, "rows_per_step_per_gpu": N, "parallelism": f"dp{world}", "hip_graph": graph_on},
            "grad_steps_per_sec": 1e3 / ms_step,
            "grad_steps_per_sec_refresh_excluded": args.steps / dt_steps,
            "timed_region": {"steps_ms": dt_steps * 1e3, "steps": args.steps,
                             "note": "ms_per_step = steps_ms / steps + refresh.ms / 5000 (one full-size refresh timed after the K steps)"},
            "refresh": {"rows_per_rank": rolled, "ms": dt_refresh * 1e3, "ms_per_step_share": dt_refresh * 1e3 / REFRESH_EVERY,
                        "transitions_per_sec": rolled * world / dt_refresh},
            "rollout_transitions_per_sec": roll_rate, "rollout_ms_per_call": roll_ms,
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
