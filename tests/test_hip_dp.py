"""Data-parallel train() on the GPU box: two ranks sharing the one GPU, gloo exchanges (RCCL refuses two ranks on one
device), through the product mirror.  Checks (a) replicas stay bit-identical, (b) the segment-graph replay of the
steady-state step (graphs between the three collectives) equals the eager exchange protocol fed with the same draws,
(c) parameters moved and are finite."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    import torch.distributed as dist
    import golden_util as gu
    from mobody_amd import synthetic
    from mobody_amd.algo import utils
    from mobody_amd.algo.call_algo import call_algo
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    S, A, bs, task = 17, 6, 256, "walker2d-medium-v2"

    def make(graph):
        # the SAME config seed on every rank and DIFFERENT initial weights: the mirror itself has to fold the rank into
        # its index streams and broadcast rank 0's replica before the first step (sync_replicas)
        cfg = gu.policy_cfg(S, A, rng="device", seed=0, penalty_type="none", batch_size=bs, graph=graph,
                            fake_batch_scale=0.5)
        torch.manual_seed(rank); np.random.seed(rank)
        pol = call_algo("mobody", cfg, 3, dev)
        if rank == 0:
            pa, pq, _ = gu.policy_params(5, S, A)
            pol.policy.load_state_dict({k: torch.from_numpy(v) for k, v in pa.items()})
            pol.q_funcs.load_state_dict({k: torch.from_numpy(v) for k, v in pq.items()})
            pol.target_q_funcs.load_state_dict({k: torch.from_numpy(v) for k, v in pq.items()})
        src = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=4000, rng="device", seed=100), 4000, task, 0)
        tar = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=500, rng="device", seed=200), 500, task, 50)
        fake = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=3000, rng="device", seed=300), 3000, task, 90)
        pol.fake_replay_buffer = fake
        pol.total_it = 1                                           # past the refresh step (no dynamics model here)
        return pol, src, tar

    from mobody_amd import ops
    out = {}
    for graph in (0, 1):
        pol, src, tar = make(graph)
        pol.train(src, tar, bs, None, None)                        # first call: eager in both modes (allocates the batch)
        for call in (1, 2, 3):
            if graph:
                pol.train(src, tar, bs, None, None)                # segment graphs + eager all-reduces
            else:                                                  # the graphs' index draws (call ids 1..3), eager protocol
                pol.total_it += 1
                c = torch.tensor([call], dtype=torch.int64, device=dev)
                fb = pol.fake_replay_buffer
                idx = [ops.sample_indices(pol._seed_for(101), 3, c, 0, bs, src.ptr_size[1:2]),
                       ops.sample_indices(pol._seed_for(102), 3, c, 0, bs, tar.ptr_size[1:2]),
                       ops.sample_indices(pol._seed_for(103), 3, c, 0, bs // 2, fb.ptr_size[1:2])]
                ops.gather_batch([src._fields(), tar._fields(), fb._fields()], idx, S, A, out=pol._batch)
                pol._update(pol._batch, int(2.5 * bs), 2 * bs)
        torch.cuda.synchronize()
        assert (pol._graph is not None) == bool(graph)
        if graph:                                                  # gloo cannot be captured: the mirror fell back to segment graphs
            assert len(pol._graph) == 4 and pol.dp_graph == "segments" and pol._ctr.tolist()[:3] == [3, 4, 4]
        out[graph] = {k: v.cpu() for k, v in list(pol.policy.state_dict().items()) + list(pol.q_funcs.state_dict().items())
                      + [("t." + k, v) for k, v in pol.target_q_funcs.state_dict().items()]}
        out[graph]["losses"] = torch.tensor(pol.losses())
    torch.save(out, os.path.join(tmp, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_eager_equals_segment_graphs(tmp_path):
    port = 29500 + os.getpid() % 400
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in (0, 1))
    import golden_util as gu
    _, pq, _ = gu.policy_params(5, 17, 6)
    for graph in (0, 1):
        for k in r0[graph]:
            if k == "losses":
                continue                                            # local shares differ per rank by construction
            assert torch.equal(r0[graph][k], r1[graph][k]), (graph, k, float((r0[graph][k] - r1[graph][k]).abs().max()))  # replicas in sync
            assert torch.isfinite(r0[graph][k]).all()
    # replay == eager protocol; the device-side Adam bias corrections use the GPU's double pow, the host path libm:
    # allow 1 ulp of fp32
    for k in r0[0]:
        if k != "losses":
            np.testing.assert_allclose(r0[1][k].numpy(), r0[0][k].numpy(), rtol=1e-6, atol=1e-8, err_msg=k)
    k = "network1.network.0.weight"
    assert not np.allclose(r0[0][k].numpy(), pq[k])                         # and the step did something
    # same config seed, yet the ranks drew different rows: their local loss shares differ
    assert not torch.equal(r0[0]["losses"], r1[0]["losses"])


def _nccl_worker(rank, world, port, tmp):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    import torch.distributed as dist
    import golden_util as gu
    from mobody_amd import synthetic
    from mobody_amd.algo import utils
    from mobody_amd.algo.call_algo import call_algo
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    S, A, bs, task = 17, 6, 256, "walker2d-medium-v2"
    res = {}
    for mode in ("single", "segments", "captured"):
        cfg = gu.policy_cfg(S, A, rng="device", seed=3, penalty_type="none", batch_size=bs, graph=1,
                            dp_graph="captured" if mode == "captured" else "segments")
        torch.manual_seed(0); np.random.seed(0)
        pol = call_algo("mobody", cfg, 3, dev)
        pol._force_segments = mode != "single"
        src = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=4000, rng="device", seed=100), 4000, task, 0)
        tar = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=500, rng="device", seed=200), 500, task, 50)
        pol.fake_replay_buffer = synthetic.fill_buffer(utils.ReplayBuffer(S, A, dev, max_size=3000, rng="device", seed=300), 3000, task, 90)
        pol.total_it = 1
        for _ in range(6):
            pol.train(src, tar, bs, None, None)
        torch.cuda.synchronize()
        assert len(pol._graph) == (4 if mode == "segments" else 1) and pol.dp_graph == ("captured" if mode == "captured" else "segments")
        res[mode] = {k: v.cpu() for k, v in list(pol.policy.state_dict().items()) + list(pol.q_funcs.state_dict().items())}
    torch.save(res, os.path.join(tmp, "nccl.pt"))
    dist.destroy_process_group()


def test_segment_replay_with_rccl_process_group_single_rank(tmp_path):
    """Graph capture and replay next to a live RCCL process group (watchdog thread, communicator streams): one rank,
    the four-segment replay with real `nccl` all-reduces between the graphs, and the ONE-graph form with the three
    all-reduces captured inside it, both equal the single-graph step bit for bit."""
    port = 29900 + os.getpid() % 90
    mp.spawn(_nccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(tmp_path / "nccl.pt")
    for k in r["single"]:
        assert torch.equal(r["single"][k], r["segments"][k]), k
        assert torch.equal(r["single"][k], r["captured"][k]), k       # RCCL all-reduces captured inside the one graph


def _dara_rows(S, A):
    """Global classifier batch of 64 source + 64 target rows with its input noise, and the way two ranks split it."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    import golden_util as gu
    src, tar = gu.gi.batch(71, 64, S, A), gu.gi.batch(72, 64, S, A)
    rng = np.random.default_rng(5)
    n_sas = rng.standard_normal((128, 2 * S + A)).astype(np.float32)
    n_sa = rng.standard_normal((128, S + A)).astype(np.float32)

    def pick(lo_src, hi_src):                        # rows [src[lo:hi] | tar[lo:hi]] and the matching noise rows
        sel = np.r_[lo_src:hi_src, 64 + lo_src:64 + hi_src]
        rows = [np.concatenate([src[k][lo_src:hi_src], tar[k][lo_src:hi_src]], 0) for k in (0, 1, 2)]
        return rows, n_sas[sel], n_sa[sel]
    return pick


def _dara_worker(rank, world, port, tmp):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    import torch.distributed as dist
    import golden_util as gu
    from mobody_amd.algo.call_algo import call_algo
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    S, A = 17, 6
    cfg = gu.policy_cfg(S, A, rng="device", seed=0, penalty_type="dara")
    torch.manual_seed(9 if world == 1 else rank)                  # ranks start from different weights; rank 0's == the single run's
    if world > 1 and rank == 0:
        torch.manual_seed(9)
    pol = call_algo("mobody", cfg, 3, dev)
    pol.sync_replicas()
    pick = _dara_rows(S, A)
    lo, hi = (0, 64) if world == 1 else (32 * rank, 32 * rank + 32)
    rows, n_sas, n_sa = pick(lo, hi)
    td = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    for _ in range(3):
        pol.update_classifier(None, None, hi - lo, rows=tuple(td(r) for r in rows), noise=(td(n_sas), td(n_sa)))
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in pol.classifier.state_dict().items()}, os.path.join(tmp, f"dara_w{world}_r{rank}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_dara_classifier_update_equals_the_single_rank_update(tmp_path):
    """with world > 1 the classifier gradients are all-reduced, so two ranks that each see half of the
    rows (32 source + 32 target) make exactly the update one rank makes on all 128."""
    port = 29900 + os.getpid() % 90
    mp.spawn(_dara_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_dara_worker, args=(1, port + 1, str(tmp_path)), nprocs=1, join=True)
    r0, r1, one = (torch.load(tmp_path / f) for f in ("dara_w2_r0.pt", "dara_w2_r1.pt", "dara_w1_r0.pt"))
    for k in one:
        assert torch.equal(r0[k], r1[k]), k                                        # replicas in sync
        # the two sums are formed in different orders; Adam's 1/sqrt(v) stretches that on near-zero-gradient entries: 2e-6 = 0.2 % of a step
        np.testing.assert_allclose(r0[k].numpy(), one[k].numpy(), rtol=1e-4, atol=2e-6, err_msg=k)
    assert any(float((one[k] - one[k].mean()).abs().max()) > 0 for k in one)


def test_bench_two_rank_path_keeps_replicas_identical(tmp_path):
    """bench.py's OWN `--gpus 2` path (its launcher, rank-salted draws, sync_replicas, the segment-graph step with the three
    all-reduces between the segments, the rank-sharded refresh, the replica check) rehearsed with two ranks sharing the one
    GPU of the test box over gloo (RCCL refuses two ranks on one device): rank 0's JSON line must report identical replicas."""
    import json
    import subprocess
    env = dict(os.environ, MOBODY_BENCH_BACKEND="gloo", MASTER_PORT=str(29900 + os.getpid() % 90))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c1", "--steps", "6", "--warmup", "2",
                        "--no_cpu_baseline", "--no_mode_sweep"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["replicas_identical"] is True and d["config"]["parallelism"] == "dp2"
    assert d["config"]["hip_graph"] is True and d["steps"] == 6
    assert d["refresh"]["rows_per_rank"] == 25000 + 1000 + 25000          # 50 000 / 2 000 init states sharded over the two ranks
    assert d["value"] > 0 and all(v == v for v in d["final_losses"])
