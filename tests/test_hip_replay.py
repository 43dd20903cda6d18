"""Device-resident replay gather / ring append vs the reference's golden ReplayBuffer behaviour (bit exact)."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


LAYOUTS = ("ring", "arrays")        # the row-interleaved ring the mirror allocates / five separate arrays (the reference's shape)


def make_buf(layout, rows, S, A, dev, fill=None):
    """An empty (or `fill`ed: five [rows][n] arrays) replay storage of either layout; both index / iterate as the five fields."""
    from mobody_amd import ops
    if layout == "arrays":
        return tuple(torch.zeros(rows, n, device=dev) if fill is None else torch.from_numpy(fill[k]).to(dev).contiguous()
                     for k, n in enumerate((S, A, S, 1, 1)))
    v = ops.RingView(torch.zeros(rows, ops.ring_pitch(S, A), device=dev), S, A)
    if fill is not None:
        for t, x in zip(v, fill):
            t.copy_(torch.from_numpy(x))
    return v


@pytest.mark.parametrize("layout", LAYOUTS)
def test_ring_append_matches_reference_add_batch(dev, layout):
    from mobody_amd import ops
    g = gu.load("g8_replay")
    cap, S, A = 50, 5, 2
    buf = make_buf(layout, cap, S, A, dev)
    ps = torch.zeros(2, dtype=torch.int64, device=dev)
    names = ("state", "action", "next_state", "reward", "not_done")
    for ci, (M, want_ptr, want_size) in enumerate(g["log"]):
        f = lambda k: torch.from_numpy(g[f"add{ci}_{k}"]).to(dev).contiguous()
        ops.ring_append(buf, cap, ps, S, A, f("obss"), f("actions"), f("next_obss"), f("rewards"),
                        f("terminals").to(torch.uint8).contiguous())
        assert ps.cpu().tolist() == [int(want_ptr), int(want_size)], ci
        for t, k in zip(buf, names):
            assert (t.cpu().numpy() == g[f"after{ci}_{k}"]).all(), (ci, k)
    # gather == reference sample() rows for the reference's own index draw
    out = ops.gather_batch([buf], [torch.from_numpy(g["sample_ind"]).to(dev)], S, A)
    for t, k in zip(out, names):
        assert (t.cpu().numpy() == g["sample_" + k]).all(), k


@pytest.mark.parametrize("layout", LAYOUTS)
def test_ring_append_with_filter_and_concat_gather(dev, layout):
    from mobody_amd import ops
    rng = np.random.default_rng(0)
    cap, S, A, M = 3000, 17, 6, 2500
    buf = make_buf(layout, cap, S, A, dev)
    ps = torch.tensor([2000, 2400], dtype=torch.int64, device=dev)
    obs, act, nxt = (rng.standard_normal((M, n)).astype(np.float32) for n in (S, A, S))
    rew = rng.standard_normal((M, 1)).astype(np.float32)
    term = (rng.uniform(size=(M, 1)) > 0.8).astype(np.uint8)
    keep = (rng.uniform(size=M) > 0.3).astype(np.uint8)
    td = lambda x: torch.from_numpy(x).to(dev).contiguous()
    ops.ring_append(buf, cap, ps, S, A, td(obs), td(act), td(nxt), td(rew), td(term), td(keep))
    from oracle import mobody_oracle as O
    K = int(keep.sum())
    segs, ptr, size = O.ring_append_plan(2000, 2400, cap, K)
    assert ps.cpu().tolist() == [ptr, size]
    want = np.zeros((cap, S), np.float32)
    kept = obs[keep.astype(bool)]
    for dst, src, ln in segs:
        want[dst:dst + ln] = kept[src:src + ln]
    got = buf[0].cpu().numpy()
    for dst, src, ln in segs:
        assert (got[dst:dst + ln] == want[dst:dst + ln]).all()
    assert (buf[4].cpu().numpy()[segs[0][0]:segs[0][0] + segs[0][2], 0] == 1.0 - term[keep.astype(bool)][:segs[0][2], 0]).all()
    # three-way concatenated gather, the middle source in the OTHER layout
    b2 = make_buf("arrays" if layout == "ring" else "ring", 100, S, A, dev,
                  fill=[rng.standard_normal((100, n)).astype(np.float32) for n in (S, A, S, 1, 1)])
    i1, i2, i3 = rng.integers(0, cap, 33), rng.integers(0, 100, 20), rng.integers(0, cap, 0)
    out = ops.gather_batch([buf, b2, buf], [td(i1), td(i2), td(i3)], S, A)
    for k in range(5):
        want = np.concatenate([buf[k].cpu().numpy()[i1], b2[k].cpu().numpy()[i2]], 0)
        assert (out[k].cpu().numpy() == want).all()


def test_replay_buffer_mirror_add_batch_sep_equals_add_batch(dev):
    """ReplayBuffer.add_batch_sep (utils.py:94-125, the driver's dataset fill) == add_batch on the same rows, incl. the wrap."""
    from mobody_amd.algo import utils
    rng = np.random.default_rng(3)
    S, A, cap = 5, 2, 50
    a, b = utils.ReplayBuffer(S, A, dev, max_size=cap), utils.ReplayBuffer(S, A, dev, max_size=cap)
    for M in (30, 35, 7):
        rows = (rng.standard_normal((M, S)).astype(np.float32), rng.standard_normal((M, A)).astype(np.float32),
                rng.standard_normal((M, S)).astype(np.float32), rng.standard_normal((M, 1)).astype(np.float32),
                (rng.uniform(size=(M, 1)) > 0.7).astype(np.float32))
        a.add_batch_sep(*rows)
        b.add_batch(dict(obss=rows[0], actions=rows[1], next_obss=rows[2], rewards=rows[3], terminals=rows[4]))
        assert (a.ptr, a.size) == (b.ptr, b.size)
    assert (a.ptr, a.size) == (22, 50)                       # 30 -> 15 after the wrap of 35 -> 22
    for x, y in zip(a._fields(), b._fields()):
        assert torch.equal(x, y)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_ring_append_random_sequences_vs_reference_arithmetic(dev, layout):
    """Random append sequences (sizes up to the capacity, random keep masks, hits of the exact-wrap cases) against a
    NumPy replay of add_batch's single-wrap slice arithmetic (utils.py:43-92) on the kept rows."""
    from mobody_amd import ops
    rng = np.random.default_rng(11)
    S, A = 3, 2
    for case in range(25):
        cap = int(rng.integers(5, 200))
        buf = make_buf(layout, cap, S, A, dev)
        ps = torch.zeros(2, dtype=torch.int64, device=dev)
        ref = [np.zeros((cap, n), np.float32) for n in (S, A, S, 1, 1)]
        ptr = size = 0
        for step in range(6):
            M = int(rng.choice([1, cap - ptr if cap > ptr else 1, cap, int(rng.integers(1, cap + 1))]))
            rows = [rng.standard_normal((M, n)).astype(np.float32) for n in (S, A, S, 1)]
            term = (rng.uniform(size=(M, 1)) > 0.6).astype(np.uint8)
            keep = (rng.uniform(size=M) > 0.35).astype(np.uint8) if step % 2 else None
            td = lambda x: torch.from_numpy(x).to(dev).contiguous()
            ops.ring_append(buf, cap, ps, S, A, td(rows[0]), td(rows[1]), td(rows[2]), td(rows[3]), td(term),
                            None if keep is None else td(keep))
            sel = np.ones(M, bool) if keep is None else keep.astype(bool)
            kept = [r[sel] for r in rows] + [1.0 - term[sel].astype(np.float32)]
            K = int(sel.sum())
            if K:                                                   # the reference's slice arithmetic on the kept rows
                end = min(ptr + K, cap)
                used = end - ptr
                for dst, src in zip(ref, kept):
                    dst[ptr:end] = src[:used]
                ptr = end % cap
                size = min(size + used, cap)
                if ptr == 0:
                    for dst, src in zip(ref, kept):
                        dst[0:K - used] = src[used:]
                    ptr = K - used
            assert ps.cpu().tolist() == [ptr, size], (case, step, cap, M, K)
            for t, want in zip(buf, ref):
                assert (t.cpu().numpy() == want).all(), (case, step)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_on_device_rollout_equals_the_step_by_step_composition(dev, layout):
    """`mobody_rollout` (H steps, fused mask, two-launch append) == the same rollout assembled from the stand-alone entry
    points (actor forward -> mobody_dyn_step -> mobody_rollout_mask -> mobody_ring_append), bit for bit, into a ring that
    wraps (cap < rows appended)."""
    import golden_util as gu
    from mobody_amd import ops, packing
    S, A, B, H, cap = 17, 6, 700, 4, 1500
    p = gu.gi.dyn_params(201, S, A)
    p["transition3.bias"][:, 0, 0] += np.float32(0.85)
    pa, _, _ = gu.policy_params(301, S, A)
    dyn = packing.pack_dynamics(p, S, A, dev)
    actor = packing.pack_mlp([{k[len("network."):]: v for k, v in pa.items()}], S, A, dev)
    init = torch.from_numpy(gu.gi.walker_like_obs(np.random.default_rng(4), B, S)).to(dev)
    elites = (0, 2, 3, 5, 6)

    def ring():
        return make_buf(layout, cap, S, A, dev), torch.tensor([100, 100], dtype=torch.int64, device=dev)

    buf1, ps1 = ring()
    ops.rollout(dyn, actor, S, A, 4, 1.0, init, H, elites, 21, 7, 0.1, True, True, 0.5, True, buf1, cap, ps1)
    buf2, ps2 = ring()
    obs, alive = init, None
    keep = torch.empty(B, dtype=torch.uint8, device=dev)
    for t in range(H):
        act = ops.mlp3_forward(actor, S, A, 1, obs, out_mode=1, max_action=1.0)[0]
        r = ops.dyn_step(dyn, S, A, 4, obs, act, alive=alive, elites=elites, seed=21, call=7 + t, penalty_coef=0.1)
        nalive = torch.empty(B, dtype=torch.uint8, device=dev)
        ops.rollout_mask(alive, r["terminal"], r["penalty"], 0.5, True, keep, nalive)
        ops.ring_append(buf2, cap, ps2, S, A, obs, act, r["next_obs"], r["reward"], r["terminal"], keep)
        obs, alive = r["next_obs"], nalive
    assert ps1.tolist() == ps2.tolist() and ps1.tolist()[1] == cap          # the ring filled up and wrapped
    for a, b in zip(buf1, buf2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("S,A", [(17, 6), (111, 8), (45, 24), (3, 1)])
def test_ring_layout_row_padding_and_wide_rows(dev, S, A):
    """The row-interleaved ring at every task width (ant rows span 15 sectors): append + gather round trip equals the
    separate-array path bit for bit, and the padding floats of every written row are zero."""
    from mobody_amd import ops
    rng = np.random.default_rng(S)
    cap, M = 400, 333
    bufs = [make_buf(l, cap, S, A, dev) for l in LAYOUTS]
    bufs[0].store.fill_(7.0)                                  # stale contents: written rows must come out clean
    rows = [rng.standard_normal((M, n)).astype(np.float32) for n in (S, A, S, 1)]
    term = (rng.uniform(size=(M, 1)) > 0.5).astype(np.uint8)
    keep = (rng.uniform(size=M) > 0.25).astype(np.uint8)
    td = lambda x: torch.from_numpy(x).to(dev).contiguous()
    pss = []
    for b in bufs:
        ps = torch.tensor([350, 360], dtype=torch.int64, device=dev)
        ops.ring_append(b, cap, ps, S, A, td(rows[0]), td(rows[1]), td(rows[2]), td(rows[3]), td(term), td(keep))
        pss.append(ps.tolist())
    assert pss[0] == pss[1]
    K = int(keep.sum())
    written = [(350 + j) % cap if 350 + j < cap else j - (cap - 350) for j in range(K)]
    W = 2 * S + A + 2
    st = bufs[0].store.cpu().numpy()
    assert (st[written, W:] == 0).all() and (st[written, :W] != 7.0).all()
    idx = td(np.asarray(written + written[::-1], np.int32))
    o1, o2 = ops.gather_batch([bufs[0]], [idx], S, A), ops.gather_batch([bufs[1]], [idx], S, A)
    for x, y in zip(o1, o2):
        assert torch.equal(x, y)


def test_replay_buffer_attribute_assignment_reaches_the_store(dev):
    """The reference mutates the public fields both ways: `buf.reward -= 1.0` (train_mobody.py:551,557) and
    `buf.reward = new_rewards` (mobody.py:381).  Either must be what sample() returns afterwards."""
    from mobody_amd.algo import utils
    rng = np.random.default_rng(8)
    S, A, n = 5, 2, 40
    rb = utils.ReplayBuffer(S, A, dev, max_size=n)
    rows = [rng.standard_normal((n, k)).astype(np.float32) for k in (S, A, S)]
    rb.convert_D4RL(dict(observations=rows[0], actions=rows[1], next_observations=rows[2],
                         rewards=np.arange(n, dtype=np.float32), terminals=np.zeros(n, bool)))
    rb.draw_indices = lambda k: torch.arange(k, dtype=torch.int32, device=dev)
    rb.reward -= 1.0
    assert (rb.sample(n)[3].cpu().numpy()[:, 0] == np.arange(n) - 1.0).all()
    rb.reward = torch.full((n, 1), 7.0)                          # a new CPU tensor, as mobody.py:381 assigns
    assert (rb.sample(n)[3].cpu().numpy() == 7.0).all() and (rb.store[:, 2 * S + A] == 7.0).all()
    rb.state = rows[2]                                           # numpy array
    assert (rb.sample(n)[0].cpu().numpy() == rows[2]).all()
    with pytest.raises(ValueError):
        rb.reward = torch.zeros(n + 1, 1)
