"""Device SumTree / PrioritizedReplayBuffer against (a) golden outputs of the reference
classes (tests/golden/sumtree_*.npz, prioritized_replay.npz — made by oracle/make_golden.py from
/root/reference), (b) the reference's own known-answer tests
(reagent/test/replay_memory/sum_tree_test.py:53-151, prioritized_replay_buffer_test.py:70-161,
restated here), (c) the numpy oracle, (d) size-independent properties at full capacity."""
import random

import numpy as np
import pytest
import torch

from golden_util import Golden
from oracle import restated as R
from reagent_amd.replay_memory import PrioritizedReplayBuffer, SumTree


def _levels(tree):
    return tree.nodes


# ---- oracle pinned to the reference (CPU) ---------------------------------------------------------
@pytest.mark.parametrize("name", ["sumtree_dyadic_100", "sumtree_real_1000"])
def test_oracle_sumtree_matches_reference_golden(name):
    g = Golden(name)
    t = R.SumTreeOracle(g.cfg["capacity"])
    for i, v in zip(g.a("set_indices"), g.a("set_values")):
        t.set(int(i), float(v))
    for d, lvl in enumerate(t.nodes):
        np.testing.assert_array_equal(lvl, g.a(f"level_{d}"))
    got = np.array([t.sample(float(q)) for q in g.a("queries")])
    np.testing.assert_array_equal(got, g.a("samples"))
    got = np.array([t.sample(q) for q in t.stratified_queries(64, random.Random(5))])
    np.testing.assert_array_equal(got, g.a("stratified_seed5_b64"))
    assert t.max_recorded_priority == float(g.a("max_recorded"))


# ---- device tree vs reference golden --------------------------------------------------------------
@pytest.mark.parametrize("mode", ["scalar_then_walk", "parallel"])
@pytest.mark.parametrize("name", ["sumtree_dyadic_100", "sumtree_real_1000"])
def test_sumtree_matches_reference_golden(backend, name, mode):
    g = Golden(name)
    t = SumTree(g.cfg["capacity"], device=backend.device)
    idx, val = g.a("set_indices"), g.a("set_values")
    if mode == "parallel":  # leaves written by the last pair naming them, levels rebuilt as left + right
        t.set_many(idx, val)
    else:  # the scalar API (a launch per call), then the rest as ONE in-order walk: reference arithmetic
        for i, v in zip(idx[:100], val[:100]):
            t.set(int(i), float(v))
        t.set_many(idx[100:], val[100:], sequential=True)
    lv = _levels(t)
    bit_exact = g.cfg["exact"] or mode == "scalar_then_walk"
    for d in range(len(lv)):
        if bit_exact:  # same operations in the same order as sum_tree.py, or sums that are exact anyway
            np.testing.assert_array_equal(lv[d], g.a(f"level_{d}"))
        else:  # children sums vs accumulated deltas: last-place differences only
            np.testing.assert_allclose(lv[d], g.a(f"level_{d}"), rtol=1e-13, atol=1e-13)
        if d < len(lv) - 1 and mode == "parallel":  # the parallel path's invariant, exactly
            np.testing.assert_array_equal(lv[d], lv[d + 1][0::2] + lv[d + 1][1::2])
    got = t.sample_many(g.a("queries")).cpu().numpy()
    agree = got == g.a("samples")
    assert agree.all() if bit_exact else agree.mean() > 0.995
    random.seed(5)
    strat = np.array(t.stratified_sample(64))
    if bit_exact:
        np.testing.assert_array_equal(strat, g.a("stratified_seed5_b64"))
    assert t.max_recorded_priority == float(g.a("max_recorded"))
    leaves = t.get_many(np.arange(g.cfg["capacity"]), dtype=torch.float64).cpu().numpy()
    if bit_exact:
        np.testing.assert_array_equal(leaves, g.a("leaves")[: g.cfg["capacity"]])


# ---- reference known-answer tests (sum_tree_test.py) ----------------------------------------------
def test_sumtree_reference_unit_tests(backend):
    with pytest.raises(ValueError, match="Sum tree capacity should be positive"):
        SumTree(capacity=-1, device=backend.device)
    tree = SumTree(capacity=100, device=backend.device)
    with pytest.raises(ValueError, match="nonnegative"):
        tree.set(node_index=0, value=-1)
    assert len(SumTree(capacity=1, device=backend.device).nodes) == 1
    assert len(SumTree(capacity=2, device=backend.device).nodes) == 2
    small = SumTree(capacity=1, device=backend.device)
    small.set(0, 1.5)
    assert small.get(0) == 1.5
    with pytest.raises(Exception, match="empty sum tree"):
        tree.sample()
    with pytest.raises(Exception, match="empty sum tree"):
        tree.stratified_sample(5)
    tree.set(node_index=0, value=1.0)
    assert tree.get(0) == 1.0
    for level in tree.nodes:  # testSetValue: the leftmost branch is 1, everything else 0
        assert level[0] == 1.0 and (level[1:] == 0.0).all()
    assert len(tree.nodes[-1]) >= 100
    tree.set(node_index=0, value=0.0)
    tree.set(node_index=5, value=1.0)
    with pytest.raises(ValueError, match=r"query_value must be in \[0, 1\]"):
        tree.sample(query_value=-0.1)
    with pytest.raises(ValueError, match=r"query_value must be in \[0, 1\]"):
        tree.sample(query_value=1.1)
    assert tree.sample() == 5  # singleton
    tree.set(node_index=5, value=0.0)
    tree.set(node_index=2, value=1.0)
    tree.set(node_index=3, value=3.0)
    for _ in range(20):
        random.seed(1)
        assert tree.sample() == 2
        assert tree.sample(query_value=0.1) == 2
    counts = {2: 0, 3: 0}
    for _ in range(60):
        counts[tree.sample()] += 1
    assert counts[2] < counts[3]


def test_sumtree_stratified_and_max_recorded(backend):
    tree = SumTree(capacity=100, device=backend.device)
    k = 32
    for i in range(k):
        tree.set(node_index=i, value=1)
    samples = tree.stratified_sample(k)
    assert samples == list(range(k))  # testStratifiedSampling
    tree2 = SumTree(capacity=100, device=backend.device)
    tree2.set(node_index=0, value=0)
    assert tree2.max_recorded_priority == 1
    for i in range(1, k):
        tree2.set(node_index=i, value=i)
        assert tree2.max_recorded_priority == i


# ---- prioritized buffer vs reference golden -------------------------------------------------------
def test_prioritized_buffer_matches_reference_golden(backend):
    g = Golden("prioritized_replay")
    c = g.cfg
    rb = PrioritizedReplayBuffer(stack_size=1, replay_capacity=c["capacity"], batch_size=c["batch"],
                                 update_horizon=c["update_horizon"], gamma=c["gamma"],
                                 max_sample_attempts=c["max_sample_attempts"], device=backend.device)
    for i in range(len(g.a("add_action"))):
        rb.add(observation=g.a("add_observation")[i], action=g.a("add_action")[i], reward=g.a("add_reward")[i],
               terminal=bool(g.a("add_terminal")[i]), priority=g.a("add_priority")[i])
    np.testing.assert_array_equal(rb._is_index_valid.numpy(), g.a("valid_mask"))
    np.testing.assert_array_equal(rb.sum_tree.nodes[-1], g.a("leaves_after_add"))
    rb.set_priority(g.a("upd_indices"), g.a("upd_values"))
    np.testing.assert_array_equal(rb.sum_tree.nodes[-1], g.a("leaves_after_set"))
    got = rb.get_priority(np.arange(c["capacity"], dtype=np.int32))
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, g.a("get_priority"))
    random.seed(9)  # same host draws as the reference run, including its unstratified re-draws
    np.testing.assert_array_equal(rb.sample_index_batch(c["batch"]).cpu().numpy(), g.a("sample_index_seed9"))
    random.seed(10)
    batch = rb.sample_transition_batch(batch_size=c["batch"])
    for k in batch._fields:
        # `priority` / `next_priority`: the reference creates that storage column but never writes it
        # (prioritized_replay_buffer.py:72-83 drops the key before _add_transition), so its output is
        # whatever the allocation held
        if not g.has(f"out_{k}") or k in ("priority", "next_priority"):
            continue
        want = g.a(f"out_{k}")
        v = getattr(batch, k).cpu().numpy()
        assert v.shape == want.shape, k
        np.testing.assert_array_equal(v, want, err_msg=k)
    assert batch.sampling_probabilities.dtype == torch.float32


# ---- reference known-answer tests (prioritized_replay_buffer_test.py) -----------------------------
SCREEN, STACK, BATCH, CAP = (8, 8), 4, 32, 100


def _mem(backend):
    return PrioritizedReplayBuffer(STACK, CAP, BATCH, max_sample_attempts=10, device=backend.device)


def _add_blank(m, action=0, reward=0.0, terminal=0, priority=1.0):
    m.add(observation=np.zeros(SCREEN), action=action, reward=reward, terminal=terminal, priority=priority)
    return (m.cursor() - 1) % CAP


def test_prioritized_reference_unit_tests(backend):
    m = _mem(backend)
    assert m.cursor() == 0
    _add_blank(m)
    assert m.cursor() == STACK and m.add_count == STACK
    with pytest.raises(ValueError, match="Add expects"):
        m.add(observation=np.zeros(SCREEN), action=0, reward=0.0, terminal=0)
    m = _mem(backend)
    index = _add_blank(m)
    for i in range(index):  # dummy frames enter with priority 0
        assert m.sum_tree.get(i) == 0.0
    with pytest.raises(AssertionError):
        m.get_priority(index)
    with pytest.raises(AssertionError):
        m.get_priority(np.array([index]))
    assert m.get_priority(np.array([index], dtype=np.int32))[0] == 1.0  # testNewElementHasHighPriority
    m = _mem(backend)
    indices = np.array([_add_blank(m) for _ in range(7)], dtype=np.int32)
    m.set_priority(indices, np.arange(7))
    fetched = m.get_priority(np.flip(indices, 0))
    assert [fetched[6 - i] for i in range(7)] == list(range(7))


def test_prioritized_low_priority_and_retries(backend):
    m = _mem(backend)
    _add_blank(m, terminal=0, priority=0.0)
    for _ in range(3):
        _add_blank(m, terminal=1)
    for _ in range(10):
        batch = m.sample_transition_batch(batch_size=2)
        assert bool((batch.terminal == 1).all())
    m = _mem(backend)
    _add_blank(m)
    with pytest.raises(RuntimeError, match="Max sample attempts: Tried 10 times"):
        m.sample_index_batch(2)
    m = _mem(backend)
    for _ in range(CAP - STACK + 2):  # cursor ends at 1
        _add_blank(m)
    assert m.cursor() == 1
    samples = m.sample_index_batch(CAP).cpu().numpy()
    assert (samples >= STACK).all() and (samples <= CAP - 1).all()


# ---- batched update == the reference's sequential loop --------------------------------------------
def test_batched_set_equals_sequential_oracle(backend):
    cap, n = 5000, 20000  # many duplicates: the LAST value written to an index must win
    rng = np.random.RandomState(2)
    idx = rng.randint(cap, size=n).astype(np.int64)
    val = rng.randint(0, 1 << 12, size=n) / 64.0  # dyadic: the oracle's delta sums are exact
    o = R.SumTreeOracle(cap)
    for i, v in zip(idx, val):
        o.set(int(i), float(v))
    t = SumTree(cap, device=backend.device)
    t.set_many(torch.from_numpy(idx).to(backend.device), torch.from_numpy(val).to(backend.device))
    for got, want in zip(t.nodes, o.nodes):
        np.testing.assert_array_equal(got, want)
    q = rng.rand(4096)
    np.testing.assert_array_equal(t.sample_many(q).cpu().numpy(), np.array([o.sample(float(x)) for x in q]))
    assert (t._claim == -1).all()  # scratch handed back clean


# ---- full-size properties (BASELINE capacity 2^20, batch 65536) -----------------------------------
@pytest.mark.gpu
def test_full_size_properties():
    dev = torch.device("cuda")
    cap, B = 1 << 20, 65536
    g = torch.Generator(device=dev).manual_seed(0)
    pri = torch.rand(cap, dtype=torch.float64, device=dev, generator=g) ** 4 + 1e-3
    t = SumTree(cap, device=dev)
    t.set_many(torch.arange(cap, device=dev), pri)
    flat = t._tree
    for d in range(t.depth):  # node == left + right at every level, exactly
        lo, hi = (1 << d) - 1, (1 << (d + 1)) - 1
        child = flat[hi : hi + (1 << (d + 1))]
        assert torch.equal(flat[lo:hi], child[0::2] + child[1::2])
    assert abs(flat[0].item() - pri.sum().item()) <= 1e-9 * pri.sum().item()
    idx = t.stratified_sample(B, generator=g)
    assert idx.shape == (B,) and int(idx.min()) >= 0 and int(idx.max()) < cap
    # stratification: query i lies in [i/B, (i+1)/B) of the cumulative mass, so sampled indices are sorted
    assert bool((idx[1:] >= idx[:-1]).all())
    # and each sample's cumulative-mass interval contains its query segment boundary region
    csum = torch.cumsum(pri, 0)
    lo = torch.where(idx > 0, csum[idx - 1], torch.zeros_like(csum[idx]))
    seg = torch.arange(B, dtype=torch.float64, device=dev) / B * flat[0]
    assert bool((csum[idx] * (1 + 1e-9) >= seg).all()) and bool((lo <= seg + flat[0] / B * (1 + 1e-9)).all())
    # priorities round-trip, and an update of a hot subset is visible immediately
    hot = torch.randint(cap, (B,), device=dev, generator=g)
    t.set_many(hot, torch.full((B,), 1e6, dtype=torch.float64, device=dev))
    idx2 = t.stratified_sample(B, generator=g)
    assert float(torch.isin(idx2, hot).double().mean()) > 0.99


def test_out_of_range_leaves_raise_or_are_skipped(backend, monkeypatch):
    """the reference indexes python lists (IndexError past the end); here host-side indices raise the same
    error, and indices that already live on the device are skipped by the kernels, never dereferenced"""
    from reagent_amd.replay_memory.sum_tree import SumTree

    t = SumTree(100, device=backend.device)
    t.set_many(np.arange(100), np.ones(100))
    for bad in ([100], [127], [-1], [5, 1 << 20]):
        with pytest.raises(IndexError):
            t.set_many(np.array(bad), np.ones(len(bad)))
        with pytest.raises(IndexError):
            t.get_many(np.array(bad))
    with pytest.raises(IndexError):
        t.set(100, 1.0)
    # device-resident indices: no host sync, out-of-range entries ignored (a padding leaf in [100, 128)
    # must never receive priority, or `sample` could return it)
    if backend.name == "emu":  # the interpreter's "device" memory is host memory: skip the host-side check
        monkeypatch.setattr(SumTree, "_check_indices", lambda self, idx: None)
    idx = torch.tensor([3, 100, 127, -5, 1 << 30, 7], dtype=torch.int64).to(backend.device)
    t.set_many(idx, torch.full((6,), 4.0, dtype=torch.float64).to(backend.device))
    leaves = t.nodes[-1]
    assert leaves[3] == 4.0 and leaves[7] == 4.0 and leaves[100:].sum() == 0.0
    assert abs(t.nodes[0][0] - (98 + 8.0)) < 1e-12
    got = t.get_many(idx, dtype=torch.float64).cpu()
    assert got.tolist() == [4.0, 0.0, 0.0, 0.0, 0.0, 4.0]
    many = torch.cat([idx, torch.arange(40, dtype=torch.int64).to(backend.device)])  # the parallel (claim) path
    t.set_many(many, torch.full((46,), 2.0, dtype=torch.float64).to(backend.device))
    assert t.nodes[-1][100:].sum() == 0.0 and t.nodes[-1][:40].sum() == 80.0
