"""reagent_amd.replay_memory.ReplayBuffer: bit-exact against (a) golden outputs of the reference
ReplayBuffer (tests/golden/replay_*.npz), (b) the reference's own known-answer tests
(reagent/test/replay_memory/circular_replay_buffer_test.py:60-352, restated here), (c) the numpy
oracle on a larger seeded case, and size-independent properties (gather == index_select)."""
import numpy as np
import pytest
import torch

from golden_util import Golden
from oracle import restated as R
from reagent_amd.replay_memory import ReplayBuffer

OBS = (4, 3)


@pytest.mark.parametrize("name", ["replay_basic", "replay_nstep", "replay_stack", "replay_all_stack", "replay_timeline",
                                  "replay_timeline_stack"])
def test_matches_reference_golden_bit_exact(backend, name):
    g = Golden(name)
    c = g.cfg
    rb = ReplayBuffer(stack_size=c["stack_size"], replay_capacity=c["replay_capacity"], batch_size=c["batch"],
                      update_horizon=c["update_horizon"], gamma=c["gamma"], device=backend.device,
                      return_everything_as_stack=c.get("return_everything_as_stack", False),
                      return_as_timeline_format=c.get("return_as_timeline_format", False))
    keys = ["observation", "action", "reward", "terminal", "possible_actions_mask", "log_prob", "mdp_id"]
    for i in range(c["n_add"]):
        tr = {k: g.a(f"add_{k}")[i] for k in keys}
        tr["terminal"] = bool(tr["terminal"])
        rb.add(**tr)
    assert int(rb.add_count) == int(g.a("add_count"))
    assert rb.size == int(g.a("size"))
    np.testing.assert_array_equal(rb._is_index_valid.numpy(), g.a("valid_mask"))
    batch = rb.sample_transition_batch(batch_size=c["batch"], indices=torch.from_numpy(g.a("indices")))
    fields = [f[len("out_"):] for f in g.z.files if f.startswith("out_") and not f.endswith("_flat")]
    ragged = [f[len("out_"):-len("_flat")] for f in g.z.files if f.startswith("out_") and f.endswith("_flat")]
    assert set(fields) | set(ragged) == set(batch._fields)
    assert bool(ragged) == bool(c.get("return_as_timeline_format"))
    for k in ragged:  # timeline format (:716-741): a list with one [steps[i], ...] tensor per transition
        got, steps = getattr(batch, k), batch.step.reshape(-1).tolist()
        assert isinstance(got, list) and [len(t) for t in got] == steps
        flat, ref = torch.cat(got, dim=0).cpu().numpy(), g.a(f"out_{k}_flat")
        assert flat.shape == ref.shape and flat.dtype == ref.dtype and flat.tobytes() == ref.tobytes(), k
    assert list(batch._fields[:9]) == ["state", "action", "reward", "next_state", "next_action",
                                       "next_reward", "terminal", "indices", "step"]  # :776-793
    for k in fields:
        got, ref = getattr(batch, k).cpu().numpy(), g.a(f"out_{k}")
        assert got.shape == ref.shape and got.dtype == ref.dtype, (k, got.shape, got.dtype, ref.shape, ref.dtype)
        assert got.tobytes() == ref.tobytes(), k  # bit exact (incl. fp32 n-step reward)


def test_reference_known_answers_add_and_errors(backend):
    m = ReplayBuffer(stack_size=4, replay_capacity=5, batch_size=8, device=backend.device)
    assert m.cursor() == 0
    m.add(observation=np.zeros(OBS), action=0, reward=0, terminal=0)
    assert m.cursor() == 4  # STACK_SIZE - 1 padding adds + 1 (test :60-68)
    m = ReplayBuffer(stack_size=4, replay_capacity=5, batch_size=8, device=backend.device)
    m.add(observation=np.zeros(OBS), action=0, reward=0, terminal=0, extra1=0, extra2=[0, 0])
    with pytest.raises(ValueError, match="Add expects"):
        m.add(observation=np.zeros(OBS), action=0, reward=0, terminal=0)
    assert m.cursor() == 4
    for kw in (dict(stack_size=10, replay_capacity=10, update_horizon=1), dict(stack_size=5, replay_capacity=10, update_horizon=10)):
        with pytest.raises(ValueError, match="There is not enough capacity"):
            ReplayBuffer(batch_size=8, gamma=1.0, device=backend.device, **kw)
    ReplayBuffer(stack_size=5, replay_capacity=10, batch_size=8, update_horizon=5, gamma=1.0, device=backend.device)


def test_reference_known_answers_nstep_sum(backend):
    m = ReplayBuffer(stack_size=4, replay_capacity=10, batch_size=8, update_horizon=5, gamma=1.0,
                     device=backend.device)
    for i in range(50):
        m.add(observation=np.full(OBS, i, dtype=np.uint8), action=0, reward=2.0, terminal=0)
    for _ in range(5):
        batch = m.sample_transition_batch()
        assert batch[2][0].item() == 10.0  # test :115-135


def test_reference_known_answers_sample_transition_batch(backend):
    C, num_adds = 10, 50
    m = ReplayBuffer(stack_size=1, replay_capacity=C, batch_size=2, device=backend.device)
    for i in range(num_adds):
        m.add(observation=np.full(OBS, i, np.uint8), action=0, reward=0, terminal=i % 4)
    assert m.sample_transition_batch()[0].shape[0] == 2
    assert m.sample_transition_batch(8)[0].shape[0] == 8
    indices = [1, 2, 3, 5, 8]
    exp_states = np.array([np.full(OBS, i, dtype=np.uint8) for i in indices])
    exp_next = (exp_states + 1) % C
    exp_states += num_adds - C
    exp_next += num_adds - C
    exp_term = np.expand_dims(np.array([min((x + num_adds - C) % 4, 1) for x in indices]), 1).astype(bool)
    b = m.sample_transition_batch(batch_size=len(indices), indices=torch.tensor(indices))
    np.testing.assert_array_equal(b.state.cpu(), exp_states)
    np.testing.assert_array_equal(b.action.cpu(), np.zeros((5, 1)))
    np.testing.assert_array_equal(b.reward.cpu(), np.zeros((5, 1)))
    np.testing.assert_array_equal(b.next_action.cpu(), np.zeros((5, 1)))
    np.testing.assert_array_equal(b.next_reward.cpu(), np.zeros((5, 1)))
    np.testing.assert_array_equal(b.next_state.cpu(), exp_next)
    np.testing.assert_array_equal(b.terminal.cpu(), exp_term)
    np.testing.assert_array_equal(b.indices.cpu(), np.expand_dims(np.array(indices), 1))


def test_reference_known_answers_multistep_and_validity(backend):
    # circular_replay_buffer_test.py:271-312: rewards [5,3,15]-style n-step sums and terminals
    m = ReplayBuffer(stack_size=1, replay_capacity=10, batch_size=2, update_horizon=5, gamma=1.0,
                     device=backend.device)
    for i in range(50):
        m.add(observation=np.full(OBS, i, np.uint8), action=0, reward=1.0, terminal=(i % 8 == 7))
    # cursor = 0; slots hold i = 40..49, terminal at i = 47 (slot 7)
    b = m.sample_transition_batch(batch_size=3, indices=torch.tensor([3, 5, 0]))
    np.testing.assert_array_equal(b.reward.cpu().numpy().ravel(), [5.0, 3.0, 5.0])
    np.testing.assert_array_equal(b.terminal.cpu().numpy().ravel(), [True, True, False])
    np.testing.assert_array_equal(b.step.cpu().numpy().ravel(), [5, 3, 5])
    # validity (:314-352 rule set): last `update_horizon` slots before the cursor are invalid unless terminal-closed
    m2 = ReplayBuffer(stack_size=1, replay_capacity=10, batch_size=2, update_horizon=3, gamma=1.0,
                      device=backend.device)
    o = R.ReplayOracle(stack_size=1, replay_capacity=10, update_horizon=3, gamma=1.0)
    for i in range(23):
        tr = dict(observation=np.full((2,), i, np.float32), action=np.int64(0), reward=np.float32(1.0), terminal=(i % 6 == 5))
        m2.add(**tr)
        o.add(**tr)
        np.testing.assert_array_equal(m2._is_index_valid.numpy(), o.valid)
        assert m2.size == int(o.valid.sum())


def test_large_seeded_case_vs_oracle_and_properties(backend):
    """C2-schema buffer (smaller C), bulk loaded: gather == index_select for every column."""
    from reagent_amd import synthetic

    C, S, A, B = 4096, 128, 16, 1000
    cols = synthetic.replay_contents(C, S, A, seed=3, p_terminal=0.01)
    rb = ReplayBuffer(replay_capacity=C, batch_size=B, device=backend.device)
    rb.load_columns({k: v.to(backend.device) for k, v in cols.items()}, mark_all_valid=False)
    o = R.ReplayOracle(replay_capacity=C)
    for i in range(C):
        o.add(**{k: v[i].numpy() for k, v in cols.items()})
    np.testing.assert_array_equal(rb._is_index_valid.numpy(), o.valid)  # closed-form == per-add rules
    idx = rb.sample_index_batch(B)
    assert bool(torch.from_numpy(o.valid)[idx.cpu()].all())
    b = rb.sample_transition_batch(B, indices=idx)
    ref = o.sample(idx.cpu().numpy(), extra_keys=["possible_actions_mask", "log_prob"])
    for k, v in ref.items():
        got = getattr(b, k).cpu().numpy()
        assert got.dtype == v.dtype and got.tobytes() == v.tobytes(), k
    assert torch.equal(b.state.cpu(), cols["observation"][idx.cpu()])
    assert torch.equal(b.next_state.cpu(), cols["observation"][(idx.cpu() + 1) % C])


def test_empty_and_unsupported(backend):
    rb = ReplayBuffer(replay_capacity=10, batch_size=2, device=backend.device)
    rb.add(observation=np.zeros(3, np.float32), action=0, reward=0.0, terminal=False)
    with pytest.raises(RuntimeError, match="no valid indices"):
        rb.sample_index_batch(2)
    rb = ReplayBuffer(replay_capacity=10, stack_size=2, device=backend.device)
    with pytest.raises(NotImplementedError):  # the reference leaves stacked sparse elements a TODO too (:150)
        rb.add(observation=np.zeros(3, np.float32), action=0, reward=0.0, terminal=False, id_list={"a": [1, 2]})


@pytest.mark.parametrize("S", [24, 32])  # 4-feature slots per row: 6 (generic loop) / 8 (divides the workgroup: slot-per-thread path)
def test_normalize_on_gather_equals_gather_then_preprocessor(backend, S):
    """Preprocessor.forward fused into the gather: bit-identical to gather -> Preprocessor (fp32),
    and to that result rounded to bf16 (the bf16 path's network-ready layout)."""
    from reagent_amd import synthetic
    from reagent_amd.core.parameters import NormalizationParameters as NP
    from reagent_amd.preprocessing import Preprocessor

    C, A, B = 2048, 4, 333
    cols = synthetic.replay_contents(C, S, A, seed=8)
    rb = ReplayBuffer(replay_capacity=C, batch_size=B, device=backend.device)
    rb.load_columns({k: v.to(backend.device) for k, v in cols.items()}, mark_all_valid=True)
    g = torch.Generator().manual_seed(1)
    norm = {}
    for i in range(S):
        if i % 3 == 0:
            norm[i] = NP(feature_type="CONTINUOUS", mean=torch.randn(1, generator=g).item(), stddev=1.3)
        elif i % 3 == 1:
            norm[i] = NP(feature_type="QUANTILE", quantiles=[-1.0, 0.0, 0.5, 2.0])
        else:
            norm[i] = NP(feature_type="BINARY")
    pre = Preprocessor(norm, device=backend.device)
    assert pre.elementwise
    idx = rb.sample_index_batch(B)
    plain = rb.sample_transition_batch(B, indices=idx)
    ones = torch.ones(B, S, dtype=torch.uint8, device=backend.device)
    # Preprocessor columns are in sorted_features order == replay column order here? build the permutation
    perm = torch.tensor(pre.sorted_features, device=backend.device)
    want_s, want_n = pre(plain.state[:, perm], ones), pre(plain.next_state[:, perm], ones)
    # normalize-on-gather applies descriptor j to column j: give it a table built for identity order
    pre_id = Preprocessor({k: norm[f] for k, f in enumerate(pre.sorted_features)}, device=backend.device)
    rb2 = ReplayBuffer(replay_capacity=C, batch_size=B, device=backend.device)
    cols2 = dict(cols)
    cols2["observation"] = cols["observation"][:, perm.cpu()]
    rb2.load_columns({k: v.to(backend.device) for k, v in cols2.items()}, mark_all_valid=True)
    fused = rb2.sample_transition_batch(B, indices=idx, state_preprocessor=pre_id)
    assert torch.equal(fused.state, want_s) and torch.equal(fused.next_state, want_n)
    assert torch.equal(fused.reward, plain.reward) and torch.equal(fused.action, plain.action)
    fused16 = rb2.sample_transition_batch(B, indices=idx, state_preprocessor=pre_id, state_dtype=torch.bfloat16)
    assert fused16.state.dtype == torch.bfloat16
    assert torch.equal(fused16.state.cpu(), want_s.cpu().to(torch.bfloat16))


# ---- checkpointing in the reference's file layout (circular_replay_buffer.py:795-890) -------------
def _filled(backend, n=37, cap=50, horizon=2):
    rb = ReplayBuffer(stack_size=1, replay_capacity=cap, batch_size=8, update_horizon=horizon, gamma=0.9,
                      device=backend.device)
    rng = np.random.RandomState(4)
    for i in range(n):
        rb.add(observation=rng.randn(*OBS).astype(np.float32), action=np.int64(rng.randint(3)),
               reward=np.float32(rng.rand()), terminal=bool(rng.rand() < 0.2), mdp_id=np.int64(i // 5))
    return rb


def test_save_load_round_trip_and_reference_layout(backend, tmp_path):
    import gzip

    rb = _filled(backend)
    rb.save(str(tmp_path), 7)
    # the reference's file set: one gzip'd np.save per storage column ("$store$_<key>") and per public
    # numpy attribute (add_count), named "<attr>_ckpt.<iteration>.gz"
    for key in rb._store:
        f = tmp_path / f"$store$_{key}_ckpt.7.gz"
        assert f.exists()
        with gzip.GzipFile(str(f)) as g:
            arr = np.load(g, allow_pickle=False)
        np.testing.assert_array_equal(arr, rb._store[key].cpu().numpy())
    with gzip.GzipFile(str(tmp_path / "add_count_ckpt.7.gz")) as g:
        assert int(np.load(g, allow_pickle=False)) == 37
    fresh = _filled(backend, n=3)  # same schema, different contents
    fresh.load(str(tmp_path), 7)
    assert int(fresh.add_count) == 37 and fresh.size == rb.size and fresh.cursor() == rb.cursor()
    np.testing.assert_array_equal(fresh._valid_host, rb._valid_host)
    idx = torch.from_numpy(np.flatnonzero(rb._valid_host)[:8].astype(np.int64))
    a, b = rb.sample_transition_batch(8, indices=idx), fresh.sample_transition_batch(8, indices=idx)
    for k in a._fields:
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    # adding continues identically after the restore
    for m in (rb, fresh):
        m.add(observation=np.ones(OBS, dtype=np.float32), action=np.int64(1), reward=np.float32(0.5), terminal=True,
              mdp_id=np.int64(99))
    np.testing.assert_array_equal(fresh._valid_host, rb._valid_host)


def test_save_garbage_collects_and_load_is_all_or_nothing(backend, tmp_path):
    rb = _filled(backend)
    rb.save(str(tmp_path), 1)
    assert (tmp_path / "add_count_ckpt.1.gz").exists()
    rb.save(str(tmp_path), 5)  # CHECKPOINT_DURATION = 4 iterations back is deleted
    assert (tmp_path / "add_count_ckpt.5.gz").exists() and not (tmp_path / "add_count_ckpt.1.gz").exists()
    rb.save(str(tmp_path / "does_not_exist"), 5)  # silently nothing, like the reference (:823-824)
    fresh = _filled(backend, n=3)
    with pytest.raises(FileNotFoundError):
        fresh.load("/does/not/exist", "3")
    (tmp_path / "$store$_reward_ckpt.5.gz").unlink()
    with pytest.raises(FileNotFoundError, match="reward"):
        fresh.load(str(tmp_path), 5)
    assert int(fresh.add_count) == 3  # nothing was loaded


def test_load_checkpoint_written_by_the_reference_layout_only(backend, tmp_path):
    """a checkpoint holding only the reference's files (no $rg$_* validity state): validity is rebuilt
    from the terminal column"""
    rb = _filled(backend)
    rb.save(str(tmp_path), 2)
    for f in tmp_path.glob("$rg$_*"):
        f.unlink()
    fresh = _filled(backend, n=3)
    fresh.load(str(tmp_path), 2)
    np.testing.assert_array_equal(fresh._valid_host, rb._valid_host)
    assert fresh.size == rb.size


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/reagent"), reason="needs the reference checkout")
def test_checkpoints_interchange_with_the_reference_class(emu_lib, tmp_path):
    """files written by the reference ReplayBuffer load here and vice versa (build container only)"""
    from oracle import reference_harness as rh

    rh._install()
    from reagent.replay_memory.circular_replay_buffer import ReplayBuffer as RefBuffer

    def fill(rb, n):
        rng = np.random.RandomState(4)
        for i in range(n):
            rb.add(observation=rng.randn(*OBS).astype(np.float32), action=np.int64(rng.randint(3)),
                   reward=np.float32(rng.rand()), terminal=bool(rng.rand() < 0.2), mdp_id=np.int64(i // 5))
        return rb

    kw = dict(stack_size=1, replay_capacity=50, batch_size=8, update_horizon=2, gamma=0.9)
    ref, mine = fill(RefBuffer(**kw), 37), fill(ReplayBuffer(device="cpu", **kw), 37)
    d_ref, d_mine = tmp_path / "ref", tmp_path / "mine"
    d_ref.mkdir(), d_mine.mkdir()
    ref.save(str(d_ref), 3)
    mine.save(str(d_mine), 3)
    assert {f.name for f in d_ref.iterdir()} <= {f.name for f in d_mine.iterdir()}  # same names (+ $rg$_ extras)
    loaded = fill(ReplayBuffer(device="cpu", **kw), 2)
    loaded.load(str(d_ref), 3)  # reference -> here
    assert int(loaded.add_count) == 37
    np.testing.assert_array_equal(loaded._valid_host, ref._is_index_valid.numpy())
    ref2 = fill(RefBuffer(**kw), 2)
    ref2.load(str(d_mine), 3)  # here -> reference
    assert int(ref2.add_count) == 37
    for k in ref._store:  # rows never written are uninitialised memory in the reference
        np.testing.assert_array_equal(ref2._store[k].numpy()[:37], ref._store[k].numpy()[:37])


def _fill_for_timeline(rb, n, discrete):
    rng = np.random.RandomState(9)
    for i in range(n):
        action = np.int64(rng.randint(3)) if discrete else rng.rand(2).astype(np.float32)
        rb.add(observation=rng.randn(4).astype(np.float32), action=action, reward=np.float32(rng.rand()),
               terminal=bool(rng.rand() < 0.25), mdp_id=np.int64(i // 6), sequence_number=np.int64(i % 6),
               log_prob=np.float32(-rng.rand()))
    return rb


@pytest.mark.parametrize("discrete", [True, False])
def test_pre_timeline_df_columns(emu_lib, discrete):
    from reagent_amd.replay_memory.utils import replay_buffer_to_pre_timeline_df

    rb = _fill_for_timeline(ReplayBuffer(device="cpu", stack_size=1, replay_capacity=40, batch_size=4), 30, discrete)
    df = replay_buffer_to_pre_timeline_df(discrete, rb)
    assert len(df) == rb.size
    cols = ["ds", "state_features", "action", "mdp_id", "sequence_number", "action_probability", "reward", "metrics"]
    assert list(df.columns)[:8] == cols
    row = df.iloc[0]
    assert row["ds"] == "2019-01-01" and isinstance(row["mdp_id"], str) and set(row["state_features"]) == {0, 1, 2, 3}
    assert row["metrics"] == {"reward": row["reward"]} and 0 < row["action_probability"] <= 1
    if discrete:
        assert isinstance(row["action"], str) and "possible_actions" in df.columns
        for pa, pam in zip(df["possible_actions"], df["possible_actions_mask"]):
            assert (pa == [] and pam == []) or (pa == ["0", "1", "2"] and pam == [1, 1, 1])
    else:
        assert set(row["action"]) == {0, 1} and "possible_actions" not in df.columns


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/reagent"), reason="needs the reference checkout")
@pytest.mark.parametrize("discrete", [True, False])
def test_pre_timeline_df_equals_the_reference(emu_lib, discrete):
    """same buffer contents + same sampled indices -> the same DataFrame as the reference function"""
    from oracle import reference_harness as rh

    rh._install()
    from reagent.replay_memory.circular_replay_buffer import ReplayBuffer as RefBuffer
    from reagent.replay_memory.utils import replay_buffer_to_pre_timeline_df as ref_fn
    from reagent_amd.replay_memory.utils import replay_buffer_to_pre_timeline_df

    kw = dict(stack_size=1, replay_capacity=40, batch_size=4)
    ref = _fill_for_timeline(RefBuffer(**kw), 30, discrete)
    mine = _fill_for_timeline(ReplayBuffer(device="cpu", **kw), 30, discrete)
    idx = ref.sample_index_batch(ref.size)
    ref.sample_index_batch = lambda n: idx
    mine.sample_index_batch = lambda n: idx.clone()
    a, b = ref_fn(discrete, ref), replay_buffer_to_pre_timeline_df(discrete, mine)
    assert list(a.columns) == list(b.columns) and len(a) == len(b)
    for c in a.columns:
        assert a[c].tolist() == b[c].tolist(), c


def _gym_buffer(device, horizon, with_mask, n=90, cap=96, F=8, A=3, seed=12):
    rb = ReplayBuffer(device=device, stack_size=1, replay_capacity=cap, batch_size=16, update_horizon=horizon, gamma=0.9)
    rng = np.random.RandomState(seed)
    for i in range(n):
        kw = dict(observation=rng.randn(F).astype(np.float32), action=np.int64(rng.randint(A)),
                  reward=np.float32(rng.rand()), terminal=bool(rng.rand() < 0.2), log_prob=np.float32(-rng.rand()),
                  mdp_id=np.int64(i // 7), sequence_number=np.int64(i % 7))
        if with_mask:
            kw["possible_actions_mask"] = (rng.rand(A) > 0.3).astype(np.float32)
        rb.add(**kw)
    return rb


@pytest.mark.parametrize("horizon,with_mask,normalize,dtype", [
    (1, True, False, torch.float32), (3, True, True, torch.float32), (3, False, True, torch.bfloat16),
    (2, False, False, torch.float32)])
def test_fused_dqn_input_equals_sample_then_maker(backend, horizon, with_mask, normalize, dtype):
    """rg_replay_dqn_batch == rg_replay_nstep + rg_replay_gather (+ normalize-on-gather) + rg_make_dqn_input,
    bit for bit, on every field of the DiscreteDqnInput (n-step windows crossing terminals included)"""
    from reagent_amd.core.parameters import NormalizationParameters as NP
    from reagent_amd.preprocessing import DiscreteDqnInputMaker, Preprocessor

    dev, F, A = backend.device, 8, 3
    rb = _gym_buffer(dev, horizon, with_mask, F=F, A=A)
    pre = None
    if normalize:
        pre = Preprocessor({i: NP(feature_type="CONTINUOUS", mean=0.1 * i, stddev=1.0 + 0.1 * i) for i in range(F)}, device=dev)
    idx = rb.sample_index_batch(70)
    fused = rb.sample_dqn_input(A, 70, indices=idx, state_preprocessor=pre, state_dtype=dtype)
    assert fused is not None
    tup = rb.sample_transition_batch(70, indices=idx, state_preprocessor=pre, state_dtype=dtype if pre is not None else None)
    ref = DiscreteDqnInputMaker(A)(tup)
    assert fused.state.float_features.dtype == dtype
    for name in ("action", "next_action", "reward", "not_terminal", "possible_actions_mask", "possible_next_actions_mask"):
        assert torch.equal(getattr(fused, name), getattr(ref, name)), name
    assert torch.equal(fused.state.float_features, ref.state.float_features)
    assert torch.equal(fused.next_state.float_features, ref.next_state.float_features)
    assert torch.equal(fused.extras.action_probability, ref.extras.action_probability)
    assert fused.step is None and fused.time_diff is None
    assert (fused.not_terminal == 0).any() and (fused.not_terminal == 1).any()


def _policy_buffer(device, horizon, n=90, cap=96, F=8, A=3, seed=21):
    rb = ReplayBuffer(device=device, stack_size=1, replay_capacity=cap, batch_size=16, update_horizon=horizon, gamma=0.9)
    rng = np.random.RandomState(seed)
    for i in range(n):
        rb.add(observation=rng.randn(F).astype(np.float32), action=(rng.rand(A) * 4 - 2).astype(np.float32),
               reward=np.float32(rng.rand()), terminal=bool(rng.rand() < 0.2), log_prob=np.float32(-rng.rand()))
    return rb


@pytest.mark.parametrize("horizon,normalize,dtype", [(1, False, torch.float32), (3, True, torch.float32), (3, True, torch.bfloat16),
                                                      (2, False, torch.float32)])
def test_fused_policy_input_equals_sample_then_maker(backend, horizon, normalize, dtype):
    """rg_replay_policy_batch (ABI 11) == rg_replay_nstep + rg_replay_gather (+ normalize-on-gather) + rg_make_policy_input, bit
    for bit, on every field of the PolicyNetworkInput (n-step windows crossing terminals, per-dimension action ranges)"""
    from reagent_amd.core.parameters import NormalizationParameters as NP
    from reagent_amd.preprocessing import PolicyNetworkInputMaker, Preprocessor

    dev, F, A = backend.device, 8, 3
    rb = _policy_buffer(dev, horizon, F=F, A=A)
    pre = None
    if normalize:
        pre = Preprocessor({i: NP(feature_type="CONTINUOUS", mean=0.1 * i, stddev=1.0 + 0.1 * i) for i in range(F)}, device=dev)
    maker = PolicyNetworkInputMaker(np.array([-2.0, -1.5, -3.0], dtype=np.float32), np.array([2.0, 2.5, 1.0], dtype=np.float32))
    idx = rb.sample_index_batch(70)
    fused = rb.sample_policy_input(maker, 70, indices=idx, state_preprocessor=pre, state_dtype=dtype)
    assert fused is not None
    ref = maker(rb.sample_transition_batch(70, indices=idx, state_preprocessor=pre, state_dtype=dtype if pre is not None else None))
    assert fused.state.float_features.dtype == dtype
    for name in ("state", "next_state", "action", "next_action"):
        assert torch.equal(getattr(fused, name).float_features, getattr(ref, name).float_features), name
    for name in ("reward", "not_terminal"):
        assert getattr(fused, name).shape == getattr(ref, name).shape and torch.equal(getattr(fused, name), getattr(ref, name)), name
    assert torch.equal(fused.extras.action_probability, ref.extras.action_probability)
    assert fused.step is None and fused.time_diff is None
    assert (fused.not_terminal == 0).any() and (fused.not_terminal == 1).any()
    assert (fused.next_action.float_features[fused.not_terminal.reshape(-1) == 0] == 0).all()


def test_fused_policy_input_declines_other_stores(backend):
    from reagent_amd.preprocessing import PolicyNetworkInputMaker
    from reagent_amd.runtime import OfflinePolicyLoop

    maker = PolicyNetworkInputMaker(np.full(2, -1.0, np.float32), np.full(2, 1.0, np.float32))
    rb = ReplayBuffer(device=backend.device, stack_size=2, replay_capacity=20, batch_size=4)
    for i in range(10):
        rb.add(observation=np.zeros(8, np.float32), action=np.zeros(2, np.float32), reward=np.float32(0), terminal=False,
               log_prob=np.float32(0))
    assert rb.sample_policy_input(maker, 4) is None  # stacked frames
    rb = ReplayBuffer(device=backend.device, stack_size=1, replay_capacity=20, batch_size=4)
    for i in range(10):
        rb.add(observation=np.zeros(6, np.float32), action=np.zeros(2, np.float32), reward=np.float32(i), terminal=False,
               log_prob=np.float32(0))
    assert rb.sample_policy_input(maker, 4) is None  # 6 features: not a multiple of 4
    # the loop then takes the three-launch path, same batch type
    loop = OfflinePolicyLoop(rb, trainer=None, batch_size=4, input_maker=maker)
    b = loop.make_batch(rb.sample_index_batch(4))
    assert b.state.float_features.shape == (4, 6) and b.action.float_features.shape == (4, 2)


def test_fused_dqn_input_declines_other_stores(backend):
    rb = ReplayBuffer(device=backend.device, stack_size=2, replay_capacity=20, batch_size=4)
    for i in range(10):
        rb.add(observation=np.zeros(8, np.float32), action=np.int64(0), reward=np.float32(0), terminal=False,
               log_prob=np.float32(0))
    assert rb.sample_dqn_input(3, 4) is None  # stacked frames
    rb = ReplayBuffer(device=backend.device, stack_size=1, replay_capacity=20, batch_size=4)
    for i in range(10):
        rb.add(observation=np.zeros(6, np.float32), action=np.int64(0), reward=np.float32(0), terminal=False,
               log_prob=np.float32(0))
    assert rb.sample_dqn_input(3, 4) is None  # 6 features: not a multiple of 4


def test_stack_slaughter(backend):
    """reagent/test/replay_memory/extra_replay_buffer_test.py:242-256 (test_stack_slaughter): stack_size 7,
    1..9 trajectories of random length, buffer sized as the reference does; every valid transition's frame
    stacks (state, action, extra) must equal the restatement :45-57 — zero frames before the trajectory
    start, never a frame of the previous trajectory.  (The reward comes back as the n-step sum here; with
    return_everything_as_stack it is stacked too: test_stack_multistep_flags_slaughter below.)"""
    stack = 7
    rng = np.random.RandomState(0)
    for n_traj in range(1, 10):
        lengths = rng.randint(1, 30, size=n_traj).tolist()
        cap = int(sum(lengths) + (n_traj + 1) * (stack - 1))
        rb = ReplayBuffer(device=backend.device, stack_size=stack, replay_capacity=cap, batch_size=1)
        i = 0
        for traj_len in lengths:
            for j in range(traj_len):
                rb.add(observation=(np.ones((3, 3)) * i).astype(np.float32), action=np.int64(i), reward=np.float32(2 * i),
                       terminal=bool(j == traj_len - 1), extra1=np.float32(3 * i))
                i += 1
        got = rb.sample_all_valid_transitions()
        exp_state, exp_scalar, exp_term, last = [], [], [], []
        i = 0
        for traj_len in lengths:
            start = i
            for j in range(traj_len):
                window = range(i - stack + 1, i + 1)
                exp_state.append(np.stack([np.zeros((3, 3)) if k < start else np.ones((3, 3)) * k for k in window], axis=-1))
                exp_scalar.append([0 if k < start else k for k in window])
                exp_term.append(j == traj_len - 1)
                last.append(i)
                i += 1
        frames = np.array(exp_scalar)
        np.testing.assert_array_equal(got.state.cpu().numpy(), np.stack(exp_state).astype(np.float32))
        np.testing.assert_array_equal(got.action.cpu().numpy(), frames)
        np.testing.assert_array_equal(got.extra1.cpu().numpy(), 3.0 * frames.astype(np.float32))
        np.testing.assert_array_equal(got.terminal.cpu().numpy().reshape(-1), np.array(exp_term))
        np.testing.assert_array_equal(got.reward.cpu().numpy().reshape(-1), 2.0 * np.array(last, dtype=np.float32))


def test_reference_known_answers_sparse_input(backend):
    """extra_replay_buffer_test.py:380-470 (test_sparse_input) restated: id-list and id-score-list elements, their
    next_ variants, offsets / ids / scores of a sampled batch"""
    C = 100
    num = C // 2
    m = ReplayBuffer(stack_size=1, replay_capacity=C, update_horizon=1, device=backend.device)

    def trans(i):
        f1, f2 = list(range(0, i % 4)), list(range(i % 4, 4))
        f3 = (list(range(0, i % 7)), [k + 0.5 for k in range(0, i % 7)])
        f4 = (list(range(i % 7, 7)), [k + 0.5 for k in range(i % 7, 7)])
        return dict(observation=np.ones(OBS, dtype=np.uint8), action=int(2 * i), reward=float(3 * i), terminal=i % 4,
                    id_list={"sparse_feat1": f1, "sparse_feat2": f2}, id_score_list={"sparse_feat3": f3, "sparse_feat4": f4})

    for i in range(num):
        m.add(**trans(i))
    indices = list(range(num - 1))
    batch = m.sample_transition_batch(len(indices), torch.tensor(indices))
    res = {"id_list": {"sparse_feat1": ([], []), "sparse_feat2": ([], [])},
           "id_score_list": {"sparse_feat3": ([], [], []), "sparse_feat4": ([], [], [])},
           "next_id_list": {"sparse_feat1": ([], []), "sparse_feat2": ([], [])},
           "next_id_score_list": {"sparse_feat3": ([], [], []), "sparse_feat4": ([], [], [])}}
    for i in range(num - 1):
        for k, src in (("id_list", trans(i)), ("id_score_list", trans(i)), ("next_id_list", trans(i + 1)), ("next_id_score_list", trans(i + 1))):
            orig = k[len("next_"):] if k.startswith("next_") else k
            for feat in res[k]:
                res[k][feat][0].append(len(res[k][feat][1]))
                if orig == "id_list":
                    res[k][feat][1].extend(src[orig][feat])
                else:
                    res[k][feat][1].extend(src[orig][feat][0])
                    res[k][feat][2].extend(src[orig][feat][1])
    for k in res:
        got = getattr(batch, k)
        for feat, want in res[k].items():
            assert len(got[feat]) == len(want)
            assert got[feat][0].dtype == torch.int32 and got[feat][1].dtype == torch.int64
            for g_, w_ in zip(got[feat], want):
                np.testing.assert_array_equal(g_.cpu().numpy(), np.asarray(w_, dtype=g_.cpu().numpy().dtype))
    m.sample_transition_batch(10)  # sample random
    with pytest.raises(AssertionError, match="not in"):
        m.add(**{**trans(3), "id_list": {"unknown_feature": [1]}})
    long = trans(5)
    long["id_list"] = {"sparse_feat1": list(range(37)), "sparse_feat2": []}  # longer than the slots: they grow
    m.add(**long)
    got = m.sample_transition_batch(1, torch.tensor([num]))
    assert got.id_list["sparse_feat1"][1].cpu().tolist() == list(range(37)) and got.id_list["sparse_feat2"][1].numel() == 0


def test_stack_multistep_flags_slaughter(backend):
    """extra_replay_buffer_test.py:137-240, :258-275 (test_stack_multistep_flags_slaughter) restated: stack_size 5,
    multi_steps 6, return_everything_as_stack and return_as_timeline_format — stacked state / action / extra of every valid
    transition, terminal = "the trajectory ends within multi_steps", and the timeline lists: `reward` holds the stacked
    rewards of the steps[i] transitions from i on, next_* those of the steps[i] transitions after i (the last entry of a
    terminal transition is undefined and skipped, as in the reference's comparison)."""
    stack, multi = 5, 6
    rng = np.random.RandomState(1)
    for n_traj in range(1, 8):
        lengths = rng.randint(1, 25, size=n_traj).tolist()
        cap = max(int(sum(lengths) + (n_traj + 1) * (stack - 1)), stack + multi)
        rb = ReplayBuffer(device=backend.device, stack_size=stack, replay_capacity=cap, batch_size=1, update_horizon=multi,
                          return_everything_as_stack=True, return_as_timeline_format=True)
        i = 0
        for traj_len in lengths:
            for j in range(traj_len):
                rb.add(observation=(np.ones((3, 3)) * i).astype(np.float32), action=np.int64(i), reward=np.float32(2 * i),
                       terminal=bool(j == traj_len - 1), extra1=np.float32(3 * i))
                i += 1
        got = rb.sample_all_valid_transitions()
        frames = lambda k, start: [0 if w < start else w for w in range(k - stack + 1, k + 1)]  # noqa: E731
        obs = lambda k, start: np.stack([np.zeros((3, 3)) if w < start else np.ones((3, 3)) * w  # noqa: E731
                                         for w in range(k - stack + 1, k + 1)], axis=-1).astype(np.float32)
        steps = got.step.reshape(-1).tolist()
        t, i = 0, 0
        for traj_len in lengths:
            start, end = i, i + traj_len
            for j in range(traj_len):
                terminal = j >= traj_len - multi
                assert bool(got.terminal[t]) == terminal
                np.testing.assert_array_equal(got.state[t].cpu().numpy(), obs(i, start))
                np.testing.assert_array_equal(got.action[t].cpu().numpy(), frames(i, start))
                np.testing.assert_array_equal(got.extra1[t].cpu().numpy(), 3.0 * np.array(frames(i, start), np.float32))
                rew = [frames(k, start) for k in range(i, i + multi) if k < end]
                nxt = [k for k in range(i + 1, i + multi + 1) if k <= end]
                assert steps[t] == len(rew) == len(nxt)
                for name in ("reward", "next_action", "next_extra1", "next_state"):
                    assert isinstance(getattr(got, name), list) and getattr(got, name)[t].shape[0] == steps[t], name
                cut = -1 if terminal else None
                np.testing.assert_array_equal(got.reward[t].cpu().numpy()[:cut], 2.0 * np.array(rew, np.float32).reshape(len(rew), stack)[:cut])
                np.testing.assert_array_equal(got.next_action[t].cpu().numpy()[:cut], np.array([frames(k, start) for k in nxt]).reshape(len(nxt), stack)[:cut])
                np.testing.assert_array_equal(got.next_extra1[t].cpu().numpy()[:cut],
                                              3.0 * np.array([frames(k, start) for k in nxt], np.float32).reshape(len(nxt), stack)[:cut])
                np.testing.assert_array_equal(got.next_state[t].cpu().numpy()[:cut], np.stack([obs(k, start) for k in nxt])[:cut])
                t += 1
                i += 1
        assert t == len(steps)


def test_reference_known_answers_replay_overflow(backend):
    """extra_replay_buffer_test.py:277-377 (test_replay_overflow), the reference's exact vectors: capacity 6, stack 2,
    multi_steps 2, timeline format — validity after every add as the cursor wraps, the stacked actions of the valid
    transitions and their next_action lists (entries the reference calls garbage are not compared)"""
    m = ReplayBuffer(stack_size=2, replay_capacity=6, batch_size=1, update_horizon=2, return_everything_as_stack=None,
                     return_as_timeline_format=True, device=backend.device)
    trans = lambda i: dict(observation=np.ones(OBS, dtype=np.uint8), action=int(2 * i), reward=float(3 * i))  # noqa: E731
    valid = lambda: m._is_index_valid.cpu().numpy().tolist()  # noqa: E731
    F, T = False, True
    assert valid() == [F] * 6
    m.add(**trans(0), terminal=False)
    assert valid() == [F] * 6
    m.add(**trans(1), terminal=False)
    assert valid() == [F] * 6
    m.add(**trans(2), terminal=False)  # s0 becomes valid once its next state is in
    assert valid() == [F, T, F, F, F, F]
    b = m.sample_all_valid_transitions()
    assert b.action.cpu().tolist() == [[0, 0]] and b.next_action[0].cpu().tolist() == [[0, 2], [2, 4]]
    m.add(**trans(3), terminal=True)  # the episode's end validates the whole episode
    assert valid() == [F, T, T, T, T, F]
    b = m.sample_all_valid_transitions()
    assert b.action.cpu().tolist() == [[0, 0], [0, 2], [2, 4], [4, 6]]
    assert b.next_action[0].cpu().tolist() == [[0, 2], [2, 4]] and b.next_action[1].cpu().tolist() == [[2, 4], [4, 6]]
    assert b.next_action[2][0].cpu().tolist() == [4, 6]
    m.add(**trans(4), terminal=False)  # wraps: s0's previous frame is overwritten
    assert valid() == [F, F, T, T, T, F]
    b = m.sample_all_valid_transitions()
    assert b.action.cpu().tolist() == [[0, 2], [2, 4], [4, 6]]
    assert b.next_action[0].cpu().tolist() == [[2, 4], [4, 6]] and b.next_action[1][0].cpu().tolist() == [4, 6]
    m.add(**trans(5), terminal=False)
    assert valid() == [F, F, F, T, T, F]
    b = m.sample_all_valid_transitions()
    assert b.action.cpu().tolist() == [[2, 4], [4, 6]] and b.next_action[0][0].cpu().tolist() == [4, 6]
    m.add(**trans(6), terminal=True)
    assert valid() == [T, T, T, F, T, F]
    b = m.sample_all_valid_transitions()
    assert b.action.cpu().tolist() == [[0, 8], [8, 10], [10, 12], [4, 6]]
    assert b.next_action[0].cpu().tolist() == [[8, 10], [10, 12]] and b.next_action[1][0].cpu().tolist() == [10, 12]


@pytest.mark.parametrize("horizon", [1, 3])
def test_add_many_equals_the_same_transitions_added_one_by_one(backend, horizon):
    """ReplayBuffer.add_many (bulk rows as one indexed device copy per column, the validity rules of add in closed form)
    against `add` called per transition — and, in the build container, against the REFERENCE class's add: same storage,
    cursor, validity mask, episode counter, over several calls, episode boundaries at every offset and two wraps of the ring."""
    import numpy as np

    from reagent_amd.replay_memory import ReplayBuffer

    C, F, A = 97, 5, 3
    rng = np.random.default_rng(7 + horizon)
    bulk = ReplayBuffer(replay_capacity=C, batch_size=4, update_horizon=horizon, gamma=0.9, device=backend.device)
    one = ReplayBuffer(replay_capacity=C, batch_size=4, update_horizon=horizon, gamma=0.9, device=backend.device)
    ref = None
    from oracle import stubs

    if stubs.reference_available():
        stubs.install()
        from reagent.replay_memory.circular_replay_buffer import ReplayBuffer as RefBuffer

        ref = RefBuffer(replay_capacity=C, batch_size=4, update_horizon=horizon, gamma=0.9)
    total = 0
    for T in (1, 2, 40, 7, 60, 97, 13, 1, 30):
        cols = dict(observation=rng.standard_normal((T, F)).astype(np.float32), action=rng.integers(0, A, T),
                    reward=rng.random(T).astype(np.float32), terminal=rng.random(T) < 0.15,
                    possible_actions_mask=(rng.random((T, A)) < 0.8).astype(np.float32), log_prob=rng.random(T).astype(np.float32))
        bulk.add_many(**{k: torch.from_numpy(v) for k, v in cols.items()})
        for t in range(T):
            row = dict(observation=cols["observation"][t], action=np.int64(cols["action"][t]), reward=np.float32(cols["reward"][t]),
                       terminal=bool(cols["terminal"][t]), possible_actions_mask=cols["possible_actions_mask"][t],
                       log_prob=np.float32(cols["log_prob"][t]))
            one.add(**row)
            if ref is not None:
                ref.add(**row)
        total += T
        assert int(bulk.add_count) == int(one.add_count) == total and bulk.cursor() == one.cursor()
        assert np.array_equal(bulk._valid_host, one._valid_host), (T, total)
        assert bulk._num_valid_indices == one._num_valid_indices == int(one._valid_host.sum())
        assert np.array_equal(bulk._terminal_host, one._terminal_host)
        assert bulk._num_transitions_in_current_episode == one._num_transitions_in_current_episode
        for k in one._store:
            assert torch.equal(bulk._store[k].cpu(), one._store[k].cpu()), k
        if ref is not None:
            assert np.array_equal(bulk._valid_host, ref._is_index_valid.numpy()), (T, total)
            assert bulk._num_valid_indices == ref._num_valid_indices
    # and the sampler reads the same batch from both
    idx = bulk._valid_indices()[:8]
    a, b = bulk.sample_transition_batch(8, indices=idx), one.sample_transition_batch(8, indices=idx)
    for x, y in zip(a, b):
        if isinstance(x, torch.Tensor):
            assert torch.equal(x, y)


def test_add_many_rejects_a_bad_column_before_touching_the_ring(emu_lib):
    """every column's shape and dtype is checked before the first copy — a bad column late in the dict must not
    leave earlier columns overwritten over transitions still marked valid; a float column for an integer element is an
    error as it is for `add`, not a silent truncation."""
    import numpy as np

    from reagent_amd.replay_memory import ReplayBuffer

    rng = np.random.default_rng(3)
    rb = ReplayBuffer(replay_capacity=16, batch_size=2, device="cpu")
    good = lambda T: dict(observation=torch.from_numpy(rng.standard_normal((T, 4)).astype(np.float32)),  # noqa: E731
                          action=torch.from_numpy(rng.integers(0, 3, T)), reward=torch.rand(T), terminal=torch.zeros(T, dtype=torch.bool))
    rb.add_many(**good(10))
    before = {k: v.clone() for k, v in rb._store.items()}
    state = (int(rb.add_count), rb._valid_host.copy(), rb._terminal_host.copy())
    bad_shape = good(4)
    bad_shape["reward"] = torch.rand(4, 2)  # the LAST-but-one key: observation / action come first in the dict
    bad_dtype = good(4)
    bad_dtype["action"] = torch.rand(4) * 3  # float for an int64 element
    for cols in (bad_shape, bad_dtype):
        with pytest.raises(ValueError, match="add_many"):
            rb.add_many(**cols)
        for k, v in rb._store.items():
            assert torch.equal(v, before[k]), k
        assert int(rb.add_count) == state[0] and np.array_equal(rb._valid_host, state[1]) and np.array_equal(rb._terminal_host, state[2])
    rb.add_many(**{k: (v.double() if k == "reward" else v) for k, v in good(3).items()})  # f64 -> f32: same-kind, as add does
    assert int(rb.add_count) == 13
