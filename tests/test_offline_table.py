"""OfflineTable + DiscreteDqnBatchPreprocessor (rg_table_dqn_batch) against what the
reference's DiscreteDqnBatchPreprocessor.forward returns for the same rows
(tests/golden/offline_table.npz: every feature type, 15 % missing features, terminal rows with an empty
possible_next_actions_mask and next_action == num_actions, repeated indices).
Pass-through / integer / one-hot fields exact; normalised features 1e-5 (ulp differences of log / pow
between libms, as for rg_normalize_dense)."""
import os
from types import SimpleNamespace

import pytest
import torch

from golden_util import Golden
from reagent_amd.data import OfflineTable
from reagent_amd.preprocessing import DiscreteDqnBatchPreprocessor, Preprocessor

EXACT = ("action", "next_action", "reward", "time_diff", "step", "not_terminal", "possible_actions_mask",
         "possible_next_actions_mask")
EXTRAS = ("mdp_id", "sequence_number", "action_probability")


def load(backend):
    g = Golden("offline_table")
    norm = {int(k): SimpleNamespace(**v) for k, v in g.cfg["norm"].items()}
    pre = Preprocessor(norm, device=backend.device)
    cols = {k[len("table_"):]: g.t(k) for k in g.z.files if k.startswith("table_")}
    return g, pre, cols


def check(out, g, feature_tol=1e-5):
    for k in EXACT:
        got, ref = getattr(out, k).cpu(), g.t(f"out_{k}")
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        assert torch.equal(got.double(), ref.double()), k
    for k in EXTRAS:
        got, ref = getattr(out.extras, k).cpu(), g.t(f"out_{k}")
        assert got.shape == ref.shape and torch.equal(got.double(), ref.double()), k
    assert out.extras.mdp_id.dtype == torch.int64
    for k, got in (("state", out.state.float_features), ("next_state", out.next_state.float_features)):
        ref = g.t(f"out_{k}")
        assert got.shape == ref.shape
        err = (got.float().cpu() - ref).abs().max()
        assert err <= feature_tol + feature_tol * ref.abs().max(), (k, err)


def test_from_table_matches_reference(backend):
    g, pre, cols = load(backend)
    A = g.cfg["num_actions"]
    table = OfflineTable(cols, A, device=backend.device)
    assert len(table) == 301 and table.num_features == 14 and table.nbytes > 0
    bp = DiscreteDqnBatchPreprocessor(A, pre)
    out = bp.from_table(table, g.t("indices").to(backend.device))
    check(out, g)
    assert out.action.dtype == torch.float32 and out.state.float_features.dtype == torch.float32


def test_forward_on_reader_batch_matches_reference(backend):
    """the dict a data loader yields (reference dtypes: bool presence, int64 masks) -> same DiscreteDqnInput"""
    g, pre, cols = load(backend)
    idx = g.t("indices")
    batch = {k: v[idx] for k, v in cols.items()}
    out = DiscreteDqnBatchPreprocessor(g.cfg["num_actions"], pre)(batch)
    check(out, g)


def test_bf16_state_rows_are_rounded_fp32(backend):
    g, pre, cols = load(backend)
    A = g.cfg["num_actions"]
    table = OfflineTable(cols, A, device=backend.device)
    idx = g.t("indices").to(backend.device)
    o32 = DiscreteDqnBatchPreprocessor(A, pre).from_table(table, idx)
    o16 = DiscreteDqnBatchPreprocessor(A, pre, state_dtype=torch.bfloat16).from_table(table, idx)
    assert o16.state.float_features.dtype == torch.bfloat16
    assert torch.equal(o16.state.float_features.cpu(), o32.state.float_features.cpu().to(torch.bfloat16))
    assert torch.equal(o16.next_state.float_features.cpu(), o32.next_state.float_features.cpu().to(torch.bfloat16))
    assert torch.equal(o16.action, o32.action)


def test_optional_columns_and_edges(backend):
    g, pre, cols = load(backend)
    A = g.cfg["num_actions"]
    slim = {k: cols[k] for k in ("state_features", "next_state_features", "action", "next_action", "reward",
                                 "possible_next_actions_mask")}
    table = OfflineTable(slim, A, device=backend.device)
    bp = DiscreteDqnBatchPreprocessor(A, pre)
    idx = torch.tensor([0, 300, 7], device=backend.device)
    out = bp.from_table(table, idx)
    full = bp.from_table(OfflineTable({**slim, "state_features_presence": torch.ones(301, 14, dtype=torch.bool),
                                       "next_state_features_presence": torch.ones(301, 14, dtype=torch.bool)},
                                      A, device=backend.device), idx)
    assert torch.equal(out.state.float_features, full.state.float_features)  # no presence column = all present
    assert float(out.time_diff.min()) == 1.0 and float(out.step.max()) == 1.0
    assert float(out.extras.action_probability.min()) == 1.0 and int(out.extras.mdp_id.abs().max()) == 0
    assert float(out.possible_actions_mask.min()) == 1.0
    empty = bp.from_table(table, torch.empty(0, dtype=torch.int64, device=backend.device))
    assert empty.state.float_features.shape == (0, pre.num_output_features) and empty.action.shape == (0, A)
    with pytest.raises(KeyError):
        OfflineTable({k: v for k, v in slim.items() if k != "reward"}, A, device=backend.device)
    bad = dict(slim)
    bad["action"] = slim["action"].clone()
    bad["action"][5] = A  # F.one_hot(action, A) raises in the reference
    with pytest.raises(RuntimeError):
        OfflineTable(bad, A, device=backend.device)
    with pytest.raises(IndexError):
        table.validate(torch.tensor([301], device=backend.device))


def test_parquet_round_trip_and_epoch(backend, tmp_path):
    g, pre, cols = load(backend)
    A = g.cfg["num_actions"]
    table = OfflineTable(cols, A, device=backend.device)
    path = os.path.join(tmp_path, "table.parquet")
    table.to_parquet(path)
    again = OfflineTable.from_parquet(path, A, device=backend.device)
    assert set(again.columns) == set(table.columns)
    for k in table.columns:
        assert torch.equal(again.columns[k], table.columns[k]), k
    check(DiscreteDqnBatchPreprocessor(A, pre).from_table(again, g.t("indices").to(backend.device)), g)
    gen = torch.Generator(device=backend.device).manual_seed(1)
    batches = list(table.epoch(64, shuffle=True, generator=gen))
    assert len(batches) == 301 // 64 and all(b.numel() == 64 for b in batches)
    seen = torch.cat(batches)
    assert seen.unique().numel() == seen.numel()  # one epoch visits a row at most once
    tail = list(table.epoch(64, shuffle=False, drop_last=False))
    assert torch.equal(torch.cat(tail).cpu(), torch.arange(301))
    rows = table.rows(batches[0])
    assert rows["state_features"].shape == (64, 14)


def test_dqn_trains_from_the_table(backend):
    """table rows -> rg_table_dqn_batch -> DQN step, compared with the step on the same fields assembled
    with torch ops from the golden outputs of the reference's batch preprocessor"""
    from reagent_amd.core import types as rlt
    from reagent_amd.core.parameters import EvaluationParameters, RLParameters
    from reagent_amd.models import FullyConnectedDQN
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.training import DQNTrainer

    g, pre, cols = load(backend)
    A, dev = g.cfg["num_actions"], backend.device
    table = OfflineTable(cols, A, device=dev)
    bp = DiscreteDqnBatchPreprocessor(A, pre)

    def trainer():
        torch.manual_seed(3)
        q = FullyConnectedDQN(pre.num_output_features, A, [32, 16], ["relu", "relu"]).to(dev)
        return DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(A)],
                          rl=RLParameters(gamma=0.9, target_update_rate=0.1, maxq_learning=True, multi_steps=3),
                          double_q_learning=True, optimizer=Optimizer__Union.default(lr=0.01),
                          evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)

    ta, tb = trainer(), trainer()
    out = bp.from_table(table, g.t("indices").to(dev))
    la = ta.train_step_native(out)
    d = lambda k: g.t(f"out_{k}").to(dev)  # noqa: E731
    ref = rlt.DiscreteDqnInput(
        state=rlt.FeatureData(out.state.float_features.clone()), next_state=rlt.FeatureData(out.next_state.float_features.clone()),
        action=d("action"), next_action=d("next_action"), reward=d("reward"), time_diff=d("time_diff"), step=d("step"),
        not_terminal=d("not_terminal"), possible_actions_mask=d("possible_actions_mask"),
        possible_next_actions_mask=d("possible_next_actions_mask"),
        extras=rlt.ExtraData(mdp_id=d("mdp_id"), sequence_number=d("sequence_number"),
                             action_probability=d("action_probability")))
    lb = tb.train_step_native(ref)
    assert torch.isfinite(la).all() and torch.equal(la.cpu(), lb.cpu())
    for pa, pb in zip(ta.q_network.parameters(), tb.q_network.parameters()):
        assert torch.equal(pa.detach().cpu(), pb.detach().cpu())


@pytest.mark.parametrize("layout", ["one_to_one", "enum_in_the_middle"])
def test_four_wide_path_equals_normalize_dense(backend, layout):
    """tables whose feature count is a multiple of 4 take the 16-byte path of rg_table_dqn_batch; it must
    give exactly what rg_normalize_dense (pinned to the reference in test_preprocessing.py) gives for the
    same rows — both the chunks that stay 1:1 and the ones an ENUM expansion shifts"""
    from reagent_amd.core.parameters import NormalizationParameters as NP

    dev, N, A, B = backend.device, 200, 3, 96
    g = torch.Generator().manual_seed(17)
    if layout == "one_to_one":
        norm = {i: NP(feature_type="CONTINUOUS", mean=0.1 * i, stddev=1.0 + 0.05 * i) for i in range(32)}
    else:  # sorted by type: BINARY, PROBABILITY, CONTINUOUS x12, ENUM (5 values), QUANTILE -> 16 in, 20 out
        norm = {0: NP(feature_type="BINARY"), 1: NP(feature_type="PROBABILITY"),
                **{i: NP(feature_type="CONTINUOUS", mean=0.0, stddev=2.0) for i in range(2, 14)},
                14: NP(feature_type="ENUM", possible_values=[0, 1, 2, 3, 4]),
                15: NP(feature_type="QUANTILE", quantiles=[0.0, 1.0, 2.0, 4.0])}
    pre = Preprocessor(norm, device=dev)
    F = len(norm)
    assert pre.num_output_features % 4 == 0 and F % 4 == 0

    def feats():
        x = torch.randn(N, F, generator=g)
        x[:, :2] = torch.rand(N, 2, generator=g)
        if layout != "one_to_one":
            x[:, pre.feature_id_to_index[14]] = torch.randint(0, 6, (N,), generator=g).float()
        return x

    cols = dict(state_features=feats(), next_state_features=feats(),
                state_features_presence=torch.rand(N, F, generator=g) > 0.2,
                next_state_features_presence=torch.rand(N, F, generator=g) > 0.2,
                action=torch.randint(A, (N,), generator=g), next_action=torch.randint(A + 1, (N,), generator=g),
                reward=torch.randn(N, generator=g),
                possible_next_actions_mask=(torch.rand(N, A, generator=g) > 0.5).long())
    table = OfflineTable(cols, A, device=dev)
    idx = torch.randint(N, (B,), generator=g).to(dev)
    out = DiscreteDqnBatchPreprocessor(A, pre).from_table(table, idx)
    for name, got in (("state", out.state.float_features), ("next_state", out.next_state.float_features)):
        rows = table.columns[f"{name}_features"][idx]
        ref = pre(rows, table.columns[f"{name}_features_presence"][idx])
        assert torch.equal(got, ref), name
    o16 = DiscreteDqnBatchPreprocessor(A, pre, state_dtype=torch.bfloat16).from_table(table, idx)
    assert torch.equal(o16.state.float_features, out.state.float_features.to(torch.bfloat16))


def test_policy_network_batch_preprocessor_matches_reference(backend):
    """PolicyNetworkBatchPreprocessor (continuous actions) against the reference class's output
    (tests/golden/policy_batch.npz): features 1e-5, every other field exact in value and shape"""
    from reagent_amd.preprocessing import PolicyNetworkBatchPreprocessor

    g = Golden("policy_batch")
    dev = backend.device
    pre = Preprocessor({int(k): SimpleNamespace(**v) for k, v in g.cfg["norm"].items()}, device=dev)
    apre = Preprocessor({int(k): SimpleNamespace(**v) for k, v in g.cfg["action_norm"].items()}, device=dev)
    batch = {k[len("in_"):]: g.t(k) for k in g.z.files if k.startswith("in_")}
    out = PolicyNetworkBatchPreprocessor(pre, apre)(batch)
    for k in ("state", "next_state", "action", "next_action"):
        got, ref = getattr(out, k).float_features.cpu(), g.t(f"out_{k}")
        assert got.shape == ref.shape and (got - ref).abs().max() <= 1e-5 + 1e-5 * ref.abs().max(), k
    for k in ("reward", "time_diff", "step", "not_terminal"):
        got, ref = getattr(out, k).cpu(), g.t(f"out_{k}")
        assert got.shape == ref.shape and torch.equal(got.double(), ref.double()), k
    for k in ("mdp_id", "sequence_number", "action_probability"):
        got, ref = getattr(out.extras, k).cpu(), g.t(f"out_{k}")
        assert got.shape == ref.shape and torch.equal(got.double(), ref.double()), k


def test_offline_table_loop_trains_over_epochs(emu_lib):
    """OfflineTableLoop: epochs of table -> rg_table_dqn_batch -> native DQN step; unshuffled epochs equal
    feeding the same index batches by hand, and the TD loss on a fixed batch goes down over epochs"""
    from types import SimpleNamespace as NS

    from reagent_amd.core.parameters import EvaluationParameters, NormalizationParameters as NP, RLParameters
    from reagent_amd.models import FullyConnectedDQN
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.runtime import OfflineTableLoop
    from reagent_amd.training import DQNTrainer

    N, F, A, B = 256, 8, 3, 64
    g = torch.Generator().manual_seed(3)
    state = torch.randn(N, F, generator=g)
    action = torch.randint(A, (N,), generator=g)
    cols = dict(state_features=state, next_state_features=state.roll(-1, 0), action=action,
                next_action=action.roll(-1, 0), reward=(action == 1).float() + 0.1 * state[:, 0],
                possible_next_actions_mask=torch.ones(N, A, dtype=torch.long))
    table = OfflineTable(cols, A, device="cpu")
    pre = Preprocessor({i: NP(feature_type="CONTINUOUS", mean=0.0, stddev=1.0) for i in range(F)}, device="cpu")
    bp = DiscreteDqnBatchPreprocessor(A, pre)

    def trainer():
        torch.manual_seed(1)
        q = FullyConnectedDQN(F, A, [32, 32], ["relu", "relu"])
        return DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(A)],
                          rl=RLParameters(gamma=0.5, target_update_rate=0.2), optimizer=Optimizer__Union.default(lr=0.01),
                          evaluation=EvaluationParameters(calc_cpe_in_training=False))

    ta, tb = trainer(), trainer()
    loop = OfflineTableLoop(table, ta, bp, B, shuffle=False)
    loop.run_epoch()
    for idx in table.epoch(B, shuffle=False):
        tb.train_step_native(bp.from_table(table, idx))
    assert loop.batches_done == N // B
    for pa, pb in zip(ta.q_network.parameters(), tb.q_network.parameters()):
        assert torch.equal(pa, pb)
    probe = bp.from_table(table, torch.arange(B))
    first = float(tb.train_step_native(probe))
    shuffled = OfflineTableLoop(table, tb, bp, B, shuffle=True, generator=torch.Generator().manual_seed(5))
    for _ in range(6):
        shuffled.run_epoch()
    assert float(tb.train_step_native(probe)) < first


def test_policy_network_input_maker_matches_reference(backend):
    """PolicyNetworkInputMaker (one launch, rg_make_policy_input) against the reference class on the same sampled batch
    (tests/golden/policy_input_maker.npz): rescaled actions, zeroed terminal next-actions and not_terminal bit-exact
    (same operations, each rounded on its own), exp(log_prob) within one ulp of torch's"""
    import collections

    from reagent_amd.preprocessing import PolicyNetworkInputMaker

    g = Golden("policy_input_maker")
    Batch = collections.namedtuple("Batch", "state next_state action next_action reward terminal log_prob")
    dev = backend.device
    batch = Batch(**{k: g.t(f"in_{k}").to(dev) for k in Batch._fields})
    out = PolicyNetworkInputMaker(g.a("action_low"), g.a("action_high"))(batch)
    assert torch.equal(out.action.float_features.cpu(), g.t("out_action"))
    assert torch.equal(out.next_action.float_features.cpu(), g.t("out_next_action"))
    assert torch.equal(out.not_terminal.cpu(), g.t("out_not_terminal")) and out.not_terminal.shape == (g.cfg["batch"], 1)
    assert torch.equal(out.state.float_features.cpu(), g.t("out_state")) and torch.equal(out.reward.cpu(), g.t("out_reward"))
    p, ref = out.extras.action_probability.cpu(), g.t("out_action_probability")
    assert p.shape == ref.shape and ((p - ref).abs() <= 1.2e-7 * ref.abs()).all()
    term = g.t("in_terminal").reshape(-1)
    assert (out.next_action.float_features.cpu()[term] == 0).all() and term.any()
