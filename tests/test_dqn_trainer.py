"""DQNTrainer (reagent_amd.training) against golden vectors of the reference DQNTrainer
(tests/golden/dqn_*.npz: losses, Q-values, post-step weights, target weights, Adam moments produced
by the unmodified reference driven through the Lightning-1.6 loop emulation).

Tolerances (BASELINE.json north_star): fp32-accurate mode — Q-values within 1e-4 abs, loss within
1e-4 rel, post-step weights within 2e-5 abs.  bf16 mode is reported, with its own looser bound.
"""
import pytest
import torch

import reagent_amd._lib as L
from golden_util import Golden
from reagent_amd import synthetic
from reagent_amd.core.parameters import EvaluationParameters, RLParameters
from reagent_amd.models import FullyConnectedDQN, set_default_precision
from reagent_amd.optimizer import Optimizer__Union
from reagent_amd.training import DQNTrainer

CASES = ["dqn_c1", "dqn_huber_masks", "dqn_sarsa_multistep", "dqn_timediff"]


def build(g: Golden, device, precision):
    c = g.cfg
    set_default_precision(precision)
    try:
        q = FullyConnectedDQN(c["state_dim"], c["num_actions"], c["sizes"], c["activations"])
    finally:
        set_default_precision(L.PREC_F32)
    with torch.no_grad():
        for p, init in zip(q.parameters(), g.seq("init_param_")):
            p.copy_(init)
    q = q.to(device)
    qt = q.get_target_network()
    rl = RLParameters(**c["rl"])
    tr = DQNTrainer(q, qt, None, actions=[str(i) for i in range(c["num_actions"])], rl=rl,
                    double_q_learning=c["double_q"], optimizer=Optimizer__Union.default(lr=c["lr"]),
                    evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(device)
    return tr


def lightning_like_step(tr, opts, batch):
    """what pl.Trainer.fit does per batch (reagent_lightning_module.py:108-133)"""
    losses = []
    for i, opt in enumerate(opts):
        loss = tr.training_step(batch, 0, i)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    return losses


@pytest.mark.parametrize("name", CASES)
def test_dqn_matches_reference_fp32_mode(backend, name):
    g = Golden(name)
    tr = build(g, backend.device, L.PREC_F32)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    assert len(opts) == 2 and type(opts[1]).__name__ == "SoftUpdate"
    for s in range(g.cfg["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        losses = lightning_like_step(tr, opts, batch)
        assert len(losses) == 2 and losses[0].shape == ()
        ref_loss = g.t(f"step{s}_loss")
        assert abs(losses[0].item() - ref_loss.item()) <= 1e-4 * abs(ref_loss.item()) + 1e-6
        assert (tr.all_action_scores.cpu() - g.t(f"step{s}_q")).abs().max() <= 1e-4  # north_star bound
        for i, p in enumerate(tr.q_network.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_param_{i}")).abs().max() <= 2e-5, (s, i)
        for i, p in enumerate(tr.q_network_target.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_target_{i}")).abs().max() <= 2e-5, (s, i)
    adam = opts[0]
    for i, p in enumerate(tr.q_network.parameters()):
        st = adam.state[p]
        ref_m, ref_v = g.t(f"final_exp_avg_{i}"), g.t(f"final_exp_avg_sq_{i}")
        assert (st["exp_avg"].cpu() - ref_m).abs().max() <= 1e-6 + 1e-4 * ref_m.abs().max()
        assert (st["exp_avg_sq"].cpu() - ref_v).abs().max() <= 1e-9 + 1e-4 * ref_v.abs().max()
        assert int(st["step"]) == g.cfg["steps"]


@pytest.mark.parametrize("name", ["dqn_c1", "dqn_huber_masks"])
def test_dqn_native_step_equals_generator_path(backend, name):
    """train_step_native (no autograd / generator) produces the same numbers as the
    Lightning-protocol path."""
    g = Golden(name)
    tr_a = build(g, backend.device, L.PREC_F32)
    tr_b = build(g, backend.device, L.PREC_F32)
    opts = [o["optimizer"] for o in tr_a.configure_optimizers()]
    for s in range(g.cfg["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        la = lightning_like_step(tr_a, opts, batch)[0]
        lb = tr_b.train_step_native(batch)
        assert torch.equal(la.cpu().reshape(()), lb.cpu().reshape(()))
        for pa, pb in zip(tr_a.q_network.parameters(), tr_b.q_network.parameters()):
            assert torch.equal(pa.detach().cpu(), pb.detach().cpu())
        for pa, pb in zip(tr_a.q_network_target.parameters(), tr_b.q_network_target.parameters()):
            assert torch.equal(pa.detach().cpu(), pb.detach().cpu())


@pytest.mark.parametrize("name", ["dqn_c1", "dqn_huber_masks"])
def test_dqn_bf16_mode_tracks_reference(backend, name):
    """bf16-MFMA throughput mode: first-step loss / Q-values within bf16-level error of the fp32
    reference (max |dQ| measured ~2e-2 on unit-scale nets)."""
    g = Golden(name)
    tr = build(g, backend.device, L.PREC_BF16)
    batch = synthetic.to_dqn_input(g.batch(0), backend.device)
    loss = tr.train_step_native(batch)
    ref_loss = g.t("step0_loss").item()
    assert abs(loss.item() - ref_loss) <= 3e-2 * abs(ref_loss) + 1e-3
    q_ref = g.t("step0_q")
    assert (tr.all_action_scores.cpu() - q_ref).abs().max() <= 5e-2 * max(1.0, q_ref.abs().max().item())


def test_generator_protocol_and_errors(backend):
    g = Golden("dqn_c1")
    tr = build(g, backend.device, L.PREC_F32)
    batch = synthetic.to_dqn_input(g.batch(0), backend.device)
    losses = list(tr.train_step_gen(batch, 0))
    assert len(losses) == 2  # td loss + soft-update trigger (test_dqn.py:123-146, CPE off)
    assert losses[0].requires_grad and losses[1].requires_grad
    assert losses[1].device.type == "cpu" and losses[1].item() == 2.0  # soft_update_result()
    # dqn_trainer_base.py:206-208
    bad = synthetic.to_dqn_input(g.batch(0), backend.device)
    bad.possible_next_actions_mask = torch.zeros_like(bad.possible_next_actions_mask)
    bad.not_terminal = torch.ones_like(bad.not_terminal)
    with pytest.raises(ValueError, match="No possible next actions"):
        next(tr.train_step_gen(bad, 0))
    # optimizer-count mismatch is detected (reagent_lightning_module.py:118-129)
    tr2 = build(g, backend.device, L.PREC_F32)
    tr2._num_optimizing_steps_cache = 1
    with pytest.raises(RuntimeError, match="yields too many times"):
        tr2.training_step(batch, 0, 0)


def test_state_dict_layout_matches_reference_names(backend):
    g = Golden("dqn_c1")
    tr = build(g, backend.device, L.PREC_F32)
    keys = list(tr.state_dict().keys())
    assert keys[:3] == ["_next_stopping_epoch", "_cleanly_stopped", "reward_boosts"]
    assert "q_network.fc.dnn.0.0.weight" in keys and "q_network_target.fc.dnn.2.0.bias" in keys
    assert len(keys) == 3 + 2 * 6
    assert tr.q_network.fc.dnn[0][0].weight.shape == (128, 4)  # nn.Linear layout


# ---- CPE heads (calc_cpe_in_training=True) --------------------------------------
CPE_NETS = ("reward_network", "q_network_cpe", "q_network_cpe_target")


def build_cpe(g: Golden, device):
    c = g.cfg
    A = c["num_actions"]
    n_out = (len(c["cpe_metrics"]) + 1) * A

    def net(out, prefix):
        m = FullyConnectedDQN(c["state_dim"], out, c["sizes"], c["activations"])
        with torch.no_grad():
            for p, init in zip(m.parameters(), g.seq(prefix)):
                p.copy_(init)
        return m.to(device)

    q = net(A, "init_param_")
    reward_net, q_cpe = net(n_out, "init_reward_network_"), net(n_out, "init_q_network_cpe_")
    q_cpe_t = q_cpe.get_target_network()
    tr = DQNTrainer(q, q.get_target_network(), reward_net, q_network_cpe=q_cpe, q_network_cpe_target=q_cpe_t,
                    metrics_to_score=list(c["cpe_metrics"]), actions=[str(i) for i in range(A)],
                    rl=RLParameters(**c["rl"]), double_q_learning=c["double_q"],
                    optimizer=Optimizer__Union.default(lr=c["lr"]),
                    evaluation=EvaluationParameters(calc_cpe_in_training=True)).to(device)
    return tr


def _check_cpe_step(tr, g, s, losses):
    ref = [g.t(f"step{s}_loss"), g.t(f"step{s}_reward_loss"), g.t(f"step{s}_cpe_loss")]
    for got, want in zip(losses[:3], ref):
        assert abs(float(got) - want.item()) <= 1e-4 * abs(want.item()) + 1e-6
    for i, p in enumerate(tr.q_network.parameters()):
        assert (p.detach().cpu() - g.t(f"step{s}_param_{i}")).abs().max() <= 2e-5, (s, i)
    for i, p in enumerate(tr.q_network_target.parameters()):
        assert (p.detach().cpu() - g.t(f"step{s}_target_{i}")).abs().max() <= 2e-5, (s, i)
    for net in CPE_NETS:
        for i, p in enumerate(getattr(tr, net).parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_{net}_{i}")).abs().max() <= 2e-5, (s, net, i)


@pytest.mark.parametrize("name", ["dqn_cpe", "dqn_cpe_sarsa_mse"])
def test_dqn_cpe_matches_reference(backend, name):
    """four losses per step in the reference's optimizer order (q, reward, cpe, soft update); the CPE
    targets see q_network(next_state) AFTER the q-network step of the same batch"""
    g = Golden(name)
    tr = build_cpe(g, backend.device)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    assert [type(o).__name__ for o in opts] == ["FusedAdam", "FusedAdam", "FusedAdam", "SoftUpdate"]
    assert tr.reward_idx_offsets.tolist() == list(range(0, (len(g.cfg["cpe_metrics"]) + 1) * g.cfg["num_actions"],
                                                        g.cfg["num_actions"]))
    for s in range(g.cfg["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        losses = lightning_like_step(tr, opts, batch)
        assert len(losses) == 4
        _check_cpe_step(tr, g, s, [l.item() for l in losses])


@pytest.mark.parametrize("name", ["dqn_cpe"])
def test_dqn_cpe_native_step(backend, name):
    g = Golden(name)
    tr = build_cpe(g, backend.device)
    for s in range(g.cfg["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        loss = tr.train_step_native(batch)
        _check_cpe_step(tr, g, s, [loss.item(), tr._cpe.losses["reward"].item(), tr._cpe.losses["cpe"].item()])


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])  # bf16x3: both planes (hi, lo) of every fragment set
@pytest.mark.parametrize("state_dim", [24, 22])  # 22: the first layer's rows are not 32-byte pieces -> per-element path
def test_fused_update_equals_separate_launches(backend, state_dim, precision):
    """rg_mlp_update_fused (Adam + soft update + bf16 re-staging of both networks in one launch, taken by
    the native step when both stacks are on the fused kernels) leaves the same bits as the four separate
    launches: parameters, target parameters, Adam moments, and the next step's Q-values (= the staged
    fragments)."""
    from reagent_amd.engine import FusedMLP

    def make():
        set_default_precision(L.PREC_BF16 if precision == "bf16" else L.PREC_BF16X3)
        try:
            torch.manual_seed(3)
            q = FullyConnectedDQN(state_dim, 5, [256, 256], ["relu", "relu"]).to(backend.device)
        finally:
            set_default_precision(L.PREC_F32)
        return DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(5)],
                          rl=RLParameters(gamma=0.9, target_update_rate=0.05, q_network_loss="huber"),
                          optimizer=Optimizer__Union.default(lr=0.003),
                          evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(backend.device)

    fused, separate = make(), make()
    separate._fused_plan = False
    for s in range(3):
        batch = synthetic.to_dqn_input(synthetic.dqn_batch(200, state_dim, 5, seed=40 + s, p_impossible=0.2), backend.device)
        la, lb = fused.train_step_native(batch), separate.train_step_native(batch)
        assert torch.equal(la, lb), s
        assert torch.equal(fused.all_action_scores, separate.all_action_scores), s
    assert isinstance(fused._qs, FusedMLP) and isinstance(fused._fused_plan, dict)  # the fused path really ran
    assert fused._qs.x3 == (precision == "bf16x3")
    for a, b in zip(fused.q_network.parameters(), separate.q_network.parameters()):
        assert torch.equal(a, b)
    for a, b in zip(fused.q_network_target.parameters(), separate.q_network_target.parameters()):
        assert torch.equal(a, b)
    oa, ob = fused.native_optimizers()[0], separate.native_optimizers()[0]
    for pa, pb in zip(fused.q_network.parameters(), separate.q_network.parameters()):
        assert torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"])
        assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"])
        assert int(oa.state[pa]["step"]) == int(ob.state[pb]["step"]) == 3
    x = batch.state
    assert torch.equal(fused.q_network(x), separate.q_network(x))
    assert torch.equal(fused.q_network_target(x), separate.q_network_target(x))
    separate._qs.stage_weights(need_transposed=True)  # (the separate path stages lazily, before the next use;
    separate._ts.stage_weights(need_transposed=False)  # the grad-mode forwards above ran on the autograd engine's own copy)
    for a, b in zip(fused._qs._wf + fused._qs._wb + fused._ts._wf, separate._qs._wf + separate._qs._wb + separate._ts._wf):
        assert torch.equal(a, b)  # the staged fragments themselves, every plane


def test_folded_tails_equal_their_own_launches(backend, monkeypatch):
    """the bias gradients' column reduce and the loss mean evaluated inside rg_mlp_wgrad_fused's reduce launch
    (rg_mlp_desc.defer_db / db_partials / sum_in) leave the bits their own launches (reduce_cols_group, rg_reduce_sum) do"""
    from reagent_amd import ops
    from reagent_amd.engine import FusedMLP

    def make():
        set_default_precision(L.PREC_BF16)
        try:
            torch.manual_seed(7)
            q = FullyConnectedDQN(24, 5, [256, 256], ["relu", "relu"]).to(backend.device)
        finally:
            set_default_precision(L.PREC_F32)
        return DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(5)],
                          rl=RLParameters(gamma=0.9, target_update_rate=0.05, q_network_loss="huber"),
                          optimizer=Optimizer__Union.default(lr=0.003),
                          evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(backend.device)

    folded, separate = make(), make()
    names = []
    real = ops._run
    monkeypatch.setattr(ops, "_run", lambda name, meta, call: (names.append(name), real(name, meta, call))[1])
    for s in range(2):
        batch = synthetic.to_dqn_input(synthetic.dqn_batch(300, 24, 5, seed=60 + s, p_impossible=0.2), backend.device)
        del names[:]
        la = folded.train_step_native(batch)
        assert "rg_reduce_sum" not in names and isinstance(folded._qs, FusedMLP)
        # the same step with both tails in their own launches
        separate.apply_pending_update()
        if s:
            separate._qs.fold_tails = False
        lb = separate._hip_forward(batch)  # (no native-step flag: rg_reduce_sum runs)
        if not s:
            separate._qs.fold_tails = False
        for p in separate._hip_params:
            p.grad = None
        separate._hip_backward(None)
        separate._update_pending = True
        separate.apply_pending_update()
        assert torch.equal(la, lb), s
        for a, b in zip(folded._slab.grad_views(), separate._slab.grad_views()):
            assert torch.equal(a, b), s
    for a, b in zip(folded.q_network.parameters(), separate.q_network.parameters()):
        assert torch.equal(a, b)


def test_dqn_bcq_matches_reference(backend):
    """batch-constrained q-learning (dqn_trainer.py:113-117, 209-215; imitator_training.py:12-25): next
    actions whose imitator probability is below drop_threshold x the row maximum leave the max.
    Golden run of the reference with a torch imitator (tests/golden/dqn_bcq.npz); here the imitator is a
    FullyConnectedNetwork with the same weights.  The batch's own mask must stay untouched."""
    from reagent_amd.models.fully_connected_network import FullyConnectedNetwork
    from reagent_amd.training import BCQConfig

    g = Golden("dqn_bcq")
    c = g.cfg
    q = FullyConnectedDQN(c["state_dim"], c["num_actions"], c["sizes"], c["activations"])
    imitator = FullyConnectedNetwork([c["state_dim"], 16, c["num_actions"]], ["relu", "linear"])
    with torch.no_grad():
        for p, init in zip(q.parameters(), g.seq("init_param_")):
            p.copy_(init)
        for p, init in zip(imitator.parameters(), g.seq("imitator_")):
            p.copy_(init)
    q, imitator = q.to(backend.device), imitator.to(backend.device)
    tr = DQNTrainer(q, q.get_target_network(), None, imitator=imitator, bcq=BCQConfig(drop_threshold=c["bcq_threshold"]),
                    actions=[str(i) for i in range(c["num_actions"])], rl=RLParameters(**c["rl"]),
                    double_q_learning=c["double_q"], optimizer=Optimizer__Union.default(lr=c["lr"]),
                    evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(backend.device)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    for s in range(c["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        before = batch.possible_next_actions_mask.clone()
        losses = lightning_like_step(tr, opts, batch)
        assert torch.equal(batch.possible_next_actions_mask, before)
        ref_loss = g.t(f"step{s}_loss").item()
        assert abs(losses[0].item() - ref_loss) <= 1e-4 * abs(ref_loss) + 1e-6
        for i, p in enumerate(tr.q_network.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_param_{i}")).abs().max() <= 2e-5, (s, i)
    # the constraint really bites in this fixture: an unconstrained trainer takes a different step
    free = build(g, backend.device, L.PREC_F32)
    free_loss = lightning_like_step(free, [o["optimizer"] for o in free.configure_optimizers()],
                                    synthetic.to_dqn_input(g.batch(0), backend.device))[0].item()
    assert abs(free_loss - g.t("step0_loss").item()) > 1e-3 * abs(free_loss)
    with pytest.raises(NotImplementedError):
        DQNTrainer(q, q.get_target_network(), None, bcq=BCQConfig(), actions=["0", "1", "2", "3", "4"],
                   evaluation=EvaluationParameters(calc_cpe_in_training=False))  # no imitator given


def test_backward_without_zero_grad_accumulates(backend):
    """PyTorch semantics for a second backward before zero_grad(): the gradients add up (the HIP backward
    overwrites its slab, so the published gradient is held and added back)"""
    g = Golden("dqn_huber_masks")
    tr = build(g, backend.device, L.PREC_F32)
    b0 = synthetic.to_dqn_input(g.batch(0), backend.device)
    b1 = synthetic.to_dqn_input(g.batch(1), backend.device)
    params = list(tr.q_network.parameters())

    def grads_of(batch, clear=True):
        if clear:
            for p in params:
                p.grad = None
        tr.compute_td_loss(batch).backward()
        return [p.grad.detach().cpu().clone() for p in params]

    g0, g1 = grads_of(b0), grads_of(b1)
    grads_of(b0)
    both = grads_of(b1, clear=False)
    for a, b, c in zip(g0, g1, both):
        assert (c - (a + b)).abs().max() <= 1e-6 * max(1.0, (a + b).abs().max().item())
    tr.q_network.zero_grad(set_to_none=False)  # zeroed in place, still aliases the slab
    again = grads_of(b0, clear=False)
    for a, c in zip(g0, again):
        assert torch.equal(a, c)


def test_logged_fields_of_the_dqn_step(backend):
    """dqn_trainer.py:306-347: what the step hands to reporter.log, pinned against the golden batch
    (logged actions / propensities / boosted rewards exactly; td_loss, model values and the masked
    arg-max against the reference run's numbers)"""
    g = Golden("dqn_huber_masks")
    tr = build(g, backend.device, L.PREC_F32)

    class Reporter:
        def __init__(self):
            self.calls = []

        def log(self, **kw):
            self.calls.append(kw)

    rep = Reporter()
    tr.set_reporter(rep)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    raw = g.batch(0)
    batch = synthetic.to_dqn_input(raw, backend.device)
    lightning_like_step(tr, opts, batch)
    assert len(rep.calls) == 1
    kw = rep.calls[0]
    assert set(kw) == {"td_loss", "logged_actions", "logged_propensities", "logged_rewards", "logged_values",
                       "model_values", "model_values_on_logged_actions", "model_action_idxs"}
    assert kw["logged_values"] is None and kw["model_values_on_logged_actions"] is None
    assert torch.equal(kw["logged_actions"].cpu(), raw["action"].argmax(dim=1, keepdim=True))
    assert torch.equal(kw["logged_propensities"].cpu(), torch.ones_like(raw["reward"]))
    boosts = torch.zeros(g.cfg["num_actions"])
    for k, v in g.cfg["rl"]["reward_boost"].items():
        boosts[int(k)] = v
    want_r = raw["reward"] + (raw["action"] * boosts).sum(dim=1, keepdim=True)  # dqn_trainer_base.py:216-241
    assert (kw["logged_rewards"].cpu() - want_r).abs().max() <= 1e-6
    ref_loss = g.t("step0_loss").item()
    assert abs(kw["td_loss"].item() - ref_loss) <= 1e-4 * abs(ref_loss) + 1e-6 and not kw["td_loss"].requires_grad
    q_ref = g.t("step0_q")
    assert (kw["model_values"].cpu() - q_ref).abs().max() <= 1e-4
    # get_max_q_values: arg-max over q - 1e9 * (1 - possible_actions_mask) (dqn_trainer_base.py:33-77)
    want_idx = (q_ref - 1e9 * (1 - raw["possible_actions_mask"])).argmax(dim=1, keepdim=True)
    assert torch.equal(kw["model_action_idxs"].cpu(), want_idx)


def test_per_step_logging_fields(backend):
    """SURVEY §8 a19 — what DQNTrainer hands its reporter after a step (reagent/training/dqn_trainer.py:306-347):
    td_loss, the logged action indices, propensities, the (boosted) rewards, the model's Q-values of that step
    and its greedy actions under the possible-actions mask.  Pinned on the golden batch of `dqn_huber_masks`."""
    g = Golden("dqn_huber_masks")
    tr = build(g, backend.device, L.PREC_F32)
    seen = []

    class Reporter:
        def log(self, **kw):
            seen.append(kw)

    tr.set_reporter(Reporter())
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    b = g.batch(0)
    batch = synthetic.to_dqn_input(b, backend.device)
    lightning_like_step(tr, opts, batch)
    assert len(seen) == 1
    rec = seen[0]
    assert set(rec) == {"td_loss", "logged_actions", "logged_propensities", "logged_rewards", "logged_values",
                        "model_values", "model_values_on_logged_actions", "model_action_idxs"}
    ref_loss = g.t("step0_loss")
    assert abs(rec["td_loss"].item() - ref_loss.item()) <= 1e-4 * abs(ref_loss.item()) + 1e-6
    assert torch.equal(rec["logged_actions"].cpu(), b["action"].argmax(dim=1, keepdim=True))
    assert torch.equal(rec["logged_propensities"].cpu(), batch.extras.action_probability.cpu())
    assert torch.equal(rec["logged_rewards"].cpu(), tr.boost_rewards(batch.reward, batch.action).cpu())
    assert rec["logged_values"] is None and rec["model_values_on_logged_actions"] is None
    q = g.t("step0_q")
    assert (rec["model_values"].cpu() - q).abs().max() <= 1e-4
    mask = b["possible_actions_mask"] if tr.maxq_learning else b["action"]
    greedy = (q + (1.0 - mask.float()) * -1e10).argmax(dim=1, keepdim=True)  # dqn_trainer_base.py:147-163
    assert torch.equal(rec["model_action_idxs"].cpu().reshape(-1, 1), greedy)


@pytest.mark.parametrize("name", ["dqn_huber_masks", "dqn_sarsa_multistep", "dqn_c1"])
def test_logger_metrics_match_the_reference(emu_lib, name):
    """dqn_trainer.py:320-347: the second logging channel — `self.logger.log_metrics` with td_loss, the mean (boosted)
    reward, the mean propensity and per-action dicts {action name: mean} of the logged actions, the model's Q-values and
    its greedy actions — key by key against what the reference's logger received on the golden batches"""
    g = Golden(name)
    tr = build(g, "cpu", L.PREC_F32)
    got = {}

    class Logger:
        def log_metrics(self, metrics, step=None):
            assert step == tr.all_batches_processed
            for k, v in metrics.items():
                if isinstance(v, dict):
                    got.update({f"{k}/{a}": x for a, x in v.items()})
                elif v is not None:
                    got[k] = v

    tr.logger = Logger()
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    for s in range(g.cfg["steps"]):
        got.clear()
        lightning_like_step(tr, opts, synthetic.to_dqn_input(g.batch(s), "cpu"))
        want = {k[len(f"step{s}_metric_"):]: g.t(k) for k in g.z.files if k.startswith(f"step{s}_metric_")}
        assert set(got) == set(want) and len(want) >= 3 + 3 * g.cfg["num_actions"], (sorted(got), sorted(want))
        for k, ref in want.items():
            v = got[k].detach().double().reshape(-1).cpu()
            assert v.shape == ref.shape and (v - ref).abs().max() <= 1e-4 * max(1.0, ref.abs().max().item()), (s, k)


@pytest.mark.parametrize("name", ["dqn_cpe", "dqn_cpe_sarsa_mse", "dqn_huber_masks", "dqn_timediff"])
def test_reporter_tensors_match_the_reference(emu_lib, name):
    """every tensor the step hands its reporter (dqn_trainer.py:306-319; with CPE heads also reward_loss,
    model_propensities, model_rewards of dqn_trainer_base.py:430-450) against what the reference's reporter received"""
    from golden_util import check_reported

    g = Golden(name)
    cpe = g.cfg.get("cpe_metrics") is not None
    tr = build_cpe(g, "cpu") if cpe else build(g, "cpu", L.PREC_F32)
    seen = {}

    class Reporter:
        def log(self, **kw):
            seen.update(kw)

    tr.set_reporter(Reporter())
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    for s in range(g.cfg["steps"]):
        seen.clear()
        lightning_like_step(tr, opts, synthetic.to_dqn_input(g.batch(s), "cpu"))
        assert check_reported(g, s, seen) >= (9 if cpe else 6)
