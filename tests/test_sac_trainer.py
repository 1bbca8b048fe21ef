"""SACTrainer (reagent_amd.training) against golden vectors of the reference SACTrainer
(tests/golden/sac_twin.npz: 3 steps of twin-critic SAC with the temperature optimizer, the reference's
randn draws recorded and injected).  The reference has NO numeric SAC test: these
vectors, produced by the unmodified reference under the Lightning-loop emulation, are the pin.
Tolerances: losses 1e-4 rel, parameters 2e-5 abs (fp32 mode)."""
import pytest
import torch

import reagent_amd._lib as L
from golden_util import Golden
from reagent_amd import synthetic
from reagent_amd.core.parameters import RLParameters
from reagent_amd.models import FullyConnectedCritic, GaussianFullyConnectedActor, set_default_precision
from reagent_amd.optimizer import Optimizer__Union
from reagent_amd.training import SACTrainer
from test_dqn_trainer import lightning_like_step


def build(g, device, precision=L.PREC_F32):
    c = g.cfg
    set_default_precision(precision)
    try:
        actor = GaussianFullyConnectedActor(c["state_dim"], c["action_dim"], c["sizes"], c["activations"])
        q1 = FullyConnectedCritic(c["state_dim"], c["action_dim"], c["sizes"], c["activations"])
        q2 = FullyConnectedCritic(c["state_dim"], c["action_dim"], c["sizes"], c["activations"])
    finally:
        set_default_precision(L.PREC_F32)
    with torch.no_grad():
        for net, name in ((actor, "actor"), (q1, "q1"), (q2, "q2")):
            for p, init in zip(net.parameters(), g.seq(f"init_{name}_")):
                p.copy_(init)
    adam = lambda: Optimizer__Union.default(lr=c["lr"])  # noqa: E731
    tr = SACTrainer(actor.to(device), q1.to(device), q2.to(device), rl=RLParameters(**c["rl"]),
                    q_network_optimizer=adam(), actor_network_optimizer=adam(), alpha_optimizer=adam(),
                    **c.get("trainer_kw", {}))
    return tr.to(device)


def check(tr, g, s, tol=2e-5):
    for n, net in dict(actor=tr.actor_network, q1=tr.q1_network, q2=tr.q2_network, q1_target=tr.q1_network_target,
                       q2_target=tr.q2_network_target).items():
        for i, p in enumerate(net.parameters()):
            err = (p.detach().cpu() - g.t(f"step{s}_{n}_{i}")).abs().max().item()
            assert err <= tol, (s, n, i, err)
    assert abs(tr.log_alpha.item() - g.t(f"step{s}_log_alpha").item()) <= 1e-6


def test_sac_matches_reference_generator_path(backend):
    g = Golden("sac_twin")
    tr = build(g, backend.device)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    assert [type(o).__name__ for o in opts] == ["FusedAdam", "FusedAdam", "FusedAdam", "AdamF64", "SoftUpdate"]
    assert tr.log_alpha.dtype == torch.float64
    for s in range(g.cfg["steps"]):
        batch = synthetic.to_policy_input(g.batch(s), backend.device)
        tr.set_noise(g.t(f"step{s}_noise_next"), g.t(f"step{s}_noise_cur"))
        losses = lightning_like_step(tr, opts, batch)
        assert len(losses) == 5
        for j, nm in enumerate(["q1_loss", "q2_loss", "actor_loss", "alpha_loss"]):
            ref = float(g.t(f"step{s}_{nm}"))
            assert abs(float(losses[j]) - ref) <= 1e-4 * abs(ref) + 2e-6, (s, nm, float(losses[j]), ref)
        check(tr, g, s)
    keys = list(tr.state_dict().keys())
    assert "log_alpha" in keys and any(k.startswith("q1_network_target.fc.dnn.0.0") for k in keys)


def test_sac_native_step_matches_reference(backend):
    g = Golden("sac_twin")
    tr = build(g, backend.device)
    for s in range(g.cfg["steps"]):
        batch = synthetic.to_policy_input(g.batch(s), backend.device)
        out = tr.train_step_native(batch, g.t(f"step{s}_noise_next"), g.t(f"step{s}_noise_cur"))
        ref = float(g.t(f"step{s}_q1_loss"))
        assert abs(out["q1_loss"].item() - ref) <= 1e-4 * abs(ref) + 2e-6
        check(tr, g, s)


def test_actor_model_surface(backend):
    g = Golden("sac_twin")
    tr = build(g, backend.device)
    batch = synthetic.to_policy_input(g.batch(0), backend.device)
    out = tr.actor_network(batch.state)
    B, A = g.cfg["batch"], g.cfg["action_dim"]
    assert out.action.shape == (B, A) and out.log_prob.shape == (B, 1) and out.squashed_mean.shape == (B, A)
    assert out.action.abs().max() <= 1 - 1e-6 + 1e-7
    lp = tr.actor_network.get_log_prob(batch.state, out.action)
    assert (lp - out.log_prob).abs().max() <= 1e-4 * max(1.0, out.log_prob.abs().max().item())
    loc, scale_log = tr.actor_network._get_loc_and_scale_log(batch.state)
    assert loc.shape == (B, A) and scale_log.min() >= -2 and scale_log.max() <= 2
    q = tr.q1_network(batch.state, batch.action)
    assert q.shape == (B, 1)


# ---- variants: value network, CRR actor weights, detached log_prob (sac_trainer.py:108-112, 214-215, 262-273, 325-340) ----
def build_variant(g, device):
    from reagent_amd.models.fully_connected_network import FloatFeatureFullyConnected
    from reagent_amd.training.sac_trainer import CRRWeightFn

    c = g.cfg
    S, A = c["state_dim"], c["action_dim"]
    actor = GaussianFullyConnectedActor(S, A, c["sizes"], c["activations"])
    q1, q2 = FullyConnectedCritic(S, A, c["sizes"], c["activations"]), FullyConnectedCritic(S, A, c["sizes"], c["activations"])
    value = FloatFeatureFullyConnected(S, 1, c["sizes"], c["activations"])
    with torch.no_grad():
        for net, name in ((actor, "actor"), (q1, "q1"), (q2, "q2"), (value, "value")):
            for p, init in zip(net.parameters(), g.seq(f"init_{name}_")):
                p.copy_(init)
    adam = lambda: Optimizer__Union.default(lr=c["lr"])  # noqa: E731
    kw = dict(c.get("trainer_kw", {}))
    if c.get("crr"):
        kw["crr_config"] = CRRWeightFn(**c["crr"])
    return SACTrainer(actor.to(device), q1.to(device), q2.to(device), value_network=value.to(device),
                      rl=RLParameters(**c["rl"]), q_network_optimizer=adam(), value_network_optimizer=adam(),
                      actor_network_optimizer=adam(), alpha_optimizer=adam(), **kw).to(device)


def check_variant(tr, g, s, tol=2e-5):
    for n, net in dict(actor=tr.actor_network, q1=tr.q1_network, q2=tr.q2_network, value=tr.value_network,
                       value_target=tr.value_network_target).items():
        for i, p in enumerate(net.parameters()):
            err = (p.detach().cpu() - g.t(f"step{s}_{n}_{i}")).abs().max().item()
            assert err <= tol, (s, n, i, err)
    assert abs(tr.log_alpha.item() - g.t(f"step{s}_log_alpha").item()) <= 1e-6


@pytest.mark.parametrize("name", ["sac_value", "sac_crr"])
@pytest.mark.parametrize("path", ["generator", "native"])
def test_sac_value_network_and_crr_variants(backend, name, path):
    g = Golden(name)
    tr = build_variant(g, backend.device)
    assert tr.q1_network_target is None and tr.value_network_target is not None  # :108-112
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    assert [type(o).__name__ for o in opts] == ["FusedAdam", "FusedAdam", "FusedAdam", "AdamF64", "FusedAdam", "SoftUpdate"]
    names = ["q1_loss", "q2_loss", "actor_loss", "alpha_loss", "value_loss"]
    for s in range(g.cfg["steps"]):
        batch = synthetic.to_policy_input(g.batch(s), backend.device)
        if path == "generator":
            tr.set_noise(g.t(f"step{s}_noise_next"), g.t(f"step{s}_noise_cur"))
            losses = lightning_like_step(tr, opts, batch)
            assert len(losses) == 6
            got = dict(zip(names, losses))
        else:
            got = tr.train_step_native(batch, g.t(f"step{s}_noise_next"), g.t(f"step{s}_noise_cur"))
        for nm in names:
            ref = float(g.t(f"step{s}_{nm}"))
            assert abs(float(got[nm]) - ref) <= 1e-4 * abs(ref) + 2e-6, (s, nm, float(got[nm]), ref)
        check_variant(tr, g, s)


def test_crr_weight_fn_is_the_reference_arithmetic():
    from reagent_amd.training.sac_trainer import CRRWeightFn

    adv = torch.linspace(-3, 3, 25)
    assert torch.equal(CRRWeightFn(indicator_fn_threshold=0.5).get_weight_from_advantage(adv), (adv >= 0.5).float())
    w = CRRWeightFn(exponent_beta=0.7, exponent_clamp=3.0).get_weight_from_advantage(adv)
    assert torch.equal(w, torch.clamp(torch.exp(adv / 0.7), 0.0, 3.0))
    with pytest.raises(AssertionError):
        CRRWeightFn()


# ---- action-embedding KLD term (sac_trainer.py:130-140, 282-306) ----
@pytest.mark.parametrize("name", ["sac_kld", "sac_kld_mean"])
@pytest.mark.parametrize("path", ["generator", "native"])
def test_sac_action_embedding_kld(backend, name, path):
    g = Golden(name)
    tr = build(g, backend.device)
    assert tr.add_kld_to_loss and "action_emb_mean" in tr.state_dict() and "action_emb_variance" in tr.state_dict()
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    names = ["q1_loss", "q2_loss", "actor_loss", "alpha_loss"]
    for s in range(g.cfg["steps"]):
        batch = synthetic.to_policy_input(g.batch(s), backend.device)
        if path == "generator":
            tr.set_noise(g.t(f"step{s}_noise_next"), g.t(f"step{s}_noise_cur"))
            got = dict(zip(names, lightning_like_step(tr, opts, batch)))
        else:
            got = tr.train_step_native(batch, g.t(f"step{s}_noise_next"), g.t(f"step{s}_noise_cur"))
        for nm in names:
            ref = float(g.t(f"step{s}_{nm}"))
            assert abs(float(got[nm]) - ref) <= 1e-4 * abs(ref) + 2e-6, (s, nm, float(got[nm]), ref)
        check(tr, g, s)


@pytest.mark.parametrize("on_mean", [False, True])
def test_kld_kernels_against_autograd(backend, on_mean):
    """rg_sac_kld + the KLD branch of the head backward against torch autograd of the reference formula"""
    from reagent_amd import ops

    B, A, w = 77, 4, 0.6
    gen = torch.Generator().manual_seed(5)
    ls = torch.randn(B, 2 * A, generator=gen)
    noise = torch.randn(B, A, generator=gen)
    mu, s2 = torch.randn(A, generator=gen) * 0.3, torch.rand(A, generator=gen) + 0.2
    lsr = ls.clone().requires_grad_()
    loc, sl = lsr[:, :A], lsr[:, A:].clamp(-2, 2)
    eps = 1e-6
    act = torch.tanh(loc + noise * sl.exp()).clamp(-1 + eps, 1 - eps)
    x = torch.tanh(loc).clamp(-1 + eps, 1 - eps) if on_mean else act
    m, v = x.mean(0), x.var(0)
    kld = 0.5 * ((v + (m - mu) ** 2) / s2 - 1 + s2.log() - v.log()).sum()
    (w * kld).backward()

    d = backend.device
    lsd, nd = ls.to(d), noise.to(d)
    action = torch.empty(B, A, device=d)
    ops.gaussian_head_forward(lsd, nd, action, None, None)
    coef, terms, out, loss = torch.empty(2 * A, device=d), torch.empty(A, device=d), torch.empty(1, device=d), torch.full((1,), 2.0, device=d)
    ops.sac_kld(lsd[:, :A] if on_mean else action, on_mean, mu.to(d), s2.to(d), w, coef, terms, out, loss)
    assert abs(out.item() - kld.item()) <= 1e-5 * max(1.0, abs(kld.item()))
    assert abs(loss.item() - (2.0 + w * kld.item())) <= 1e-5 * max(1.0, abs(kld.item()))
    dls = torch.empty(B, 2 * A, device=d)
    ops.gaussian_head_backward(lsd, nd, None, None, dls, kld_coef=coef, kld_on_mean=on_mean)
    assert (dls.cpu() - lsr.grad).abs().max() <= 1e-6 + 1e-4 * lsr.grad.abs().max()


def test_fused_network_updates_equal_the_separate_launches(backend, monkeypatch):
    """native step with Adam + soft update + re-staging in one launch per network (engine.FusedUpdate, fused bf16 stacks)
    == the same step with rg_adam_step / rg_soft_update / rg_mlp_stage_weights_fused launches, bit for bit"""
    from reagent_amd.engine import FusedMLP

    def make(flag):
        torch.manual_seed(7)
        set_default_precision(L.PREC_BF16)
        try:
            S, A, H = 64, 32, [256, 256]
            actor = GaussianFullyConnectedActor(S, A, H, ["relu", "relu"])
            q1, q2 = FullyConnectedCritic(S, A, H, ["relu", "relu"]), FullyConnectedCritic(S, A, H, ["relu", "relu"])
        finally:
            set_default_precision(L.PREC_F32)
        adam = lambda: Optimizer__Union.default(lr=1e-3)  # noqa: E731
        d = backend.device
        tr = SACTrainer(actor.to(d), q1.to(d), q2.to(d), rl=RLParameters(gamma=0.97, target_update_rate=0.05),
                        q_network_optimizer=adam(), actor_network_optimizer=adam(), alpha_optimizer=adam()).to(d)
        tr.use_fused_update = flag
        return tr

    ta, tb = make(True), make(False)
    for s in range(3):
        b = synthetic.to_policy_input(synthetic.policy_batch(160, 64, 32, seed=900 + s), backend.device)
        g = torch.Generator().manual_seed(50 + s)
        n1, n2 = torch.randn(160, 32, generator=g), torch.randn(160, 32, generator=g)
        oa, ob = ta.train_step_native(b, n1, n2), tb.train_step_native(b, n1, n2)
        for k in oa:
            assert torch.equal(oa[k].cpu(), ob[k].cpu()), (s, k)
        for na, nb in ((ta.q1_network, tb.q1_network), (ta.q2_network_target, tb.q2_network_target), (ta.actor_network, tb.actor_network),
                       (ta.q1_network_target, tb.q1_network_target)):
            for pa, pb in zip(na.parameters(), nb.parameters()):
                assert torch.equal(pa.detach().cpu(), pb.detach().cpu()), s
    assert isinstance(ta._e["q1"]["stack"], FusedMLP) and ta._fused_plan and tb._fused_plan is False
    launches = []
    from reagent_amd import ops
    real = ops._run
    monkeypatch.setattr(ops, "_run", lambda name, meta, call: (launches.append(name), real(name, meta, call))[1])
    ta.train_step_native(b, n1, n2)
    assert launches.count("rg_mlp_update_fused") == 3 and "rg_soft_update" not in launches and "rg_mlp_stage_weights_fused" not in launches
    # the three mean losses (q1, q2, actor) ride in the reduce launches of their networks' weight gradients: no launch of
    # their own in the native step (alpha's loss is not a reduce_sum); the generator path keeps them
    assert "rg_reduce_sum" not in launches


def test_bf16_state_rows_from_the_sampler_equal_fp32_rows(backend):
    """OfflinePolicyLoop(state_dtype=bfloat16): the gather writes the normalized state rows in the fused kernels' operand
    type (the critic's state panel bf16 next to its fp32 action panel) — the kernels round the fp32 rows to the same
    values, so losses and weights are bit-identical to the fp32-row loop"""
    import numpy as np

    from reagent_amd.core.parameters import CONTINUOUS_TRAINING_ACTION_RANGE as R
    from reagent_amd.core.parameters import NormalizationParameters
    from reagent_amd.engine import FusedMLP
    from reagent_amd.preprocessing import PolicyNetworkInputMaker, Preprocessor
    from reagent_amd.replay_memory import ReplayBuffer
    from reagent_amd.runtime import OfflinePolicyLoop

    dev = backend.device
    S, A, B, C = 64, 32, 192, 1024

    def build(state_dtype):
        torch.manual_seed(5)
        set_default_precision(L.PREC_BF16)
        try:
            nets = [GaussianFullyConnectedActor(S, A, [256, 256], ["relu", "relu"]), FullyConnectedCritic(S, A, [256, 256], ["relu", "relu"]),
                    FullyConnectedCritic(S, A, [256, 256], ["relu", "relu"])]
        finally:
            set_default_precision(L.PREC_F32)
        adam = lambda: Optimizer__Union.default(lr=1e-3)  # noqa: E731
        tr = SACTrainer(nets[0].to(dev), nets[1].to(dev), nets[2].to(dev), rl=RLParameters(gamma=0.99, target_update_rate=0.05),
                        q_network_optimizer=adam(), actor_network_optimizer=adam(), alpha_optimizer=adam()).to(dev)
        cols = synthetic.replay_contents(C, S, A, seed=3)
        cols["action"] = torch.rand(C, A, generator=torch.Generator().manual_seed(4)) * 1.8 - 0.9
        del cols["possible_actions_mask"]
        rb = ReplayBuffer(replay_capacity=C, batch_size=B, device=dev)
        rb.load_columns({k: v.to(dev) for k, v in cols.items()}, mark_all_valid=True)
        mean, std = synthetic.normalization_table(S, 7)
        pre = Preprocessor({i: NormalizationParameters(feature_type="CONTINUOUS", mean=mean[i].item(), stddev=std[i].item())
                            for i in range(S)}, device=dev)
        maker = PolicyNetworkInputMaker(np.full(A, R[0], dtype=np.float32), np.full(A, R[1], dtype=np.float32))
        return OfflinePolicyLoop(rb, tr, B, maker, pre, state_dtype=state_dtype), tr

    (la, ta), (lb, tb) = build(torch.bfloat16), build(None)
    assert isinstance(ta._e["q1"]["stack"] if hasattr(ta, "_e") else ta.q1_network.fc.stack(), FusedMLP)
    g = torch.Generator().manual_seed(9)
    for s in range(3):
        idx = torch.randint(C, (B,), generator=g)
        n1, n2 = torch.randn(B, A, generator=g), torch.randn(B, A, generator=g)
        oa, ob = la.step(idx, noise_next=n1, noise_cur=n2), lb.step(idx, noise_next=n1, noise_cur=n2)
        for k in oa:
            assert torch.equal(oa[k].cpu(), ob[k].cpu()), (s, k)
    assert la.make_batch(idx).state.float_features.dtype == torch.bfloat16 and ta._panels
    for pa, pb in zip(ta.parameters(), tb.parameters()):
        assert torch.equal(pa.detach().cpu(), pb.detach().cpu())


@pytest.mark.parametrize("name", ["sac_twin", "sac_value", "sac_crr", "sac_kld", "sac_kld_mean"])
def test_per_step_metrics_match_the_reference(backend, name):
    """SURVEY §8 a19 — what SACTrainer hands `self.logger.log_metrics` each step (reagent/training/sac_trainer.py:343-380):
    every key and value the reference logged on the golden batches (recorded by the oracle's loop): td_loss, reward /
    Q-value / target means, entropy temperature, log-prob means, next-state value, min-Q of the actor's action, actor loss
    (before the KLD term), q2_value, target_state_value (value network), the KLD statistics."""
    g = Golden(name)
    tr = build_variant(g, backend.device) if g.cfg.get("value") else build(g, backend.device)
    logged = {}

    class Logger:
        def log_metrics(self, metrics, step=None):
            assert step == tr.all_batches_processed
            logged.update(metrics)

    tr.logger = Logger()
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    for s in range(g.cfg["steps"]):
        batch = synthetic.to_policy_input(g.batch(s), backend.device)
        tr.set_noise(g.t(f"step{s}_noise_next"), g.t(f"step{s}_noise_cur"))
        logged.clear()
        lightning_like_step(tr, opts, batch)
        want = {k[len(f"step{s}_metric_"):]: float(g.t(k)) for k in g.z.files if k.startswith(f"step{s}_metric_")}
        assert set(logged) == set(want) and len(want) >= 11, (sorted(logged), sorted(want))
        for k, ref in want.items():
            got = float(logged[k])
            assert abs(got - ref) <= 2e-5 * max(1.0, abs(ref)), (s, k, got, ref)
    tr.logger = None  # no logger: the step evaluates none of it
    logged.clear()
    lightning_like_step(tr, opts, batch)
    assert not logged


def test_device_scheduled_step_ticks_its_schedules_in_one_launch(backend, monkeypatch):
    """The four optimizers of a SAC step (q1, q2, actor, temperature) each count their Adam steps in a device-resident schedule
    (graph mode).  Their ticks leave as ONE rg_sched_tick_many launch at the end of the native step (ops.deferred_ticks) — and the
    scheduled steps stay bit-identical to the scalar-coefficient steps."""
    from reagent_amd import ops
    from reagent_amd.training.dqn_trainer import enable_graph_mode

    g = Golden("sac_twin")
    res = {}
    for sched in (False, True):
        tr = build(g, backend.device)
        if sched:
            enable_graph_mode(tr)
        launches = []
        real_run = ops._run

        def counting_run(name, meta, call):
            launches.append((name, dict(meta)))
            return real_run(name, meta, call)

        monkeypatch.setattr(ops, "_run", counting_run)
        for s in range(g.cfg["steps"]):
            batch = synthetic.to_policy_input(g.batch(s), backend.device)
            del launches[:]
            tr.train_step_native(batch, g.t(f"step{s}_noise_next").to(backend.device), g.t(f"step{s}_noise_cur").to(backend.device))
            ticks = [m for n, m in launches if n == "rg_sched_tick"]
            assert ticks == ([{"n": 4}] if sched else []), ticks
            assert launches[-1][0] == ("rg_sched_tick" if sched else launches[-1][0])  # the last launch of the step
        monkeypatch.setattr(ops, "_run", real_run)
        res[sched] = [p.detach().cpu().clone() for p in tr.parameters()] + [tr.log_alpha.detach().cpu().clone()]
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)
