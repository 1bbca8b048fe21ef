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
                    q_network_optimizer=adam(), actor_network_optimizer=adam(), alpha_optimizer=adam())
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
