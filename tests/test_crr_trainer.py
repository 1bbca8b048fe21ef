"""DiscreteCRRTrainer (reagent_amd.training) against golden vectors of the reference
DiscreteCRRTrainer (tests/golden/crr_*.npz, produced by the unmodified reference under the
Lightning-loop emulation), the two head kernels against the reference's formulas written out in torch
with autograd (discrete_crr_trainer.py:191-285), and the reference's own structural test
(reagent/test/training/test_crr.py, which pins counts and order but no numbers) restated at the end.
Tolerances: losses 1e-4 rel, parameters 2e-5 abs (fp32 mode)."""
import pytest
import torch
import torch.nn.functional as F
from torch import distributions as pyd

import reagent_amd._lib as L
from golden_util import Golden
from reagent_amd import ops, synthetic
from reagent_amd.core.parameters import EvaluationParameters, RLParameters
from reagent_amd.models import FullyConnectedActor, FullyConnectedDQN, set_default_precision
from reagent_amd.optimizer import Optimizer__Union
from reagent_amd.training import DiscreteCRRTrainer
from test_td3_trainer import lightning_like_step

CASES = ["crr_twin_entropy_cpe", "crr_single_delayed"]


def build(g, device, precision=L.PREC_F32):
    c = g.cfg
    S, A, sizes, acts = c["state_dim"], c["num_actions"], c["sizes"], c["activations"]
    cpe = c["cpe_metrics"] is not None
    set_default_precision(precision)
    try:
        nets = dict(actor=FullyConnectedActor(S, A, sizes, acts, action_activation=c.get("actor_activation", "tanh")),
                    q1=FullyConnectedDQN(S, A, sizes, acts))
        if c["twin"]:
            nets["q2"] = FullyConnectedDQN(S, A, sizes, acts)
        if cpe:
            n_out = (len(c["cpe_metrics"]) + 1) * A
            nets["reward"] = FullyConnectedDQN(S, n_out, sizes, acts)
            nets["cpe"] = FullyConnectedDQN(S, n_out, sizes, acts)
    finally:
        set_default_precision(L.PREC_F32)
    with torch.no_grad():
        for name, net in nets.items():
            for p, init in zip(net.parameters(), g.seq(f"init_{name}_")):
                p.copy_(init)
    nets = {k: n.to(device) for k, n in nets.items()}
    adam = lambda: Optimizer__Union.default(lr=c["lr"])  # noqa: E731
    tr = DiscreteCRRTrainer(
        actor_network=nets["actor"], actor_network_target=nets["actor"].get_target_network(), q1_network=nets["q1"],
        q1_network_target=nets["q1"].get_target_network(), reward_network=nets.get("reward"),
        q2_network=nets.get("q2"), q2_network_target=nets["q2"].get_target_network() if c["twin"] else None,
        q_network_cpe=nets.get("cpe"), q_network_cpe_target=nets["cpe"].get_target_network() if cpe else None,
        metrics_to_score=list(c["cpe_metrics"]) if cpe else None,
        evaluation=EvaluationParameters(calc_cpe_in_training=cpe), rl=RLParameters(**c["rl"]),
        q_network_optimizer=adam(), actor_network_optimizer=adam(), actions=[str(i) for i in range(A)], **c["trainer"])
    return tr.to(device)


def nets_of(tr):
    nets = dict(actor=tr.actor_network, actor_target=tr.actor_network_target, q1=tr.q1_network,
                q1_target=tr.q1_network_target)
    if tr.q2_network is not None:
        nets.update(q2=tr.q2_network, q2_target=tr.q2_network_target)
    if tr.calc_cpe_in_training:
        nets.update(reward=tr.reward_network, cpe=tr.q_network_cpe, cpe_target=tr.q_network_cpe_target)
    return nets


def check(tr, g, s, tol=2e-5):
    for n, net in nets_of(tr).items():
        for i, p in enumerate(net.parameters()):
            err = (p.detach().cpu() - g.t(f"step{s}_{n}_{i}")).abs().max().item()
            assert err <= tol, (s, n, i, err)


def loss_names(c):
    names = ["q1_loss"] + (["q2_loss"] if c["twin"] else []) + ["actor_loss"]
    return names + (["reward_loss", "cpe_loss"] if c["cpe_metrics"] is not None else [])


@pytest.mark.parametrize("name", CASES)
def test_crr_matches_reference_generator_path(backend, name):
    g = Golden(name)
    c = g.cfg
    tr = build(g, backend.device)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    names = loss_names(c)
    assert len(opts) == len(names) + 1 and type(opts[-1]).__name__ == "SoftUpdate"
    for s in range(c["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        losses = lightning_like_step(tr, opts, batch, s)
        delayed = s % c["trainer"]["delayed_policy_update"] != 0
        for nm, l in zip(names, losses):
            if nm == "actor_loss" and delayed:
                assert l is None
                continue
            ref = float(g.t(f"step{s}_{nm}"))
            assert abs(float(l) - ref) <= 1e-4 * abs(ref) + 2e-6, (s, nm, float(l), ref)
        check(tr, g, s)
    keys = tr.state_dict().keys()
    assert any(k.startswith("actor_network_target.fc.dnn.0.0") for k in keys) and "reward_boosts" in keys
    scores, none = tr.get_detached_model_outputs(batch.state)
    assert none is None and scores.shape == (c["batch"], c["num_actions"])


@pytest.mark.parametrize("name", CASES)
def test_crr_native_step_matches_reference(backend, name):
    g = Golden(name)
    c = g.cfg
    tr = build(g, backend.device)
    for s in range(c["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        out = tr.train_step_native(batch)
        for nm in loss_names(c):
            if f"step{s}_{nm}" not in g.z:
                assert out[nm] is None
                continue
            ref = float(g.t(f"step{s}_{nm}"))
            assert abs(out[nm].item() - ref) <= 1e-4 * abs(ref) + 2e-6, (s, nm)
        check(tr, g, s)


@pytest.mark.parametrize("twin", [True, False])
def test_crr_critic_head_against_torch(backend, twin):
    dev, B, A, gamma = backend.device, 300, 5, 0.9
    gen = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=gen)  # noqa: E731
    q1, q2, q1n, q2n, nl = (r(B, A) for _ in range(5))
    action = F.one_hot(torch.randint(A, (B,), generator=gen), A).float()
    reward, nt = torch.rand(B, generator=gen), (torch.rand(B, generator=gen) > 0.2).float()
    boosts = r(A) * 0.1
    # discrete_crr_trainer.py:191-212
    q1r, q2r = q1.clone().requires_grad_(), q2.clone().requires_grad_()
    probs = pyd.Categorical(logits=nl).probs
    v = (q1n * probs).sum(1, keepdim=True)
    if twin:
        v = torch.min(v, (q2n * probs).sum(1, keepdim=True))
    target = (reward + (action * boosts).sum(1)).unsqueeze(1) + gamma * v * nt.unsqueeze(1)
    l1 = F.mse_loss((q1r * action).sum(1, keepdim=True), target)
    l2 = F.mse_loss((q2r * action).sum(1, keepdim=True), target)
    l1.backward()
    l2.backward()
    d = lambda t: t.to(dev).contiguous()  # noqa: E731
    P = ops.crr_partials(B)
    tgt, dq1, dq2 = torch.empty(B, device=dev), torch.empty(B, A, device=dev), torch.empty(B, A, device=dev)
    p1, p2 = torch.empty(P, device=dev), torch.empty(P, device=dev)
    ops.crr_critic_head(d(q1), d(q2) if twin else None, d(q1n), d(q2n) if twin else None, d(nl), d(action), d(reward),
                        d(boosts), d(nt), gamma, tgt, dq1, dq2 if twin else None, p1, p2 if twin else None)
    assert (tgt.cpu() - target.squeeze(1)).abs().max() <= 2e-6
    assert abs(p1.sum().item() / B - l1.item()) <= 1e-5 * abs(l1.item())
    assert (dq1.cpu() - q1r.grad).abs().max() <= 1e-7
    if twin:
        assert abs(p2.sum().item() / B - l2.item()) <= 1e-5 * abs(l2.item())
        assert (dq2.cpu() - q2r.grad).abs().max() <= 1e-7


@pytest.mark.parametrize("entropy_coeff,clip_limit", [(0.0, 10.0), (0.3, 1.5)])
def test_crr_actor_head_against_torch(backend, entropy_coeff, clip_limit):
    dev, B, A, beta, max_weight = backend.device, 300, 4, 0.6, 2.5
    gen = torch.Generator().manual_seed(5)
    q, z = torch.randn(B, A, generator=gen), torch.randn(B, A, generator=gen) * 2
    action = F.one_hot(torch.randint(A, (B,), generator=gen), A).float()
    pi_b = 0.02 + 0.98 * torch.rand(B, generator=gen)  # small values drive the ratio into the upper clip
    # discrete_crr_trainer.py:224-285
    zr = z.clone().requires_grad_()
    dist = pyd.Categorical(logits=zr)
    values = (q * dist.probs).sum(1, keepdim=True)
    weight = torch.clamp(((1 / beta) * ((q - values) * action).sum(1, keepdim=True)).exp(), 0, max_weight)
    idx = torch.argmax(action, dim=1, keepdim=True)
    log_pi_b = dist.log_prob(idx.squeeze(1)).unsqueeze(1)
    pi_t = (dist.probs * action).sum(1, keepdim=True)
    entropy = 0
    if entropy_coeff > 0:
        ratio = torch.clip(pi_t / pi_b.view(pi_t.shape), min=1e-4, max=clip_limit)
        assert (ratio == clip_limit).any() and (ratio < clip_limit).any()
        entropy = (ratio * log_pi_b).mean()
    plain = (-log_pi_b * weight.detach()).mean()
    loss = plain + entropy_coeff * entropy
    loss.backward()
    d = lambda t: t.to(dev).contiguous()  # noqa: E731
    P = ops.crr_partials(B)
    dz, pp, pe = torch.empty(B, A, device=dev), torch.empty(P, device=dev), torch.empty(P, device=dev)
    ops.crr_actor_head(d(q), d(z), d(action), d(pi_b) if entropy_coeff > 0 else None, beta, max_weight, entropy_coeff,
                       clip_limit, dz, pp, pe if entropy_coeff > 0 else None)
    assert abs(pp.sum().item() / B - plain.item()) <= 1e-5 * abs(plain.item())
    if entropy_coeff > 0:
        assert abs(pe.sum().item() / B - entropy.item()) <= 1e-5 * abs(entropy.item())
    assert (dz.cpu() - zr.grad).abs().max() <= 2e-7, (dz.cpu() - zr.grad).abs().max()


def test_crr_rejects_bad_arguments(backend):
    dev = backend.device
    t = torch.zeros(4, 3, device=dev)
    v, p = torch.zeros(4, device=dev), torch.zeros(ops.crr_partials(4), device=dev)
    with pytest.raises(Exception):  # q2 without its target
        ops.crr_critic_head(t, t, t, None, t, t, v, None, v, 0.9, v, t.clone(), t.clone(), p, p.clone())
    with pytest.raises(Exception):  # entropy term without logged propensities
        ops.crr_actor_head(t, t, t, None, 1.0, 20.0, 0.1, 10.0, t.clone(), p, p.clone())


# ---- reagent/test/training/test_crr.py restated (same sizes, same assertions on the result level) ----
class _RefCrrCase:
    """setUp of test_crr.py:19-103: batch 3, state 10, 2 actions, two layers of 20, an exploring actor
    (exploration_variance 1e-10), int64 one-hot actions, CPE nets for the reward metric only"""

    def __init__(self, device):
        self.B, self.S, self.A = 3, 10, 2
        sizes, acts = [20, 20], ["relu", "relu"]
        self.actions = [str(i) for i in range(self.A)]
        mk = lambda out: FullyConnectedDQN(self.S, out, sizes, acts).to(device)  # noqa: E731
        self.actor = FullyConnectedActor(self.S, self.A, sizes, acts, exploration_variance=1e-10).to(device)
        self.q1, self.q2 = mk(self.A), mk(self.A)
        n_out = 1 * self.A  # get_metrics_to_score(RewardOptions().metric_reward_values) is empty: reward only
        self.reward_net, self.q_cpe = mk(n_out), mk(n_out)
        from reagent_amd.core import types as rlt

        g = torch.Generator().manual_seed(0)
        t = lambda x: x.to(device)  # noqa: E731
        self.inp = rlt.DiscreteDqnInput(
            state=rlt.FeatureData(t(torch.rand(self.B, self.S, generator=g))),
            next_state=rlt.FeatureData(t(torch.rand(self.B, self.S, generator=g))),
            reward=t(torch.ones(self.B, 1)), time_diff=t(torch.ones(self.B, 1) * 2), step=t(torch.ones(self.B, 1) * 2),
            not_terminal=t(torch.ones(self.B, 1)), action=t(torch.tensor([[0, 1], [1, 0], [0, 1]])),
            next_action=t(torch.tensor([[1, 0], [0, 1], [1, 0]])), possible_actions_mask=t(torch.ones(self.B, self.A)),
            possible_next_actions_mask=t(torch.ones(self.B, self.A)),
            extras=rlt.ExtraData(action_probability=t(torch.ones(self.B, 1))))

    def trainer(self, no_cpe=False, no_q2=False, **params):
        return DiscreteCRRTrainer(
            actor_network=self.actor, actor_network_target=self.actor.get_target_network(), q1_network=self.q1,
            q1_network_target=self.q1.get_target_network(), q2_network=None if no_q2 else self.q2,
            q2_network_target=None if no_q2 else self.q2.get_target_network(),
            reward_network=None if no_cpe else self.reward_net, q_network_cpe=None if no_cpe else self.q_cpe,
            q_network_cpe_target=None if no_cpe else self.q_cpe.get_target_network(), metrics_to_score=[],
            evaluation=EvaluationParameters(calc_cpe_in_training=not no_cpe), actions=self.actions, **params)


def test_reference_crr_test_init_and_properties(backend):
    c = _RefCrrCase(backend.device)
    tr = c.trainer()
    assert torch.isclose(tr.reward_boosts, torch.zeros(2)).all()  # test_init
    boosted = c.trainer(rl=RLParameters(reward_boost={i: int(i) + 1 for i in c.actions}))
    assert torch.isclose(boosted.reward_boosts, torch.tensor([1.0, 2.0])).all()
    assert tr.q_network is tr.q1_network  # test_q_network_property
    scores, _ = tr.get_detached_model_outputs(c.inp.state)  # test_get_detached_model_outputs
    assert scores.shape == (c.B, c.A)


def test_reference_crr_test_configure_optimizers(backend):
    c = _RefCrrCase(backend.device)
    tr = c.trainer()
    optimizers = tr.configure_optimizers()
    assert len(optimizers) == 6
    order = [tr.q1_network, tr.q2_network, tr.actor_network, tr.reward_network, tr.q_network_cpe, tr.q1_network]
    for opt, net in zip(optimizers, order):  # the soft update's first group starts with q1's target/source
        opt_param = opt["optimizer"].param_groups[0]["params"][0]
        assert torch.isclose(opt_param, list(net.parameters())[0]).all()
    assert len(c.trainer(no_cpe=True).configure_optimizers()) == 4
    assert len(c.trainer(no_q2=True).configure_optimizers()) == 5


def test_reference_crr_test_train_step_gen(backend):
    c = _RefCrrCase(backend.device)
    losses = list(c.trainer().train_step_gen(c.inp, batch_idx=1))
    assert len(losses) == 6 and all(l.requires_grad for l in losses)  # (grad_fn types cannot hold for fused losses)
    assert all(torch.isfinite(l).all() for l in losses)
    assert len(list(c.trainer(no_cpe=True).train_step_gen(c.inp, batch_idx=1))) == 4
    assert len(list(c.trainer(no_q2=True).train_step_gen(c.inp, batch_idx=1))) == 5
    assert len(list(c.trainer(use_target_actor=True).train_step_gen(c.inp, batch_idx=1))) == 6
    delayed = list(c.trainer(delayed_policy_update=2).train_step_gen(c.inp, batch_idx=1))
    assert len(delayed) == 6 and delayed[2] is None
    assert len(list(c.trainer(entropy_coeff=1.0).train_step_gen(c.inp, batch_idx=1))) == 6
    # the losses drive their optimizers: one Lightning-style pass changes every online network
    tr = c.trainer()
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    before = [p.detach().clone() for p in tr.actor_network.parameters()]
    lightning_like_step(tr, opts, c.inp, 0)
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, tr.actor_network.parameters()))


@pytest.mark.parametrize("name", ["crr_twin_entropy_cpe", "crr_single_delayed"])
def test_reporter_fields_match_the_reference(emu_lib, name):
    """discrete_crr_trainer.py:375-382 (and the CPE heads' reporter fields, dqn_trainer_base.py:243-452): the tensors
    handed to the reporter against what the reference's reporter received"""
    from golden_util import check_reported

    g = Golden(name)
    tr = build(g, "cpu")
    seen = {}

    class Reporter:
        def log(self, **kw):
            seen.update(kw)

    tr.set_reporter(Reporter())
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    total = 0
    for s in range(g.cfg["steps"]):
        seen.clear()
        lightning_like_step(tr, opts, synthetic.to_dqn_input(g.batch(s), "cpu"), s)
        total += check_reported(g, s, seen)
    assert total >= 6
