"""C51Trainer (reagent_amd.training) against golden vectors of the reference
C51Trainer (tests/golden/c51_*.npz): loss within 1e-4 rel, post-step weights within 2e-5 abs (fp32
mode).  The cases use a tight [qmin, qmax] so that clamped targets exercise the l == b == u fix-ups
of the categorical projection (c51_trainer.py:136-155)."""
import pytest
import torch

import reagent_amd._lib as L
from golden_util import Golden
from reagent_amd import synthetic
from reagent_amd.core.parameters import RLParameters
from reagent_amd.models import CategoricalDQN, FullyConnectedDQN, set_default_precision
from reagent_amd.optimizer import Optimizer__Union
from reagent_amd.training import C51Trainer
from test_dqn_trainer import lightning_like_step


def build(g, device, precision=L.PREC_F32):
    c = g.cfg
    set_default_precision(precision)
    try:
        dist = FullyConnectedDQN(c["state_dim"], c["num_actions"], c["sizes"], c["activations"], num_atoms=c["num_atoms"])
    finally:
        set_default_precision(L.PREC_F32)
    q = CategoricalDQN(dist, qmin=c["qmin"], qmax=c["qmax"], num_atoms=c["num_atoms"])
    with torch.no_grad():
        for p, init in zip(q.parameters(), g.seq("init_param_")):
            p.copy_(init)
    q = q.to(device)
    return C51Trainer(q, q.get_target_network(), actions=[str(i) for i in range(c["num_actions"])],
                      rl=RLParameters(**c["rl"]), double_q_learning=c["double_q"], num_atoms=c["num_atoms"],
                      qmin=c["qmin"], qmax=c["qmax"], optimizer=Optimizer__Union.default(lr=c["lr"])).to(device)


@pytest.mark.parametrize("name", ["c51_double", "c51_sarsa"])
def test_c51_matches_reference_fp32_mode(backend, name):
    g = Golden(name)
    tr = build(g, backend.device)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    assert len(opts) == 2
    for s in range(g.cfg["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        losses = lightning_like_step(tr, opts, batch)
        ref = g.t(f"step{s}_loss").item()
        assert abs(losses[0].item() - ref) <= 1e-4 * abs(ref) + 1e-6, (losses[0].item(), ref)
        for i, p in enumerate(tr.q_network.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_param_{i}")).abs().max() <= 2e-5, (s, i)
        for i, p in enumerate(tr.q_network_target.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_target_{i}")).abs().max() <= 2e-5, (s, i)
    assert tr.state_dict()["support"].shape == (g.cfg["num_atoms"],)
    qv = tr.q_network(batch.state)  # expected values, (B, A)
    assert qv.shape == (g.cfg["batch"], g.cfg["num_actions"])
    assert (tr.all_q_values.cpu() - 0).shape == qv.shape
    ld = tr.q_network.log_dist(batch.state)
    assert ld.shape == (g.cfg["batch"], g.cfg["num_actions"], g.cfg["num_atoms"])
    assert (ld.exp().sum(-1) - 1).abs().max() <= 1e-5


def test_c51_native_step(backend):
    g = Golden("c51_double")
    tr_a, tr_b = build(g, backend.device), build(g, backend.device)
    opts = [o["optimizer"] for o in tr_a.configure_optimizers()]
    batch = synthetic.to_dqn_input(g.batch(0), backend.device)
    la = lightning_like_step(tr_a, opts, batch)[0]
    lb = tr_b.train_step_native(batch)
    assert torch.equal(la.cpu().reshape(()), lb.cpu().reshape(()))
    for pa, pb in zip(tr_a.q_network.parameters(), tr_b.q_network.parameters()):
        assert torch.equal(pa.detach().cpu(), pb.detach().cpu())


@pytest.mark.parametrize("name", ["c51_double", "c51_sarsa"])
def test_reporter_fields_match_the_reference(emu_lib, name):
    """c51_trainer.py:179-186: the tensors handed to the reporter against what the reference's reporter received"""
    from golden_util import check_reported

    g = Golden(name)
    tr = build(g, "cpu")
    seen = {}

    class Reporter:
        def log(self, **kw):
            seen.update(kw)

    tr.set_reporter(Reporter())
    from test_td3_trainer import lightning_like_step as step_with_batch_idx

    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    total = 0
    for s in range(g.cfg["steps"]):  # reported every log_every_n_steps batches (:178): batch 0 here, nothing afterwards
        seen.clear()
        step_with_batch_idx(tr, opts, synthetic.to_dqn_input(g.batch(s), "cpu"), s)
        total += check_reported(g, s, seen)
    assert total == 6
