"""QRDQNTrainer (reagent_amd.training) against golden vectors of the reference QRDQNTrainer
(tests/golden/qrdqn_*.npz) — loss within 1e-4 rel, post-step weights within 2e-5 abs (fp32 mode)."""
import pytest
import torch

import reagent_amd._lib as L
from golden_util import Golden
from reagent_amd import synthetic
from reagent_amd.core.parameters import EvaluationParameters, RLParameters
from reagent_amd.models import FullyConnectedDQN, set_default_precision
from reagent_amd.optimizer import Optimizer__Union
from reagent_amd.training import QRDQNTrainer
from test_dqn_trainer import lightning_like_step


def build(g, device, precision):
    c = g.cfg
    set_default_precision(precision)
    try:
        q = FullyConnectedDQN(c["state_dim"], c["num_actions"], c["sizes"], c["activations"], num_atoms=c["num_atoms"])
    finally:
        set_default_precision(L.PREC_F32)
    with torch.no_grad():
        for p, init in zip(q.parameters(), g.seq("init_param_")):
            p.copy_(init)
    q = q.to(device)
    return QRDQNTrainer(q, q.get_target_network(), actions=[str(i) for i in range(c["num_actions"])],
                        rl=RLParameters(**c["rl"]), double_q_learning=c["double_q"], num_atoms=c["num_atoms"],
                        optimizer=Optimizer__Union.default(lr=c["lr"]),
                        evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(device)


@pytest.mark.parametrize("name", ["qrdqn_double", "qrdqn_single_sarsa"])
def test_qrdqn_matches_reference_fp32_mode(backend, name):
    g = Golden(name)
    tr = build(g, backend.device, L.PREC_F32)
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
    assert "quantiles" in tr.state_dict() and tr.state_dict()["quantiles"].shape == (1, g.cfg["num_atoms"])
    out = tr.q_network(batch.state)
    assert out.shape == (g.cfg["batch"], g.cfg["num_actions"], g.cfg["num_atoms"])  # (B, A, N) view


def test_qrdqn_native_step_and_bf16(backend):
    g = Golden("qrdqn_double")
    tr_a, tr_b = build(g, backend.device, L.PREC_F32), build(g, backend.device, L.PREC_F32)
    opts = [o["optimizer"] for o in tr_a.configure_optimizers()]
    batch = synthetic.to_dqn_input(g.batch(0), backend.device)
    la = lightning_like_step(tr_a, opts, batch)[0]
    lb = tr_b.train_step_native(batch)
    assert torch.equal(la.cpu().reshape(()), lb.cpu().reshape(()))
    for pa, pb in zip(tr_a.q_network.parameters(), tr_b.q_network.parameters()):
        assert torch.equal(pa.detach().cpu(), pb.detach().cpu())
    tr_c = build(g, backend.device, L.PREC_BF16)
    lc = tr_c.train_step_native(batch)
    ref = g.t("step0_loss").item()
    assert abs(lc.item() - ref) <= 3e-2 * abs(ref) + 1e-3


def test_qrdqn_cpe_matches_reference(backend):
    """calc_cpe_in_training for QR-DQN: the CPE targets use q_network(next_state).mean(atoms) after the
    q-network step (qrdqn_trainer.py:161-177); two extra metrics"""
    from test_dqn_trainer import CPE_NETS

    g = Golden("qrdqn_cpe")
    c = g.cfg
    A, n_out = c["num_actions"], (len(c["cpe_metrics"]) + 1) * c["num_actions"]

    def net(out, prefix, num_atoms=None):
        m = FullyConnectedDQN(c["state_dim"], out, c["sizes"], c["activations"], num_atoms=num_atoms)
        with torch.no_grad():
            for p, init in zip(m.parameters(), g.seq(prefix)):
                p.copy_(init)
        return m.to(backend.device)

    q = net(A, "init_param_", c["num_atoms"])
    reward_net, q_cpe = net(n_out, "init_reward_network_"), net(n_out, "init_q_network_cpe_")
    tr = QRDQNTrainer(q, q.get_target_network(), metrics_to_score=list(c["cpe_metrics"]), reward_network=reward_net,
                      q_network_cpe=q_cpe, q_network_cpe_target=q_cpe.get_target_network(),
                      actions=[str(i) for i in range(A)], rl=RLParameters(**c["rl"]),
                      double_q_learning=c["double_q"], num_atoms=c["num_atoms"],
                      optimizer=Optimizer__Union.default(lr=c["lr"]), cpe_optimizer=Optimizer__Union.default(lr=c["lr"]),
                      evaluation=EvaluationParameters(calc_cpe_in_training=True)).to(backend.device)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    assert len(opts) == 4
    for s in range(c["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        losses = lightning_like_step(tr, opts, batch)
        for got, key in zip(losses[:3], ["loss", "reward_loss", "cpe_loss"]):
            want = g.t(f"step{s}_{key}").item()
            assert abs(got.item() - want) <= 1e-4 * abs(want) + 1e-6, (key, got.item(), want)
        for netname in CPE_NETS:
            for i, p in enumerate(getattr(tr, netname).parameters()):
                assert (p.detach().cpu() - g.t(f"step{s}_{netname}_{i}")).abs().max() <= 2e-5, (s, netname, i)
