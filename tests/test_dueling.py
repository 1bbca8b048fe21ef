"""DuelingQNetwork (reagent/models/dueling_q_network.py) under DQNTrainer and QRDQNTrainer against golden vectors of
the unmodified reference (tests/golden/dqn_dueling.npz, qrdqn_dueling.npz: loss, Q, autograd gradients, post-step
weights and targets).  Tolerances as in test_dqn_trainer.py; the gradients are pinned directly because Adam's first
steps (lr * g / (|g| + eps)) amplify rounding of near-zero gradient entries.
"""
import pytest
import torch

import reagent_amd._lib as L
from golden_util import Golden
from reagent_amd import ops, synthetic
from reagent_amd.core import types as rlt
from reagent_amd.core.parameters import EvaluationParameters, RLParameters
from reagent_amd.models import DuelingQNetwork, set_default_precision
from reagent_amd.optimizer import Optimizer__Union
from reagent_amd.training import DQNTrainer, QRDQNTrainer
from test_dqn_trainer import lightning_like_step


def build(g: Golden, device, precision):
    c = g.cfg
    set_default_precision(precision)
    try:
        q = DuelingQNetwork.make_fully_connected(c["state_dim"], c["num_actions"], c["sizes"], c["activations"],
                                                 num_atoms=c.get("num_atoms"))
    finally:
        set_default_precision(L.PREC_F32)
    inits = g.seq("init_param_")
    assert len(inits) == len(list(q.parameters()))
    with torch.no_grad():
        for p, init in zip(q.parameters(), inits):
            assert p.shape == init.shape  # same parameter order and layout as the reference's module
            p.copy_(init)
    q = q.to(device)
    common = dict(actions=[str(i) for i in range(c["num_actions"])], rl=RLParameters(**c["rl"]),
                  double_q_learning=c["double_q"], optimizer=Optimizer__Union.default(lr=c["lr"]),
                  evaluation=EvaluationParameters(calc_cpe_in_training=False))
    if c.get("num_atoms"):
        return QRDQNTrainer(q, q.get_target_network(), num_atoms=c["num_atoms"], **common).to(device)
    return DQNTrainer(q, q.get_target_network(), None, **common).to(device)


def test_state_dict_names_and_structure(backend):
    q = DuelingQNetwork.make_fully_connected(12, 4, [48, 32], ["relu", "leaky_relu"])
    keys = list(q.state_dict().keys())
    assert keys == [f"{net}.fc.dnn.{i}.0.{w}" for net in ("shared_network", "advantage_network", "value_network")
                    for i in range(2) for w in ("weight", "bias")]
    assert q.shared_network.fc.dnn[1][0].weight.shape == (32, 48)  # the embedding layer is linear, width layers[-1]
    assert q.shared_network.fc.activation_names == ["relu", "linear"]
    assert q.advantage_network.fc.dnn[0][0].weight.shape == (16, 32) and q.advantage_network.fc.activation_names[0] == "leaky_relu"
    assert q.value_network.fc.dnn[1][0].weight.shape == (1, 16)
    with pytest.raises(AssertionError, match="divisible by 2"):
        DuelingQNetwork.make_fully_connected(12, 4, [48, 31], ["relu", "relu"])
    t = q.get_target_network()
    assert t is not q and all(a is not b for a, b in zip(t.parameters(), q.parameters()))
    assert isinstance(q.input_prototype(), rlt.FeatureData)


@pytest.mark.parametrize("atoms", [1, 5])
def test_combine_and_split_kernels(backend, atoms):
    B, A, N = 37, 6, atoms
    gen = torch.Generator().manual_seed(3)
    val, adv, dq = torch.randn(B, N, generator=gen), torch.randn(B, A * N, generator=gen), torch.randn(B, A * N, generator=gen)
    d = backend.device
    q = torch.empty(B, A * N, device=d)
    ops.dueling_combine(val.to(d), adv.to(d), A, N, q)
    adv3 = adv.view(B, A, N).double()
    ref = val.view(B, 1, N).double() + adv3 - adv3.mean(dim=(1, 2), keepdim=True)
    assert (q.cpu().double() - ref.reshape(B, -1)).abs().max() <= 1e-6
    dadv, dval = torch.empty(B, A * N, device=d), torch.empty(B, N, device=d)
    ops.dueling_split(dq.to(d), A, N, dadv, dval)
    dq3 = dq.view(B, A, N).double()
    assert (dval.cpu().double() - dq3.sum(1)).abs().max() <= 1e-6
    assert (dadv.cpu().double() - (dq3 - dq3.mean(dim=(1, 2), keepdim=True)).reshape(B, -1)).abs().max() <= 1e-6


@pytest.mark.parametrize("name", ["dqn_dueling", "qrdqn_dueling"])
def test_dueling_matches_reference_fp32_mode(backend, name):
    g = Golden(name)
    tr = build(g, backend.device, L.PREC_F32)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    for s in range(g.cfg["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        loss = tr.training_step(batch, 0, 0)
        opts[0].zero_grad()
        loss.backward()
        ref_loss = g.t(f"step{s}_loss")
        assert abs(loss.item() - ref_loss.item()) <= 1e-4 * abs(ref_loss.item()) + 1e-6
        if name == "dqn_dueling":
            assert (tr.all_action_scores.cpu() - g.t(f"step{s}_q")).abs().max() <= 1e-4  # north_star bound
        if s == 0:  # later steps start from weights that already differ in the last bits
            for i, p in enumerate(tr.q_network.parameters()):
                ref = g.t(f"step{s}_grad_{i}")
                assert (p.grad.cpu() - ref).abs().max() <= 3e-5 * max(1.0, ref.abs().max().item()), (s, i)
        opts[0].step()
        opts[1].zero_grad()
        tr.training_step(batch, 0, 1).backward()
        opts[1].step()
        for i, p in enumerate(tr.q_network.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_param_{i}")).abs().max() <= 2e-5, (s, i)
        for i, p in enumerate(tr.q_network_target.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_target_{i}")).abs().max() <= 2e-5, (s, i)


@pytest.mark.parametrize("name", ["dqn_dueling", "qrdqn_dueling"])
def test_dueling_native_step_equals_generator_path(backend, name):
    g = Golden(name)
    tr_a, tr_b = build(g, backend.device, L.PREC_F32), build(g, backend.device, L.PREC_F32)
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


def test_dueling_module_forward_matches_reference_q(backend):
    """the module's own forward (inference entry point) = the golden's step-0 Q before any update"""
    g = Golden("dqn_dueling")
    tr = build(g, backend.device, L.PREC_F32)
    b = synthetic.to_dqn_input(g.batch(0), backend.device)
    q = tr.q_network(b.state)
    assert q.shape == (g.cfg["batch"], g.cfg["num_actions"])
    assert (q.cpu() - g.t("step0_q")).abs().max() <= 1e-4
    mask = b.possible_actions_mask
    qm = tr.q_network(b.state, mask)
    assert torch.all(qm[mask == 0] < -1e9) and torch.equal(qm[mask == 1], q[mask == 1])


# gradient bounds in the Frobenius norm (measured over seeds: bf16 0.03-0.09 per layer — dZ is rounded at five layers and a
# few relu masks flip; bf16x3 0 to 2e-3, one flipped mask being a rank-1 difference)
@pytest.mark.parametrize("precision,tol,gtol", [(L.PREC_BF16, 4e-2, 0.15), (L.PREC_BF16X3, 2e-4, 5e-3)])
def test_dueling_mixed_engines_against_torch(backend, precision, tol, gtol):
    """layers [256, 256, 256]: the trunk runs on the fused kernels, the [256 -> 128 -> .] streams on the per-layer
    GEMMs; Q and every parameter gradient against torch fp32 autograd of the reference's formula"""
    from reagent_amd.engine import FCStack, FusedMLP

    S, A, B = 40, 5, 130
    torch.manual_seed(0)
    set_default_precision(precision)
    try:
        q = DuelingQNetwork.make_fully_connected(S, A, [256, 256, 256], ["relu", "relu", "relu"]).to(backend.device)
    finally:
        set_default_precision(L.PREC_F32)
    st = q.fc.stack()
    assert isinstance(st.s, FusedMLP) and isinstance(st.a, FCStack) and st.s.x3 == (precision == L.PREC_BF16X3)
    gen = torch.Generator().manual_seed(11)
    x, dq = torch.randn(B, S, generator=gen), torch.randn(B, A, generator=gen) / B
    lin = q.fc.linears()
    ws = [l.weight.detach().cpu().clone().requires_grad_() for l in lin]
    bs = [l.bias.detach().cpu().clone().requires_grad_() for l in lin]

    def run(h, idx, acts):
        for i, a in zip(idx, acts):
            h = torch.nn.functional.linear(h, ws[i], bs[i])
            h = torch.relu(h) if a == "relu" else h
        return h

    e = run(x, [0, 1, 2], ["relu", "relu", "linear"])
    adv, val = run(e, [3, 4], ["relu", "linear"]), run(e, [5, 6], ["relu", "linear"])
    ref = val + adv - adv.mean(dim=1, keepdim=True)
    ref.backward(dq)

    xd = x.to(backend.device)
    st.stage_weights(need_transposed=True)
    xc, xt = st.stage_input(xd, need_transposed=True)
    out = torch.empty(B, A, device=backend.device)
    st.forward(xc, out, save=True)
    scale = max(1.0, ref.abs().max().item())
    assert (out.cpu() - ref.detach()).abs().max() <= tol * scale
    dw = [torch.empty_like(l.weight) for l in lin]
    db = [torch.empty_like(l.bias) for l in lin]
    st.backward(dq.to(backend.device), xt, dw, db)
    for i in range(len(lin)):
        gw, gb = ws[i].grad, bs[i].grad
        assert (dw[i].cpu() - gw).norm() <= gtol * gw.norm(), i
        assert (db[i].cpu() - gb).norm() <= gtol * gb.norm() + 1e-7, i
