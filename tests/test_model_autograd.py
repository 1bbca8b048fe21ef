"""Autograd through the model classes (reagent/models/dqn.py:52-63, critic.py:79-92 are plain nn.Modules in the reference:
`q_network(state).sum().backward()` fills .grad).  Here a Linear -> activation stack records one autograd node whose
backward is the stack's HIP backward; outputs of HIP heads that have no autograd form carry a node that RAISES on
backward instead of silently yielding no gradient."""
import pytest
import torch

import reagent_amd._lib as L
from reagent_amd.core import types as rlt
from reagent_amd.models import (FullyConnectedCritic, FullyConnectedDQN, GaussianFullyConnectedActor,
                                set_default_precision)


def _torch_forward(params, acts, x):
    h = x
    for i, a in enumerate(acts):
        h = torch.nn.functional.linear(h, params[2 * i], params[2 * i + 1])
        if a == "relu":
            h = torch.relu(h)
    return h


@pytest.mark.parametrize("precision,sizes,tol", [("f32", [24, 16], 2e-5), ("bf16x3", [256, 256], 2e-4), ("bf16", [256, 256], 6e-2)])
def test_dqn_forward_is_differentiable(backend, precision, sizes, tol):
    dev = backend.device
    set_default_precision({"f32": L.PREC_F32, "bf16x3": L.PREC_BF16X3, "bf16": L.PREC_BF16}[precision])
    try:
        torch.manual_seed(1)
        q = FullyConnectedDQN(12, 5, sizes, ["relu", "relu"]).to(dev)
    finally:
        set_default_precision(L.PREC_F32)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(40, 12, generator=g).to(dev).requires_grad_()
    coef = torch.randn(40, 5, generator=g).to(dev)
    mask = (torch.rand(40, 5, generator=g) > 0.3).float().to(dev)
    out = q(rlt.FeatureData(x), mask)
    assert out.requires_grad and out.grad_fn is not None
    (out * coef * mask).sum().backward()
    # torch reference on copies
    ref_p = [p.detach().cpu().double().requires_grad_() for p in q.parameters()]
    xr = x.detach().cpu().double().requires_grad_()
    ref = _torch_forward(ref_p, ["relu", "relu", "linear"], xr) + (1 - mask.cpu().double()) * -1e10
    (ref * coef.cpu().double() * mask.cpu().double()).sum().backward()
    assert ((out.detach().cpu().double() - ref.detach()) * mask.cpu().double()).abs().max() <= tol * 50  # (possible actions)
    for p, r in zip(q.parameters(), ref_p):
        assert p.grad is not None and p.grad.shape == p.shape
        scale = max(1.0, r.grad.abs().max().item())
        if precision == "bf16":  # 8-bit operands (and a few ReLU masks that flip): the gradient as a whole, not per element
            assert (p.grad.cpu().double() - r.grad).norm() <= 0.1 * r.grad.norm() + 1e-3, (precision, p.shape)
        else:
            assert (p.grad.cpu().double() - r.grad).abs().max() <= tol * scale, (precision, p.shape)
    assert (x.grad.cpu().double() - xr.grad).norm() <= (0.1 if precision == "bf16" else max(tol, 1e-4)) * xr.grad.norm()
    # gradients accumulate like any autograd leaf
    before = [p.grad.clone() for p in q.parameters()]
    (q(rlt.FeatureData(x.detach())) * coef).sum().backward()
    assert all(not torch.equal(b, p.grad) for b, p in zip(before, q.parameters()))
    # no graph under no_grad
    with torch.no_grad():
        assert not q(rlt.FeatureData(x)).requires_grad


def test_critic_is_differentiable_through_cat(backend):
    dev = backend.device
    torch.manual_seed(3)
    c = FullyConnectedCritic(6, 2, [16, 16], ["relu", "relu"]).to(dev)
    s = torch.randn(20, 6).to(dev)
    a = torch.randn(20, 2).to(dev).requires_grad_()
    c(rlt.FeatureData(s), rlt.FeatureData(a)).sum().backward()
    ref_p = [p.detach().cpu().double() for p in c.parameters()]
    ar = a.detach().cpu().double().requires_grad_()
    _torch_forward(ref_p, ["relu", "relu", "linear"], torch.cat((s.cpu().double(), ar), 1)).sum().backward()
    assert (a.grad.cpu().double() - ar.grad).abs().max() <= 2e-5


def test_a_stale_node_raises(backend):
    dev = backend.device
    torch.manual_seed(4)
    q = FullyConnectedDQN(8, 3, [16, 16], ["relu", "relu"]).to(dev)
    x = torch.randn(10, 8).to(dev)
    first = q(rlt.FeatureData(x))
    second = q(rlt.FeatureData(x * 2))  # overwrites the stack's saved activations
    second.sum().backward()
    with pytest.raises(RuntimeError, match="another recorded forward"):
        first.sum().backward()


def test_hip_heads_refuse_backward_instead_of_yielding_nothing(backend):
    dev = backend.device
    torch.manual_seed(5)
    actor = GaussianFullyConnectedActor(6, 2, [16, 16], ["relu", "relu"]).to(dev)
    out = actor(rlt.FeatureData(torch.randn(10, 6).to(dev)))
    assert out.action.requires_grad
    with pytest.raises(NotImplementedError, match="outside autograd"):
        out.log_prob.sum().backward()
    bn = FullyConnectedDQN(6, 2, [16, 16], ["relu", "relu"], use_batch_norm=True).to(dev)
    y = bn(rlt.FeatureData(torch.randn(10, 6).to(dev)))
    with pytest.raises(NotImplementedError, match="outside autograd"):
        y.sum().backward()
    with torch.no_grad():
        assert not actor(rlt.FeatureData(torch.randn(10, 6).to(dev))).action.requires_grad


def test_grad_mode_inference_leaves_the_trainers_workspace_alone(backend):
    """`q_network(x)` in default grad mode used to run the SAVING forward on the stack the trainer owns
    (`trainer._qs`), replacing its (batch, training) workspace — the buffers a captured HIP graph holds addresses of and the
    activations a yielded loss's backward reads.  The autograd node now has its own engine instance."""
    from reagent_amd import synthetic
    from reagent_amd.core.parameters import EvaluationParameters, RLParameters
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.training import DQNTrainer

    dev = backend.device
    S, A, B = 24, 4, 200
    set_default_precision(L.PREC_BF16)
    try:
        torch.manual_seed(3)
        q = FullyConnectedDQN(S, A, [256, 256], ["relu", "relu"]).to(dev)
    finally:
        set_default_precision(L.PREC_F32)
    tr = DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(A)],
                    rl=RLParameters(gamma=0.9, target_update_rate=0.1), optimizer=Optimizer__Union.default(lr=1e-3),
                    evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)
    batch = synthetic.to_dqn_input(synthetic.dqn_batch(B, S, A, seed=5), dev)
    tr.train_step_native(batch)
    st = tr._qs
    assert st is q.fc.stack()
    key, ptrs = st._ws["key"], [t.data_ptr() for t in st._ws["act_frag"]]
    x = torch.randn(1, S).to(dev)
    out = q(rlt.FeatureData(x))  # default grad mode: a recorded forward
    assert out.requires_grad
    assert st._ws["key"] == key and [t.data_ptr() for t in st._ws["act_frag"]] == ptrs
    assert q.fc.autograd_stack() is not st
    # a recorded forward between a yielded loss and its backward does not disturb the trainer's saved activations
    gen = tr.train_step_gen(batch, 0)
    loss = next(gen)
    q(rlt.FeatureData(x)).sum().backward()
    grads_side = [p.grad.clone() for p in q.parameters()]
    for p in q.parameters():
        p.grad = None
    loss.backward()
    got = [p.grad.clone() for p in q.parameters()]
    # same step without the interleaved forward
    for p in q.parameters():
        p.grad = None
    gen2 = tr.train_step_gen(batch, 0)
    next(gen2).backward()
    for a, b in zip(got, q.parameters()):
        assert torch.equal(a, b.grad)
    assert any(g.abs().sum() > 0 for g in grads_side)
