"""The optional layer components of FullyConnectedNetwork (reagent/models/fully_connected_network.py:101-153): batch-norm on
a layer's input, dropout after the activation, the residual wrapper — kernels against torch, whole stacks against torch
autograd of the same layer sequence, module layout against the reference's parameter names."""
import pytest
import torch
import torch.nn.functional as F

import reagent_amd._lib as L
from reagent_amd import ops

_ACT = {"relu": torch.relu, "tanh": torch.tanh, "linear": lambda t: t, "leaky_relu": F.leaky_relu, "sigmoid": torch.sigmoid}


@pytest.mark.parametrize("B,n", [(5, 3), (300, 70), (1000, 129)])
def test_batch_norm_kernels_against_torch(backend, B, n):
    gen = torch.Generator().manual_seed(B)
    d = backend.device
    x, g = torch.randn(B, n, generator=gen) * 2 + 1, torch.randn(B, n, generator=gen)
    bn = torch.nn.BatchNorm1d(n)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    rm, rv = bn.running_mean.clone().to(d), bn.running_var.clone().to(d)
    w, b = bn.weight.detach().to(d), bn.bias.detach().to(d)
    xr = x.clone().requires_grad_()
    bn(xr).backward(g)  # training mode: batch statistics, running statistics move
    ws = ops.batch_norm_workspace(B, n, d)
    y, sm, sr = torch.empty(B, n, device=d), torch.empty(n, device=d), torch.empty(n, device=d)
    ops.batch_norm_forward(x.to(d), w, b, rm, rv, True, 0.1, 1e-5, y, sm, sr, ws)
    s = lambda t: max(1.0, t.abs().max().item())  # noqa: E731
    with torch.no_grad():
        assert (y.cpu() - bn.train()(x)).abs().max() <= 4e-6 * s(y)  # (this second torch call moves torch's running stats again)
    dx, dg, db = torch.empty(B, n, device=d), torch.empty(n, device=d), torch.empty(n, device=d)
    ops.batch_norm_backward(g.to(d), x.to(d), w, sm, sr, None, True, 1e-5, ws, dx, dg, db)
    assert (dx.cpu() - xr.grad).abs().max() <= 1e-5 * s(xr.grad)
    assert (dg.cpu() - bn.weight.grad).abs().max() <= 1e-5 * s(bn.weight.grad) * max(1, B // 100)
    assert (db.cpu() - bn.bias.grad).abs().max() <= 1e-5 * s(bn.bias.grad) * max(1, B // 100)
    # running statistics after ONE training forward
    bn2 = torch.nn.BatchNorm1d(n)
    bn2(x)
    assert (rm.cpu() - bn2.running_mean).abs().max() <= 1e-6 and (rv.cpu() - bn2.running_var).abs().max() <= 2e-6 * s(bn2.running_var)
    # eval mode: the running statistics normalise, gradients pass through the frozen scale
    bn2.weight.data.copy_(bn.weight.data)
    bn2.bias.data.copy_(bn.bias.data)
    bn2.eval()
    xr = x.clone().requires_grad_()
    ye = bn2(xr)
    ye.backward(g)
    ops.batch_norm_forward(x.to(d), w, b, rm, rv, False, 0.1, 1e-5, y)
    assert (y.cpu() - ye.detach()).abs().max() <= 4e-6 * s(ye)
    ops.batch_norm_backward(g.to(d), x.to(d), w, rm, None, rv, False, 1e-5, ws, dx, dg, db)
    assert (dx.cpu() - xr.grad).abs().max() <= 1e-5 * s(xr.grad)
    assert (dg.cpu() - bn2.weight.grad).abs().max() <= 1e-5 * s(bn2.weight.grad) * max(1, B // 100)


@pytest.mark.parametrize("B,n,p", [(7, 5, 0.3), (512, 64, 0.5), (100, 33, 0.0), (2048, 96, 0.1)])
def test_dropout_kernel(backend, B, n, p):
    d = backend.device
    x = torch.randn(B, n, generator=torch.Generator().manual_seed(2)).to(d)
    keep, y = torch.empty(B * n, dtype=torch.uint8, device=d), torch.empty(B, n, device=d)
    ops.dropout(x, p, keep, y, seed=99, offset=1)
    k = keep.view(B, n)
    scale = torch.tensor(1.0 / (1.0 - p), dtype=torch.float32)
    assert torch.equal(y.cpu(), torch.where(k.cpu() > 0, x.cpu() * scale, torch.zeros(())))
    frac = k.float().mean().item()
    assert abs(frac - (1 - p)) <= 4 * (p * (1 - p) / (B * n)) ** 0.5 + 1e-9  # 4 sigma of the binomial
    keep_b, keep_c = torch.empty_like(keep), torch.empty_like(keep)
    ops.dropout(x, p, keep_b, y, seed=99, offset=1)
    ops.dropout(x, p, keep_c, y, seed=99, offset=2)
    assert torch.equal(keep_b, keep)  # the same (seed, offset) draws the same mask ...
    if p > 0 and B * n > 1000:
        assert abs((keep_c != keep).float().mean().item() - 2 * p * (1 - p)) < 0.05  # ... another offset an independent one
    g, dx = torch.randn(B, n, generator=torch.Generator().manual_seed(3)).to(d), torch.empty(B, n, device=d)
    ops.dropout(g, p, keep, dx, generate=False)
    assert torch.equal(dx.cpu(), torch.where(k.cpu() > 0, g.cpu() * scale, torch.zeros(())))


def _reference_pass(net, x, dout, keeps, training):
    """torch autograd of the reference's layer sequence with this network's parameters (fp32 leaves):
    [BatchNorm1d] -> Linear -> [LayerNorm] -> activation -> [Dropout with the given keep masks] -> [+ input]"""
    leaves = {k: v.detach().cpu().clone().requires_grad_() for k, v in net.named_parameters()}
    name = {id(p): k for k, p in net.named_parameters()}
    P = lambda p: leaves[name[id(p)]]  # noqa: E731
    xr = x.clone().requires_grad_()
    h = xr
    stats = []
    for i, (lin, ln, bn, p, res) in enumerate(zip(net.linears(), net.layer_norms(), net.batch_norms(), net.dropouts(), net.residuals())):
        h_in = h
        if bn is not None:
            rm, rv = bn.running_mean.detach().clone().cpu(), bn.running_var.detach().clone().cpu()
            h = F.batch_norm(h, rm, rv, P(bn.weight), P(bn.bias), training, 0.1, bn.eps)
            stats.append((rm, rv))
        h = F.linear(h, P(lin.weight), P(lin.bias))
        if ln is not None:
            h = F.layer_norm(h, (lin.out_features,), P(ln.weight), P(ln.bias), ln.eps)
        h = _ACT[net.activation_names[i]](h)
        if p > 0.0 and training:
            h = h * keeps[i].float() / (1.0 - p)
        if res:
            h = h_in + h
    h.backward(dout)
    return h.detach(), xr.grad, {k: v.grad for k, v in leaves.items()}, stats


CASES = {
    "batch_norm": dict(layers=[12, 32, 32, 5], acts=["relu", "tanh", "linear"], kw=dict(use_batch_norm=True)),
    "skip": dict(layers=[24, 24, 24, 3], acts=["relu", "relu", "linear"], kw=dict(use_skip_connections=True)),
    "dropout": dict(layers=[10, 40, 24, 4], acts=["relu", "leaky_relu", "linear"], kw=dict(dropout_ratio=0.25)),
    "everything": dict(layers=[16, 16, 16, 16], acts=["relu", "tanh", "linear"],
                       kw=dict(use_batch_norm=True, use_layer_norm=True, dropout_ratio=0.2, use_skip_connections=True,
                               normalize_output=True)),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("precision,tol", [(L.PREC_F32, 5e-5), (L.PREC_BF16, 8e-2)])
@pytest.mark.parametrize("training", [True, False])
def test_general_stack_against_autograd(backend, case, precision, tol, training):
    from reagent_amd.engine_general import GeneralFCStack
    from reagent_amd.models import FullyConnectedNetwork, set_default_precision

    c = CASES[case]
    torch.manual_seed(11)
    set_default_precision(precision)
    try:
        net = FullyConnectedNetwork(c["layers"], c["acts"], **c["kw"])
    finally:
        set_default_precision(L.PREC_F32)
    with torch.no_grad():
        for m in [m for m in net.layer_norms() + net.batch_norms() if m is not None]:
            m.weight.uniform_(0.5, 1.5)
            m.bias.normal_(0, 0.2)
        for bn in [b for b in net.batch_norms() if b is not None]:
            bn.running_mean.normal_(0, 0.3)
            bn.running_var.uniform_(0.5, 2.0)
        for l in net.linears():
            l.bias.normal_(0, 0.1)
    net.train(training)
    B = 64
    gen = torch.Generator().manual_seed(5)
    x, dout = torch.randn(B, c["layers"][0], generator=gen), torch.randn(B, c["layers"][-1], generator=gen) / B
    dev = backend.device
    net = net.to(dev)
    st = net.stack()
    assert isinstance(st, GeneralFCStack)
    before = [(b.running_mean.clone().cpu(), b.running_var.clone().cpu()) for b in net.batch_norms() if b is not None]
    st.set_need_input_grad(True)
    st.stage_weights(need_transposed=True)
    xc, xt = st.stage_input(x.to(dev), need_transposed=True)
    out = torch.empty(B, c["layers"][-1], device=dev)
    st.forward(xc, out, save=True)
    keeps = [st._bufs[("keep/s", i)].view(B, -1).cpu() if (p > 0 and training) else None for i, p in enumerate(net.dropouts())]
    # the reference pass starts from the running statistics the forward found
    after = [(b.running_mean.clone(), b.running_var.clone()) for b in net.batch_norms() if b is not None]
    for b, (rm, rv) in zip([b for b in net.batch_norms() if b is not None], before):
        b.running_mean.copy_(rm.to(dev))
        b.running_var.copy_(rv.to(dev))
    out_ref, dx_ref, grads_ref, stats = _reference_pass(net, x, dout, keeps, training)
    s = lambda t: max(1.0, t.abs().max().item())  # noqa: E731
    assert (out.cpu() - out_ref).abs().max() <= tol * s(out_ref)
    for (rm, rv), (rm_ref, rv_ref) in zip(after, stats):  # running statistics: moved in training mode, untouched in eval
        assert (rm.cpu() - rm_ref).abs().max() <= max(tol, 1e-5) * s(rm_ref) and (rv.cpu() - rv_ref).abs().max() <= max(tol, 1e-5) * s(rv_ref)
    params = list(net.parameters())
    from reagent_amd.engine import ensure_slab, grad_views

    slab = ensure_slab(params)
    dw, db = grad_views(net, slab, params)
    slab.grad.zero_()
    dx = torch.empty(B, c["layers"][0], device=dev)
    st.backward(dout.to(dev), xt, dw, db, dx32=dx)
    assert (dx.cpu() - dx_ref).abs().max() <= tol * s(dx_ref)
    for i, (k, p) in enumerate(net.named_parameters()):
        got = slab.view(slab.grad, i).cpu()
        assert (got - grads_ref[k]).abs().max() <= tol * s(grads_ref[k]), k
    # a frozen pass (input gradient only) leaves the parameter gradients alone
    keep_grad = slab.grad.clone()
    dx2 = torch.empty_like(dx)
    st.backward(dout.to(dev), xt, dw, db, dx32=dx2, skip_wgrad=True)
    assert torch.equal(slab.grad, keep_grad) and torch.equal(dx2, dx)
    # module-level inference: forward() follows the module's mode
    y = net(x.to(dev))
    if not training:
        assert (y.cpu() - out_ref).abs().max() <= tol * s(out_ref)


def test_module_layout_matches_the_reference_names():
    """parameter / buffer names of the reference's module with every option on (probed from the reference:
    oracle/make_golden.py::gen_fc_options)"""
    from reagent_amd.models import FullyConnectedNetwork

    net = FullyConnectedNetwork([8, 8, 4], ["relu", "linear"], use_batch_norm=True, use_layer_norm=True, dropout_ratio=0.1,
                                use_skip_connections=True)
    assert list(net.state_dict().keys()) == [
        "dnn.0.module.0.vanilla.weight", "dnn.0.module.0.vanilla.bias", "dnn.0.module.0.vanilla.running_mean",
        "dnn.0.module.0.vanilla.running_var", "dnn.0.module.0.vanilla.num_batches_tracked", "dnn.0.module.1.weight",
        "dnn.0.module.1.bias", "dnn.0.module.2.weight", "dnn.0.module.2.bias",
        "dnn.1.0.vanilla.weight", "dnn.1.0.vanilla.bias", "dnn.1.0.vanilla.running_mean", "dnn.1.0.vanilla.running_var",
        "dnn.1.0.vanilla.num_batches_tracked", "dnn.1.1.weight", "dnn.1.1.bias"]
    assert net.residuals() == [True, False] and net.dropouts() == [0.1, 0.0] and not net.is_plain()


def test_all_options_match_the_reference_module(backend):
    """golden fc_options: the reference's FullyConnectedNetwork with batch-norm + layer-norm + residual wrappers —
    same state_dict names (loaded strictly), training-mode forward / backward (batch statistics, running statistics
    after the forward), eval-mode forward; with dropout layers the names stay and eval mode is the identity"""
    from golden_util import Golden
    from reagent_amd.engine import ensure_slab, grad_views
    from reagent_amd.models import FullyConnectedNetwork

    g = Golden("fc_options")
    c = g.cfg
    names = [str(n) for n in g.a("names")]
    make = lambda p: FullyConnectedNetwork(c["layers"], c["activations"], use_batch_norm=True, use_layer_norm=True,  # noqa: E731
                                           dropout_ratio=p, use_skip_connections=True)
    net = make(0.0)
    assert list(net.state_dict().keys()) == names
    assert list(make(c["dropout_ratio"]).state_dict().keys()) == [str(n) for n in g.a("names_dropout")]
    net.load_state_dict({n: g.t(f"init_{i}") for i, n in enumerate(names)}, strict=True)
    dev = backend.device
    net = net.to(dev)
    x, dout = g.t("x").to(dev), g.t("dout").to(dev)
    st = net.stack()
    st.set_need_input_grad(True)
    st.stage_weights(need_transposed=True)
    xc, xt = st.stage_input(x, need_transposed=True)
    out = torch.empty(x.shape[0], c["layers"][-1], device=dev)
    st.forward(xc, out, save=True)
    s = lambda t: max(1.0, t.abs().max().item())  # noqa: E731
    assert (out.cpu() - g.t("train_out")).abs().max() <= 3e-5 * s(g.t("train_out"))
    for i, (n, v) in enumerate(net.state_dict().items()):  # running_mean / running_var / num_batches_tracked moved once
        ref = g.t(f"after_{i}")
        assert (v.cpu().double() - ref.double()).abs().max() <= 1e-5 * s(ref.double()), n
    params = list(net.parameters())
    slab = ensure_slab(params)
    dw, db = grad_views(net, slab, params)
    dx = torch.empty_like(x)
    st.backward(dout, xt, dw, db, dx32=dx)
    assert (dx.cpu() - g.t("train_dx")).abs().max() <= 3e-5 * s(g.t("train_dx"))
    for i, (n, _) in enumerate(net.named_parameters()):
        ref = g.t(f"train_grad_{i}")
        assert (slab.view(slab.grad, i).cpu() - ref).abs().max() <= 3e-5 * s(ref), n
    net.eval()
    assert (net(x).cpu() - g.t("eval_out")).abs().max() <= 3e-5 * s(g.t("eval_out"))
    net_d = make(c["dropout_ratio"]).to(dev)
    net_d.load_state_dict(net.state_dict())
    net_d.eval()
    assert (net_d(x).cpu() - g.t("eval_out")).abs().max() <= 3e-5 * s(g.t("eval_out"))
    net_d.train()  # training mode drops activations: a different output, same expectation scale
    assert (net_d(x).cpu() - g.t("train_out")).abs().max() > 1e-3


@pytest.mark.parametrize("name", ["dqn_batchnorm", "dqn_dueling_bn", "qrdqn_bn"])
@pytest.mark.parametrize("path", ["generator", "native"])
def test_dqn_with_batch_norm_matches_reference(backend, path, name):
    """golden dqn_batchnorm (dqn_dueling_bn: the default builder's dueling network, batch-normed trunk; qrdqn_bn: a
    quantile network under QRDQNTrainer, single-Q): FullyConnectedDQN(use_batch_norm=True) under DQNTrainer — losses, Q-values, gradients,
    parameters, target parameters and BOTH networks' running statistics (which include the reference's post-step
    q_network(next_state) forward, dqn_trainer.py:268) over three steps"""
    from golden_util import Golden
    from reagent_amd import synthetic
    from reagent_amd.core.parameters import EvaluationParameters, RLParameters
    from reagent_amd.models import FullyConnectedDQN
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.training import DQNTrainer

    g = Golden(name)
    c = g.cfg
    if c.get("dueling"):
        from reagent_amd.models import DuelingQNetwork

        q = DuelingQNetwork.make_fully_connected(c["state_dim"], c["num_actions"], c["sizes"], c["activations"], use_batch_norm=True)
    else:
        q = FullyConnectedDQN(c["state_dim"], c["num_actions"], c["sizes"], c["activations"], use_batch_norm=True,
                              num_atoms=c.get("num_atoms"))
    inits = g.seq("init_param_")
    assert [tuple(p.shape) for p in q.parameters()] == [tuple(t.shape) for t in inits]  # BatchNorm, Linear, ... in module order
    with torch.no_grad():
        for p, init in zip(q.parameters(), inits):
            p.copy_(init)
    q = q.to(backend.device)
    common = dict(actions=[str(i) for i in range(c["num_actions"])], rl=RLParameters(**c["rl"]),
                  double_q_learning=c["double_q"], optimizer=Optimizer__Union.default(lr=c["lr"]),
                  evaluation=EvaluationParameters(calc_cpe_in_training=False))
    if c.get("num_atoms"):
        from reagent_amd.training import QRDQNTrainer

        tr = QRDQNTrainer(q, q.get_target_network(), num_atoms=c["num_atoms"], **common).to(backend.device)
    else:
        tr = DQNTrainer(q, q.get_target_network(), None, **common).to(backend.device)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    for s in range(c["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)

        def grads_match():
            for i, p in enumerate(tr.q_network.parameters()):
                ref = g.t(f"step{s}_grad_{i}")
                assert (p.grad.cpu() - ref).abs().max() <= 3e-5 * max(1.0, ref.abs().max().item()), i

        if path == "generator":
            loss = tr.training_step(batch, 0, 0)
            opts[0].zero_grad()
            loss.backward()
            grads_match()
            opts[0].step()
            opts[1].zero_grad()
            tr.training_step(batch, 0, 1).backward()
            opts[1].step()
        else:
            loss = tr.train_step_native(batch)
            grads_match()
        ref_loss = g.t(f"step{s}_loss")
        assert abs(loss.item() - ref_loss.item()) <= 1e-4 * abs(ref_loss.item()) + 1e-6
        if g.has(f"step{s}_q"):
            assert (tr.all_action_scores.cpu() - g.t(f"step{s}_q")).abs().max() <= 1e-4
        for i, p in enumerate(tr.q_network.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_param_{i}")).abs().max() <= 2e-5, (s, i)
        for i, p in enumerate(tr.q_network_target.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_target_{i}")).abs().max() <= 2e-5, (s, i)
        for key, net in (("qbuf", tr.q_network), ("tbuf", tr.q_network_target)):
            for i, bf in enumerate(net.buffers()):
                ref = g.t(f"step{s}_{key}_{i}")
                assert (bf.cpu().double() - ref.double()).abs().max() <= 2e-5 * max(1.0, ref.double().abs().max().item()), (s, key, i)


@pytest.mark.parametrize("path", ["generator", "native"])
def test_sac_with_batch_normed_networks_matches_reference(backend, path):
    """golden sac_bn: batch-normed critics and Gaussian actor under SACTrainer (every network in training mode) — losses,
    parameters, target parameters and all five networks' running statistics (the actor's move twice per forward: the
    reference evaluates its stack in forward() and again in get_log_prob())"""
    from golden_util import Golden
    from reagent_amd import synthetic
    from reagent_amd.core.parameters import RLParameters
    from reagent_amd.models import FullyConnectedCritic, GaussianFullyConnectedActor
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.training import SACTrainer
    from test_layer_norm import _lightning_step
    from test_sac_trainer import check

    g = Golden("sac_bn")
    c = g.cfg
    S, A = c["state_dim"], c["action_dim"]
    actor = GaussianFullyConnectedActor(S, A, c["sizes"], c["activations"], use_batch_norm=True)
    q1 = FullyConnectedCritic(S, A, c["sizes"], c["activations"], use_batch_norm=True)
    q2 = FullyConnectedCritic(S, A, c["sizes"], c["activations"], use_batch_norm=True)
    with torch.no_grad():
        for net, name in ((actor, "actor"), (q1, "q1"), (q2, "q2")):
            inits = g.seq(f"init_{name}_")
            assert len(inits) == len(list(net.parameters()))
            for p, init in zip(net.parameters(), inits):
                p.copy_(init)
    adam = lambda: Optimizer__Union.default(lr=c["lr"])  # noqa: E731
    d = backend.device
    tr = SACTrainer(actor.to(d), q1.to(d), q2.to(d), rl=RLParameters(**c["rl"]), q_network_optimizer=adam(),
                    actor_network_optimizer=adam(), alpha_optimizer=adam()).to(d)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    names = ["q1_loss", "q2_loss", "actor_loss", "alpha_loss"]
    for s in range(c["steps"]):
        batch = synthetic.to_policy_input(g.batch(s), d)
        if path == "generator":
            tr.set_noise(g.t(f"step{s}_noise_next"), g.t(f"step{s}_noise_cur"))
            got = dict(zip(names, _lightning_step(tr, opts, batch)))
        else:
            got = tr.train_step_native(batch, g.t(f"step{s}_noise_next"), g.t(f"step{s}_noise_cur"))
        for nm in names:
            ref = float(g.t(f"step{s}_{nm}"))
            assert abs(float(got[nm]) - ref) <= 1e-4 * abs(ref) + 2e-6, (s, nm, float(got[nm]), ref)
        check(tr, g, s, tol=3e-5)
        for n, net in dict(actor=tr.actor_network, q1=tr.q1_network, q2=tr.q2_network, q1_target=tr.q1_network_target,
                           q2_target=tr.q2_network_target).items():
            for i, bf in enumerate(net.buffers()):
                ref = g.t(f"step{s}_{n}_buf_{i}")
                assert (bf.cpu().double() - ref.double()).abs().max() <= 3e-5 * max(1.0, ref.double().abs().max().item()), (s, n, i)


@pytest.mark.parametrize("path", ["generator", "native"])
def test_td3_with_batch_normed_networks_matches_reference(backend, path):
    """golden td3_bn: batch-normed deterministic actor under TD3Trainer (delayed policy update; plain critics — with
    batch-normed critics the actor loss -mean_b q1(s, actor(s)) does not depend on the action at all): losses, parameters of
    all six networks and the actors' running statistics over four steps"""
    from golden_util import Golden
    from reagent_amd import synthetic
    from reagent_amd.core.parameters import RLParameters
    from reagent_amd.models import FullyConnectedActor, FullyConnectedCritic
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.training import TD3Trainer
    from test_td3_trainer import check, lightning_like_step, nets

    g = Golden("td3_bn")
    c = g.cfg
    S, A = c["state_dim"], c["action_dim"]
    actor = FullyConnectedActor(S, A, c["sizes"], c["activations"], use_batch_norm=True)
    q1 = FullyConnectedCritic(S, A, c["sizes"], c["activations"])
    q2 = FullyConnectedCritic(S, A, c["sizes"], c["activations"])
    with torch.no_grad():
        for net, name in ((actor, "actor"), (q1, "q1"), (q2, "q2")):
            inits = g.seq(f"init_{name}_")
            assert len(inits) == len(list(net.parameters()))
            for p, init in zip(net.parameters(), inits):
                p.copy_(init)
    adam = lambda: Optimizer__Union.default(lr=c["lr"])  # noqa: E731
    d = backend.device
    tr = TD3Trainer(actor.to(d), q1.to(d), q2.to(d), rl=RLParameters(**c["rl"]), q_network_optimizer=adam(),
                    actor_network_optimizer=adam(), noise_variance=c["noise_variance"], noise_clip=c["noise_clip"],
                    delayed_policy_update=c["delayed_policy_update"]).to(d)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    for s in range(c["steps"]):
        batch = synthetic.to_policy_input(g.batch(s), d)
        if path == "generator":
            tr.set_noise(g.t(f"step{s}_noise"))
            losses = lightning_like_step(tr, opts, batch, s)
            q1_loss = losses[0]
        else:
            q1_loss = tr.train_step_native(batch, g.t(f"step{s}_noise"))["q1_loss"]
        ref = float(g.t(f"step{s}_q1_loss"))
        assert abs(float(q1_loss) - ref) <= 1e-4 * abs(ref) + 2e-6
        check(tr, g, s, tol=3e-5)
        for n, net in nets(tr).items():
            for i, bf in enumerate(net.buffers()):
                ref = g.t(f"step{s}_{n}_buf_{i}")
                assert (bf.cpu().double() - ref.double()).abs().max() <= 3e-5 * max(1.0, ref.double().abs().max().item()), (s, n, i)


def test_dropout_networks_refuse_graph_capture(backend):
    """a captured step would replay ONE dropout mask (the Philox offset is a launch argument): refused loudly"""
    from reagent_amd.core.parameters import EvaluationParameters, RLParameters
    from reagent_amd.models import FullyConnectedDQN
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.training import DQNTrainer
    from reagent_amd.training.dqn_trainer import enable_graph_mode

    q = FullyConnectedDQN(6, 3, [16, 16], ["relu", "relu"], dropout_ratio=0.2).to(backend.device)
    tr = DQNTrainer(q, q.get_target_network(), None, actions=["a", "b", "c"], rl=RLParameters(),
                    optimizer=Optimizer__Union.default(), evaluation=EvaluationParameters(calc_cpe_in_training=False))
    with pytest.raises(NotImplementedError, match="dropout"):
        enable_graph_mode(tr)
