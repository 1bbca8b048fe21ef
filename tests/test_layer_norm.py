"""use_layer_norm (reagent/models/fully_connected_network.py:128-130): the LayerNorm kernels against torch, FC stacks with
LayerNorm against torch autograd, and trainers with layer-normed networks against golden runs of the reference."""
import pytest
import torch

import reagent_amd._lib as L
from reagent_amd import ops


@pytest.mark.parametrize("n,act,dtype", [(48, "relu", torch.float32), (512, "tanh", torch.bfloat16), (200, "linear", torch.float32)])
def test_layer_norm_kernels_against_torch(backend, n, act, dtype):
    B = 37
    gen = torch.Generator().manual_seed(1)
    z = torch.randn(B, n, generator=gen) * 2 + 0.5
    gamma, beta = torch.rand(n, generator=gen) + 0.5, torch.randn(n, generator=gen) * 0.2
    g = torch.randn(B, n, generator=gen)
    zr, gr, br = z.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ln = torch.nn.functional.layer_norm(zr, (n,), gr, br, 1e-5)
    ref = {"relu": torch.relu, "tanh": torch.tanh, "linear": lambda t: t}[act](ln)
    ln.backward(g)  # g = gradient at the LayerNorm output (the activation's derivative is applied by the caller)
    d = backend.device
    y, y32 = torch.empty(B, n, dtype=dtype, device=d), torch.empty(B, n, device=d)
    mean, rstd = torch.empty(B, device=d), torch.empty(B, device=d)
    zd, gd, bd = z.to(d), gamma.to(d), beta.to(d)
    ops.layer_norm_forward(zd, gd, bd, 1e-5, L.ACT[act], y=y, y32=y32, mean=mean, rstd=rstd)
    assert (y32.cpu() - ref.detach()).abs().max() <= 2e-6 * max(1.0, ref.abs().max().item())
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-6
    assert (y.float().cpu() - ref.detach()).abs().max() <= tol * max(1.0, ref.abs().max().item())
    assert (mean.cpu() - z.mean(1)).abs().max() <= 1e-6
    ws = torch.empty(L.lib().rg_layer_norm_backward_workspace_bytes(B, n) // 4, device=d)
    dz, dz32 = torch.empty(B, n, dtype=dtype, device=d), torch.empty(B, n, device=d)
    dgamma, dbeta = torch.empty(n, device=d), torch.empty(n, device=d)
    ops.layer_norm_backward(g.to(d), zd, mean, rstd, gd, dgamma, dbeta, ws, dz=dz, dz32=dz32)
    assert (dz32.cpu() - zr.grad).abs().max() <= 1e-5 * max(1.0, zr.grad.abs().max().item())
    assert (dgamma.cpu() - gr.grad).abs().max() <= 1e-5 * max(1.0, gr.grad.abs().max().item())
    assert (dbeta.cpu() - br.grad).abs().max() <= 1e-5 * max(1.0, br.grad.abs().max().item())


@pytest.mark.parametrize("precision,tol", [(L.PREC_F32, 3e-5), (L.PREC_BF16, 6e-2)])
def test_layer_normed_stack_against_autograd(backend, precision, tol):
    """FullyConnectedNetwork(use_layer_norm=True, normalize_output=True): forward and every gradient (weights, biases,
    LayerNorm gamma / beta, input) against torch autograd of Linear -> LayerNorm -> activation"""
    from reagent_amd.engine import FCStack
    from reagent_amd.models import FullyConnectedNetwork, set_default_precision

    torch.manual_seed(3)
    set_default_precision(precision)
    try:
        net = FullyConnectedNetwork([20, 64, 48, 6], ["relu", "tanh", "linear"], use_layer_norm=True, normalize_output=True)
    finally:
        set_default_precision(L.PREC_F32)
    lns = net.layer_norms()
    assert all(ln is not None for ln in lns) and [k for k, _ in net.named_parameters()][:4] == [
        "dnn.0.0.weight", "dnn.0.0.bias", "dnn.0.1.weight", "dnn.0.1.bias"]
    with torch.no_grad():
        for ln in lns:
            ln.weight.uniform_(0.5, 1.5)
            ln.bias.normal_(0, 0.2)
        for l in net.linears():
            l.bias.normal_(0, 0.1)
    ref = torch.nn.Sequential(*[torch.nn.Sequential(torch.nn.Linear(l.in_features, l.out_features), torch.nn.LayerNorm(l.out_features),
                                                    {"relu": torch.nn.ReLU(), "tanh": torch.nn.Tanh(), "linear": torch.nn.Identity()}[a])
                                for l, a in zip(net.linears(), net.activation_names)])
    ref.load_state_dict({k.replace("dnn.", ""): v for k, v in net.state_dict().items()})
    gen = torch.Generator().manual_seed(8)
    x, dout = torch.randn(50, 20, generator=gen), torch.randn(50, 6, generator=gen) / 50
    xr = x.clone().requires_grad_()
    out_ref = ref(xr)
    out_ref.backward(dout)
    net = net.to(backend.device)
    st = net.stack()
    assert isinstance(st, FCStack)
    st.set_need_input_grad(True)
    st.stage_weights(need_transposed=True)
    xc, xt = st.stage_input(x.to(backend.device), need_transposed=True)
    out = torch.empty(50, 6, device=backend.device)
    st.forward(xc, out, save=True)
    scale = lambda t: max(1.0, t.abs().max().item())  # noqa: E731
    assert (out.cpu() - out_ref.detach()).abs().max() <= tol * scale(out_ref)
    lin = net.linears()
    dw, db = [torch.empty_like(l.weight) for l in lin], [torch.empty_like(l.bias) for l in lin]
    dln = [(torch.empty_like(ln.weight), torch.empty_like(ln.bias)) for ln in net.layer_norms()]
    st.bind_ln_grads(dln)
    dx = torch.empty(50, 20, device=backend.device)
    st.backward(dout.to(backend.device), xt, dw, db, dx32=dx)
    gtol = tol if precision == L.PREC_F32 else 0.15
    for i, m in enumerate(ref):
        for got, want in ((dw[i], m[0].weight.grad), (db[i], m[0].bias.grad), (dln[i][0], m[1].weight.grad), (dln[i][1], m[1].bias.grad)):
            assert (got.cpu() - want).norm() <= gtol * want.norm() + 1e-7, i
    assert (dx.cpu() - xr.grad).norm() <= gtol * xr.grad.norm()


def _lightning_step(tr, opts, batch):
    losses = []
    for i, opt in enumerate(opts):
        loss = tr.training_step(batch, 0, i)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    return losses


@pytest.mark.parametrize("path", ["generator", "native"])
def test_dqn_with_layer_norm_matches_reference(backend, path):
    from golden_util import Golden
    from reagent_amd import synthetic
    from reagent_amd.core.parameters import EvaluationParameters, RLParameters
    from reagent_amd.models import FullyConnectedDQN
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.training import DQNTrainer

    g = Golden("dqn_layernorm")
    c = g.cfg
    q = FullyConnectedDQN(c["state_dim"], c["num_actions"], c["sizes"], c["activations"], use_layer_norm=True)
    inits = g.seq("init_param_")
    assert [tuple(p.shape) for p in q.parameters()] == [tuple(t.shape) for t in inits]  # Linear, LayerNorm, ... in module order
    with torch.no_grad():
        for p, init in zip(q.parameters(), inits):
            p.copy_(init)
    q = q.to(backend.device)
    tr = DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(c["num_actions"])], rl=RLParameters(**c["rl"]),
                    double_q_learning=c["double_q"], optimizer=Optimizer__Union.default(lr=c["lr"]),
                    evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(backend.device)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    for s in range(c["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        def grads_match():  # LayerNorm gamma / beta included (parameter order = module order)
            for i, p in enumerate(tr.q_network.parameters()):
                ref = g.t(f"step{s}_grad_{i}")
                assert (p.grad.cpu() - ref).abs().max() <= 3e-5 * max(1.0, ref.abs().max().item()), i

        if path == "generator":
            loss = tr.training_step(batch, 0, 0)
            opts[0].zero_grad()
            loss.backward()
            if s == 0:
                grads_match()
            opts[0].step()
            opts[1].zero_grad()
            tr.training_step(batch, 0, 1).backward()
            opts[1].step()
        else:
            loss = tr.train_step_native(batch)
            if s == 0:
                grads_match()
        ref_loss = g.t(f"step{s}_loss")
        assert abs(loss.item() - ref_loss.item()) <= 1e-4 * abs(ref_loss.item()) + 1e-6
        assert (tr.all_action_scores.cpu() - g.t(f"step{s}_q")).abs().max() <= 1e-4
        for i, p in enumerate(tr.q_network.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_param_{i}")).abs().max() <= 2e-5, (s, i)
        for i, p in enumerate(tr.q_network_target.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_target_{i}")).abs().max() <= 2e-5, (s, i)


@pytest.mark.parametrize("path", ["generator", "native"])
@pytest.mark.parametrize("name", ["sac_ln_critics", "sac_ln_actor"])
def test_sac_with_layer_normed_networks_matches_reference(backend, path, name):
    from golden_util import Golden
    from reagent_amd import synthetic
    from reagent_amd.core.parameters import RLParameters
    from reagent_amd.models import FullyConnectedCritic, GaussianFullyConnectedActor
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.training import SACTrainer
    from test_sac_trainer import check

    g = Golden(name)
    c = g.cfg
    S, A = c["state_dim"], c["action_dim"]
    ln_a, ln_c = c.get("actor_layer_norm", False), c.get("critic_layer_norm", False)
    actor = GaussianFullyConnectedActor(S, A, c["sizes"], c["activations"], use_layer_norm=ln_a)
    q1 = FullyConnectedCritic(S, A, c["sizes"], c["activations"], use_layer_norm=ln_c)
    q2 = FullyConnectedCritic(S, A, c["sizes"], c["activations"], use_layer_norm=ln_c)
    if ln_a:  # FC stack's LayerNorms, then loc_layer_norm / scale_layer_norm: the reference's parameter order
        assert [k for k, _ in actor.named_parameters()][-4:] == ["loc_layer_norm.weight", "loc_layer_norm.bias",
                                                                "scale_layer_norm.weight", "scale_layer_norm.bias"]
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
        check(tr, g, s)
