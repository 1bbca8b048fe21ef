"""FusedAdam (rg_adam_step) against torch.optim.Adam — the arithmetic the reference delegates to
(reagent/optimizer/uninferrable_optimizers.py:23-33) — and SoftUpdate (rg_soft_update) against the
formula of reagent/optimizer/soft_update.py:60-70.  Neither is pinned by a reference test
("parity unpinned"): tolerance = a few fp32 ulps over 5 steps."""
import pytest
import torch

from reagent_amd.optimizer import FusedAdam, Optimizer__Union, SoftUpdate


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_matches_torch_adam(backend, wd):
    g = torch.Generator().manual_seed(0)
    shapes = [(33, 7), (33,), (5, 33), (5,)]
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).to(backend.device)) for s in shapes]
    rs = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    fused = FusedAdam(ps, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    ref = torch.optim.Adam(rs, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    for step in range(5):
        for p, r in zip(ps, rs):
            gr = torch.randn(*r.shape, generator=g) * (10.0 ** (step - 3))
            p.grad, r.grad = gr.to(backend.device), gr.clone()
        fused.step()
        ref.step()
        for p, r in zip(ps, rs):
            assert (p.detach().cpu() - r.detach()).abs().max() <= 2e-7 + 2e-6 * r.detach().abs().max()
    for p, r in zip(ps, rs):
        assert int(fused.state[p]["step"]) == 5
        assert (fused.state[p]["exp_avg"].cpu() - ref.state[r]["exp_avg"]).abs().max() <= 1e-6 * ref.state[r]["exp_avg"].abs().max() + 1e-12
        assert (fused.state[p]["exp_avg_sq"].cpu() - ref.state[r]["exp_avg_sq"]).abs().max() <= 1e-6 * ref.state[r]["exp_avg_sq"].abs().max() + 1e-12
    # parameters stay views of ONE flat slab (what the RCCL all-reduce and the kernels see)
    slab = fused.slab_for(0)
    assert slab.is_bound() and all(p.is_contiguous() for p in ps)
    assert "exp_avg" in fused.state_dict()["state"][0]


def test_params_without_grad_are_skipped_like_torch(backend):
    ps = [torch.nn.Parameter(torch.ones(4, 4).to(backend.device)), torch.nn.Parameter(torch.ones(3).to(backend.device))]
    opt = FusedAdam(ps, lr=0.1)
    ps[1].grad = torch.ones(3).to(backend.device)
    opt.step()
    assert torch.equal(ps[0].detach().cpu(), torch.ones(4, 4))
    assert (ps[1].detach().cpu() - 0.9).abs().max() < 1e-6


def test_soft_update(backend):
    g = torch.Generator().manual_seed(1)
    src = [torch.nn.Parameter(torch.randn(9, 4, generator=g).to(backend.device)), torch.nn.Parameter(torch.randn(9, generator=g).to(backend.device))]
    tgt = [torch.nn.Parameter(torch.randn(9, 4, generator=g).to(backend.device)), torch.nn.Parameter(torch.randn(9, generator=g).to(backend.device))]
    s0, t0 = [p.detach().cpu().clone() for p in src], [p.detach().cpu().clone() for p in tgt]
    su = SoftUpdate.make_optimizer_scheduler(tgt, src, tau=0.3)["optimizer"]
    su.step()
    for t, a, b in zip(tgt, s0, t0):
        assert (t.detach().cpu() - (0.3 * a + (1.0 - 0.3) * b)).abs().max() <= 1e-7
    # slab-resident sources (after an Adam step) take the single-launch path; same numbers
    adam = FusedAdam(src, lr=0.0)
    for p in src:
        p.grad = torch.zeros_like(p)
    adam.step()
    t1 = [p.detach().cpu().clone() for p in tgt]
    su.step()
    for t, a, b in zip(tgt, s0, t1):
        assert (t.detach().cpu() - (0.3 * a + (1.0 - 0.3) * b)).abs().max() <= 1e-7
    with pytest.raises(ValueError, match="tau should be in"):
        SoftUpdate(tgt, src, tau=1.5)
    with pytest.raises(ValueError, match="same number of parameters"):
        SoftUpdate(tgt, src[:1], tau=0.5)


def test_optimizer_union_default_is_adam(backend):
    p = [torch.nn.Parameter(torch.zeros(2, 2).to(backend.device))]
    o = Optimizer__Union.default().make_optimizer_scheduler(p)
    assert set(o) == {"optimizer"} and isinstance(o["optimizer"], FusedAdam)
    assert o["optimizer"].defaults["lr"] == 0.001 and o["optimizer"].defaults["betas"] == (0.9, 0.999)


def test_lr_scheduler_drives_the_fused_adam(backend):
    """Optimizer__Union(Adam=Adam(lr_schedulers=[StepLR(...)])) as in reagent/optimizer/optimizer.py:61-85:
    the torch scheduler object changes group["lr"], which the fused kernel reads at every step — the
    parameters follow torch.optim.Adam + the same scheduler bit for bit"""
    from reagent_amd.optimizer import Adam, Optimizer__Union, StepLR

    dev = backend.device
    torch.manual_seed(0)
    p_ref = torch.nn.Parameter(torch.randn(300))
    p_hip = torch.nn.Parameter(p_ref.detach().clone().to(dev))
    made = Optimizer__Union(Adam=Adam(lr=0.05, lr_schedulers=[StepLR(step_size=2, gamma=0.5)])).make_optimizer_scheduler([p_hip])
    opt, sched = made["optimizer"], made["lr_scheduler"]
    ref_opt = torch.optim.Adam([p_ref], lr=0.05)
    ref_sched = torch.optim.lr_scheduler.StepLR(ref_opt, step_size=2, gamma=0.5)
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        grad = torch.randn(300, generator=g)
        p_ref.grad, p_hip.grad = grad.clone(), grad.clone().to(dev)
        ref_opt.step()
        opt.step()
        ref_sched.step()
        sched.step()
        assert opt.param_groups[0]["lr"] == ref_opt.param_groups[0]["lr"]
    assert opt.param_groups[0]["lr"] == 0.05 * 0.5 ** 3
    assert (p_hip.detach().cpu() - p_ref.detach()).abs().max() <= 1e-6
    assert "lr_scheduler" not in Optimizer__Union.default(lr=0.1).make_optimizer_scheduler([torch.nn.Parameter(torch.zeros(4, device=dev))])


def _grads(shapes, g, scale=1.0):
    return [torch.randn(*s, generator=g) * scale for s in shapes]


def test_checkpoint_resume_continues_like_torch_adam(backend):
    """3 steps, state_dict() -> a FRESH FusedAdam over fresh parameters -> load_state_dict(), 3 more steps:
    identical to 6 uninterrupted torch.optim.Adam steps; the state_dict has torch.optim.Adam's layout in
    both directions (a torch.optim.Adam state_dict loads into FusedAdam)"""
    g = torch.Generator().manual_seed(3)
    shapes = [(17, 5), (17,), (3, 17), (3,)]
    init = [torch.randn(*s, generator=g) for s in shapes]
    rs = [torch.nn.Parameter(w.clone()) for w in init]
    ref = torch.optim.Adam(rs, lr=0.01)
    ps = [torch.nn.Parameter(w.clone().to(backend.device)) for w in init]
    fused = FusedAdam(ps, lr=0.01)
    for step in range(3):
        gr = _grads(shapes, g)
        for p, r, x in zip(ps, rs, gr):
            p.grad, r.grad = x.to(backend.device), x.clone()
        fused.step()
        ref.step()
    sd = fused.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}  # torch.optim.Adam's keys, nothing private
    # resume A: our own state_dict into a fresh optimizer over fresh parameter objects
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    fused2 = FusedAdam(ps2, lr=0.01)
    fused2.load_state_dict(sd)
    # resume B: torch.optim.Adam's state_dict into a fresh FusedAdam
    ps3 = [torch.nn.Parameter(r.detach().clone().to(backend.device)) for r in rs]
    fused3 = FusedAdam(ps3, lr=0.01)
    fused3.load_state_dict(ref.state_dict())
    for step in range(3):
        gr = _grads(shapes, g)
        for p2, p3, r, x in zip(ps2, ps3, rs, gr):
            p2.grad, p3.grad, r.grad = x.to(backend.device), x.to(backend.device), x.clone()
        fused2.step()
        fused3.step()
        ref.step()
    for p2, p3, r in zip(ps2, ps3, rs):
        assert (p2.detach().cpu() - r.detach()).abs().max() <= 2e-6, "resumed run diverged from torch.optim.Adam"
        assert (p3.detach().cpu() - r.detach()).abs().max() <= 2e-6
        assert int(fused2.state[p2]["step"]) == 6 and int(fused3.state[p3]["step"]) == 6
        # state[p] stays a view of the flat buffer the kernel updates
        m = fused2.moments_for(0)[1]
        assert fused2.state[p2]["exp_avg"].data_ptr() >= m.data_ptr()
        assert (fused2.state[p2]["exp_avg"].cpu() - ref.state[r]["exp_avg"]).abs().max() <= 1e-6


def test_a_second_optimizer_over_the_same_parameters_starts_from_zero_moments(backend):
    """configure_optimizers() may be called more than once per network: every FusedAdam owns its moments"""
    g = torch.Generator().manual_seed(4)
    shapes = [(9, 4), (9,)]
    init = [torch.randn(*s, generator=g) for s in shapes]
    ps = [torch.nn.Parameter(w.clone().to(backend.device)) for w in init]
    first = FusedAdam(ps, lr=0.01)
    for _ in range(3):
        for p, x in zip(ps, _grads(shapes, g)):
            p.grad = x.to(backend.device)
        first.step()
    rs = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    ref = torch.optim.Adam(rs, lr=0.01)  # a fresh torch optimizer at the same point
    second = FusedAdam(ps, lr=0.01)
    gr = _grads(shapes, g)
    for p, r, x in zip(ps, rs, gr):
        p.grad, r.grad = x.to(backend.device), x.clone()
    second.step()
    ref.step()
    for p, r in zip(ps, rs):
        assert (p.detach().cpu() - r.detach()).abs().max() <= 1e-6
    assert float(first.state[ps[0]]["exp_avg"].abs().max()) > 0  # and the first one kept its own
    assert first.moments_for(0)[1].data_ptr() != second.moments_for(0)[1].data_ptr()


# ---- the other members of Optimizer__Union (missing 5) -------------------------------------------------------------
@pytest.mark.parametrize("name,kwargs", [
    ("SGD", dict(lr=0.05, momentum=0.9, nesterov=True)),
    ("RMSprop", dict(lr=0.01, momentum=0.5, centered=True)),
    ("AdamW", dict(lr=0.003, weight_decay=0.05)),
    ("Adagrad", dict(lr=0.05)),
])
def test_union_members_besides_adam_step_like_torch(backend, name, kwargs):
    """Optimizer__Union(<member>=...) as in reagent/optimizer/union.py: the member's torch.optim class over the trainer's
    parameters.  Three native DQN steps (exact-fp32 engine) against the same network stepped on the CPU by torch autograd and
    the same torch optimizer: the gradients come from the HIP backward, the update is torch's own arithmetic."""
    import reagent_amd.optimizer as O
    from oracle import restated as R
    from reagent_amd import synthetic
    from reagent_amd.core.parameters import EvaluationParameters, RLParameters
    from reagent_amd.models import FullyConnectedDQN
    from reagent_amd.training import DQNTrainer

    dev = backend.device
    S, A, B = 12, 4, 64
    torch.manual_seed(3)
    q = FullyConnectedDQN(S, A, [32, 24], ["relu", "relu"])
    init = [p.detach().clone() for p in q.parameters()]
    q = q.to(dev)
    union = O.Optimizer__Union(**{name: getattr(O, name)(**kwargs)})
    assert union.selected_field == name and type(union.value).__name__ == name
    tr = DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(A)],
                    rl=RLParameters(gamma=0.9, target_update_rate=0.1, q_network_loss="mse"), optimizer=union,
                    evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)
    assert isinstance(tr.native_optimizers()[0], getattr(torch.optim, name))
    o = R.DQNOracle(init, init, ["relu", "relu", "linear"], gamma=0.9, tau=0.1, loss="mse")
    o.opt = getattr(torch.optim, name)(o.params, **kwargs)  # the reference's arithmetic for this member IS torch's
    for s in range(3):
        b = synthetic.dqn_batch(B, S, A, seed=30 + s, p_impossible=0.2)
        loss = tr.train_step_native(synthetic.to_dqn_input(b, dev))
        ref = o.step(b)
        assert abs(loss.item() - ref["loss"].item()) <= 1e-4 * max(1.0, abs(ref["loss"].item())), (name, s)
    for p, r in zip(tr.q_network.parameters(), o.params):
        assert (p.detach().cpu() - r.detach()).abs().max() <= 2e-5, name
    for p, r in zip(tr.q_network_target.parameters(), o.target):
        assert (p.detach().cpu() - r).abs().max() <= 2e-5, name


def test_union_rejects_what_it_cannot_serve():
    import reagent_amd.optimizer as O

    with pytest.raises(ValueError, match="exactly one member"):
        O.Optimizer__Union(Adam=O.Adam(), SGD=O.SGD())
    with pytest.raises(ValueError, match="no member"):
        O.Optimizer__Union(Lion=object())
    with pytest.raises(NotImplementedError, match="closure"):
        O.Optimizer__Union(LBFGS=object())
    assert isinstance(O.Optimizer__Union.default(lr=0.01).value, O.Adam) and O.Optimizer__Union().selected_field == "Adam"
    made = O.Optimizer__Union(SGD=O.SGD(lr=0.1, lr_schedulers=[O.StepLR(step_size=2)])).make_optimizer_scheduler(
        [torch.nn.Parameter(torch.zeros(3))])
    assert isinstance(made["optimizer"], torch.optim.SGD) and isinstance(made["lr_scheduler"], torch.optim.lr_scheduler.StepLR)


def test_union_members_take_the_reference_field_lists():
    """uninferrable_optimizers.py:23-114: every field a reference config or YAML may pass is accepted; a field the installed
    torch.optim class lacks is dropped only at its default (never a silent `maximize=True`); the fused Adam refuses the two
    fields that would change its arithmetic"""
    import dataclasses

    import reagent_amd.optimizer as O

    want = dict(Adam={"maximize", "foreach", "capturable", "differentiable"}, NAdam={"maximize", "foreach", "momentum_decay"},
                RAdam={"maximize", "foreach"}, SGD={"maximize", "foreach", "differentiable", "nesterov", "dampening"},
                AdamW={"maximize", "foreach", "capturable", "amsgrad"}, Adamax={"maximize", "foreach"},
                Rprop={"maximize", "foreach", "etas", "step_sizes"})
    for name, fields in want.items():
        have = {f.name for f in dataclasses.fields(getattr(O, name))}
        assert fields <= have, (name, fields - have)
    p = [torch.nn.Parameter(torch.zeros(3))]
    opt = O.SGD(lr=0.1, maximize=True, foreach=False).make_optimizer_scheduler(p)["optimizer"]
    assert opt.defaults["maximize"] is True and opt.defaults["foreach"] is False
    assert isinstance(O.Adam(foreach=True, capturable=True).make_optimizer_scheduler(p)["optimizer"], O.FusedAdam)
    with pytest.raises(NotImplementedError, match="maximize"):
        O.Adam(maximize=True).make_optimizer_scheduler(p)
    Odd = O._torch_config("SGD", lr=0.1, not_a_torch_argument=False)
    assert isinstance(Odd().make_optimizer_scheduler(p)["optimizer"], torch.optim.SGD)  # at its default: dropped
    with pytest.raises(TypeError, match="not_a_torch_argument"):
        Odd(not_a_torch_argument=True).make_optimizer_scheduler(p)


def test_sched_tick_many_ticks_each_schedule_once(backend):
    """rg_sched_tick_many (ABI 11): n distinct device schedules, [0] += 1 each, one launch; a repeated schedule or n > 8 is refused;
    ops.deferred_ticks collects a scope's ticks, flushes before a second tick of the same schedule and at 8"""
    import ctypes

    import reagent_amd._lib as L
    from reagent_amd import ops

    dev = backend.device
    sch = [torch.tensor([float(i), 1e-3, 0.0, 0.0], dtype=torch.float64, device=dev) for i in range(10)]
    lib = L.lib()
    arr = (ctypes.c_void_p * 3)(*[s.data_ptr() for s in sch[:3]])
    assert lib.rg_sched_tick_many(arr, 3, L.stream_ptr()) == 0
    assert [float(s[0]) for s in sch[:4]] == [1.0, 2.0, 3.0, 3.0]
    dup = (ctypes.c_void_p * 2)(sch[0].data_ptr(), sch[0].data_ptr())
    assert lib.rg_sched_tick_many(dup, 2, L.stream_ptr()) == -1  # RG_EINVAL
    nine = (ctypes.c_void_p * 9)(*[s.data_ptr() for s in sch[:9]])
    assert lib.rg_sched_tick_many(nine, 9, L.stream_ptr()) == -1 and lib.rg_sched_tick_many(nine, 0, L.stream_ptr()) == 0
    with ops.deferred_ticks():
        for s in sch:          # ten schedules: eight leave when the ninth arrives, two at the end of the scope
            ops.sched_tick(s)
        assert float(sch[9][0]) == 9.0 and float(sch[0][0]) == 2.0  # 0..7 already ticked, 8 and 9 still waiting
        ops.sched_tick(sch[9])  # a second tick of a waiting schedule flushes first
        ops.tick_fence(sch[9])  # ... and a reader of it sees both
        assert float(sch[9][0]) == 11.0 and float(sch[8][0]) == 9.0
    ops.sched_tick(sch[0])  # outside a scope: at once
    assert float(sch[0][0]) == 3.0
