"""SAC elementwise kernels against torch autograd of the reference formulas
(oracle/restated.py::GaussianActorOracle follows reagent/models/actor.py:166-261)."""
import numpy as np
import pytest
import torch

from oracle import restated as R
from reagent_amd import ops


def _head_ref(ls, noise, A):
    pi = R.GaussianActorOracle(["linear"], A)
    loc, sl = ls[:, :A], ls[:, A:].clamp(-2, 2)
    raw = loc + noise * sl.exp()
    a = pi.squash(raw)
    raw2 = torch.atanh(a)
    r = (raw2 - loc) / sl.exp()
    lp = torch.sum((-(r**2) / 2 - sl - pi.const) - (1 - a**2 + pi.eps).log(), dim=1)
    return a, lp


@pytest.mark.parametrize("B,A", [(70, 2), (300, 32), (33, 5), (1000, 3), (40, 64), (5, 300)])  # 300 > one workgroup: row-per-thread kernels
def test_gaussian_head_forward_backward(backend, B, A):
    g = torch.Generator().manual_seed(B)
    ls = torch.randn(B, 2 * A, generator=g) * 1.5
    ls[0, A] = 5.0     # scale_log clamp (upper) -> zero gradient
    ls[1, A + 1] = -7.0  # scale_log clamp (lower)
    noise = torch.randn(B, A, generator=g)
    ga = torch.randn(B, A, generator=g)
    glp = torch.randn(B, generator=g)
    dev = backend.device
    act = torch.zeros(B, A, device=dev)
    lp = torch.zeros(B, device=dev)
    sm = torch.zeros(B, A, device=dev)
    ops.gaussian_head_forward(ls.to(dev), noise.to(dev), act, lp, sm)
    lsr = ls.clone().double().requires_grad_(True)
    a_ref, lp_ref = _head_ref(lsr, noise.double(), A)
    a32, lp32 = _head_ref(ls, noise, A)  # what the reference computes (fp32)
    assert (act.cpu().double() - a_ref.detach()).abs().max() <= 2e-6
    # near tanh saturation (|a| -> 1 - 1e-6) fp32 itself is only good to ~1e-2 in log(1 - a^2 + eps):
    # require fp32-class agreement with the fp32 reference there, tight agreement elsewhere
    sat = (a32.abs() > 0.999).any(dim=1)
    if (~sat).any():  # wide rows: some element saturates in every row
        assert (lp.cpu() - lp32)[~sat].abs().max() <= 2e-5 * max(1.0, lp32.abs().max().item())
    assert (lp.cpu() - lp32)[sat].abs().max() <= 2e-2 if sat.any() else True
    assert (sm.cpu() - torch.clamp(torch.tanh(ls[:, :A]), -1 + 1e-6, 1 - 1e-6)).abs().max() <= 2e-6
    # log-prob of a given action equals the forward's own log-prob (reagent/test/models/test_actor.py:162-178)
    lp2 = torch.zeros(B, device=dev)
    ops.gaussian_log_prob(ls.to(dev), act, lp2)
    assert (lp2 - lp).abs().max() <= 1e-5 * max(1.0, lp.abs().max().item())
    # backward vs autograd
    ((a_ref * ga.double()).sum() + (lp_ref * glp.double()).sum()).backward()
    d = torch.zeros(B, 2 * A, device=dev)
    ops.gaussian_head_backward(ls.to(dev), noise.to(dev), ga.to(dev), glp.to(dev), d)
    ref = lsr.grad
    err = (d.cpu().double() - ref).abs()
    tol = 2e-4 * (1.0 + ref.abs())
    ok_rows = ~sat
    assert (err[ok_rows] <= tol[ok_rows]).all(), (err[ok_rows] / tol[ok_rows]).max() if ok_rows.any() else 0
    assert d[0, A].item() == 0.0 and d[1, A + 1].item() == 0.0


def test_sac_loss_heads(backend):
    B = 700
    g = torch.Generator().manual_seed(0)
    dev = backend.device
    f = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    q1, q2, q1t, q2t = f(B), f(B), f(B), f(B)
    lpn, lp = f(B) * 2, f(B) * 2
    r, nt = torch.rand(B, generator=g), (torch.rand(B, generator=g) > 0.1).float()
    alpha = torch.tensor([0.0123], dtype=torch.float64)
    P = ops.sac_partials(B)
    tgt, dq1, dq2 = (torch.zeros(B, device=dev) for _ in range(3))
    l1, l2 = torch.zeros(P, device=dev), torch.zeros(P, device=dev)
    ops.sac_critic_head(q1.to(dev), q2.to(dev), q1t.to(dev), q2t.to(dev), lpn.to(dev), r.to(dev), nt.to(dev), 0.99,
                        alpha.to(dev), tgt, dq1, dq2, l1, l2)
    v = torch.min(q1t, q2t) - (alpha * lpn.clamp(-2, 2).double()).float()
    y = r + 0.99 * v * nt
    assert (tgt.cpu() - y).abs().max() <= 1e-6
    assert abs(l1.sum().item() / B - torch.nn.functional.mse_loss(q1, y).item()) <= 1e-5
    assert abs(l2.sum().item() / B - torch.nn.functional.mse_loss(q2, y).item()) <= 1e-5
    assert (dq1.cpu() - 2 * (q1 - y) / B).abs().max() <= 1e-8
    # actor + temperature heads
    q1a, q2a = f(B), f(B)
    q2a[:5] = q1a[:5]  # ties split the gradient like torch.minimum
    glp, d1, d2 = (torch.zeros(B, device=dev) for _ in range(3))
    lpart, epart = torch.zeros(P, device=dev), torch.zeros(P, device=dev)
    ops.sac_actor_head(lp.to(dev), q1a.to(dev), q2a.to(dev), alpha.to(dev), -1.0, glp, d1, d2, lpart, epart)
    lpr, q1r, q2r = lp.clone().requires_grad_(True), q1a.clone().requires_grad_(True), q2a.clone().requires_grad_(True)
    loss = (alpha.float() * lpr.clamp(-2, 2) - torch.min(q1r, q2r)).mean()
    loss.backward()
    assert abs(lpart.sum().item() / B - loss.item()) <= 1e-5
    assert (glp.cpu() - lpr.grad).abs().max() <= 1e-8
    assert (d1.cpu() - q1r.grad).abs().max() <= 1e-8 and (d2.cpu() - q2r.grad).abs().max() <= 1e-8
    log_alpha = torch.tensor([np.log(0.0123)], dtype=torch.float64)
    grad, al = torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
    ops.sac_alpha_grad(epart, B, log_alpha.to(dev), grad, al)
    m = (lp.clamp(-2, 2) + (-1.0)).double().mean()
    assert abs(grad.item() + m.item()) <= 1e-6 and abs(al.item() + (log_alpha * m).item()) <= 1e-6
    # fp64 Adam on log_alpha vs torch.optim.Adam
    p = torch.nn.Parameter(log_alpha.clone())
    opt = torch.optim.Adam([p], lr=3e-3)
    pd, md, vd = log_alpha.clone().to(dev), torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
    ex = torch.zeros(1, dtype=torch.float64, device=dev)
    for step in range(1, 4):
        p.grad = grad.cpu().clone() * step
        opt.step()
        ops.adam_step_f64(pd, grad * step, md, vd, 3e-3, 0.9, 0.999, 1e-8, 1 - 0.9**step, (1 - 0.999**step) ** 0.5, ex)
        assert abs(pd.item() - p.item()) <= 1e-12 and abs(ex.item() - p.detach().exp().item()) <= 1e-12
