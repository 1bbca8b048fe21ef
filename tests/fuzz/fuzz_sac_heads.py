import os, sys, random, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import emu_backend
emu_backend.install()
from conftest import Backend
import test_sac_heads as T

# the Gaussian policy head (forward, log-prob of a given action, backward) on random (batch, action_dim) — rows of one
# element, widths around the wavefront and beyond one workgroup — through the checks of tests/test_sac_heads.py
# (float64 autograd of the reference's formulas, actor.py:166-261)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
random.seed(seed)
bad = 0
for case in range(cases):
    B = random.choice([2, 3, 63, 64, 65, 257, 1000])  # (the test plants its clamp probes in rows 0 and 1)
    A = random.choice([2, 3, 16, 31, 32, 33, 64, 65, 128, 255, 256, 257, 300])
    try:
        T.test_gaussian_head_forward_backward(Backend("emu", "cpu"), B, A)
        print("OK ", B, A)
    except Exception:
        bad += 1
        print("BAD", B, A)
        traceback.print_exc(limit=2)

# the SAC loss heads (critic targets / losses / gradients, actor loss with min(q1, q2) and its tie split, the temperature
# gradient) at batch sizes around the workgroup and partial-sum boundaries — sac_trainer.py:230-340 under torch autograd
import numpy as np
import torch
from reagent_amd import ops

for case in range(cases):
    B = random.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 4097])
    g = torch.Generator().manual_seed(seed * 100 + case)
    f = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    q1, q2, q1t, q2t, lpn, lp = f(B), f(B), f(B), f(B), f(B) * 2, f(B) * 2
    r, nt = torch.rand(B, generator=g), (torch.rand(B, generator=g) > 0.1).float()
    gamma, a0 = random.choice([0.0, 0.9, 0.99]), random.choice([1e-4, 0.0123, 0.7])
    alpha = torch.tensor([a0], dtype=torch.float64)
    P = ops.sac_partials(B)
    tgt, dq1, dq2 = (torch.zeros(B) for _ in range(3))
    l1, l2 = torch.zeros(P), torch.zeros(P)
    ops.sac_critic_head(q1, q2, q1t, q2t, lpn, r, nt, gamma, alpha, tgt, dq1, dq2, l1, l2)
    v = torch.min(q1t, q2t) - (alpha * lpn.clamp(-2, 2).double()).float()
    y = r + gamma * v * nt
    ok = bool((tgt - y).abs().max() <= 2e-6 * max(1.0, y.abs().max().item()))
    ok &= abs(l1.sum().item() / B - torch.nn.functional.mse_loss(q1, y).item()) <= 2e-5 * max(1.0, torch.nn.functional.mse_loss(q1, y).item())
    ok &= abs(l2.sum().item() / B - torch.nn.functional.mse_loss(q2, y).item()) <= 2e-5 * max(1.0, torch.nn.functional.mse_loss(q2, y).item())
    # (y is re-derived here in another rounding order: two fp32 ulps of 2 (q - y) / B are within the harness, not a kernel matter)
    ok &= bool((dq1 - 2 * (q1 - y) / B).abs().max() <= 2e-6 / B + 2e-9) and bool((dq2 - 2 * (q2 - y) / B).abs().max() <= 2e-6 / B + 2e-9)
    q1a, q2a = f(B), f(B)
    q2a[: min(5, B)] = q1a[: min(5, B)]  # ties split the gradient like torch.minimum
    target_entropy = random.choice([-1.0, -3.5])
    glp, d1, d2 = (torch.zeros(B) for _ in range(3))
    lpart, epart = torch.zeros(P), torch.zeros(P)
    ops.sac_actor_head(lp, q1a, q2a, alpha, target_entropy, glp, d1, d2, lpart, epart)
    lpr, q1r, q2r = lp.clone().requires_grad_(True), q1a.clone().requires_grad_(True), q2a.clone().requires_grad_(True)
    loss = (alpha.float() * lpr.clamp(-2, 2) - torch.min(q1r, q2r)).mean()
    loss.backward()
    ok &= abs(lpart.sum().item() / B - loss.item()) <= 2e-5 * max(1.0, abs(loss.item()))
    ok &= bool((glp - lpr.grad).abs().max() <= 1e-7 / B + 1e-10) and bool((d1 - q1r.grad).abs().max() <= 1e-7) and bool((d2 - q2r.grad).abs().max() <= 1e-7)
    log_alpha = torch.tensor([np.log(a0)], dtype=torch.float64)
    grad, al = torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)
    ops.sac_alpha_grad(epart, B, log_alpha, grad, al)
    m = (lp.clamp(-2, 2) + target_entropy).double().mean()
    ok &= abs(grad.item() + m.item()) <= 2e-6 and abs(al.item() + (log_alpha * m).item()) <= 2e-5
    print("OK " if ok else "BAD", "loss heads", dict(B=B, gamma=gamma, alpha=a0, target_entropy=target_entropy))
    bad += 0 if ok else 1
print("bad cases:", bad)
sys.exit(1 if bad else 0)
