import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/fuzz")
import torch
import torch.distributions as pyd
import torch.nn.functional as F
import gpu_ops  # FUZZ_ON_GPU=1: the real library on cuda:0 instead of the interpreter
ops = gpu_ops.select()

# the discrete-CRR heads on random (batch, actions): critic targets under softmax(actor(s')) with the twin minimum, both
# critics' losses and gradients; the actor's clamped exp-advantage weight, the clipped importance-ratio entropy term and
# d loss / d scores — discrete_crr_trainer.py:191-285 under torch autograd (float64)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
random.seed(seed)
bad = 0
for case in range(cases):
    gen = torch.Generator().manual_seed(seed * 1000 + case)
    B, A = random.choice([1, 2, 63, 64, 65, 256, 257, 1000]), random.choice([1, 2, 3, 5, 16, 17, 64])
    r = lambda *s: torch.randn(*s, generator=gen)  # noqa: E731
    twin, gamma = random.random() < 0.5, random.choice([0.0, 0.9, 1.0])
    q1, q2, q1n, q2n, nl = (r(B, A) * random.choice([0.3, 1.0, 5.0]) for _ in range(5))
    action = F.one_hot(torch.randint(A, (B,), generator=gen), A).float()
    reward, nt = torch.rand(B, generator=gen), (torch.rand(B, generator=gen) > 0.2).float()
    boosts = r(A) * 0.1 if random.random() < 0.5 else None
    D = lambda t: t.double()  # noqa: E731
    q1r, q2r = D(q1).requires_grad_(), D(q2).requires_grad_()
    probs = pyd.Categorical(logits=D(nl)).probs
    v = (D(q1n) * probs).sum(1, keepdim=True)
    if twin:
        v = torch.min(v, (D(q2n) * probs).sum(1, keepdim=True))
    rw = D(reward) + ((D(action) * D(boosts)).sum(1) if boosts is not None else 0)
    target = rw.unsqueeze(1) + gamma * v * D(nt).unsqueeze(1)
    l1 = F.mse_loss((q1r * D(action)).sum(1, keepdim=True), target)
    l2 = F.mse_loss((q2r * D(action)).sum(1, keepdim=True), target)
    l1.backward(); l2.backward()
    P = ops.crr_partials(B)
    tgt, dq1, dq2, p1, p2 = torch.empty(B), torch.empty(B, A), torch.empty(B, A), torch.empty(P), torch.empty(P)
    ops.crr_critic_head(q1, q2 if twin else None, q1n, q2n if twin else None, nl, action, reward, boosts, nt, gamma, tgt,
                        dq1, dq2 if twin else None, p1, p2 if twin else None)
    sc = max(1.0, target.abs().max().item())
    ok = bool((D(tgt) - target.squeeze(1)).abs().max() <= 3e-6 * sc) and abs(p1.sum().item() / B - l1.item()) <= 2e-5 * max(1.0, l1.item())
    # d loss / d q = 2 (q_sel - target) / B in fp32: the rounding scales with |q| + |target|, not with the target alone (seed 11's
    # last case: |q| up to 20 against targets below 1 — 1.4e-8 where 3e-6 * sc / B allowed 1.3e-8)
    scq = sc + max(q1.abs().max().item(), q2.abs().max().item())
    ok &= bool((D(dq1) - q1r.grad).abs().max() <= 3e-6 * scq / B + 1e-9)
    if twin:
        ok &= abs(p2.sum().item() / B - l2.item()) <= 2e-5 * max(1.0, l2.item()) and bool((D(dq2) - q2r.grad).abs().max() <= 3e-6 * scq / B + 1e-9)
    if not ok:  # which of the critic checks
        print("   critic: tgt err %.2e (allowed %.2e) | loss1 %.8f vs %.8f | dq1 err %.2e (allowed %.2e)%s" % (
            (D(tgt) - target.squeeze(1)).abs().max().item(), 3e-6 * sc, p1.sum().item() / B, l1.item(),
            (D(dq1) - q1r.grad).abs().max().item(), 3e-6 * scq / B + 1e-9,
            " | loss2 %.8f vs %.8f | dq2 err %.2e" % (p2.sum().item() / B, l2.item(), (D(dq2) - q2r.grad).abs().max().item()) if twin else ""))
    # ---- actor head
    beta, max_weight = random.choice([0.3, 0.6, 2.0]), random.choice([1.5, 2.5, 20.0])
    entropy_coeff, clip_limit = random.choice([0.0, 0.3]), random.choice([1.5, 10.0])
    q, z = r(B, A), r(B, A) * 2
    pi_b = 0.02 + 0.98 * torch.rand(B, generator=gen)
    zr = D(z).requires_grad_()
    dist = pyd.Categorical(logits=zr)
    values = (D(q) * dist.probs).sum(1, keepdim=True)
    weight = torch.clamp(((1 / beta) * ((D(q) - values) * D(action)).sum(1, keepdim=True)).exp(), 0, max_weight)
    idx = torch.argmax(action, dim=1, keepdim=True)
    log_pi = dist.log_prob(idx.squeeze(1)).unsqueeze(1)
    pi_t = (dist.probs * D(action)).sum(1, keepdim=True)
    entropy = torch.zeros((), dtype=torch.float64)
    if entropy_coeff > 0:
        entropy = (torch.clip(pi_t / D(pi_b).view(pi_t.shape), min=1e-4, max=clip_limit) * log_pi).mean()
    plain = (-log_pi * weight.detach()).mean()
    (plain + entropy_coeff * entropy).backward()
    dz, pp, pe = torch.empty(B, A), torch.empty(P), torch.empty(P)
    ops.crr_actor_head(q, z, action, pi_b if entropy_coeff > 0 else None, beta, max_weight, entropy_coeff, clip_limit, dz, pp,
                       pe if entropy_coeff > 0 else None)
    ok2 = abs(pp.sum().item() / B - plain.item()) <= 3e-5 * max(1.0, abs(plain.item()))
    if entropy_coeff > 0:
        ok2 &= abs(pe.sum().item() / B - entropy.item()) <= 3e-5 * max(1.0, abs(entropy.item()))
    ok2 &= bool((D(dz) - zr.grad).abs().max() <= 3e-6 * max(1.0, max_weight) / B + 1e-9)
    print(("OK " if ok and ok2 else "BAD"), dict(B=B, A=A, twin=twin, gamma=gamma, beta=beta, max_weight=max_weight, entropy=entropy_coeff, clip=clip_limit),
          "critic", ok, "actor", ok2, "dz err %.1e" % (D(dz) - zr.grad).abs().max().item())
    bad += 0 if ok and ok2 else 1
print("bad cases:", bad)
sys.exit(1 if bad else 0)
