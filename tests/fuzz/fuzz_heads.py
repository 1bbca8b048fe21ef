import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/fuzz")
import torch
import torch.nn.functional as F
import gpu_ops  # FUZZ_ON_GPU=1: the real library on cuda:0 instead of the interpreter
ops = gpu_ops.select()
import reagent_amd._lib as L

# rg_dqn_head on random shapes (batch sizes around the workgroup size, 1 .. 100 actions, masks with rows that allow a single
# action, terminal rows, n-step discount exponents, reward boosts, both losses, double-Q on / off) against the reference's
# formulas under torch autograd in float64 (dqn_trainer_base.py:33-77, dqn_trainer.py:179-239): the masked (double-Q) next
# value and its index, the mean loss, d loss / d Q.
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
random.seed(seed)
bad = 0
for case in range(cases):
    g = torch.Generator().manual_seed(seed * 1000 + case)
    B = random.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 4097])
    A = random.choice([1, 2, 3, 16, 17, 31, 32, 33, 64, 100])
    double_q, loss = random.random() < 0.5, random.choice(["mse", "huber"])
    gamma = random.choice([0.0, 0.9, 0.99, 1.0])
    q, qo, qt = (torch.randn(B, A, generator=g) * random.choice([0.1, 1.0, 30.0]) for _ in range(3))
    if random.random() < 0.3:  # exact ties between actions: the first maximal index wins (torch.max / argmax on CPU)
        qo[:, 1:] = qo[:, :1]
        qt[:, 1:] = qt[:, :1]
    act = F.one_hot(torch.randint(0, A, (B,), generator=g), A).float()
    mask = (torch.rand(B, A, generator=g) < random.choice([0.2, 0.7, 1.0])).float()
    mask[torch.arange(B), torch.randint(0, A, (B,), generator=g)] = 1.0  # at least one possible next action per row
    reward = torch.randn(B, generator=g)
    boosts = torch.randn(A, generator=g) if random.random() < 0.4 else None
    nt = (torch.rand(B, generator=g) < 0.8).float()
    gexp = torch.randint(1, 4, (B,), generator=g).float() if random.random() < 0.4 else None
    dq, parts = torch.empty(B, A), torch.empty(ops.dqn_head_partials(B))
    nq, ni, qs = torch.empty(B), torch.empty(B, dtype=torch.int64), torch.empty(B)
    ops.dqn_head(q.contiguous(), qo.contiguous(), qt.contiguous(), act, mask, reward, boosts, nt, gamma, gexp, double_q,
                 L.LOSS[loss], dq, parts, nq, ni, qs)
    # ---- the reference's formulas, float64
    qd = q.double().requires_grad_()
    on, tg = qo.double() + -1e9 * (1 - mask.double()), qt.double() + -1e9 * (1 - mask.double())
    if double_q:
        idx = on.argmax(dim=1, keepdim=True)
        nxt = tg.gather(1, idx)
    else:
        nxt, idx = tg.max(dim=1, keepdim=True)
    r = reward.double().reshape(-1, 1)
    if boosts is not None:
        r = r + (act.double() * boosts.double().reshape(1, -1)).sum(1, keepdim=True)
    disc = torch.full((B, 1), gamma, dtype=torch.float64) if gexp is None else torch.pow(torch.tensor(gamma, dtype=torch.float64), gexp.double().reshape(-1, 1))
    target = r + nt.double().reshape(-1, 1) * disc * nxt
    q_sel = (qd * act.double()).sum(1, keepdim=True)
    lref = F.mse_loss(q_sel, target.detach()) if loss == "mse" else F.smooth_l1_loss(q_sel, target.detach())
    lref.backward()
    # the fp32 kernel adds -1e9 to masked entries like the reference (fp32 there too): with every entry of a row masked but
    # one, ties cannot arise; with exact ties the FIRST index must win
    scale = max(1.0, q.abs().max().item(), qt.abs().max().item())
    got_loss = parts.double().sum().item() / B  # per-workgroup sums; the step's reduce launch takes the mean
    ok = torch.equal(ni, idx.reshape(-1)) and bool((nq.double() - nxt.reshape(-1)).abs().max() <= 1e-6 * scale)
    ok &= bool((qs.double() - q_sel.detach().reshape(-1)).abs().max() <= 1e-6 * scale)
    ok &= abs(got_loss - lref.item()) <= 2e-5 * max(1.0, abs(lref.item()))
    ok &= bool((dq.double() - qd.grad).abs().max() <= 2e-6 * max(1.0, qd.grad.abs().max().item()) + 1e-7 * scale)
    print(("OK " if ok else "BAD"), dict(B=B, A=A, double_q=double_q, loss=loss, gamma=gamma, boosts=boosts is not None, gexp=gexp is not None),
          "dq err %.2e" % (dq.double() - qd.grad).abs().max().item(), "loss", got_loss, lref.item())
    bad += 0 if ok else 1

# rg_qr_head (the dense QR-DQN head of the fp32 mode): the reference's (N, B, N) quantile-Huber pair loss, its gradient and the
# masked next-action choice on random (B, A, N) — qrdqn_trainer.py:108-160, :210-218 — in float64 under autograd
for case in range(max(1, cases // 2)):
    g = torch.Generator().manual_seed(seed * 1000 + 500 + case)
    B, A = random.choice([1, 2, 7, 64, 65, 130]), random.choice([1, 2, 3, 16, 17])
    N = random.choice([1, 2, 3, 7, 32, 51, 64, 200])
    double_q, maxq, gamma = random.random() < 0.5, random.random() < 0.7, random.choice([0.0, 0.9, 1.0])
    q, qo, qt = (torch.randn(B, A * N, generator=g) * random.choice([0.2, 1.0, 5.0]) for _ in range(3))
    if random.random() < 0.3:
        q = (q * 4).round() / 4  # exact ties between target and current quantiles (td == 0 sits on the indicator's edge)
        qt = (qt * 4).round() / 4
    act = F.one_hot(torch.randint(0, A, (B,), generator=g), A).float()
    if maxq:
        mask = (torch.rand(B, A, generator=g) < 0.6).float()
        mask[torch.arange(B), torch.randint(0, A, (B,), generator=g)] = 1.0
    else:
        mask = F.one_hot(torch.randint(0, A, (B,), generator=g), A).float()  # SARSA: the logged next action
    reward = (torch.randn(B, generator=g) * 4).round() / 4
    boosts = torch.randn(A, generator=g) if random.random() < 0.3 else None
    nt = (torch.rand(B, generator=g) < 0.8).float()
    gexp = torch.randint(1, 4, (B,), generator=g).float() if random.random() < 0.3 else None
    quant = ((0.5 + torch.arange(N)) / float(N)).float()
    dq, parts, allq = torch.empty(B, A * N), torch.empty(B), torch.empty(B, A)
    ops.qr_head(q.contiguous(), qo.contiguous() if double_q else None, qt.contiguous(), act, mask, reward, boosts, nt, gamma, gexp,
                quant, N, maxq, dq, parts, allq)
    qd = q.double().requires_grad_()
    cur3, on3, tg3 = qd.view(B, A, N), qo.double().view(B, A, N), qt.double().view(B, A, N)
    if maxq:
        sel = (on3 if double_q else tg3).mean(2) + -1e9 * (1 - mask.double())
        nxt = tg3[torch.arange(B), sel.argmax(1)]
    else:
        nxt = (tg3 * mask.double().unsqueeze(-1)).sum(1)
    r = reward.double().reshape(-1, 1)
    if boosts is not None:
        r = r + (act.double() * boosts.double().reshape(1, -1)).sum(1, keepdim=True)
    disc = torch.full((B, 1), gamma, dtype=torch.float64) if gexp is None else torch.pow(torch.tensor(gamma, dtype=torch.float64), gexp.double().reshape(-1, 1))
    target = (r + disc * nt.double().reshape(-1, 1) * nxt).detach()
    cur = (cur3 * act.double().unsqueeze(-1)).sum(1)
    td = target.t().unsqueeze(-1) - cur
    hub = torch.where(td.abs() < 1, 0.5 * td.pow(2), td.abs() - 0.5)
    lref = (hub * (quant.double() - (td.detach() < 0).double()).abs()).mean()
    lref.backward()
    gs = max(1e-30, qd.grad.abs().max().item())
    ok = abs(parts.double().sum().item() - lref.item()) <= 2e-5 * max(1.0, abs(lref.item()))
    ok &= bool((dq.double() - qd.grad).abs().max() <= 3e-5 * gs + 1e-9)
    ok &= bool((allq.double() - q.double().view(B, A, N).mean(2)).abs().max() <= 1e-5 * max(1.0, q.abs().max().item()))
    print(("OK " if ok else "BAD"), "qr", dict(B=B, A=A, N=N, double_q=double_q, maxq=maxq, gamma=gamma),
          "dq err %.2e of %.2e" % ((dq.double() - qd.grad).abs().max().item(), gs), "loss", parts.double().sum().item(), lref.item())
    bad += 0 if ok else 1

# rg_c51_head: softmax over atoms, masked next action by expected value, the categorical projection with the reference's
# l == b == u fix-ups (targets planted ON the support grid: reward 0 / terminal rows / gamma 1), cross-entropy and its logit
# gradient — c51_trainer.py:98-187 in float64 under autograd
for case in range(max(1, cases // 2)):
    g = torch.Generator().manual_seed(seed * 1000 + 800 + case)
    B, A = random.choice([1, 2, 7, 64, 65, 130]), random.choice([1, 2, 3, 16, 17])
    N = random.choice([2, 3, 7, 32, 51, 64, 200])
    double_q, maxq, gamma = random.random() < 0.5, random.random() < 0.7, random.choice([0.0, 0.5, 0.9, 1.0])
    qmin, qmax = random.choice([(-10.0, 10.0), (0.0, 5.0), (-100.0, 200.0)])
    q, qo, qt = (torch.randn(B, A * N, generator=g) * random.choice([0.2, 1.0, 4.0]) for _ in range(3))
    act = F.one_hot(torch.randint(0, A, (B,), generator=g), A).float()
    if maxq:
        mask = (torch.rand(B, A, generator=g) < 0.6).float()
        mask[torch.arange(B), torch.randint(0, A, (B,), generator=g)] = 1.0
    else:
        mask = F.one_hot(torch.randint(0, A, (B,), generator=g), A).float()
    reward = torch.randn(B, generator=g) * (qmax - qmin) / 4
    reward[torch.rand(B, generator=g) < 0.3] = 0.0  # with gamma 1 the target atoms sit exactly on the support
    boosts = torch.randn(A, generator=g) if random.random() < 0.3 else None
    nt = (torch.rand(B, generator=g) < 0.8).float()
    gexp = torch.randint(1, 4, (B,), generator=g).float() if random.random() < 0.3 else None
    support = torch.linspace(qmin, qmax, N)
    dq, parts, allq = torch.empty(B, A * N), torch.empty(B), torch.empty(B, A)
    ops.c51_head(q.contiguous(), qo.contiguous() if double_q else None, qt.contiguous(), act, mask, reward, boosts, nt, gamma, gexp,
                 support, qmin, qmax, N, maxq, dq, parts, allq)
    sd = support.double()
    qd = q.double().requires_grad_()
    logd = F.log_softmax(qd.view(B, A, N), dim=2)
    next_dist = F.softmax(qt.double().view(B, A, N), dim=2)
    if maxq:
        nq = ((F.softmax(qo.double().view(B, A, N), dim=2) if double_q else next_dist) * sd).sum(2)
        nd = next_dist[torch.arange(B), (nq + -1e9 * (1 - mask.double())).argmax(1)]
    else:
        nd = (next_dist * mask.double().unsqueeze(-1)).sum(1)
    r = reward.double().reshape(-1, 1)
    if boosts is not None:
        r = r + (act.double() * boosts.double().reshape(1, -1)).sum(1, keepdim=True)
    disc = torch.full((B, 1), gamma, dtype=torch.float64) if gexp is None else torch.pow(torch.tensor(gamma, dtype=torch.float64), gexp.double().reshape(-1, 1))
    tq = (r + disc * nt.double().reshape(-1, 1) * sd).clamp(qmin, qmax)
    b = (tq - qmin) / ((qmax - qmin) / (N - 1.0))
    lo, up = b.floor().to(torch.int64), b.ceil().to(torch.int64)
    lo[(up > 0) * (lo == up)] -= 1
    up[(lo < (N - 1)) * (lo == up)] += 1
    m = torch.zeros_like(nd)
    m.scatter_add_(1, lo, nd * (up.double() - b))
    m.scatter_add_(1, up, nd * (b - lo.double()))
    lref = -(m.detach() * (logd * act.double().unsqueeze(-1)).sum(1)).sum(1).mean()
    lref.backward()
    gs = max(1e-30, qd.grad.abs().max().item())
    want_q = (logd.detach().exp() * sd).sum(2)
    # fp32 places a target atom that lies within rounding of a grid point on either side of it: the mass moves between
    # neighbours continuously, so loss and gradient agree to fp32 accuracy of b (~N * 1e-7 of a bin)
    ok = abs(parts.double().sum().item() - lref.item()) <= 5e-5 * max(1.0, abs(lref.item()))
    ok &= bool((dq.double() - qd.grad).abs().max() <= 2e-4 * gs + 1e-9)
    ok &= bool((allq.double() - want_q).abs().max() <= 2e-5 * max(1.0, abs(qmin), abs(qmax)))
    print(("OK " if ok else "BAD"), "c51", dict(B=B, A=A, N=N, double_q=double_q, maxq=maxq, gamma=gamma, range=(qmin, qmax)),
          "dq err %.2e of %.2e" % ((dq.double() - qd.grad).abs().max().item(), gs), "loss", parts.double().sum().item(), lref.item())
    bad += 0 if ok else 1
print("bad cases:", bad)
sys.exit(1 if bad else 0)
