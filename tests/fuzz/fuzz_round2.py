import sys, random
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/fuzz")
import numpy as np
import torch
import gpu_ops  # FUZZ_ON_GPU=1: the real library on cuda:0 instead of the interpreter
ops = gpu_ops.select()
import reagent_amd._lib as L

# random shapes through the round-2 kernels: layer norm (forward / backward), ragged gather, dueling combine / split,
# the policy input maker, the SAC KLD term — each against torch on the same inputs
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
random.seed(seed)
bad = 0
for case in range(cases):
    g = torch.Generator().manual_seed(seed * 100 + case)
    ok = True
    # ---- layer norm
    B, n = random.choice([1, 3, 4, 5, 63, 130]), random.choice([1, 2, 31, 64, 65, 200, 512, 777, 2048])
    z, gamma, beta, gy = torch.randn(B, n, generator=g) * 3, torch.rand(n, generator=g) + 0.3, torch.randn(n, generator=g), torch.randn(B, n, generator=g)
    zr, gr, br = z.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ref = torch.nn.functional.layer_norm(zr, (n,), gr, br, 1e-5)
    ref.backward(gy)
    y, mean, rstd = torch.empty(B, n), torch.empty(B), torch.empty(B)
    ops.layer_norm_forward(z, gamma, beta, 1e-5, L.ACT["linear"], y32=y, mean=mean, rstd=rstd)
    dz, dg, db = torch.empty(B, n), torch.empty(n), torch.empty(n)
    ws = torch.empty(L.lib().rg_layer_norm_backward_workspace_bytes(B, n) // 4)
    ops.layer_norm_backward(gy, z, mean, rstd, gamma, dg, db, ws, dz32=dz)
    s = lambda t: max(1.0, t.abs().max().item())  # noqa: E731
    # Gradients are held to a float64 evaluation, with torch's own fp32 error as the yardstick: a 1- or 2-wide layer norm
    # is constant / +-1, its input gradient is rounding noise times rstd (up to 316 = 1 / sqrt(eps)) — torch's fp32 result
    # is 3e-5 .. 3e-4 off there (the kernel 3e-14 at width 1) and no fixed bound against it means anything.
    zd, gd, bd = z.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    torch.nn.functional.layer_norm(zd, (n,), gd, bd, 1e-5).backward(gy.double())
    near = lambda got, t32, t64: (got.double() - t64).abs().max() <= 2e-5 * s(t64) + 2 * (t32.double() - t64).abs().max()  # noqa: E731
    ok &= bool((y - ref.detach()).abs().max() <= 5e-6 * s(ref) and near(dz, zr.grad, zd.grad)
               and near(dg, gr.grad, gd.grad) and near(db, br.grad, bd.grad))
    # ---- ragged gather
    C, W = random.choice([5, 64, 300]), random.choice([1, 4, 16, 40])
    lens = torch.randint(0, W + 1, (C,), generator=g).int()
    ids, sc = torch.randint(0, 1 << 40, (C, W), generator=g), torch.rand(C, W, generator=g)
    Bq = random.choice([0, 1, 7, 1024, 1500])
    idx = torch.randint(0, C, (Bq,), generator=g)
    off, out_ids, out_sc = ops.ragged_gather(ids, sc, lens, idx)
    want_off, want_ids, want_sc, acc = [], [], [], 0
    for b in idx.tolist():
        want_off.append(acc); k = int(lens[b]); acc += k
        want_ids += ids[b, :k].tolist(); want_sc += sc[b, :k].tolist()
    ok &= off.tolist() == want_off and out_ids.tolist() == want_ids and np.allclose(out_sc.numpy(), np.array(want_sc, dtype=np.float32))
    # ---- dueling
    Bd, A, N = random.choice([1, 9, 70]), random.choice([1, 2, 16, 33]), random.choice([1, 7, 51, 200])
    val, adv, dq = torch.randn(Bd, N, generator=g), torch.randn(Bd, A * N, generator=g), torch.randn(Bd, A * N, generator=g)
    q, dadv, dval = torch.empty(Bd, A * N), torch.empty(Bd, A * N), torch.empty(Bd, N)
    ops.dueling_combine(val, adv, A, N, q)
    ops.dueling_split(dq, A, N, dadv, dval)
    a3, d3 = adv.view(Bd, A, N).double(), dq.view(Bd, A, N).double()
    ok &= bool((q.double() - (val.view(Bd, 1, N).double() + a3 - a3.mean(dim=(1, 2), keepdim=True)).reshape(Bd, -1)).abs().max() <= 3e-6
               and (dval.double() - d3.sum(1)).abs().max() <= 3e-6 * max(1, A)
               and (dadv.double() - (d3 - d3.mean(dim=(1, 2), keepdim=True)).reshape(Bd, -1)).abs().max() <= 3e-6)
    # ---- policy input maker arithmetic
    Bp, Ap = random.choice([1, 6, 257]), random.choice([1, 3, 32])
    lo = torch.randn(Ap, generator=g) - 2
    hi = lo + torch.rand(Ap, generator=g) * 4 + 0.5
    tl, th = torch.full((Ap,), -1.0), torch.full((Ap,), 1.0)
    act = lo + torch.rand(Bp, Ap, generator=g) * (hi - lo)
    nact = lo + torch.rand(Bp, Ap, generator=g) * (hi - lo)
    term, lp = torch.rand(Bp, 1, generator=g) > 0.6, -torch.rand(Bp, 1, generator=g)
    ao, no, nt, pr = torch.empty(Bp, Ap), torch.empty(Bp, Ap), torch.empty(Bp, 1), torch.empty(Bp, 1)
    ops.make_policy_input(act, nact, term, lp, torch.stack([lo, hi, tl, th]).contiguous(), ao, no, nt, pr)
    resc = lambda x: ((x - lo) / (hi - lo)) * (th - tl) + tl  # noqa: E731
    want_n = resc(nact)
    want_n[term.reshape(-1)] = 0
    ok &= torch.equal(ao, resc(act)) and torch.equal(no, want_n) and torch.equal(nt, 1.0 - term.float()) and bool(((pr - lp.exp()).abs() <= 2e-7).all())
    print(f"case {case}: ln {B}x{n}  ragged C={C} W={W} B={Bq}  dueling {Bd}x{A}x{N}  policy {Bp}x{Ap}  ->", "ok" if ok else "BAD")
    bad += 0 if ok else 1
print("bad cases:", bad)
sys.exit(1 if bad else 0)
