"""CPU experiment (no kernel): what a TWO-product weight gradient (dZ kept as ONE bf16 plane: dZ_hi . (x_hi + x_lo)) would do to
the split-bf16 mode's gradients and first-Adam-step weights at C2's shapes.  fp64 reference; operands from the oracle's net."""
import sys, torch
sys.path.insert(0, '/root/repo')
from reagent_amd import synthetic
torch.manual_seed(0)
def run(B):
    S, A, H = 128, 16, [512, 512, 512]
    dims = [S] + H + [A]
    init = synthetic.fc_init(dims, ["relu"] * 3 + ["linear"], seed=40)
    Ws = [w.double() for w in init[0::2]]; bs = [b.double() for b in init[1::2]]
    b = synthetic.dqn_batch(B, S, A, seed=1)
    x = b["state"].double()
    acts = [x]
    h = x
    for l, (W, bb) in enumerate(zip(Ws, bs)):
        z = h @ W.T + bb
        h = torch.relu(z) if l < 3 else z
        acts.append(h)
    q = h
    # a TD-like dZ at the output: (q_sel - target) / B on the logged action, Huber-clipped
    a = b["action"].double()
    tgt = b["reward"].double() + 0.99 * torch.randn(B, 1, dtype=torch.double) * 0.5
    d = ((q * a).sum(1, keepdim=True) - tgt).clamp(-1, 1) / B
    dz = d * a
    out = []
    for l in (3, 2, 1, 0):
        X = acts[l]
        g_exact = dz.T @ X
        split = lambda t: (t.float().bfloat16().double(), (t.float() - t.float().bfloat16().float()).bfloat16().double())
        dz_hi, dz_lo = split(dz)
        x_hi, x_lo = split(X)
        g3 = dz_lo.T @ x_hi + dz_hi.T @ x_lo + dz_hi.T @ x_hi          # the shipped three-product form
        g2 = dz_hi.T @ x_lo + dz_hi.T @ x_hi                            # dZ as one bf16 plane
        gm = g_exact.abs().max()
        e3, e2 = (g3 - g_exact).abs(), (g2 - g_exact).abs()
        flip3 = ((g3 * g_exact) < 0).double().mean().item()
        flip2 = ((g2 * g_exact) < 0).double().mean().item()
        out.append((l, (e3.max() / gm).item(), (e2.max() / gm).item(), flip3, flip2))
        if l > 0:
            dz = (dz @ Ws[l]) * (acts[l] > 0)
    return out
for B in (2048, 65536):
    print("B =", B)
    for l, e3, e2, f3, f2 in run(B):
        print(f"  layer {l}: max|dg|/max|g|  three-product {e3:.2e}   two-product {e2:.2e}   |  first-Adam-step direction flips  {100*f3:.4f} %  vs  {100*f2:.3f} %")
