import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
import emu_backend
emu_backend.install()
import reagent_amd._lib as L
from reagent_amd.engine import FusedMLP, make_stack
import test_fused_mlp as T

# the split-bf16 ("bf16x3") fused stack on random shapes — input width 1..512 (the K = 16 padding), output width 1..128,
# 2-4 hidden layers of 256 / 512, batches around the 64-row tile — against the exact float64 statement: this is the mode
# held to the reference's 1e-4, so outputs, every weight / bias gradient and the input gradient must be fp32-class
# (relative Frobenius error <= 3e-5, scaled up only where float32 arithmetic itself is that far off in an ill-conditioned draw)
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
acts_pool = ["relu", "leaky_relu", "tanh"]
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    H = random.choice([256, 512])
    nl = random.choice([2, 3, 4]) if H == 256 else random.choice([2, 3])
    dims = [random.randint(1, 512)] + [H] * nl + [random.randint(1, 128)]
    acts = [random.choice(acts_pool) for _ in range(nl)] + ["linear"]
    batch = random.choice([1, 7, 63, 64, 65, 128, 129, 200])
    ws, bs = T._net(dims, acts, case, "cpu")
    codes = [L.ACT[a] for a in acts]
    st = make_stack(ws, bs, codes, L.PREC_BF16X3)
    if not (isinstance(st, FusedMLP) and st.x3):
        print("unserved", dims); continue
    st.set_need_input_grad(True)
    st.stage_weights(need_transposed=True)
    g = torch.Generator().manual_seed(case)
    x = torch.randn(batch, dims[0], generator=g)
    dout = torch.randn(batch, dims[-1], generator=g) / batch
    out = torch.zeros(batch, dims[-1])
    xc, xt = st.stage_input(x, True)
    st.forward(xc, out, save=True)
    dw = [torch.zeros_like(w) for w in ws]; db = [torch.zeros_like(b) for b in bs]
    dx = torch.zeros(batch, dims[0])
    st.backward(dout, xt, dw, db, dx32=dx)
    ro, rdw, rdb, rdx = T._ref64(ws, bs, acts, x, dout)
    # float32 torch on the same case: the yardstick for what fp32-class means here
    W32 = [w.detach().clone().requires_grad_() for w in ws]; B32 = [b.detach().clone().requires_grad_() for b in bs]
    x32 = x.clone().requires_grad_()
    h = x32
    for w, b, a in zip(W32, B32, acts):
        h = T.ACTS[a](h @ w.t() + b)
    h.backward(dout)
    f32 = max([T._rel(h.detach(), ro)] + [T._rel(w.grad, r) for w, r in zip(W32, rdw)] + [T._rel(x32.grad, rdx)])
    errs = [T._rel(out, ro)] + [T._rel(a, b) for a, b in zip(dw, rdw)] + [T._rel(a, b) for a, b in zip(db, rdb)] + [T._rel(dx, rdx)]
    out2 = torch.zeros_like(out)
    st.forward(xc, out2, save=False)
    # A ReLU / leaky-ReLU unit whose pre-activation lies inside the split-bf16 noise (~1e-5 of the sum of |terms|) can take
    # the other branch of the derivative than float64 does: one such unit moves dZ of its layer by ~sqrt(2 / (B H)) of its
    # norm (seen: |z| = 1.0e-6 among 65 536 values of mean magnitude 1.2 -> 1.4e-3 on that layer's gradients, every
    # other tensor at 1e-5).  Count the units within 2e-5 of the mean magnitude (the noise level) and allow for them; cases without any stay at 3e-5.
    hd, flips = x.double(), 0
    for w, b, a in zip(ws[:-1], bs[:-1], acts[:-1]):
        z = hd @ w.double().t() + b.double()
        if a in ("relu", "leaky_relu"):
            flips += int((z.abs() < 2e-5 * z.abs().mean()).sum())
        hd = T.ACTS[a](z)
    allow = max(3e-5, 30 * f32) + 3 * (2 * flips / (batch * H)) ** 0.5
    ok = errs[0] < max(3e-5, 30 * f32) and max(errs) < allow and torch.equal(out2, out)
    bad += not ok
    print("OK " if ok else "BAD", dims, acts, batch, "max err %.1e (fp32 torch %.1e, %d borderline units)" % (max(errs), f32, flips), "max|dout| %.1e" % (out.double() - ro).abs().max().item())
print("bad cases:", bad)
sys.exit(1 if bad else 0)
