import sys, random
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
import emu_backend
emu_backend.install()
import reagent_amd._lib as L
from reagent_amd.engine import FusedMLP, make_stack
import test_fused_mlp as T

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
acts_pool = ["relu", "leaky_relu", "tanh"]
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    H = random.choice([256, 512])
    nl = random.choice([2, 3, 4]) if H == 256 else random.choice([2, 3])
    dims = [random.randint(1, 512)] + [H] * nl + [random.randint(1, 128)]
    acts = [random.choice(acts_pool) for _ in range(nl)] + ["linear"]
    batch = random.choice([1, 7, 64, 127, 128, 129, 200, 300])
    ws, bs = T._net(dims, acts, case, "cpu")
    codes = [L.ACT[a] for a in acts]
    if not FusedMLP.supported(ws, codes):
        print("unsupported", dims); continue
    st = make_stack(ws, bs, codes, L.PREC_BF16)
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
    ro, rdw, rdb, rdx = T._ref(ws, bs, acts, x, dout)
    errs = [T._rel(out, ro)] + [T._rel(a, b) for a, b in zip(dw, rdw)] + [T._rel(a, b) for a, b in zip(db, rdb)] + [T._rel(dx, rdx)]
    # how far bf16 operands themselves put this case from the exact result: in an ill-conditioned draw (saturated tanh
    # layers under large weights: model 5-8 % off float64) kernel and model disagree by a tenth of that, both equally wrong
    e64 = T._ref64(ws, bs, acts, x, dout)
    intrinsic = max([T._rel(a.float(), b) for a, b in zip(rdw, e64[1])] + [T._rel(rdx.float(), e64[3])])
    ok = errs[0] < 3e-3 and max(errs[1:]) < max(8e-3, 0.2 * intrinsic)
    bad += not ok
    print("OK " if ok else "BAD", dims, acts, batch, ["%.1e" % e for e in errs])
print("bad cases:", bad)
