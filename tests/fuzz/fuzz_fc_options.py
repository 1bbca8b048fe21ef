import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
import emu_backend
emu_backend.install()
import reagent_amd._lib as L
from reagent_amd.engine import ensure_slab, grad_views
from reagent_amd.models import FullyConnectedNetwork
from test_fc_options import _reference_pass

# random FullyConnectedNetwork configurations with batch-norm / layer-norm / dropout / residual wrappers in random
# combinations, training and eval mode: forward, input gradient, every parameter gradient and the running statistics
# against torch autograd of the same layer sequence (with the kernel's own dropout masks)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 9
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 16
random.seed(seed)
acts = ["relu", "tanh", "leaky_relu", "linear", "sigmoid"]
bad = 0
for case in range(cases):
    nl = random.randint(1, 4)
    widths = [random.choice([3, 8, 17, 32, 70])]  # (a width-1 layer norm is constant: the batch norm after it divides rounding noise by sqrt(eps))
    for _ in range(nl):
        widths.append(widths[-1] if random.random() < 0.5 else random.choice([5, 16, 33, 64]))
    kw = dict(use_batch_norm=random.random() < 0.6, use_layer_norm=random.random() < 0.4,
              dropout_ratio=random.choice([0.0, 0.0, 0.1, 0.5]), use_skip_connections=random.random() < 0.6,
              normalize_output=random.random() < 0.5)
    if not (kw["use_batch_norm"] or kw["dropout_ratio"] > 0 or (kw["use_skip_connections"] and any(a == b for a, b in zip(widths, widths[1:])))):
        kw["use_batch_norm"] = True
    training, B = random.random() < 0.7, random.choice([7, 64, 300])  # (2-row batches / 2-wide layer norms normalise to +-1: their
    # gradients are rounding noise amplified by 1 / sqrt(eps), nothing to compare)
    torch.manual_seed(seed * 1000 + case)
    net = FullyConnectedNetwork(widths, [random.choice(acts) for _ in range(nl)], **kw)
    with torch.no_grad():
        for m in [m for m in net.layer_norms() + net.batch_norms() if m is not None]:
            m.weight.uniform_(0.5, 1.5)
            m.bias.normal_(0, 0.2)
        for bn in [b for b in net.batch_norms() if b is not None]:
            bn.running_mean.normal_(0, 0.3)
            bn.running_var.uniform_(0.5, 2.0)
    net.train(training)
    x, dout = torch.randn(B, widths[0]), torch.randn(B, widths[-1]) / B
    st = net.stack()
    before = [(b.running_mean.clone(), b.running_var.clone()) for b in net.batch_norms() if b is not None]
    st.set_need_input_grad(True)
    st.stage_weights(need_transposed=True)
    xc, xt = st.stage_input(x, need_transposed=True)
    out = torch.empty(B, widths[-1])
    st.forward(xc, out, save=True)
    keeps = [st._bufs[("keep/s", i)].view(B, -1) if (p > 0 and training) else None for i, p in enumerate(net.dropouts())]
    after = [(b.running_mean.clone(), b.running_var.clone()) for b in net.batch_norms() if b is not None]
    for b, (rm, rv) in zip([b for b in net.batch_norms() if b is not None], before):
        b.running_mean.copy_(rm)
        b.running_var.copy_(rv)
    out_ref, dx_ref, grads_ref, stats = _reference_pass(net, x, dout, keeps, training)
    s = lambda t: max(1.0, t.abs().max().item())  # noqa: E731
    tol = 2e-4
    ok = bool((out - out_ref).abs().max() <= tol * s(out_ref))
    ok &= all(bool((a[0] - r[0]).abs().max() <= 1e-5 * s(r[0]) and (a[1] - r[1]).abs().max() <= 1e-5 * s(r[1])) for a, r in zip(after, stats))
    params = list(net.parameters())
    slab = ensure_slab(params)
    dw, db = grad_views(net, slab, params)
    dx = torch.empty(B, widths[0])
    st.backward(dout, xt, dw, db, dx32=dx, out32=out)
    ok &= bool((dx - dx_ref).abs().max() <= tol * s(dx_ref))
    for i, (k, p) in enumerate(net.named_parameters()):
        ok &= bool((slab.view(slab.grad, i) - grads_ref[k]).abs().max() <= tol * s(grads_ref[k]))
    print(f"case {case}: widths {widths} B={B} {'train' if training else 'eval'} "
          + " ".join(k for k, v in kw.items() if v) + f" p={kw['dropout_ratio']} ->", "ok" if ok else "BAD")
    bad += 0 if ok else 1
print("bad cases:", bad)
sys.exit(1 if bad else 0)
