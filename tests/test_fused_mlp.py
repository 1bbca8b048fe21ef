"""The fused bf16 FullyConnected-stack kernels (rg_mlp_forward_fused / rg_mlp_backward_fused /
rg_fc_wgrad_frag / rg_stage_weights_frag) against a float64 statement of the same math with bf16
rounding applied at the same points (weights, activations and dZ between layers).
Tolerance: relative Frobenius error < 0.5 % (fp32 accumulation order + rare rounding ties that flip
a ReLU mask)."""
import pytest
import torch
import torch.nn.functional as F

import reagent_amd._lib as L
from reagent_amd.engine import FCStack, FusedMLP, make_stack

ACTS = {"relu": F.relu, "leaky_relu": F.leaky_relu, "tanh": torch.tanh, "linear": lambda x: x}


def _net(dims, acts, seed, dev):
    g = torch.Generator().manual_seed(seed)
    ws = [torch.nn.Parameter((torch.randn(o, i, generator=g) * (1.5 / i ** 0.5)).to(dev)) for i, o in zip(dims, dims[1:])]
    bs = [torch.nn.Parameter((torch.randn(o, generator=g) * 0.1).to(dev)) for o in dims[1:]]
    return ws, bs


DACT = {"relu": lambda h: (h > 0).double(), "leaky_relu": lambda h: torch.where(h > 0, 1.0, 0.01).double(),
        "tanh": lambda h: 1 - h * h, "linear": lambda h: torch.ones_like(h)}


def _bf(t):
    return t.float().to(torch.bfloat16).double()


def _ref(ws, bs, acts, x, dout):
    """float64 statement of the kernels' math with bf16 rounding at the same points: weights,
    activations between layers, dZ between layers (accumulation itself is fp32 on the GPU)."""
    W = [_bf(w.detach().cpu()) for w in ws]
    Bv = [b.detach().cpu().double() for b in bs]
    hs = [_bf(x.cpu())]
    for l, (w, b, a) in enumerate(zip(W, Bv, acts)):
        z = ACTS[a](hs[-1] @ w.t() + b)
        hs.append(z if l == len(W) - 1 else _bf(z))
    dz = _bf(dout.cpu())
    dws, dbs = [None] * len(W), [None] * len(W)
    for l in range(len(W) - 1, -1, -1):
        dws[l] = dz.t() @ hs[l]
        dbs[l] = dz.sum(0)
        dh = dz @ W[l]
        if l > 0:
            dz = _bf(dh * DACT[acts[l - 1]](hs[l]))
    return hs[-1], dws, dbs, dh


def _rel(a, b):
    return ((a.double().cpu() - b).norm() / (b.norm() + 1e-12)).item()


@pytest.mark.parametrize("dims,acts,batch", [
    ([24, 256, 256, 16], ["relu", "relu", "linear"], 100),
    ([128, 512, 512, 512, 16], ["relu", "relu", "relu", "linear"], 300),
    ([128, 256, 256, 8], ["relu", "relu", "linear"], 700),
    ([130, 256, 256, 70], ["leaky_relu", "tanh", "linear"], 129),
    ([40, 512, 512, 3], ["tanh", "relu", "linear"], 64),
    ([64, 256, 256, 200], ["relu", "relu", "linear"], 140),  # 7 output column tiles: the pipelined output layer
    # weight-gradient tile shapes (wgrad_shape_core): the cases above take 8 x 8, 16 x 4 (dW0 of a 512-wide stack) and
    # 1 x 16 (<= 32 outputs over 512); 64 and 128 outputs over 512 take 2 x 16 and 4 x 16
    ([32, 512, 512, 64], ["relu", "relu", "linear"], 260),
    ([96, 512, 512, 128], ["relu", "leaky_relu", "linear"], 130),
])
def test_fused_forward_backward_wgrad(backend, dims, acts, batch):
    dev = backend.device
    ws, bs = _net(dims, acts, 1, dev)
    codes = [L.ACT[a] for a in acts]
    assert FusedMLP.supported(ws, codes)
    st = make_stack(ws, bs, codes, L.PREC_BF16)
    assert isinstance(st, FusedMLP)
    st.set_need_input_grad(True)
    st.stage_weights(need_transposed=True)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(batch, dims[0], generator=g).to(dev)
    dout = (torch.randn(batch, dims[-1], generator=g) / batch).to(dev)
    out = torch.zeros(batch, dims[-1], device=dev)
    xc, xt = st.stage_input(x, True)
    st.forward(xc, out, save=True)
    dw = [torch.zeros_like(w) for w in ws]
    db = [torch.zeros_like(b) for b in bs]
    dx = torch.zeros(batch, dims[0], device=dev)
    st.backward(dout, xt, dw, db, dx32=dx)
    ref_out, ref_dw, ref_db, ref_dx = _ref(ws, bs, acts, x, dout)
    assert _rel(out, ref_out) < 2e-3, _rel(out, ref_out)
    for l in range(len(ws)):
        assert _rel(dw[l], ref_dw[l]) < 5e-3, (l, _rel(dw[l], ref_dw[l]))
        assert _rel(db[l], ref_db[l]) < 5e-3, (l, _rel(db[l], ref_db[l]))
    assert _rel(dx, ref_dx) < 5e-3
    # non-saving forward gives the same outputs (bit for bit: same kernel, same order)
    out2 = torch.zeros_like(out)
    st.forward(xc, out2, save=False)
    assert torch.equal(out2, out)
    # bf16 input is accepted as well
    out3 = torch.zeros_like(out)
    st.forward(x.to(torch.bfloat16), out3, save=False)
    assert torch.equal(out3, out)
    # ... also when saving for backward
    out4, dw4 = torch.zeros_like(out), [torch.zeros_like(w) for w in ws]
    db4, dx4 = [torch.zeros_like(b) for b in bs], torch.zeros_like(dx)
    st.forward(x.to(torch.bfloat16), out4, save=True)
    st.backward(dout, xt, dw4, db4, dx32=dx4)
    assert torch.equal(out4, out) and torch.equal(dx4, dx)
    for l in range(len(ws)):
        assert torch.equal(dw4[l], dw[l]) and torch.equal(db4[l], db[l]), l


def test_fused_matches_per_layer_bf16_engine(backend):
    """Same math as the per-layer bf16 GEMM path (both round activations to bf16 between layers)."""
    dev = backend.device
    dims, acts = [64, 256, 256, 8], ["relu", "relu", "linear"]
    ws, bs = _net(dims, acts, 5, dev)
    codes = [L.ACT[a] for a in acts]
    fused, plain = make_stack(ws, bs, codes, L.PREC_BF16), FCStack(ws, bs, codes, L.PREC_BF16)
    fused.stage_weights(True)
    plain.stage_weights(True)
    x = torch.randn(200, 64, generator=torch.Generator().manual_seed(3)).to(dev)
    o1, o2 = torch.zeros(200, 8, device=dev), torch.zeros(200, 8, device=dev)
    fused.forward(fused.stage_input(x, False)[0], o1)
    plain.forward(plain.stage_input(x, False)[0], o2)
    assert (o1 - o2).abs().max() <= 2e-3 * max(1.0, o2.abs().max().item())


def test_unsupported_shapes_fall_back_to_per_layer(backend):
    dev = backend.device
    ws, bs = _net([16, 128, 64, 4], ["relu", "relu", "linear"], 0, dev)
    assert isinstance(make_stack(ws, bs, [1, 1, 0], L.PREC_BF16), FCStack)
    ws, bs = _net([16, 256, 256, 4], ["relu", "relu", "linear"], 0, dev)
    assert isinstance(make_stack(ws, bs, [1, 1, 0], L.PREC_F32), FCStack)


# ---- split-bf16 ("bf16x3") mode: fp32-class results from the bf16 MFMA pipe ----------------------------
def _ref64(ws, bs, acts, x, dout):
    """float64 statement of the exact math (no rounding anywhere)"""
    W = [w.detach().cpu().double() for w in ws]
    Bv = [b.detach().cpu().double() for b in bs]
    hs = [x.cpu().double()]
    for w, b, a in zip(W, Bv, acts):
        hs.append(ACTS[a](hs[-1] @ w.t() + b))
    dz = dout.cpu().double()
    dws, dbs = [None] * len(W), [None] * len(W)
    for l in range(len(W) - 1, -1, -1):
        dws[l] = dz.t() @ hs[l]
        dbs[l] = dz.sum(0)
        dh = dz @ W[l]
        if l > 0:
            dz = dh * DACT[acts[l - 1]](hs[l])
    return hs[-1], dws, dbs, dh


@pytest.mark.parametrize("dims,acts,batch", [
    ([24, 256, 256, 16], ["relu", "relu", "linear"], 100),
    ([128, 512, 512, 512, 16], ["relu", "relu", "relu", "linear"], 200),
    ([130, 256, 256, 70], ["leaky_relu", "tanh", "linear"], 129),
    ([288, 512, 512, 1], ["relu", "relu", "linear"], 70),
])
def test_fused_x3_is_fp32_class(backend, dims, acts, batch):
    """hi/lo split operands, three MFMAs per product: every result within ~1e-5 (relative Frobenius) of the
    float64 statement — against 2e-3 for plain bf16 operands — and max-abs output error <= 1e-4."""
    dev = backend.device
    ws, bs = _net(dims, acts, 1, dev)
    codes = [L.ACT[a] for a in acts]
    st = make_stack(ws, bs, codes, L.PREC_BF16X3)
    assert isinstance(st, FusedMLP) and st.x3
    st.set_need_input_grad(True)
    st.stage_weights(need_transposed=True)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(batch, dims[0], generator=g).to(dev)
    dout = (torch.randn(batch, dims[-1], generator=g) / batch).to(dev)
    out = torch.zeros(batch, dims[-1], device=dev)
    xc, xt = st.stage_input(x, True)
    st.forward(xc, out, save=True)
    dw = [torch.zeros_like(w) for w in ws]
    db = [torch.zeros_like(b) for b in bs]
    dx = torch.zeros(batch, dims[0], device=dev)
    st.backward(dout, xt, dw, db, dx32=dx)
    ref_out, ref_dw, ref_db, ref_dx = _ref64(ws, bs, acts, x, dout)
    print(f"\n[x3 {dims}] out rel {_rel(out, ref_out):.2e} max-abs {(out.double().cpu() - ref_out).abs().max():.2e}; "
          f"dw rel {[float('%.1e' % _rel(dw[l], ref_dw[l])) for l in range(len(ws))]}; dx rel {_rel(dx, ref_dx):.2e}")
    assert _rel(out, ref_out) < 2e-5 and (out.double().cpu() - ref_out).abs().max() <= 1e-4
    for l in range(len(ws)):
        assert _rel(dw[l], ref_dw[l]) < 3e-5, (l, _rel(dw[l], ref_dw[l]))
        assert _rel(db[l], ref_db[l]) < 3e-5, (l, _rel(db[l], ref_db[l]))
    assert _rel(dx, ref_dx) < 3e-5
    out2 = torch.zeros_like(out)
    st.forward(xc, out2, save=False)
    assert torch.equal(out2, out)


_ONE_PLANE_SNIPPET = r"""
import sys, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
if {emu}:
    import emu_backend; emu_backend.install()
import reagent_amd._lib as L
import test_fused_mlp as T
from reagent_amd.engine import FusedMLP, make_stack
dev = "cpu" if {emu} else "cuda"
dims, acts, batch = [128, 512, 512, 16], ["relu", "relu", "linear"], 192
ws, bs = T._net(dims, acts, 1, dev)
st = make_stack(ws, bs, [L.ACT[a] for a in acts], L.PREC_BF16X3)
assert isinstance(st, FusedMLP) and st.x3
st.set_need_input_grad(True); st.stage_weights(need_transposed=True)
g = torch.Generator().manual_seed(2)
x = torch.randn(batch, dims[0], generator=g).to(dev)
dout = (torch.randn(batch, dims[-1], generator=g) / batch).to(dev)
out = torch.zeros(batch, dims[-1], device=dev)
xc, xt = st.stage_input(x, True)
st.forward(xc, out, save=True)
dw = [torch.zeros_like(w) for w in ws]; db = [torch.zeros_like(b) for b in bs]
dx = torch.zeros(batch, dims[0], device=dev)
st.backward(dout, xt, dw, db, dx32=dx)
ref_out, ref_dw, ref_db, ref_dx = T._ref64(ws, bs, acts, x, dout)
print("RESULT", T._rel(out, ref_out), max(T._rel(dw[l], ref_dw[l]) for l in range(3)), max(T._rel(db[l], ref_db[l]) for l in range(3)), T._rel(dx, ref_dx))
"""


def test_x3_one_plane_dz_option(backend):
    """RG_X3_DZ_PLANES=1 (rg_mlp_frag.h: x3_dz_planes — opt-in, NOT the default): the stack's weight gradient multiplies dZ as ONE
    bf16 plane (two MFMAs per tile pair, no dZ lo plane written or read: -6 % on the C2 split-bf16 step, profiles/r05_*).
    What it keeps: the forward, dgrad (dx) and the bias gradients stay three-product / fp32-class.  What it gives up: dW is
    bf16-rounded-dZ class, ~2e-3 relative — this test pins both (and that the default build is NOT that, test above)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for planes in ("1", "2"):
        env = dict(os.environ, RG_X3_DZ_PLANES=planes)
        p = subprocess.run([sys.executable, "-c", _ONE_PLANE_SNIPPET.format(root=root, emu=backend.name == "emu")],
                           capture_output=True, text=True, env=env, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        res[planes] = [float(v) for v in [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][0].split()[1:]]
    print(f"\n[x3 dZ planes] (out, dW, db, dx) relative errors: two planes {res['2']}, one plane {res['1']}")
    out1, dw1, db1, dx1 = res["1"]
    out2, dw2, db2, dx2 = res["2"]
    assert out1 == out2 and dx1 == dx2 and db1 == db2  # only the weight gradient changes
    assert out1 < 2e-5 and dx1 < 3e-5 and db1 < 3e-5 and dw2 < 3e-5
    assert 1e-4 < dw1 < 5e-3  # one plane of dZ: 2^-9 per element


def test_x3_on_an_unserved_shape_runs_exact_fp32(backend):
    ws, bs = _net([16, 128, 64, 4], ["relu", "relu", "linear"], 0, backend.device)
    st = make_stack(ws, bs, [1, 1, 0], L.PREC_BF16X3)
    assert isinstance(st, FCStack) and st.precision == L.PREC_F32


@pytest.mark.parametrize("prec", [L.PREC_BF16, L.PREC_BF16X3])
def test_two_panel_input_and_partial_input_gradient(backend, prec):
    """cat(state, action) read in place as two K-panels == the assembled matrix, bit for bit; dx restricted to
    the action columns == that slice of the full input gradient (reagent/models/critic.py:79-92 without the cat)."""
    dev = backend.device
    S, A, batch = 64, 20, 150
    dims, acts = [S + A, 256, 256, 1], ["relu", "relu", "linear"]
    ws, bs = _net(dims, acts, 4, dev)
    codes = [L.ACT[a] for a in acts]
    st = make_stack(ws, bs, codes, prec)
    assert isinstance(st, FusedMLP)
    st.set_need_input_grad(True)
    st.stage_weights(need_transposed=True)
    g = torch.Generator().manual_seed(6)
    state, action = torch.randn(batch, S, generator=g).to(dev), torch.randn(batch, A, generator=g).to(dev)
    dout = (torch.randn(batch, 1, generator=g) / batch).to(dev)
    cat = torch.cat((state, action), dim=1).contiguous()

    def run(two_panel, col0):
        out = torch.zeros(batch, 1, device=dev)
        if two_panel:
            st.forward(state, out, save=True, x2=action)
        else:
            st.forward(cat, out, save=True)
        dw = [torch.zeros_like(w) for w in ws]
        db = [torch.zeros_like(b) for b in bs]
        dx = torch.zeros(batch, S + A - col0, device=dev)
        st.backward(dout, None, dw, db, dx32=dx, dx_col0=col0)
        return out, dw, db, dx

    o1, dw1, db1, dx1 = run(False, 0)
    o2, dw2, db2, dx2 = run(True, S)
    assert torch.equal(o1, o2)
    for a, b in zip(dw1 + db1, dw2 + db2):
        assert torch.equal(a, b)
    assert torch.equal(dx1[:, S:], dx2)
    out3 = torch.zeros(batch, 1, device=dev)
    st.forward(cat, out3, save=False)  # a plain call after a two-panel one: the panel fields do not stick
    assert torch.equal(out3, o1)


@pytest.mark.parametrize("prec", [L.PREC_BF16, L.PREC_BF16X3])
@pytest.mark.parametrize("acts", [["relu", "leaky_relu", "linear"], ["tanh", "relu", "linear"]])
def test_dx_only_backward_of_a_frozen_stack(backend, prec, acts):
    """forward(save=SAVE_FOR_DX) + backward(skip_wgrad) — SAC's / TD3's frozen critics in the actor step — returns the
    input gradient of the full saving path bit for bit without writing activation or dZ fragments (buffers poisoned
    to prove it); a tanh layer still gets its activations saved (its gradient needs the values)."""
    from reagent_amd.engine import SAVE_FOR_DX

    dev = backend.device
    S, A, batch = 64, 32, 140
    dims = [S + A, 256, 256, 1]
    ws, bs = _net(dims, acts, 9, dev)
    st = make_stack(ws, bs, [L.ACT[a] for a in acts], prec)
    assert isinstance(st, FusedMLP)
    st.set_need_input_grad(True)
    st.stage_weights(need_transposed=True)
    g = torch.Generator().manual_seed(12)
    state, action = torch.randn(batch, S, generator=g).to(dev), torch.randn(batch, A, generator=g).to(dev)
    dout = (torch.randn(batch, 1, generator=g) / batch).to(dev)
    out_full, out_dx = torch.zeros(batch, 1, device=dev), torch.zeros(batch, 1, device=dev)
    dx_full, dx_only = torch.zeros(batch, A, device=dev), torch.zeros(batch, A, device=dev)
    st.forward(state, out_full, save=True, x2=action)
    st.backward(dout, None, None, None, dx32=dx_full, skip_wgrad=True, dx_col0=S)
    # poison what the reduced path must neither read (sign-based layers' activations, layer 0's input) nor write
    frag_names = [k for k, v in st._ws.items() if isinstance(v, (list, tuple))]
    before = {}
    for k in frag_names:
        for i, t in enumerate(st._ws[k]):
            if isinstance(t, torch.Tensor) and t.dtype == torch.bfloat16:
                keep = k.startswith("act") and i >= 1 and acts[i - 1] == "tanh"
                if not keep:
                    t.fill_(float("nan"))
                before[(k, i)] = t.clone()
    st.forward(state, out_dx, save=SAVE_FOR_DX, x2=action)
    st.backward(dout, None, None, None, dx32=dx_only, skip_wgrad=True, dx_col0=S)
    assert torch.equal(out_full, out_dx) and torch.equal(dx_full, dx_only) and torch.isfinite(dx_only).all()
    for (k, i), t in before.items():
        now = st._ws[k][i]
        same = torch.equal(now.view(torch.int16), t.view(torch.int16))
        rewritten_ok = k.startswith("act") and i >= 1 and acts[i - 1] == "tanh"
        assert same or rewritten_ok, (k, i)
    with pytest.raises(AssertionError, match="SAVE_FOR_DX"):
        st.backward(dout, None, [torch.zeros_like(w) for w in ws], [torch.zeros_like(b) for b in bs])


@pytest.mark.parametrize("prec", [L.PREC_BF16, L.PREC_BF16X3])
@pytest.mark.parametrize("n_out,batch", [(16, 256), (16, 200), (1, 256), (1, 130), (3, 128), (8, 192)])
def test_thin_output_layer_store_forms_agree_bit_for_bit(backend, prec, n_out, batch):
    """RG_OUT_ROWSTORE (round 5): a thin output layer's result leaves as whole 16-byte pieces through LDS — the tile's block as one
    run for a dense output and a full tile (a critic's single column), whole rows where N % 4 == 0 — and straight from the
    accumulators otherwise (a misaligned base, a row pitch that is no multiple of 4 floats, the last partial tile of a dense
    output).  Which form runs is decided per workgroup from the output's address and pitch; the VALUES must not depend on it:
    the same forward into a dense aligned buffer, a buffer 4 bytes off 16-byte alignment and a wider-pitch buffer, bit for bit."""
    dev = backend.device
    dims, acts = [64, 256, 256, n_out], ["relu", "relu", "linear"]
    ws, bs = _net(dims, acts, 1, dev)
    st = make_stack(ws, bs, [L.ACT[a] for a in acts], prec)
    assert isinstance(st, FusedMLP)
    st.stage_weights(need_transposed=False)
    x = torch.randn(batch, dims[0], generator=torch.Generator().manual_seed(4)).to(dev)
    xc, _ = st.stage_input(x, False)
    dense = torch.zeros(batch, n_out, device=dev)
    st.forward(xc, dense, save=False)
    off = torch.zeros(batch * n_out + 1, device=dev)[1:].view(batch, n_out)  # 4 bytes off: straight from the accumulators
    st.forward(xc, off, save=False)
    wide = torch.zeros(batch, n_out + 4, device=dev)  # pitch N + 4: rows of whole pieces (N % 4 == 0) or the accumulator path
    st.forward(xc, wide[:, :n_out], save=False)
    odd = torch.zeros(batch, n_out + 1, device=dev)  # pitch N + 1: never a multiple of 4 floats for these N
    st.forward(xc, odd[:, :n_out], save=False)
    assert torch.isfinite(dense).all() and dense.abs().max() > 0
    assert torch.equal(off, dense) and torch.equal(wide[:, :n_out], dense) and torch.equal(odd[:, :n_out], dense)
    assert torch.equal(wide[:, n_out:], torch.zeros(batch, 4, device=dev))  # nothing written past the row's N columns
    assert torch.equal(odd[:, n_out:], torch.zeros(batch, 1, device=dev))


_UNEVEN_SNIPPET = r"""
import sys, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
if {emu}:
    import emu_backend; emu_backend.install()
import reagent_amd._lib as L
import test_fused_mlp as T
from reagent_amd.engine import FusedMLP, make_stack
dev = "cpu" if {emu} else "cuda"
dims, acts, batch = [64, 512, 512, 8], ["relu", "relu", "linear"], 2048
ws, bs = T._net(dims, acts, 1, dev)
st = make_stack(ws, bs, [L.ACT[a] for a in acts], {prec})
assert isinstance(st, FusedMLP)
st.stage_weights(need_transposed=True)
g = torch.Generator().manual_seed(2)
x = torch.randn(batch, dims[0], generator=g).to(dev)
dout = (torch.randn(batch, dims[-1], generator=g) / batch).to(dev)
out = torch.zeros(batch, dims[-1], device=dev)
st.forward(x, out, save=True)
dw = [torch.zeros_like(w) for w in ws]; db = [torch.zeros_like(b) for b in bs]
st.backward(dout, None, dw, db)
torch.save([t.cpu() for t in dw + db], {path!r})
"""


@pytest.mark.parametrize("prec", [L.PREC_BF16, L.PREC_BF16X3])
def test_unevenly_split_weight_gradient_launch(backend, prec, tmp_path):
    """rg_mlp_wgrad_fused's round-5 launch plan (mlp_fused.hip: "entries"): when the multi-tile layers' workgroups are exactly one
    round of the chip and only single-tile layers follow, a part of each multi-tile layer's splits is made shorter and the
    rest longer, as two ENTRIES of the launch (the second with mb_base > 0 and its partial slabs after the first's).  The plan
    keys on the CU count, so only C2's full-size stack meets it on the GPU; here RG_WGRAD_TOTAL / RG_WGRAD_THIN make a small
    stack meet it (128 workgroups of the 512 x 512 layer, 64 single-tile ones) on either backend.  Against the same launch
    with even splits (RG_WGRAD_UNEVEN=0): the layers whose entries did not change are bit-identical, the unevenly split one
    differs by the rounding of its partial sums only — bf16 partial tiles: 2^-9 of each of the 32 partials, which cut the batch
    elsewhere (~2e-3 of max|dW| here; the bf16 operands themselves cost 4e-3); split-bf16 keeps fp32 partials: summation order."""
    import os
    import subprocess
    import sys

    if backend.name == "emu" and prec == L.PREC_BF16X3:
        pytest.skip("a minute on the emulator for the same host-side plan; this case runs on the GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got, plans = {}, {}
    for uneven in ("125", "0"):
        path = str(tmp_path / f"dw_{uneven}.pt")
        env = dict(os.environ, RG_WGRAD_UNEVEN=uneven, RG_WGRAD_TOTAL="128", RG_WGRAD_THIN="32", RG_WGRAD_DEBUG="1")
        p = subprocess.run([sys.executable, "-c", _UNEVEN_SNIPPET.format(root=root, emu=backend.name == "emu", prec=prec, path=path)],
                           capture_output=True, text=True, env=env, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        plans[uneven] = [ln for ln in p.stderr.splitlines() if ln.startswith("rg_mlp_wgrad_fused: entry")]
        got[uneven] = torch.load(path)
    print("\n" + "\n".join(plans["125"]))
    assert len(plans["0"]) == 3 and len(plans["125"]) == 4  # the 512 x 512 layer as two entries ...
    hidden = [ln for ln in plans["125"] if "layer 1 " in ln]
    assert len(hidden) == 2 and "blocks [0, 48) in 16 splits of 3" in hidden[0] and "blocks [48, 64) in 16 splits of 1" in hidden[1]
    assert plans["125"].index(hidden[0]) == 0 and plans["125"].index(hidden[1]) == 3  # ... the long class first, the short one last
    (dw0, dw1, dw2, *db), (ew0, ew1, ew2, *eb) = got["125"], got["0"]
    assert torch.equal(dw0, ew0) and torch.equal(dw2, ew2) and all(torch.equal(a, b) for a, b in zip(db, eb))
    rel = float((dw1 - ew1).abs().max() / ew1.abs().max())
    print(f"[uneven wgrad] hidden layer's dW, uneven vs even splits: {rel:.2e} of max|dW|")
    assert 0 < rel < (6e-3 if prec == L.PREC_BF16 else 2e-6)
