"""FullyConnected layer ops (rg_fc_forward / rg_fc_dgrad / rg_fc_wgrad / rg_transpose_cast) through
the C ABI against a float64 numpy statement of the same math.
Reference semantics: nn.Linear + activation, reagent/models/fully_connected_network.py:101-153, and
its autograd backward.  Tolerances: fp32 mode 2e-5 relative to the result scale; bf16 mode is checked
on bf16-rounded operands (so only accumulation order / output rounding differ)."""
import numpy as np
import pytest
import torch

import reagent_amd._lib as L
from reagent_amd import ops

PRECS = [pytest.param(L.PREC_F32, id="f32"), pytest.param(L.PREC_BF16, id="bf16")]


def _act(z, a):
    return {0: z, 1: np.maximum(z, 0), 2: np.where(z > 0, z, 0.01 * z), 3: np.tanh(z),
            4: 1 / (1 + np.exp(-z)), 5: np.log1p(np.exp(z))}[a]


def _dact(h, a):
    return {0: np.ones_like(h), 1: (h > 0) * 1.0, 2: np.where(h > 0, 1.0, 0.01), 3: 1 - h * h,
            4: h * (1 - h), 5: 1 - np.exp(-h)}[a]


def _c(t, prec, dev):  # fp32 cpu tensor -> compute-type tensor on device
    return t.to(dev).to(ops.compute_dtype(prec)).contiguous()


def _f64(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def _tol(prec, scale):
    return (2e-5 if prec == L.PREC_F32 else 1e-2) * scale + 1e-6


SHAPES_FWD = [(70, 40, 24, 1), (200, 150, 100, 2), (33, 2, 4, 0), (300, 16, 130, 3), (129, 200, 72, 4),
              (64, 33, 9, 5), (1, 5, 3, 1)]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,N,K,act", SHAPES_FWD)
def test_fc_forward(backend, prec, M, N, K, act):
    g = torch.Generator().manual_seed(M * 1000 + N)
    dev = backend.device
    x = _c(torch.randn(M, K, generator=g), prec, dev)
    w = _c(torch.randn(N, K, generator=g) * 0.3, prec, dev)
    b = torch.randn(N, generator=g).to(dev)
    cd = ops.compute_dtype(prec)
    y = torch.zeros(M, N, dtype=cd, device=dev)
    y32 = torch.zeros(M, N, device=dev)
    yt = torch.zeros(N, M, dtype=cd, device=dev)
    ops.fc_forward(x, w, b, act, prec, y=y, y32=y32, yt=yt)
    ref = _act(_f64(x) @ _f64(w).T + _f64(b), act)
    scale = np.abs(ref).max()
    assert np.abs(_f64(y32) - ref).max() <= 2e-5 * scale + 1e-6  # fp32 output: accumulation only
    assert np.abs(_f64(y) - ref).max() <= _tol(prec, scale)
    assert np.abs(_f64(yt).T - ref).max() <= _tol(prec, scale)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,N_out,K_in,act", [(70, 40, 24, 1), (200, 16, 150, 2), (100, 150, 20, 3),
                                              (65, 3, 200, 0), (130, 130, 130, 4)])
def test_fc_dgrad(backend, prec, M, N_out, K_in, act):
    g = torch.Generator().manual_seed(7 + M)
    dev = backend.device
    dz = _c(torch.randn(M, N_out, generator=g), prec, dev)
    wt = _c((torch.randn(N_out, K_in, generator=g) * 0.3).t().contiguous(), prec, dev)
    ht = _c(torch.rand(K_in, M, generator=g) * 2 - 1, prec, dev)
    cd = ops.compute_dtype(prec)
    dx = torch.zeros(M, K_in, dtype=cd, device=dev)
    dx32 = torch.zeros(M, K_in, device=dev)
    dxt = torch.zeros(K_in, M, dtype=cd, device=dev)
    ops.fc_dgrad(dz, wt, ht, act, prec, dx=dx, dx32=dx32, dxt=dxt)
    ref = (_f64(dz) @ _f64(wt).T) * _dact(_f64(ht).T, act)
    scale = np.abs(ref).max()
    assert np.abs(_f64(dx32) - ref).max() <= 2e-5 * scale + 1e-6
    assert np.abs(_f64(dx) - ref).max() <= _tol(prec, scale)
    assert np.abs(_f64(dxt).T - ref).max() <= _tol(prec, scale)
    # no mask (network input / SAC action gradient)
    ops.fc_dgrad(dz, wt, None, 0, prec, dx=None, dx32=dx32, dxt=None)
    assert np.abs(_f64(dx32) - _f64(dz) @ _f64(wt).T).max() <= 2e-5 * scale + 1e-6


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,N_out,K_in,bias", [(300, 40, 24, True), (1000, 150, 130, True), (700, 16, 300, True),
                                               (700, 2, 4, True), (129, 140, 8, False), (64, 1, 40, True)])
def test_fc_wgrad(backend, prec, M, N_out, K_in, bias):
    g = torch.Generator().manual_seed(11 + M)
    dev = backend.device
    dzt = _c(torch.randn(N_out, M, generator=g), prec, dev)
    xt = _c(torch.randn(K_in, M, generator=g), prec, dev)
    dw = torch.full((N_out, K_in), 7.0, device=dev)
    db = torch.full((N_out,), 7.0, device=dev) if bias else None
    ws = torch.empty(ops.fc_wgrad_workspace_bytes(N_out, K_in, M, prec) // 4 + 4, device=dev)
    ops.fc_wgrad(dzt, xt, dw, db, ws, prec)
    ref = _f64(dzt) @ _f64(xt).T
    assert np.abs(_f64(dw) - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-5
    if bias:
        refb = _f64(dzt).sum(1)
        assert np.abs(_f64(db) - refb).max() <= 2e-5 * (np.abs(_f64(dzt)).sum(1).max()) + 1e-5


def test_fc_wgrad_is_deterministic(backend):
    g = torch.Generator().manual_seed(3)
    dev = backend.device
    dzt = torch.randn(48, 900, generator=g).to(dev)
    xt = torch.randn(70, 900, generator=g).to(dev)
    outs = []
    for _ in range(2):
        dw = torch.zeros(48, 70, device=dev)
        ws = torch.empty(ops.fc_wgrad_workspace_bytes(48, 70, 900, L.PREC_F32) // 4 + 4, device=dev)
        ops.fc_wgrad(dzt, xt, dw, None, ws, L.PREC_F32)
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("src_dt,dst_dt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                           (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)])
def test_transpose_cast(backend, src_dt, dst_dt):
    dev = backend.device
    src = torch.randn(77, 45, generator=torch.Generator().manual_seed(1)).to(src_dt).to(dev)
    dst = torch.zeros(77, 45, dtype=dst_dt, device=dev)
    dst_t = torch.zeros(45, 77, dtype=dst_dt, device=dev)
    ops.transpose_cast(src, dst, dst_t)
    want = src.float().cpu().to(dst_dt)  # torch's RNE cast == v_cvt_pk_bf16_f32
    assert torch.equal(dst.cpu(), want)
    assert torch.equal(dst_t.cpu(), want.t().contiguous())


@pytest.mark.parametrize("M,N,K,act", [(2088, 200, 1024, "relu"), (2304, 1024, 1056, "linear")])
def test_large_bf16_shapes_take_the_dma_kernel_bit_identically(backend, M, N, K, act):
    if backend.name == "emu" and N > 256:
        pytest.skip("the interpreter runs the smaller case only (this one is ~2.5e9 MACs)")
    """bf16 forward / dgrad of large shapes run on the 256x256 LDS-DMA kernel (rg_gemm.h); a leading
    dimension that is not a multiple of 8 elements forces the 128x128 kernel for the same data: the two
    must agree bit for bit (every accumulator sees its K chunks in the same order) — M and N tails,
    bias, activation, fp32 + bf16 + transposed outputs included"""
    dev = backend.device
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev)
    wide = torch.zeros(M, K + 4, dtype=torch.bfloat16)
    wide[:, :K] = x
    x_dev, x_odd = x.to(dev).contiguous(), wide.to(dev)[:, :K]  # same values, leading dimension K + 4
    assert x_odd.stride(0) % 8 != 0
    w_dev = w.to(dev).contiguous()
    outs = []
    for xin in (x_dev, x_odd):
        y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        y32 = torch.zeros(M, N, device=dev)
        yt = torch.zeros(N, M, dtype=torch.bfloat16, device=dev)
        ops.fc_forward(xin, w_dev, bias, L.ACT[act], L.PREC_BF16, y=y, y32=y32, yt=yt)
        outs.append((y, y32, yt))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    ref = _act((x.double() @ w.double().t() + bias.cpu().double()).numpy(), L.ACT[act])
    assert np.abs(outs[0][1].cpu().double().numpy() - ref).max() <= 2e-3 * (1 + np.abs(ref).max())
    if backend.name == "emu":
        return  # forward covers the kernel on the interpreter; dgrad shares everything but the epilogue functor
    # dgrad: dx = dz . wt^T-layout, masked by act'(h) from the transposed activation copy
    dz = (torch.randn(M, N, generator=g) / N ** 0.5).to(torch.bfloat16)
    wt = w.t().contiguous()  # [K, N] = W^T, K-contiguous along N
    dzw = torch.zeros(M, N + 4, dtype=torch.bfloat16)
    dzw[:, :N] = dz
    ht = (torch.randn(K, M, generator=g)).to(torch.bfloat16).to(dev)
    res = []
    for dzin in (dz.to(dev).contiguous(), dzw.to(dev)[:, :N]):
        dx = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
        dx32 = torch.zeros(M, K, device=dev)
        ops.fc_dgrad(dzin, wt.to(dev), ht, L.ACT["relu"], L.PREC_BF16, dx=dx, dx32=dx32)
        res.append((dx, dx32))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,N,K,act", [(300, 150, 100, "relu"), (257, 131, 72, "tanh"), (128, 64, 64, "linear")])
def test_swapped_accumulator_epilogue_is_bit_identical(backend, M, N, K, act):
    """a bf16 forward that wants no transposed copy runs with the MFMA operands swapped (lane = output
    row, 16-byte fp32 / 8-byte bf16 row stores); asking for the transposed copy too selects the other
    epilogue.  Same products in the same order: y and y32 must agree bit for bit (tails, odd N, bias and
    activation included)."""
    dev = backend.device
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)

    def run(with_yt):
        ldy = (N + 7) // 8 * 8  # row pitch a multiple of 16 bytes, as the engine allocates it
        y = torch.zeros(M, ldy, dtype=torch.bfloat16, device=dev)[:, :N]
        y32 = torch.zeros(M, ldy, device=dev)[:, :N]
        yt = torch.zeros(N, M, dtype=torch.bfloat16, device=dev) if with_yt else None
        ops.fc_forward(x, w, bias, L.ACT[act], L.PREC_BF16, y=y, y32=y32, yt=yt)
        return y, y32

    (ya, y32a), (yb, y32b) = run(False), run(True)
    assert torch.equal(ya, yb) and torch.equal(y32a, y32b)
    ref = _act((x.cpu().double() @ w.cpu().double().t() + bias.cpu().double()).numpy(), L.ACT[act])
    assert np.abs(y32a.cpu().double().numpy() - ref).max() <= 2e-3 * (1 + np.abs(ref).max())
    # fp32 output alone (the last layer of an inference forward)
    only32 = torch.zeros(M, N, device=dev)
    ops.fc_forward(x, w, bias, L.ACT[act], L.PREC_BF16, y32=only32)
    assert torch.equal(only32, y32a.contiguous())
