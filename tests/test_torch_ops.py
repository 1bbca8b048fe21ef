"""torch.ops.reagent_amd.* (reagent_amd/torch_ops.py): the registered custom ops against plain torch on the same inputs."""
import pytest
import torch

import reagent_amd.torch_ops as T  # noqa: F401  (registers the library)

R = torch.ops.reagent_amd


@pytest.fixture
def dev(backend):
    if backend.name == "emu":
        import emu_backend

        emu_backend.serve_torch_ops_on_cpu()  # the interpreter backend serves CPU tensors, registered from the test side
    return backend.device


def test_ops_are_registered_for_the_gpu_only():
    for name in ("mlp_forward", "adam_step_", "soft_update_", "gaussian_head", "dueling_combine", "max_q_values_with_target"):
        assert hasattr(R, name)
    schema = R.adam_step_.default._schema
    assert schema.arguments[0].alias_info is not None and schema.arguments[0].alias_info.is_write  # in-place on param


def test_mlp_forward_and_optimizer_ops(dev):
    gen = torch.Generator().manual_seed(2)
    ws = [torch.randn(24, 10, generator=gen).to(dev) * 0.3, torch.randn(5, 24, generator=gen).to(dev) * 0.3]
    bs = [torch.randn(24, generator=gen).to(dev) * 0.1, torch.randn(5, generator=gen).to(dev) * 0.1]
    x = torch.randn(33, 10, generator=gen).to(dev)
    q = R.mlp_forward(x, ws, bs, ["relu", "linear"], "f32")
    ref = torch.relu(x.cpu() @ ws[0].cpu().T + bs[0].cpu()) @ ws[1].cpu().T + bs[1].cpu()
    assert q.shape == (33, 5) and (q.cpu() - ref).abs().max() <= 1e-5

    p = torch.randn(1000, generator=gen)
    g = torch.randn(1000, generator=gen)
    pt = p.clone().requires_grad_()
    opt = torch.optim.Adam([pt], lr=1e-2)
    pd, m, v = p.to(dev), torch.zeros(1000, device=dev), torch.zeros(1000, device=dev)
    for step in (1, 2, 3):
        pt.grad = g.clone()
        opt.step()
        R.adam_step_(pd, g.to(dev), m, v, 1e-2, 0.9, 0.999, 1e-8, 0.0, step)
    assert (pd.cpu() - pt.detach()).abs().max() <= 1e-6
    tgt = torch.zeros(1000, device=dev)
    R.soft_update_(tgt, pd, 0.25)
    assert torch.allclose(tgt.cpu(), 0.25 * pd.cpu(), atol=1e-7)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_mlp_forward_sees_weights_written_by_the_in_place_ops(dev, precision):
    """mlp_forward -> adam_step_(w) / soft_update_(w) -> mlp_forward: the staged bf16 fragments of the cached stack must
    follow weights that this package's own in-place ops wrote (they bump no torch version counter); `.detach()` views
    of the weights included"""
    gen = torch.Generator().manual_seed(5)
    dims = [16, 256, 256, 4]
    ws = [(torch.randn(o, i, generator=gen) * 0.2).to(dev) for i, o in zip(dims, dims[1:])]
    bs = [(torch.randn(o, generator=gen) * 0.1).to(dev) for o in dims[1:]]
    x = torch.randn(40, dims[0], generator=gen).to(dev)
    acts = ["relu", "relu", "linear"]
    views = [w.detach() for w in ws]
    q0 = R.mlp_forward(x, views, bs, acts, precision).clone()
    g = torch.randn(ws[1].shape, generator=gen).to(dev)
    m, v = torch.zeros_like(ws[1]), torch.zeros_like(ws[1])
    R.adam_step_(ws[1], g, m, v, 5e-2, 0.9, 0.999, 1e-8, 0.0, 1)
    q1 = R.mlp_forward(x, views, bs, acts, precision).clone()
    fresh = R.mlp_forward(x, [w.clone() for w in ws], [b.clone() for b in bs], acts, precision)
    assert (q1 - q0).abs().max() > 1e-3  # the step moved every weight of layer 1 by lr
    assert torch.equal(q1, fresh)
    R.soft_update_(ws[2], torch.zeros_like(ws[2]), 0.5)
    q2 = R.mlp_forward(x, views, bs, acts, precision)
    fresh = R.mlp_forward(x, [w.clone() for w in ws], [b.clone() for b in bs], acts, precision)
    assert torch.equal(q2, fresh) and not torch.equal(q2, q1)


def test_head_ops(dev):
    gen = torch.Generator().manual_seed(4)
    B, A = 50, 6
    qo, qt = torch.randn(B, A, generator=gen), torch.randn(B, A, generator=gen)
    mask = (torch.rand(B, A, generator=gen) > 0.3).float()
    mask[:, 0] = 1
    for double_q in (True, False):
        mq, idx = R.max_q_values_with_target(qo.to(dev), qt.to(dev), mask.to(dev), double_q)
        pen = -1e9 * (1 - mask)
        ref_idx = ((qo if double_q else qt) + pen).argmax(1, keepdim=True)
        assert torch.equal(idx.cpu(), ref_idx) and torch.equal(mq.cpu(), qt.gather(1, ref_idx))
    val, adv = torch.randn(B, 1, generator=gen), torch.randn(B, A, generator=gen)
    q = R.dueling_combine(val.to(dev), adv.to(dev), A, 1)
    assert (q.cpu() - (val + adv - adv.mean(1, keepdim=True))).abs().max() <= 1e-6
    ls, noise = torch.randn(B, 2 * A, generator=gen), torch.randn(B, A, generator=gen)
    action, lp = R.gaussian_head(ls.to(dev), noise.to(dev))
    ref_a = torch.tanh(ls[:, :A] + noise * ls[:, A:].clamp(-2, 2).exp()).clamp(-1 + 1e-6, 1 - 1e-6)
    assert action.shape == (B, A) and lp.shape == (B, 1) and (action.cpu() - ref_a).abs().max() <= 1e-6


@pytest.mark.gpu
def test_cpu_tensors_are_refused_by_the_dispatcher():
    """no CPU kernel is registered by the product: the dispatcher raises (run where no test hook touched the library)"""
    import subprocess
    import sys

    code = ("import torch, reagent_amd.torch_ops\n"
            "try:\n    torch.ops.reagent_amd.soft_update_(torch.zeros(4), torch.ones(4), 0.5)\n"
            "except NotImplementedError as e:\n    print('refused')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/root/repo")
    assert "refused" in out.stdout, out.stderr[-500:]
