"""Edge cases of the round-2 kernels: empty and minimal batches, widths off the wave size, the widest shapes."""
import numpy as np
import pytest
import torch

import reagent_amd._lib as L
from reagent_amd import ops


def test_policy_input_maker_empty_and_single_column(backend):
    import collections

    from reagent_amd.preprocessing import PolicyNetworkInputMaker

    Batch = collections.namedtuple("Batch", "state next_state action next_action reward terminal log_prob")
    d = backend.device
    for B, A in ((0, 3), (1, 1), (5, 1)):
        g = torch.Generator().manual_seed(B + A)
        lo, hi = np.full(A, -2.0, dtype=np.float32), np.full(A, 3.0, dtype=np.float32)
        b = Batch(state=torch.zeros(B, 2, device=d), next_state=torch.zeros(B, 2, device=d),
                  action=(torch.rand(B, A, generator=g) * 5 - 2).to(d), next_action=(torch.rand(B, A, generator=g) * 5 - 2).to(d),
                  reward=torch.zeros(B, 1, device=d), terminal=(torch.rand(B, 1, generator=g) > 0.5).to(d),
                  log_prob=-torch.rand(B, 1, generator=g).to(d))
        out = PolicyNetworkInputMaker(lo, hi)(b)
        assert out.action.float_features.shape == (B, A) and out.not_terminal.shape == (B, 1)
        from reagent_amd.core.parameters import CONTINUOUS_TRAINING_ACTION_RANGE as R

        tl, th = torch.tensor(R[0]), torch.tensor(R[1])  # fp32 scalars, as the reference's maker holds them
        ref = ((b.action.cpu() - torch.tensor(lo)) / (torch.tensor(hi) - torch.tensor(lo))) * (th - tl) + tl
        assert torch.equal(out.action.float_features.cpu(), ref)
        assert torch.equal(out.not_terminal.cpu(), 1.0 - b.terminal.float().cpu())


@pytest.mark.parametrize("n", [1, 3, 65, 2048])
def test_layer_norm_odd_widths(backend, n):
    B = 5
    g = torch.Generator().manual_seed(n)
    z, gamma, beta = torch.randn(B, n, generator=g), torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    ref = torch.nn.functional.layer_norm(z, (n,), gamma, beta, 1e-5)
    d = backend.device
    y = torch.empty(B, n, device=d)
    mean, rstd = torch.empty(B, device=d), torch.empty(B, device=d)
    ops.layer_norm_forward(z.to(d), gamma.to(d), beta.to(d), 1e-5, L.ACT["linear"], y32=y, mean=mean, rstd=rstd)
    assert (y.cpu() - ref).abs().max() <= 3e-6 * max(1.0, ref.abs().max().item())
    with pytest.raises(L.ReagentHipError):  # rows wider than a wave holds in registers are refused, not truncated
        ops.layer_norm_forward(torch.zeros(2, 2049, device=d), torch.ones(2049, device=d), torch.zeros(2049, device=d), 1e-5,
                               L.ACT["linear"], y32=torch.empty(2, 2049, device=d))


def test_dueling_combine_at_the_widest_qr_shape(backend):
    B, A, N = 9, 16, 200
    g = torch.Generator().manual_seed(2)
    val, adv = torch.randn(B, N, generator=g), torch.randn(B, A * N, generator=g)
    d = backend.device
    q = torch.empty(B, A * N, device=d)
    ops.dueling_combine(val.to(d), adv.to(d), A, N, q)
    a3 = adv.view(B, A, N).double()
    ref = (val.view(B, 1, N).double() + a3 - a3.mean(dim=(1, 2), keepdim=True)).reshape(B, -1)
    assert (q.cpu().double() - ref).abs().max() <= 2e-6


def test_sac_kld_minimal_batch(backend):
    d = backend.device
    x = torch.tensor([[0.2, -0.4], [0.6, 0.1]])
    mu, s2 = torch.tensor([0.0, 0.1]), torch.tensor([0.5, 0.7])
    coef, terms, out = torch.empty(4, device=d), torch.empty(2, device=d), torch.empty(1, device=d)
    ops.sac_kld(x.to(d), False, mu.to(d), s2.to(d), 0.3, coef, terms, out, None)
    m, v = x.mean(0), x.var(0)
    ref = 0.5 * ((v + (m - mu) ** 2) / s2 - 1 + s2.log() - v.log()).sum()
    assert abs(out.item() - ref.item()) <= 1e-6 * max(1.0, abs(ref.item()))
    with pytest.raises(L.ReagentHipError):  # the unbiased variance needs two rows
        ops.sac_kld(x[:1].to(d), False, mu.to(d), s2.to(d), 0.3, coef, terms, out, None)
