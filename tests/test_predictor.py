"""Serving wrappers: DiscreteDqnWithPreprocessor == q_network(Preprocessor(x, presence)),
checked against the oracle's CPU restatement of both stages; the HIP-graph replay (GPU only) returns
the eager result bit for bit."""
import pytest
import torch

import reagent_amd._lib as L
from oracle import restated as R
from reagent_amd.core.parameters import NormalizationParameters as NP
from reagent_amd.models import FullyConnectedDQN
from reagent_amd.prediction import DiscreteDqnPredictorWrapper, DiscreteDqnWithPreprocessor, ServingFeatureData
from reagent_amd.preprocessing import Preprocessor


def _build(device, S=12, A=5):
    g = torch.Generator().manual_seed(3)
    norm = {}
    for i in range(S):
        norm[i] = (NP(feature_type="CONTINUOUS", mean=torch.randn(1, generator=g).item(), stddev=1.5) if i % 2 == 0
                   else NP(feature_type="BINARY"))
    torch.manual_seed(0)
    q = FullyConnectedDQN(S, A, [32, 16], ["relu", "relu"]).to(device)
    pre = Preprocessor(norm, device=device)
    names = [f"a{i}" for i in range(A)]
    return norm, q, DiscreteDqnPredictorWrapper(DiscreteDqnWithPreprocessor(q, pre), names), names


def _inputs(B, S, device, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, S, generator=g)
    presence = (torch.rand(B, S, generator=g) > 0.1).to(torch.uint8)
    return x.to(device), presence.to(device)


def test_predictor_equals_preprocessor_then_network(backend):
    norm, q, wrapper, names = _build(backend.device)
    x, presence = _inputs(37, 12, backend.device)
    got_names, qv = wrapper(ServingFeatureData(float_features_with_presence=(x, presence)))
    assert got_names == names and qv.shape == (37, 5)
    feats = R.preprocess(norm, x.cpu(), presence.cpu())  # oracle: reference Preprocessor restated
    params = [p.detach().cpu() for p in q.parameters()]
    want = R.fc_forward(params, ["relu", "relu", "linear"], feats)
    assert (qv.cpu() - want).abs().max() <= 1e-4
    with pytest.raises(NotImplementedError):
        wrapper(ServingFeatureData(float_features_with_presence=(x, presence), id_list_features={1: (x, x)}))
    proto = wrapper.dqn_with_preprocessor.input_prototype()
    assert proto[0].float_features_with_presence[0].shape == (1, 12)


@pytest.mark.gpu
def test_hip_graph_replay_matches_eager():
    L.lib()
    dev = torch.device("cuda")
    _, _, wrapper, names = _build(dev)
    for B in (1, 64, 300):
        x, presence = _inputs(B, 12, dev, seed=B)
        state = ServingFeatureData(float_features_with_presence=(x, presence))
        eager = wrapper(state)[1].clone()
        wrapper.capture(B)
        for _ in range(3):
            got_names, replayed = wrapper(state)
            assert got_names == names and torch.equal(replayed, eager)
        x2, p2 = _inputs(B, 12, dev, seed=B + 1000)  # same graph, new request
        state2 = ServingFeatureData(float_features_with_presence=(x2, p2))
        want2 = wrapper.dqn_with_preprocessor(state2)
        assert torch.equal(wrapper(state2)[1], want2)


def test_predictor_matches_the_reference_wrapper(backend):
    """golden predictor_dqn: the reference's own DiscreteDqnPredictorWrapper (jit-traced DiscreteDqnWithPreprocessor over a
    Preprocessor with every feature type and a FullyConnectedDQN, prediction/predictor_wrapper.py:94-152) on a batch with
    missing features: same action names, Q-values within 1e-4"""
    from types import SimpleNamespace

    from golden_util import Golden

    g = Golden("predictor_dqn")
    c = g.cfg
    norm = {int(k): SimpleNamespace(**v) for k, v in c["norm"].items()}
    pre = Preprocessor(norm, device=backend.device)
    assert pre.sorted_features == c["sorted_features"]
    q = FullyConnectedDQN(c["state_dim"], c["num_actions"], c["sizes"], c["activations"])
    with torch.no_grad():
        for p, ref in zip(q.parameters(), g.seq("param_")):
            p.copy_(ref)
    wrapper = DiscreteDqnPredictorWrapper(DiscreteDqnWithPreprocessor(q.to(backend.device), pre), c["action_names"])
    state = ServingFeatureData(float_features_with_presence=(g.t("x").to(backend.device), g.t("presence").to(backend.device)))
    names, qv = wrapper(state)
    assert list(names) == c["action_names"]
    ref = g.t("q_values")
    assert qv.shape == ref.shape and (qv.cpu() - ref).abs().max() <= 1e-4
    if str(backend.device).startswith("cuda"):  # the captured HIP graph replays the same numbers
        wrapper.capture(c["batch"])
        assert torch.equal(wrapper(state)[1], qv)
