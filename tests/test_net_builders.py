"""reagent_amd.net_builder (§8(b)'s "called by" column): every builder against the reference's builder of the same
name — constructor fields and defaults, `build_*` parameter lists, and the networks they build for the same
hyper-parameters and normalization data (class, parameter names and shapes; an ENUM feature widens the input).  The
comparison runs where the reference tree is (the build container), in a subprocess so that `reagent` imports through
oracle/stubs.py; the layouts it pins are also committed (tests/golden/net_builders.json) and checked everywhere."""
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "net_builders.json")

# (namespace, class name, reference module, builder kwargs, build method, build arguments as an expression over S / A)
CASES = [
    ("discrete_dqn", "FullyConnected", "discrete_dqn.fully_connected", dict(sizes=[16, 8], activations=["relu", "tanh"], use_batch_norm=True),
     "build_q_network", "(None, S, 3)"),
    ("discrete_dqn", "Dueling", "discrete_dqn.dueling", dict(sizes=[16, 8], activations=["relu", "relu"]), "build_q_network", "(None, S, 3)"),
    ("quantile_dqn", "Quantile", "quantile_dqn.quantile", dict(sizes=[12], activations=["leaky_relu"]), "build_q_network", "(S, 3, 7)"),
    ("quantile_dqn", "DuelingQuantile", "quantile_dqn.dueling_quantile", dict(), "build_q_network", "(S, 2, 5)"),
    ("categorical_dqn", "Categorical", "categorical_dqn.categorical", dict(sizes=[10, 10], activations=["relu", "relu"]),
     "build_q_network", "(S, 3, 11, -5, 5)"),
    ("continuous_actor", "GaussianFullyConnected", "continuous_actor.gaussian_fully_connected", dict(use_layer_norm=True),
     "build_actor", "(None, S, A)"),
    ("continuous_actor", "FullyConnected", "continuous_actor.fully_connected", dict(sizes=[8], activations=["tanh"], action_activation="linear"),
     "build_actor", "(None, S, A)"),
    ("discrete_actor", "FullyConnected", "discrete_actor.fully_connected", dict(), "build_actor", "(S, 4)"),
    ("parametric_dqn", "FullyConnected", "parametric_dqn.fully_connected", dict(use_layer_norm=True, final_activation="tanh"),
     "build_q_network", "(S, A, 2)"),
    ("value", "FullyConnected", "value.fully_connected", dict(sizes=[6, 6], activations=["relu", "relu"], use_layer_norm=True),
     "build_value_network", "(S,)"),
]
NORM = dict(S={1: dict(feature_type="CONTINUOUS", mean=0.5, stddev=2.0), 2: dict(feature_type="ENUM", possible_values=[3, 5, 7]),
               4: dict(feature_type="BINARY"), 9: dict(feature_type="QUANTILE", quantiles=[0.0, 1.0, 2.0])},
            A={10: dict(feature_type="CONTINUOUS_ACTION", min_value=-1.0, max_value=1.0),
               11: dict(feature_type="CONTINUOUS_ACTION", min_value=0.0, max_value=2.0)})

SCRIPT = """
    import dataclasses, importlib, inspect, json, sys
    sys.path.insert(0, %(root)r)
    REF = %(ref)r
    if REF:
        from oracle import stubs
        stubs.install(); stubs.install_gym()
        from reagent.core.parameters import NormalizationData, NormalizationParameters
    else:
        from reagent_amd.core.parameters import NormalizationData, NormalizationParameters
    CASES, NORM = %(cases)r, %(norm)r
    S, A = (NormalizationData({k: NormalizationParameters(**v) for k, v in NORM[n].items()}) for n in "SA")
    out = {}
    for ns, name, ref_mod, kw, method, args in CASES:
        if REF:
            cls = getattr(importlib.import_module("reagent.net_builder." + ref_mod), name)
        else:
            cls = getattr(importlib.import_module("reagent_amd.net_builder." + ns), name)
        b = cls(**kw)
        net = getattr(b, method)(*eval(args))
        fields = [(f.name, repr(getattr(cls(), f.name))) for f in dataclasses.fields(cls)]
        params = [n for n in inspect.signature(getattr(cls, method)).parameters if n != "self"]
        out[ns + "." + name] = dict(net=type(net).__name__, fields=fields, build_params=params,
                                    state=[(k, list(v.shape)) for k, v in net.state_dict().items()],
                                    action_preprocessing=getattr(b, "default_action_preprocessing", None))
    print("RESULT" + json.dumps(out))
"""


def _run(ref: bool):
    code = textwrap.dedent(SCRIPT) % dict(root=ROOT, ref=ref, cases=CASES, norm=NORM)
    env = {k: v for k, v in os.environ.items() if k != "REAGENT_AMD_OWN_TYPES"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
    assert r.returncode == 0 and lines, r.stdout[-3000:] + r.stderr[-3000:]
    return json.loads(lines[-1][len("RESULT"):])


def _normalize(d):
    return json.loads(json.dumps(d))


def test_builders_build_the_committed_layouts():
    """the layouts the reference's builders produce (recorded by the test below) from this package's builders"""
    want = json.load(open(GOLDEN))
    got = _normalize(_run(ref=False))
    assert set(got) == set(want)
    for k in want:
        assert got[k] == want[k], (k, got[k], want[k])


@pytest.mark.skipif(not os.path.isdir("/root/reference/reagent"), reason="needs the reference tree (build container)")
def test_committed_layouts_are_what_the_reference_builds():
    ref = _normalize(_run(ref=True))
    if os.environ.get("RG_WRITE_GOLDEN") == "1":
        with open(GOLDEN, "w") as f:
            json.dump(ref, f, indent=1, sort_keys=True)
    assert ref == json.load(open(GOLDEN))


def test_built_networks_run_and_errors(emu_lib):
    from reagent_amd.core import types as rlt
    from reagent_amd.core.parameters import NormalizationData, NormalizationParameters as NP
    from reagent_amd.net_builder import continuous_actor, discrete_dqn, get_num_output_features, parametric_dqn

    S = NormalizationData({k: NP(**v) for k, v in NORM["S"].items()})
    A = NormalizationData({k: NP(**v) for k, v in NORM["A"].items()})
    assert get_num_output_features(S.dense_normalization_parameters) == 6  # 1 + 3 (ENUM) + 1 + 1
    q = discrete_dqn.FullyConnected(sizes=[16, 8], activations=["relu", "relu"]).build_q_network(None, S, 3)
    x = torch.randn(5, 6)
    assert q(rlt.FeatureData(float_features=x)).shape == (5, 3)
    critic = parametric_dqn.FullyConnected().build_q_network(S, A)
    assert critic(rlt.FeatureData(x), rlt.FeatureData(torch.rand(5, 2))).shape == (5, 1)
    with pytest.raises(AssertionError):  # one activation per layer, as in every reference builder
        discrete_dqn.Dueling(sizes=[8, 8], activations=["relu"])
    with pytest.raises(NotImplementedError):
        continuous_actor.GaussianFullyConnected(embedding_dim=4)
    # the serving module of the discrete builders: Preprocessor -> Q-network -> (action names, Q-values)
    serving = discrete_dqn.FullyConnected(sizes=[8], activations=["relu"])
    net = serving.build_q_network(None, S, 2)
    module = serving.build_serving_module(net, S, ["a", "b"], None)
    raw = torch.tensor([[0.5, 5.0, 1.0, 1.5]])
    from reagent_amd.prediction.predictor_wrapper import ServingFeatureData

    names, qv = module(ServingFeatureData(float_features_with_presence=(raw, torch.ones_like(raw))))
    assert names == ["a", "b"] and qv.shape == (1, 2)
