"""reagent_amd.training.parameters: the generated `*TrainerParameters` dataclasses against the reference's
(reagent/training/parameters.py:28-128) — same field names in the same order, same defaults (the factories produce equal
RLParameters / a default Adam union / an empty action list) — and the round trip a model manager performs:
`Trainer(networks..., **trainer_param.asdict())`.  The reference side is rebuilt in a subprocess (oracle/stubs.py) from
the reference's own decorator arguments, read out of its source with `ast` (the module itself imports every trainer of
the repository, most of which need packages that are not installed)."""
import dataclasses
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["DQNTrainerParameters", "QRDQNTrainerParameters", "C51TrainerParameters", "SACTrainerParameters",
         "TD3TrainerParameters", "CRRTrainerParameters"]


def _norm(v):
    if v is None or isinstance(v, (bool, int, float, str)):
        return repr(float(v)) if isinstance(v, (int, float)) and not isinstance(v, bool) else repr(v)
    if isinstance(v, (list, tuple)):
        return [_norm(x) for x in v]
    if dataclasses.is_dataclass(v) and type(v).__name__ == "RLParameters":
        return ["RLParameters"] + [[f.name, _norm(getattr(v, f.name))] for f in dataclasses.fields(v)]
    return type(v).__name__


def _describe(cls):
    inst = cls() if all(f.default is not dataclasses.MISSING or f.default_factory is not dataclasses.MISSING
                        for f in dataclasses.fields(cls)) else None
    return [[f.name, _norm(getattr(inst, f.name)) if inst is not None else "required"] for f in dataclasses.fields(cls)]


@pytest.mark.skipif(not os.path.isdir("/root/reference/reagent"), reason="needs the reference tree (build container)")
def test_parameter_classes_equal_the_reference():
    code = textwrap.dedent("""
        import ast, dataclasses, importlib, json, sys
        sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
        from oracle import stubs
        stubs.install()
        from reagent.core.configuration import make_config_class
        from test_trainer_parameters import NAMES, _describe
        src = open(stubs.REFERENCE_ROOT + "/reagent/training/parameters.py").read()
        imports = {}
        out = {}
        tree = ast.parse(src)
        for node in tree.body:
            if isinstance(node, ast.ImportFrom) and node.level == 1:
                for a in node.names:
                    imports[a.name] = "reagent.training." + node.module
            if isinstance(node, ast.ClassDef) and node.name in NAMES:
                call = node.decorator_list[0]
                trainer = call.args[0].value.id                     # <Trainer>.__init__
                blocklist = ast.literal_eval([k.value for k in call.keywords if k.arg == "blocklist"][0])
                T = getattr(importlib.import_module(imports[trainer]), trainer)
                out[node.name] = _describe(make_config_class(T.__init__, blocklist=blocklist)(type(node.name, (), {})))
        print("RESULT" + json.dumps(out))
    """) % (ROOT, ROOT)
    env = {k: v for k, v in os.environ.items() if k != "REAGENT_AMD_OWN_TYPES"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
    assert r.returncode == 0 and lines, r.stdout[-3000:] + r.stderr[-3000:]
    import json

    ref = json.loads(lines[-1][len("RESULT"):])
    import reagent_amd.training.parameters as P

    assert sorted(ref) == sorted(NAMES)
    for n in NAMES:
        own = json.loads(json.dumps(_describe(getattr(P, n))))
        assert own == ref[n], (n, own, ref[n])


def test_parameters_spread_into_the_trainers(emu_lib):
    """what DiscreteDQN.build_trainer / SAC.build_trainer do with a trainer_param (discrete_dqn.py:105-115, sac.py:104-112)"""
    from reagent_amd.core.parameters import EvaluationParameters, NormalizationData, NormalizationParameters as NP, RLParameters
    from reagent_amd.net_builder import continuous_actor, discrete_dqn, parametric_dqn
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.training import DQNTrainer, SACTrainer
    from reagent_amd.training.parameters import DQNTrainerParameters, SACTrainerParameters

    S = NormalizationData({i: NP("CONTINUOUS", mean=0.0, stddev=1.0) for i in range(8)})
    A = NormalizationData({100 + i: NP("CONTINUOUS_ACTION", min_value=-1.0, max_value=1.0) for i in range(2)})
    tp = DQNTrainerParameters(actions=["l", "r"], rl=RLParameters(gamma=0.5), optimizer=Optimizer__Union.default(lr=0.01))
    assert tp.rl.gamma == 0.5 and tp.double_q_learning and tp.minibatch_size == 1024  # readable before any trainer exists
    q = discrete_dqn.FullyConnected(sizes=[16], activations=["relu"]).build_q_network(None, S, len(tp.actions))
    tr = DQNTrainer(q_network=q, q_network_target=q.get_target_network(), reward_network=None,
                    evaluation=EvaluationParameters(calc_cpe_in_training=False), **tp.asdict())
    assert tr.gamma == 0.5 and tr.num_actions == 2 and tr.double_q_learning
    assert tr.configure_optimizers()[0]["optimizer"].param_groups[0]["lr"] == 0.01
    sp = SACTrainerParameters(entropy_temperature=0.2)
    actor = continuous_actor.GaussianFullyConnected(sizes=[16], activations=["relu"]).build_actor(None, S, A)
    critic = parametric_dqn.FullyConnected(sizes=[16], activations=["relu"])
    sac = SACTrainer(actor_network=actor, q1_network=critic.build_q_network(S, A), q2_network=critic.build_q_network(S, A),
                     **sp.asdict())
    assert float(sac.entropy_temperature) == pytest.approx(0.2) and sac.gamma == RLParameters().gamma
    assert [f.name for f in dataclasses.fields(DQNTrainerParameters)] == [
        "actions", "rl", "double_q_learning", "bcq", "minibatch_size", "minibatches_per_step", "optimizer"]
