"""Drop-in detail of SURVEY §8 a8: inside a ReAgent installation the trainers must be annotated with the
reference's OWN batch classes, because `make_trainer_preprocessor` looks the input maker up by class object
(reagent/gym/preprocessors/trainer_preprocessor.py:39-48).  Runs where the reference tree is present (the build
container), in a subprocess so that `reagent` is importable BEFORE reagent_amd is imported."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/reagent"), reason="needs the reference tree (build container)")
def test_reference_batch_classes_are_used_when_reagent_is_importable():
    code = textwrap.dedent("""
        import inspect, sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from oracle import stubs
        stubs.install()                       # puts /root/reference on sys.path with fakes for uninstalled deps
        import reagent.core.types as ref
        import reagent_amd.core.types as rlt
        assert rlt.USING_REFERENCE_TYPES and rlt.DiscreteDqnInput is ref.DiscreteDqnInput
        assert rlt.PolicyNetworkInput is ref.PolicyNetworkInput and rlt.FeatureData is ref.FeatureData
        from reagent_amd.training import DQNTrainer, SACTrainer
        ann = inspect.signature(DQNTrainer.train_step_gen).parameters["training_batch"].annotation
        assert ann is ref.DiscreteDqnInput, ann     # what make_trainer_preprocessor keys its maker map with
        ann = inspect.signature(SACTrainer.train_step_gen).parameters["training_batch"].annotation
        assert ann is ref.PolicyNetworkInput, ann
        # and a native step runs on a batch of the reference's class (SIMT interpreter backend)
        sys.path.insert(0, %r)
        import emu_backend, torch
        class MP:
            def setattr(self, obj, name, value, raising=True): setattr(obj, name, value)
        emu_backend.install(MP())
        from reagent_amd import synthetic
        from reagent_amd.core.parameters import EvaluationParameters, RLParameters
        from reagent_amd.models import FullyConnectedDQN
        from reagent_amd.optimizer import Optimizer__Union
        q = FullyConnectedDQN(12, 4, [32, 16], ["relu", "relu"])
        tr = DQNTrainer(q, q.get_target_network(), None, actions=list("abcd"), rl=RLParameters(gamma=0.9),
                        optimizer=Optimizer__Union.default(lr=0.01),
                        evaluation=EvaluationParameters(calc_cpe_in_training=False))
        batch = synthetic.to_dqn_input(synthetic.dqn_batch(32, 12, 4, seed=1), "cpu")
        assert type(batch) is ref.DiscreteDqnInput
        loss = tr.train_step_native(batch)
        assert torch.isfinite(loss).all()
        print("ok")
    """) % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests"))
    env = {k: v for k, v in os.environ.items() if k != "REAGENT_AMD_OWN_TYPES"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-1500:] + out.stderr[-3000:]


def test_own_types_by_default_outside_a_reagent_installation():
    import reagent_amd.core.types as rlt

    assert rlt.DiscreteDqnInput.__module__ in ("reagent_amd.core.types", "reagent.core.types")
