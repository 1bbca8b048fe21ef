"""The input maker a trainer needs, chosen from the annotation of its `train_step_gen`
(reagent/gym/preprocessors/trainer_preprocessor.py:32-69, :476-484).  The makers themselves are the one-launch HIP
forms in reagent_amd/preprocessing/trainer_preprocessor.py, whose `create_for_env` reads the action space without
importing gym (a discrete space has `n`, a box `low` / `high`).
"""
import inspect

import torch

from ...core import types as rlt
from ...preprocessing.trainer_preprocessor import DiscreteDqnInputMaker, PolicyNetworkInputMaker  # noqa: F401


# trainer_preprocessor.py:478-484 minus the memory-network / parametric / slate inputs (not on the path)
REPLAY_BUFFER_MAKER_MAP = {
    rlt.DiscreteDqnInput: DiscreteDqnInputMaker,
    rlt.PolicyNetworkInput: PolicyNetworkInputMaker,
}


def make_trainer_preprocessor(trainer, device: torch.device, env, maker_map):
    """trainer_preprocessor.py:32-57: `training_batch` must be the first parameter of `train_step_gen` and carry its
    type; an input type without a maker is a KeyError, as there."""
    sig = inspect.signature(trainer.train_step_gen)
    assert list(sig.parameters.keys())[0] == "training_batch", f"{sig.parameters} doesn't have training batch in first position."
    training_batch_type = sig.parameters["training_batch"].annotation
    assert training_batch_type != inspect.Parameter.empty
    maker = maker_map[training_batch_type].create_for_env(env)

    def trainer_preprocessor(batch):
        return maker(batch).to(device)  # the batch of a device-resident buffer is already there: no copy

    trainer_preprocessor.maker = maker  # the datasets look here for the one-launch sampler + maker form
    return trainer_preprocessor


def make_replay_buffer_trainer_preprocessor(trainer, device: torch.device, env):
    return make_trainer_preprocessor(trainer, device, env, REPLAY_BUFFER_MAKER_MAP)
