"""Trainer parameter classes: the hyper-parameter half of each trainer's constructor as a dataclass — what a ReAgent
config holds as `trainer_param` and a model manager spreads into the trainer next to the networks it built
(`DQNTrainer(q_network=..., **trainer_param.asdict())`, reagent/model_managers/discrete/discrete_dqn.py:105-115).  As in
reagent/training/parameters.py:28-150 the classes are generated from the constructors' own signatures (annotated
parameters minus the networks and evaluation objects), so they cannot drift from the trainers.

The reference's constructors carry their mutable defaults as `field(default_factory=...)` under `@resolve_defaults`;
here such a parameter defaults to None and the constructor builds the default (tests/test_reference_signatures.py).  A
parameter object, though, is read before any trainer exists (`trainer_param.rl.gamma`, `trainer_param.actions`), so the
generated fields get the reference's factories back: RLParameters(), Optimizer__Union.default(), list().
"""
import dataclasses
import inspect
from typing import get_args

from ..core.parameters import RLParameters
from ..optimizer import Optimizer__Union
from .c51_trainer import C51Trainer
from .discrete_crr_trainer import DiscreteCRRTrainer
from .dqn_trainer import DQNTrainer
from .qrdqn_trainer import QRDQNTrainer
from .sac_trainer import SACTrainer
from .td3_trainer import TD3Trainer

_FACTORY_BY_TYPE = {RLParameters: RLParameters, Optimizer__Union: Optimizer__Union.default}


def _factory_for(name, annotation):
    """the default factory of a None-defaulted constructor parameter, from its Optional[...] annotation"""
    if name == "actions":
        return list
    for t in get_args(annotation) or (annotation,):
        if t in _FACTORY_BY_TYPE:
            return _FACTORY_BY_TYPE[t]
    return None


def make_config_class(func, blocklist):
    """decorator: a dataclass whose fields are `func`'s annotated parameters outside `blocklist`
    (reagent/core/configuration.py:40-103), with `asdict()` = the keyword arguments to pass on"""
    blocked = set(blocklist) | {"self"}

    def wrap(cls):
        fields = []
        for p in inspect.signature(func).parameters.values():
            if p.name in blocked or (p.annotation is inspect.Parameter.empty and p.default is inspect.Parameter.empty):
                continue  # (configuration.py:75-83: a parameter needs an annotation or a default to become a field)
            factory = _factory_for(p.name, p.annotation) if p.default in (None, "default") else None
            if factory is not None:
                fields.append((p.name, p.annotation, dataclasses.field(default_factory=factory)))
            elif p.annotation is inspect.Parameter.empty:
                fields.append((p.name, type(p.default), dataclasses.field(default=p.default)))
            elif p.default is inspect.Parameter.empty:
                fields.append((p.name, p.annotation))
            else:
                fields.append((p.name, p.annotation, dataclasses.field(default=p.default)))
        fields.sort(key=lambda f: len(f) == 3)  # required fields first, order otherwise kept (sort is stable)

        def asdict(self):
            return {f.name: getattr(self, f.name) for f in dataclasses.fields(self)}  # shallow: objects stay objects

        return dataclasses.make_dataclass(cls.__name__, fields, namespace={"asdict": asdict, "__doc__": cls.__doc__,
                                                                          "__module__": cls.__module__})

    return wrap


@make_config_class(SACTrainer.__init__, blocklist=["use_gpu", "actor_network", "q1_network", "q2_network", "value_network"])
class SACTrainerParameters:
    """parameters.py:28-33"""


@make_config_class(TD3Trainer.__init__, blocklist=["use_gpu", "actor_network", "q1_network", "q2_network"])
class TD3TrainerParameters:
    """parameters.py:36-41"""


@make_config_class(DiscreteCRRTrainer.__init__, blocklist=[
    "use_gpu", "actor_network", "actor_network_target", "q1_network", "q1_network_target", "reward_network", "q2_network",
    "q2_network_target", "q_network_cpe", "q_network_cpe_target", "metrics_to_score", "evaluation"])
class CRRTrainerParameters:
    """parameters.py:44-62"""


@make_config_class(DQNTrainer.__init__, blocklist=[
    "use_gpu", "q_network", "q_network_target", "reward_network", "q_network_cpe", "q_network_cpe_target",
    "metrics_to_score", "imitator", "loss_reporter", "evaluation"])
class DQNTrainerParameters:
    """parameters.py:79-95"""


@make_config_class(QRDQNTrainer.__init__, blocklist=[
    "use_gpu", "q_network", "q_network_target", "metrics_to_score", "reward_network", "q_network_cpe",
    "q_network_cpe_target", "loss_reporter", "evaluation"])
class QRDQNTrainerParameters:
    """parameters.py:98-113"""


@make_config_class(C51Trainer.__init__, blocklist=[
    "use_gpu", "q_network", "q_network_target", "metrics_to_score", "loss_reporter", "evaluation"])
class C51TrainerParameters:
    """parameters.py:116-128"""
