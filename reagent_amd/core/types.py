"""Batch containers with the field names of the reference's ``reagent.core.types`` ("rlt").

Only what the DQN / QR-DQN / SAC hot path touches: FeatureData (:312-347), ExtraData (:440-450),
ActorOutput (:245-249), BaseInput (:688-769), DiscreteDqnInput (:772-816), PolicyNetworkInput
(:899-915) and the tensor-method forwarding of TensorDataClass (:49-108).  The trainers in this
package only read attributes, so instances of the reference's own classes work as well.

When the reference package itself is importable (a ReAgent installation this package is dropped into), its OWN
classes are re-exported from here instead of the restatements below: the reference looks trainers' input types up
by class OBJECT — `make_trainer_preprocessor` reads the annotation of `train_step_gen` and indexes a map keyed by
`rlt.DiscreteDqnInput` / `rlt.PolicyNetworkInput` (reagent/gym/preprocessors/trainer_preprocessor.py:39-48) — so
`DQNTrainer.train_step_gen(training_batch: rlt.DiscreteDqnInput, ...)` must name the reference's class there.
`REAGENT_AMD_OWN_TYPES=1` keeps the restatements.
"""
import dataclasses
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F


@dataclass
class TensorDataClass:
    def __getattr__(self, attr):
        if attr.startswith("__") and attr.endswith("__"):
            raise AttributeError(attr)
        tensor_attr = getattr(torch.Tensor, attr, None)
        if tensor_attr is None or not callable(tensor_attr):
            raise AttributeError(f"{type(self).__name__} doesn't have {attr} attribute.")

        def continuation(*args, **kwargs):
            def f(v):
                if isinstance(v, (torch.Tensor, TensorDataClass)) and getattr(v, attr, None) is not None:
                    return getattr(v, attr)(*args, **kwargs)
                if isinstance(v, dict):
                    return {kk: f(vv) for kk, vv in v.items()}
                if isinstance(v, tuple):
                    return tuple(f(vv) for vv in v)
                return v

            return type(self)(**f(self.__dict__))

        return continuation

    def cuda(self, *args, **kwargs):
        out = {}
        for k, v in self.__dict__.items():
            if isinstance(v, torch.Tensor):
                kwargs["non_blocking"] = kwargs.get("non_blocking", True)
                out[k] = v.cuda(*args, **kwargs)
            elif isinstance(v, TensorDataClass):
                out[k] = v.cuda(*args, **kwargs)
            else:
                out[k] = v
        return type(self)(**out)

    def cpu(self):
        out = {}
        for k, v in self.__dict__.items():
            out[k] = v.cpu() if isinstance(v, (torch.Tensor, TensorDataClass)) else v
        return type(self)(**out)


@dataclass
class ActorOutput(TensorDataClass):
    action: torch.Tensor
    log_prob: Optional[torch.Tensor] = None
    squashed_mean: Optional[torch.Tensor] = None


@dataclass
class FeatureData(TensorDataClass):
    float_features: torch.Tensor
    # sparse / sequence members of the reference type are outside this hot path
    id_list_features: Optional[object] = None
    id_score_list_features: Optional[object] = None
    stacked_float_features: Optional[torch.Tensor] = None
    candidate_docs: Optional[object] = None
    time_since_first: Optional[torch.Tensor] = None

    def __post_init__(self):
        ff = self.float_features
        if isinstance(ff, torch.Tensor) and ff.ndim != 2:
            raise ValueError(f"float_features should be 2D; got {ff.shape}.")


@dataclass
class ExtraData(TensorDataClass):
    mdp_id: Optional[torch.Tensor] = None
    sequence_number: Optional[torch.Tensor] = None
    action_probability: Optional[torch.Tensor] = None
    max_num_actions: Optional[int] = None
    metrics: Optional[torch.Tensor] = None

    @classmethod
    def from_dict(cls, d):
        return cls(**{f.name: d.get(f.name, None) for f in dataclasses.fields(cls)})


@dataclass
class BaseInput(TensorDataClass):
    state: FeatureData
    next_state: FeatureData
    reward: torch.Tensor
    time_diff: Optional[torch.Tensor]
    step: Optional[torch.Tensor]
    not_terminal: torch.Tensor

    def __len__(self):
        assert self.state.float_features.ndim == 2
        return self.state.float_features.size()[0]

    def batch_size(self):
        return len(self)

    def as_dict_shallow(self):
        return {
            "state": self.state,
            "next_state": self.next_state,
            "reward": self.reward,
            "time_diff": self.time_diff,
            "step": self.step,
            "not_terminal": self.not_terminal,
        }

    @staticmethod
    def from_dict(batch):
        return BaseInput(
            state=FeatureData(float_features=batch["state_features"]),
            next_state=FeatureData(float_features=batch["next_state_features"]),
            reward=batch["reward"],
            time_diff=batch["time_diff"],
            step=batch.get("step", None),
            not_terminal=batch["not_terminal"],
        )


@dataclass
class DiscreteDqnInput(BaseInput):
    action: torch.Tensor
    next_action: torch.Tensor
    possible_actions_mask: torch.Tensor
    possible_next_actions_mask: torch.Tensor
    extras: ExtraData

    @classmethod
    def input_prototype(cls, action_dim=2, batch_size=10, state_dim=3):
        return cls(
            state=FeatureData(float_features=torch.randn(batch_size, state_dim)),
            next_state=FeatureData(float_features=torch.randn(batch_size, state_dim)),
            reward=torch.rand(batch_size, 1),
            time_diff=torch.ones(batch_size, 1),
            step=torch.ones(batch_size, 1),
            not_terminal=torch.ones(batch_size, 1),
            action=F.one_hot(torch.randint(high=action_dim, size=(batch_size,)), num_classes=action_dim),
            next_action=F.one_hot(torch.randint(high=action_dim, size=(batch_size,)), num_classes=action_dim),
            possible_actions_mask=torch.ones(batch_size, action_dim),
            possible_next_actions_mask=torch.ones(batch_size, action_dim),
            extras=ExtraData(action_probability=torch.ones(batch_size, 1)),
        )

    @classmethod
    def from_dict(cls, batch):
        base = BaseInput.from_dict(batch)
        return cls(
            action=batch["action"],
            next_action=batch["next_action"],
            possible_actions_mask=batch["possible_actions_mask"],
            possible_next_actions_mask=batch["possible_next_actions_mask"],
            extras=ExtraData.from_dict(batch),
            **base.as_dict_shallow(),
        )


@dataclass
class PolicyNetworkInput(BaseInput):
    action: FeatureData
    next_action: FeatureData
    extras: Optional[ExtraData] = None

    @classmethod
    def from_dict(cls, batch):
        base = BaseInput.from_dict(batch)
        return cls(
            action=FeatureData(float_features=batch["action"]),
            next_action=FeatureData(float_features=batch["next_action"]),
            extras=batch.get("extras", None),
            **base.as_dict_shallow(),
        )


# ---- the reference's own classes, when it is importable (see the module docstring) ---------------------------
def _reference_types():
    import importlib.util
    import os

    if os.environ.get("REAGENT_AMD_OWN_TYPES") == "1":
        return None
    try:
        if importlib.util.find_spec("reagent") is None or importlib.util.find_spec("reagent.core.types") is None:
            return None
        import reagent.core.types as ref

        return ref
    except Exception:  # a partial installation (missing dependency of the reference): keep the restatements
        return None


USING_REFERENCE_TYPES = False
_ref = _reference_types()
if _ref is not None:
    for _name in ("TensorDataClass", "ActorOutput", "FeatureData", "ExtraData", "BaseInput", "DiscreteDqnInput",
                  "PolicyNetworkInput"):
        globals()[_name] = getattr(_ref, _name)
    USING_REFERENCE_TYPES = True
del _ref
