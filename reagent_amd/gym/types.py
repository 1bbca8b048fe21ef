"""Transition / Trajectory records with the fields and checks of reagent/gym/types.py:19-106 (no gym import there
either).  `Transition.asdict()` is what BasicReplayBufferInserter spreads into `ReplayBuffer.add`."""
import dataclasses
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class Transition:
    mdp_id: int
    sequence_number: int
    observation: Any
    action: Any
    reward: float
    terminal: bool
    log_prob: Optional[float] = None
    possible_actions_mask: Optional[np.ndarray] = None
    info: Optional[Dict] = None

    def asdict(self):
        """dataclasses.asdict minus the fields that are None (types.py:31-33)"""
        return {k: v for k, v in dataclasses.asdict(self).items() if v is not None}


def get_optional_fields(cls) -> List[str]:
    """names of the Optional[...] fields of a dataclass (types.py:36-43)"""
    return [f.name for f in dataclasses.fields(cls) if type(None) in getattr(f.type, "__args__", ())]


@dataclass
class Trajectory:
    transitions: List[Transition] = field(default_factory=list)

    def __post_init__(self) -> None:
        self.optional_field_exist: Dict[str, bool] = {f: False for f in get_optional_fields(Transition)}

    def __len__(self) -> int:
        return len(self.transitions)

    def add_transition(self, transition: Transition) -> None:
        """the first transition decides which optional fields every later one must (not) fill (types.py:57-73)"""
        if len(self) == 0:
            for f in self.optional_field_exist:
                if getattr(transition, f, None) is not None:
                    self.optional_field_exist[f] = True
        for f, should_exist in self.optional_field_exist.items():
            val = getattr(transition, f, None)
            if (val is not None) != should_exist:
                raise ValueError(f"Field {f} given val {val} whereas should_exist is {should_exist}.")
        self.transitions.append(transition)

    def __getattr__(self, attr: str):
        if attr.startswith("__") or attr in ("transitions", "optional_field_exist"):
            raise AttributeError(attr)
        return [getattr(t, attr) for t in self.transitions]

    def calculate_cumulative_reward(self, gamma: float = 1.0):
        assert len(self) > 0, "called on empty trajectory"
        return sum(r * gamma**i for i, r in enumerate(self.reward))

    def to_dict(self):
        """types.py:88-106 (the reference one-hots the action over 2 classes there)"""
        d = {"action": F.one_hot(torch.from_numpy(np.stack(self.action)), 2)}
        for f in ("observation", "reward", "terminal", "log_prob", "possible_actions_mask"):
            if self.optional_field_exist.get(f, True):
                vals = getattr(self, f)
                d[f] = torch.tensor(vals) if np.isscalar(vals[0]) else torch.from_numpy(np.stack(vals)).float()
        return d


# Transform ReplayBuffer's transition batch to the trainer's input type
TrainerPreprocessor = Callable[[Any], Any]
# Called after env.step(action)
PostStep = Callable[[Transition], None]
# Called after the end of an episode
PostEpisode = Callable[[Trajectory, Dict], None]
