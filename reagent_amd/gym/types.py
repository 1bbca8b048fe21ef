"""Transition / Trajectory records with the fields and checks of reagent/gym/types.py:19-106 (no gym import there
either).  `Transition.asdict()` is what BasicReplayBufferInserter spreads into `ReplayBuffer.add`."""
import dataclasses
from dataclasses import dataclass
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


_OPTIONAL = tuple(get_optional_fields(Transition))  # log_prob, possible_actions_mask, info


class Trajectory:
    """The transitions of one episode (types.py:46-106).  The first transition decides which optional fields every later
    one must — or must not — fill; `trajectory.<field>` is the list of that field over the episode."""

    def __init__(self, transitions: Optional[List[Transition]] = None):
        self.optional_field_exist: Dict[str, bool] = dict.fromkeys(_OPTIONAL, False)
        self.transitions: List[Transition] = []
        for t in transitions or ():
            self.add_transition(t)

    def __len__(self) -> int:
        return len(self.transitions)

    def add_transition(self, transition: Transition) -> None:
        filled = {f: getattr(transition, f, None) is not None for f in _OPTIONAL}
        if not self.transitions:
            self.optional_field_exist = filled
        for f, should_exist in self.optional_field_exist.items():
            if filled[f] != should_exist:
                raise ValueError(f"Field {f} given val {getattr(transition, f, None)} whereas should_exist is {should_exist}.")
        self.transitions.append(transition)

    def __getattr__(self, attr: str):
        if attr.startswith("__") or attr in ("transitions", "optional_field_exist"):
            raise AttributeError(attr)
        return [getattr(t, attr) for t in self.transitions]

    def calculate_cumulative_reward(self, gamma: float = 1.0):
        """(discounted) sum of the episode's rewards"""
        assert len(self) > 0, "called on empty trajectory"
        total, weight = 0.0, 1.0
        for r in self.reward:
            total, weight = total + weight * r, weight * gamma
        return total

    def to_dict(self):
        """tensors of the episode (the reference one-hots the action over 2 classes here, types.py:88-106)"""
        out = {"action": F.one_hot(torch.from_numpy(np.stack(self.action)), 2)}
        for name in ("observation", "reward", "terminal", "log_prob", "possible_actions_mask"):
            if not self.optional_field_exist.get(name, True):
                continue
            vals = getattr(self, name)
            out[name] = torch.tensor(vals) if np.isscalar(vals[0]) else torch.from_numpy(np.stack(vals)).float()
        return out


# Transform ReplayBuffer's transition batch to the trainer's input type
TrainerPreprocessor = Callable[[Any], Any]
# Called after env.step(action)
PostStep = Callable[[Transition], None]
# Called after the end of an episode
PostEpisode = Callable[[Trajectory, Dict], None]
