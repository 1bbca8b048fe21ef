"""Transition -> ReplayBuffer.add (reagent/gym/preprocessors/replay_buffer_inserters.py:27-41)."""
from typing import Callable

from ...replay_memory.circular_replay_buffer import ReplayBuffer
from ..types import Transition

ReplayBufferInserter = Callable[[ReplayBuffer, Transition], None]


class BasicReplayBufferInserter:
    def __call__(self, replay_buffer: ReplayBuffer, transition: Transition):
        replay_buffer.add(**transition.asdict())


def make_replay_buffer_inserter(env) -> ReplayBufferInserter:
    """replay_buffer_inserters.py:31-34.  A RecSim environment (observations that are dictionaries of user / doc /
    response spaces) would need the reference's RecSimReplayBufferInserter: slate data is outside SURVEY.md §8."""
    spaces = getattr(getattr(env, "observation_space", None), "spaces", None)
    if isinstance(spaces, dict) and "doc" in spaces:
        raise NotImplementedError("RecSim observation spaces (user / doc / response) are not covered (SURVEY.md §8)")
    return BasicReplayBufferInserter()
