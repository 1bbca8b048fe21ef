"""Replay-buffer tuple -> trainer input makers with the call surface of
reagent/gym/preprocessors/trainer_preprocessor.py:100-227, running on the device the batch lives on.
"""
import numpy as np
import torch

from .. import ops
from ..core import types as rlt
from ..core.parameters import CONTINUOUS_TRAINING_ACTION_RANGE


class DiscreteDqnInputMaker:
    """trainer_preprocessor.py:100-158; the one-hot / not_terminal / exp(log_prob) work is one
    rg_make_dqn_input launch."""

    def __init__(self, num_actions: int, trainer_preprocessor=None):
        self.num_actions = num_actions
        self.trainer_preprocessor = trainer_preprocessor

    @classmethod
    def create_for_env(cls, env):
        """trainer_preprocessor.py:105-117 without importing gym: a discrete action space is one with `n`; an
        environment may bring its own state preprocessor"""
        space = env.action_space
        assert hasattr(space, "n"), f"a discrete action space (with `n`) is needed, got {type(space)}"
        return cls(num_actions=int(space.n), trainer_preprocessor=getattr(env, "trainer_preprocessor", None))

    def __call__(self, batch):
        action, next_action, terminal = batch.action, batch.next_action, batch.terminal
        assert (len(action.shape) == 2 and action.shape[1] == 1 and next_action.shape == action.shape), (
            f"Must be action with stack_size = 1, but got shapes {action.shape}, {next_action.shape}"
        )
        B, dev, A = action.shape[0], action.device, self.num_actions
        f32 = dict(dtype=torch.float32, device=dev)
        a1h, na1h = torch.empty(B, A, **f32), torch.empty(B, A, **f32)
        not_terminal, prob = torch.empty(B, 1, **f32), torch.empty(B, 1, **f32)
        term_u8 = terminal.view(torch.uint8) if terminal.dtype == torch.bool else terminal.to(torch.uint8)
        ops.make_dqn_input(action.reshape(-1).contiguous(), next_action.reshape(-1).contiguous(),
                           term_u8.reshape(-1).contiguous(), batch.log_prob.reshape(-1).float().contiguous(),
                           A, a1h, na1h, not_terminal, prob)
        if self.trainer_preprocessor is not None:
            state = self.trainer_preprocessor(batch.state)
            next_state = self.trainer_preprocessor(batch.next_state)
        else:
            state = rlt.FeatureData(float_features=batch.state)
            next_state = rlt.FeatureData(float_features=batch.next_state)
        try:
            possible_actions_mask = batch.possible_actions_mask.float()
        except AttributeError:
            possible_actions_mask = torch.ones_like(a1h)
        try:
            possible_next_actions_mask = batch.next_possible_actions_mask.float()
        except AttributeError:
            possible_next_actions_mask = torch.ones_like(na1h)
        return rlt.DiscreteDqnInput(
            state=state, action=a1h, next_state=next_state, next_action=na1h,
            possible_actions_mask=possible_actions_mask,
            possible_next_actions_mask=possible_next_actions_mask, reward=batch.reward,
            not_terminal=not_terminal, step=None, time_diff=None,
            extras=rlt.ExtraData(mdp_id=None, sequence_number=None, action_probability=prob,
                                 max_num_actions=None, metrics=None),
        )


def rescale_actions(actions, new_min, new_max, prev_min, prev_max):
    """reagent/training/utils.py:13-29 (range asserts dropped: they force a host sync)."""
    prev_range = prev_max - prev_min
    new_range = new_max - new_min
    return ((actions - prev_min) / prev_range) * new_range + new_min


class PolicyNetworkInputMaker:
    """trainer_preprocessor.py:161-227 (dense path; candidate-doc features are out of scope)."""

    def __init__(self, action_low: np.ndarray, action_high: np.ndarray):
        self.action_low = torch.tensor(action_low)
        self.action_high = torch.tensor(action_high)
        (train_low, train_high) = CONTINUOUS_TRAINING_ACTION_RANGE
        self.train_low = torch.tensor(train_low)
        self.train_high = torch.tensor(train_high)
        self._on = {}  # device -> the four range tensors resident there (no host-to-device copy per batch)

    @classmethod
    def create_for_env(cls, env):
        """trainer_preprocessor.py:169-173: a box action space is one with `low` / `high`"""
        space = env.action_space
        assert hasattr(space, "low") and hasattr(space, "high"), f"a box action space (low / high) is needed, got {type(space)}"
        return cls(space.low, space.high)

    def _ranges(self, dev, A):
        """[4, A] = prev_min, prev_max, new_min, new_max per action dimension, resident on `dev`"""
        r = self._on.get((dev, A))
        if r is None:
            rows = [t.reshape(-1).float().expand(A) if t.numel() == 1 else t.reshape(-1).float()
                    for t in (self.action_low, self.action_high, self.train_low, self.train_high)]
            r = self._on[(dev, A)] = torch.stack(rows).contiguous().to(dev)
        return r

    def __call__(self, batch):
        """one launch (rg_make_policy_input): the operations and roundings of
        rescale_actions(...) / next_action * not_terminal / 1 - terminal / log_prob.exp()"""
        def f32(t):
            t = t if t.dtype == torch.float32 else t.float()
            return t if t.stride(-1) == 1 else t.contiguous()

        a, na, lp = f32(batch.action), f32(batch.next_action), f32(batch.log_prob).contiguous()
        term = batch.terminal if batch.terminal.element_size() == 1 else (batch.terminal != 0)
        term = term.contiguous()
        assert a.dim() == 2 and na.shape == a.shape, "dense [batch, action_dim] actions"
        dev = a.device
        B, A = a.shape
        action = torch.empty(B, A, dtype=torch.float32, device=dev)
        next_action = torch.empty(B, A, dtype=torch.float32, device=dev)
        not_terminal = torch.empty(batch.terminal.shape, dtype=torch.float32, device=dev)
        prob = torch.empty(lp.shape, dtype=torch.float32, device=dev)
        ops.make_policy_input(a, na, term, lp, self._ranges(dev, A), action, next_action, not_terminal, prob)
        return self._pack(batch, action, next_action, not_terminal, prob)

    @staticmethod
    def _pack(batch, action, next_action, not_terminal, action_probability):
        return rlt.PolicyNetworkInput(
            state=rlt.FeatureData(batch.state), next_state=rlt.FeatureData(batch.next_state),
            action=rlt.FeatureData(action), next_action=rlt.FeatureData(next_action), reward=batch.reward,
            not_terminal=not_terminal, step=None, time_diff=None,
            extras=rlt.ExtraData(mdp_id=None, sequence_number=None, action_probability=action_probability,
                                 max_num_actions=None, metrics=None),
        )
