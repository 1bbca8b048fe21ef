"""Seeded synthetic transition data of SURVEY.md §8(d): the same CPU generator feeds the oracle,
the parity tests and bench.py (tensors are created on the CPU from a seeded torch.Generator and
then moved), so reference and HIP paths see identical inputs."""
from typing import Dict

import numpy as np

import torch
import torch.nn.functional as F


def dqn_batch(batch: int, state_dim: int, num_actions: int, seed: int = 0, p_terminal: float = 0.1,
              p_impossible: float = 0.0, with_steps: bool = False,
              n_extra_metrics: int = 0, with_propensity: bool = False) -> Dict[str, torch.Tensor]:
    """Fields of rlt.DiscreteDqnInput as a plain dict of CPU fp32 tensors."""
    g = torch.Generator().manual_seed(seed)
    state = torch.randn(batch, state_dim, generator=g)
    next_state = torch.randn(batch, state_dim, generator=g)
    reward = torch.rand(batch, 1, generator=g)
    not_terminal = (torch.rand(batch, 1, generator=g) > p_terminal).float()
    action = F.one_hot(torch.randint(num_actions, (batch,), generator=g), num_actions).float()
    next_action = F.one_hot(torch.randint(num_actions, (batch,), generator=g), num_actions).float()
    pna = torch.ones(batch, num_actions)
    if p_impossible > 0:
        pna = (torch.rand(batch, num_actions, generator=g) >= p_impossible).float()
        keep = torch.randint(num_actions, (batch,), generator=g)
        pna[torch.arange(batch), keep] = 1.0  # never a non-terminal state without a possible action
    if with_steps:
        step = torch.randint(1, 4, (batch, 1), generator=g).float()
        time_diff = torch.randint(1, 5, (batch, 1), generator=g).float()
    else:
        step = torch.ones(batch, 1)
        time_diff = torch.ones(batch, 1)
    out = dict(state=state, next_state=next_state, reward=reward, not_terminal=not_terminal,
               action=action, next_action=next_action, possible_actions_mask=torch.ones(batch, num_actions),
               possible_next_actions_mask=pna, step=step, time_diff=time_diff)
    if n_extra_metrics:  # extras.metrics of the CPE heads (drawn last: earlier fields keep their values)
        out["metrics"] = torch.rand(batch, n_extra_metrics, generator=g)
    if with_propensity:  # extras.action_probability of the logging policy, in (0.05, 1]
        out["action_probability"] = 0.05 + 0.95 * torch.rand(batch, 1, generator=g)
    return out


def policy_batch(batch: int, state_dim: int, action_dim: int, seed: int = 0,
                 p_terminal: float = 0.1) -> Dict[str, torch.Tensor]:
    """Fields of rlt.PolicyNetworkInput (actions already in the training range)."""
    g = torch.Generator().manual_seed(seed)
    return dict(
        state=torch.randn(batch, state_dim, generator=g),
        next_state=torch.randn(batch, state_dim, generator=g),
        action=torch.rand(batch, action_dim, generator=g) * 1.8 - 0.9,
        next_action=torch.rand(batch, action_dim, generator=g) * 1.8 - 0.9,
        reward=torch.rand(batch, 1, generator=g),
        not_terminal=(torch.rand(batch, 1, generator=g) > p_terminal).float(),
        step=torch.ones(batch, 1), time_diff=torch.ones(batch, 1),
    )


def to_dqn_input(d: Dict[str, torch.Tensor], device=None):
    from .core import types as rlt

    t = (lambda x: x.to(device)) if device is not None else (lambda x: x)
    return rlt.DiscreteDqnInput(
        state=rlt.FeatureData(t(d["state"])), next_state=rlt.FeatureData(t(d["next_state"])),
        reward=t(d["reward"]), time_diff=t(d["time_diff"]), step=t(d["step"]),
        not_terminal=t(d["not_terminal"]), action=t(d["action"]), next_action=t(d["next_action"]),
        possible_actions_mask=t(d["possible_actions_mask"]),
        possible_next_actions_mask=t(d["possible_next_actions_mask"]),
        extras=rlt.ExtraData(action_probability=t(d["action_probability"] if "action_probability" in d
                                                  else torch.ones_like(d["reward"])),
                             metrics=t(d["metrics"]) if "metrics" in d else None),
    )


def to_policy_input(d: Dict[str, torch.Tensor], device=None):
    from .core import types as rlt

    t = (lambda x: x.to(device)) if device is not None else (lambda x: x)
    return rlt.PolicyNetworkInput(
        state=rlt.FeatureData(t(d["state"])), next_state=rlt.FeatureData(t(d["next_state"])),
        reward=t(d["reward"]), time_diff=t(d["time_diff"]), step=t(d["step"]),
        not_terminal=t(d["not_terminal"]), action=rlt.FeatureData(t(d["action"])),
        next_action=rlt.FeatureData(t(d["next_action"])), extras=None,
    )


def replay_contents(capacity: int, obs_dim: int, num_actions: int, seed: int = 0,
                    p_terminal: float = 0.005) -> Dict[str, torch.Tensor]:
    """Column contents of a full replay buffer in the gym `Transition` schema."""
    g = torch.Generator().manual_seed(seed)
    return dict(
        observation=torch.randn(capacity, obs_dim, generator=g),
        action=torch.randint(num_actions, (capacity,), generator=g),
        reward=torch.rand(capacity, generator=g),
        terminal=torch.rand(capacity, generator=g) < p_terminal,
        possible_actions_mask=torch.ones(capacity, num_actions),
        log_prob=torch.zeros(capacity),
    )


def fc_init(dims, activations, seed: int = 0):
    """Seeded weights for a FullyConnected stack, drawn like the reference's initialiser
    (reagent/models/fully_connected_network.py:21-23,122-126: W ~ N(0, (gain * sqrt(1/fan_in))^2), b = 0)
    from an explicit CPU generator.  The BASELINE-shape goldens (oracle/make_golden.py, `baseline_*`) load
    these into the reference networks and the parity tests load them into the HIP-backed ones, so fixtures
    of 0.6-2.2 M parameters do not have to carry their initial weights."""
    import math

    g = torch.Generator().manual_seed(seed)
    out = []
    for i, (fan_in, fan_out) in enumerate(zip(dims, dims[1:])):
        try:
            gain = torch.nn.init.calculate_gain(activations[i])
        except ValueError:
            gain = 1.0
        out.append(torch.randn(fan_out, fan_in, generator=g) * (gain * math.sqrt(1.0 / fan_in)))
        out.append(torch.zeros(fan_out))
    return out


def normalization_table(n_features: int, seed: int = 7):
    """(mean, stddev) of n CONTINUOUS features: mean ~ N(0,1), stddev ~ U[0.5, 2)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n_features, generator=g), torch.rand(n_features, generator=g) * 1.5 + 0.5


class ScriptedEnv:
    """A deterministic stand-in for a gym environment with the attributes the replay-buffer training flow touches
    (`reset`, `step`, `possible_actions_mask`, `action_space`; reagent/gym/datasets/replay_buffer_dataset.py:96-137):
    observations, rewards and masks are closed forms of (episode, step) with exactly representable values, episodes
    end after `episode_lengths[episode % len]` steps.  Used by the parity tests and by oracle/make_golden.py, which
    drives the reference's ReplayBufferDataset with the same script."""

    class _Discrete:
        def __init__(self, n):
            self.n = n

    class _Box:
        def __init__(self, low, high):
            self.low, self.high = np.asarray(low, dtype=np.float32), np.asarray(high, dtype=np.float32)
            self.shape = self.low.shape

    def __init__(self, obs_dim: int, num_actions: int = None, action_low=None, action_high=None,
                 episode_lengths=(5, 3, 7), with_mask: bool = True):
        self.obs_dim, self.episode_lengths = obs_dim, tuple(episode_lengths)
        self.action_space = (self._Discrete(num_actions) if num_actions is not None else self._Box(action_low, action_high))
        self.num_actions, self.with_mask = num_actions, with_mask and num_actions is not None
        self.episode, self.t = -1, 0

    def _obs(self):
        i = np.arange(self.obs_dim)
        return (((self.episode * 31 + self.t * 17 + i * 7) % 23) / 8.0 - 1.0).astype(np.float32)

    @property
    def possible_actions_mask(self):
        if not self.with_mask:
            return None
        m = np.ones(self.num_actions, dtype=np.float32)
        m[(self.episode * 3 + self.t) % self.num_actions] = float((self.episode + self.t) % 3 != 0)
        return m

    def reset(self):
        self.episode += 1
        self.t = 0
        return self._obs()

    def step(self, action):
        a = float(np.asarray(action, dtype=np.float64).sum())
        self.t += 1
        reward = ((self.episode * 5 + self.t * 3) % 11) / 4.0 - 1.0 + 0.125 * a
        terminal = self.t >= self.episode_lengths[self.episode % len(self.episode_lengths)]
        return self._obs(), reward, terminal, {"episode": self.episode, "t": self.t}


class ScriptedAgent:
    """(action, log_prob) as a closed form of a call counter; `post_step` is None (replay_buffer_dataset.py:139)"""

    post_step = None

    def __init__(self, env: ScriptedEnv):
        self.env, self.calls = env, 0

    def act(self, obs, possible_actions_mask=None):
        k = self.calls
        self.calls += 1
        log_prob = -0.125 - 0.25 * (k % 5)
        if self.env.num_actions is not None:
            return int((k * 5 + 3) % self.env.num_actions), log_prob
        sp = self.env.action_space
        frac = ((k * 3 + np.arange(sp.low.size) * 5) % 8) / 8.0  # inside [low, high): rescale_actions asserts the range
        return (sp.low + frac.astype(np.float32) * (sp.high - sp.low)).astype(np.float32), log_prob
