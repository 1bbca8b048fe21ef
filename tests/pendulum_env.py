"""The inverted-pendulum swing-up task with the constants and episode rules of gym's `Pendulum-v0` — what the reference's
SAC / TD3 integration tests train on (reagent/gym/tests/configs/pendulum/*.yaml) — as plain numpy, with the attributes
the replay-buffer training flow touches (the surface of reagent_amd.synthetic.ScriptedEnv with a box action space).

State (theta, theta'), theta = 0 upright; observation (cos theta, sin theta, theta'); one action, a torque in [-2, 2]
(clipped); theta'' = 3 g / (2 l) sin theta + 3 u / (m l^2) with g = 10, m = l = 1, semi-implicit Euler steps of 0.05 s,
|theta'| <= 8; reward -(normalised theta^2 + 0.1 theta'^2 + 0.001 u^2); never terminal, the caller cuts an episode at
`max_steps` = 200.  Start state uniform in [-pi, pi] x [-1, 1] from the environment's own seeded generator.
"""
import numpy as np


class _Box:
    def __init__(self, low, high):
        self.low, self.high = np.asarray(low, dtype=np.float32), np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape


class PendulumEnv:
    MAX_SPEED, MAX_TORQUE, DT, G, M, L = 8.0, 2.0, 0.05, 10.0, 1.0, 1.0
    max_steps = 200
    possible_actions_mask = None
    num_actions = None

    def __init__(self, seed: int = 0):
        self.rng = np.random.RandomState(seed)
        self.action_space = _Box([-self.MAX_TORQUE], [self.MAX_TORQUE])
        self.theta, self.theta_dot, self.t = 0.0, 0.0, 0

    def _obs(self):
        return np.array([np.cos(self.theta), np.sin(self.theta), self.theta_dot], dtype=np.float32)

    def reset(self):
        self.theta, self.theta_dot = self.rng.uniform(-np.pi, np.pi), self.rng.uniform(-1.0, 1.0)
        self.t = 0
        return self._obs()

    def step(self, action):
        u = float(np.clip(np.asarray(action, dtype=np.float64).reshape(-1)[0], -self.MAX_TORQUE, self.MAX_TORQUE))
        th, thd = self.theta, self.theta_dot
        norm_th = ((th + np.pi) % (2 * np.pi)) - np.pi
        cost = norm_th ** 2 + 0.1 * thd ** 2 + 0.001 * u ** 2
        thd = thd + (3 * self.G / (2 * self.L) * np.sin(th) + 3.0 / (self.M * self.L ** 2) * u) * self.DT
        self.theta = th + thd * self.DT  # (the new speed enters the angle before it is clipped, as in gym's v0)
        self.theta_dot = float(np.clip(thd, -self.MAX_SPEED, self.MAX_SPEED))
        self.t += 1
        return self._obs(), -float(cost), False, {"t": self.t}
