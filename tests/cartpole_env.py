"""The cart-pole balancing task (Barto, Sutton & Anderson 1983) with the constants and episode rules of gym's
`CartPole-v0` — what the reference's integration tests train on (reagent/gym/tests/configs/cartpole/*.yaml) — as plain
numpy, with the attributes the replay-buffer training flow touches (reagent_amd.synthetic.ScriptedEnv has the same
surface): `reset`, `step`, `possible_actions_mask`, `action_space.n`, `max_steps`.  gym itself is not installed here.

State (x, x', theta, theta'); actions 0 / 1 push the cart with -10 / +10 N; explicit Euler steps of 0.02 s; an episode
ends when |x| > 2.4 m or |theta| > 12 degrees (terminal) and is cut at 200 steps by the caller's `max_steps`; the reward
is 1 for every step taken.  A uniform(-0.05, 0.05) start state from the environment's own seeded generator.
"""
import math

import numpy as np


class _Discrete:
    def __init__(self, n):
        self.n = n


class CartPoleEnv:
    GRAVITY, MASS_CART, MASS_POLE, HALF_LENGTH, FORCE, DT = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    X_LIMIT, THETA_LIMIT = 2.4, 12 * 2 * math.pi / 360
    max_steps = 200

    def __init__(self, seed: int = 0):
        self.rng = np.random.RandomState(seed)
        self.action_space = _Discrete(2)
        self.num_actions = 2
        self.state = np.zeros(4)
        self.t = 0

    @property
    def possible_actions_mask(self):
        return np.ones(2, dtype=np.float32)

    def reset(self):
        self.state = self.rng.uniform(-0.05, 0.05, size=4)
        self.t = 0
        return self.state.astype(np.float32)

    def step(self, action):
        a = int(np.asarray(action).reshape(-1)[0]) if not isinstance(action, int) else action
        x, x_dot, th, th_dot = self.state
        force = self.FORCE if a == 1 else -self.FORCE
        cos_t, sin_t = math.cos(th), math.sin(th)
        total = self.MASS_CART + self.MASS_POLE
        pole_ml = self.MASS_POLE * self.HALF_LENGTH
        tmp = (force + pole_ml * th_dot * th_dot * sin_t) / total
        th_acc = (self.GRAVITY * sin_t - cos_t * tmp) / (self.HALF_LENGTH * (4.0 / 3.0 - self.MASS_POLE * cos_t * cos_t / total))
        x_acc = tmp - pole_ml * th_acc * cos_t / total
        self.state = np.array([x + self.DT * x_dot, x_dot + self.DT * x_acc, th + self.DT * th_dot, th_dot + self.DT * th_acc])
        self.t += 1
        terminal = bool(abs(self.state[0]) > self.X_LIMIT or abs(self.state[2]) > self.THETA_LIMIT)
        return self.state.astype(np.float32), 1.0, terminal, {"t": self.t}
