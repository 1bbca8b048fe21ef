"""TEST INFRASTRUCTURE (oracle).  The UNMODIFIED reference SACTrainer (imported from /root/reference through oracle/stubs.py,
stepped by the Lightning-loop emulation of oracle/reference_harness.py, torch-CPU) trained online on tests/pendulum_env.py
through the same loop tests/test_zz_learning_curve.py drives this package through: how many random transitions and
training episodes the reference ITSELF needs to clear its -500 bar with a margin (sac_pendulum_online.yaml).  Used once to
size that test; nothing imports it.

    python -m oracle.learning_probe <prefill> <episodes> <seed> [<seed> ...]

Round 5, this container: 5000 / 40 -> sampled-policy evaluation -262 (worst episode -1135) and -141 (-356) on seeds 0 / 1;
10000 / 60 -> -129 (-302) and -133 (-357).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import reference_harness as H  # noqa: E402
from pendulum_env import PendulumEnv  # noqa: E402


def run(seed, prefill, episodes, batch=256):
    tr = H.build_sac(3, 1, [64, 64], ["leaky_relu", "leaky_relu"], dict(gamma=0.99, target_update_rate=0.005, softmax_policy=True),
                     1e-3, seed=seed, value=True, entropy_temperature=0.3)
    loop = H.PLLoop(tr)
    import reagent.core.types as rlt

    env, rng = PendulumEnv(seed), np.random.RandomState(seed + 1)
    N = prefill + 201 * episodes + 1
    S, S2 = np.zeros((N, 3), np.float32), np.zeros((N, 3), np.float32)
    A, R, NT = np.zeros((N, 1), np.float32), np.zeros((N, 1), np.float32), np.ones((N, 1), np.float32)
    n = 0
    lo, hi = -1 + 1e-6, 1 - 1e-6  # CONTINUOUS_TRAINING_ACTION_RANGE; the environment's range is [-2, 2]

    def push(s, a, r, s2, last):
        nonlocal n
        S[n], A[n], R[n], S2[n], NT[n] = s, a, r, s2, 0.0 if last else 1.0
        n += 1

    def act(s, mean=False):
        with torch.no_grad():
            out = tr.actor_network(rlt.FeatureData(torch.from_numpy(s)[None]))
        a = (out.squashed_mean if mean else out.action).reshape(-1).numpy()
        return ((a - lo) / (hi - lo) * 4.0 - 2.0).astype(np.float32)

    def train():
        idx = rng.randint(0, n - 1, size=batch)
        nt = torch.from_numpy(NT[idx])
        to_model = lambda a: torch.from_numpy((a + 2.0) / 4.0 * (hi - lo) + lo)  # noqa: E731
        loop.step(H.policy_batch_to_reference(dict(
            state=torch.from_numpy(S[idx]), next_state=torch.from_numpy(S2[idx]), reward=torch.from_numpy(R[idx]),
            time_diff=torch.ones(batch, 1), step=torch.ones(batch, 1), not_terminal=nt, action=to_model(A[idx]),
            next_action=to_model(A[idx + 1]) * nt)))

    while n < prefill:  # whole episodes of the uniform policy; the dataset's rule: step index >= max_steps is terminal
        s = env.reset()
        for t in range(201):
            a = rng.uniform(-2, 2, size=1).astype(np.float32)
            s2, r, _, _ = env.step(a)
            push(s, a, r, s2, t >= 200)
            s = s2
            if n >= prefill:
                break
    rewards = []
    for _ in range(episodes):
        s, tot = env.reset(), 0.0
        for t in range(201):
            a = act(s)
            s2, r, _, _ = env.step(a)
            push(s, a, r, s2, t >= 200)
            train()
            tot, s = tot + r, s2
        rewards.append(tot)

    def evaluate(mean):
        out = []
        for _ in range(20):
            s, tot = env.reset(), 0.0
            for _ in range(200):
                s, r, _, _ = env.step(act(s, mean))
                tot += r
            out.append(tot)
        return np.array(out)

    return rewards, evaluate(False), evaluate(True)


if __name__ == "__main__":
    torch.set_num_threads(2)
    prefill, episodes, seeds = int(sys.argv[1]), int(sys.argv[2]), [int(x) for x in sys.argv[3:]]
    for seed in seeds:
        t0 = time.time()
        rw, sampled, mean = run(seed, prefill, episodes)
        print(f"seed {seed} prefill {prefill} episodes {episodes}: sampled-policy evaluation {sampled.mean():.0f} (worst {sampled.min():.0f}), "
              f"squashed mean {mean.mean():.0f} | training first five {np.mean(rw[:5]):.0f}, last ten {np.mean(rw[-10:]):.0f} "
              f"({time.time() - t0:.0f} s)", flush=True)
