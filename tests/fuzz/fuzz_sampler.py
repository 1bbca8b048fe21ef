import sys, random
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, torch
import gpu_ops  # FUZZ_ON_GPU=1: the real library on cuda instead of the interpreter
DEV = gpu_ops.device()
from reagent_amd.core.parameters import NormalizationParameters as NP
from reagent_amd.preprocessing import DiscreteDqnInputMaker, Preprocessor
from reagent_amd.replay_memory import ReplayBuffer

random.seed(3)
bad = 0
for case in range(20):
    # (widths whose 16-byte chunks divide the workgroup — 4 .. 128, 256, 512 — take the register-descriptor path of round 5)
    F = random.choice([4 * random.randint(1, 40), 4 * random.randint(1, 40), 8, 32, 64, 128, 256]); A = random.randint(1, 20); H = random.randint(1, 5)
    cap = random.randint(H + 40, 300); n = random.randint(H + 5, cap + 50); B = random.choice([1, 5, 63, 64, 65, 130])
    with_mask = random.random() < 0.5; norm = random.random() < 0.6
    dt = torch.bfloat16 if (norm and random.random() < 0.5) else torch.float32
    rb = ReplayBuffer(device=DEV, stack_size=1, replay_capacity=cap, batch_size=B, update_horizon=H, gamma=0.93)
    rng = np.random.RandomState(case)
    for i in range(n):
        kw = dict(observation=rng.randn(F).astype(np.float32), action=np.int64(rng.randint(A)), reward=np.float32(rng.rand()),
                  terminal=bool(rng.rand() < 0.15), log_prob=np.float32(-rng.rand()))
        if with_mask: kw["possible_actions_mask"] = (rng.rand(A) > 0.3).astype(np.float32)
        rb.add(**kw)
    if rb.size == 0: continue
    pre = Preprocessor({i: NP(feature_type="CONTINUOUS", mean=0.1 * (i % 5), stddev=1.0 + 0.1 * (i % 3)) for i in range(F)}, device=DEV) if norm else None
    idx = rb.sample_index_batch(B)
    fused = rb.sample_dqn_input(A, B, indices=idx, state_preprocessor=pre, state_dtype=dt if norm else None)
    tup = rb.sample_transition_batch(B, indices=idx, state_preprocessor=pre, state_dtype=dt if norm else None)
    ref = DiscreteDqnInputMaker(A)(tup)
    ok = fused is not None
    if ok:
        for name in ("action", "next_action", "reward", "not_terminal", "possible_actions_mask", "possible_next_actions_mask"):
            ok = ok and torch.equal(getattr(fused, name), getattr(ref, name))
        ok = ok and torch.equal(fused.state.float_features, ref.state.float_features) and torch.equal(fused.next_state.float_features, ref.next_state.float_features)
        ok = ok and torch.equal(fused.extras.action_probability, ref.extras.action_probability)
    bad += not ok
    print("OK " if ok else "BAD", dict(F=F, A=A, H=H, cap=cap, n=n, B=B, mask=with_mask, norm=norm, dt=str(dt)))
print("bad cases:", bad)
