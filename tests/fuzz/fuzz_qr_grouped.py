import os, sys, random, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
import gpu_ops  # FUZZ_ON_GPU=1: the real library on cuda instead of the interpreter
DEV = gpu_ops.device()
from reagent_amd import synthetic
from reagent_amd.qr_engine import GroupedQR
import test_qrdqn_trainer as T

# QR-DQN's grouped wide layer (qr_engine.py / qr_grouped.hip) against the dense [B, A * N] path of the same trunk on random
# (batch, actions, quantiles): batches below one 128-row tile and off its multiples, actions that no row selects (empty
# groups), quantile counts off the 8-element records — with the next action forced by the mask, so that both paths regress
# the same targets and the comparison is tight (tests/test_qrdqn_trainer.py::test_grouped_head_equals_dense_path, part 1)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 8
random.seed(seed)
bad = 0
for case in range(cases):
    S = random.choice([8, 24, 100])
    A = random.choice([2, 3, 4, 7, 16])
    N = random.choice([1, 3, 8, 10, 51, 72, 200])
    B = random.choice([1, 5, 127, 128, 129, 300, 640])
    maxq, double_q = random.random() < 0.7, random.random() < 0.5
    rl = dict(gamma=0.9, target_update_rate=0.1, maxq_learning=maxq)
    try:
        tg, td = T._qr_pair(DEV, S, A, N, [256, 256], rl, double_q, seed=case)
        if not GroupedQR.eligible(tg):
            print("not eligible", dict(S=S, A=A, N=N, B=B)); continue
        b = synthetic.dqn_batch(B, S, A, seed=case, p_impossible=0.3)
        g = torch.Generator().manual_seed(case)
        used = random.sample(range(A), random.randint(1, A))  # next actions come from a subset: the other groups stay empty
        pick = torch.tensor(used)[torch.randint(len(used), (B,), generator=g)]
        forced = torch.nn.functional.one_hot(pick, A).float()
        b1 = dict(b, possible_next_actions_mask=forced, next_action=forced * b["not_terminal"])
        batch = synthetic.to_dqn_input(b1, DEV)
        lg, ld = tg.train_step_native(batch).item(), td.train_step_native(batch).item()  # (the loss lives in a reused device buffer)
        assert tg._gq_active is not None and getattr(td, "_gq_active", None) is None
        ok = abs(lg - ld) <= 2e-5 * abs(ld) + 1e-7
        rels = [((x - y).norm() / (y.norm() + 1e-12)).item() for x, y in zip(tg._slab.grad_views(), td._slab.grad_views())]
        ok &= max(rels) <= 4e-3
        # a second step runs the one-launch update's weights through both paths again
        lg2, ld2 = tg.train_step_native(batch).item(), td.train_step_native(batch).item()
        ok &= abs(lg2 - ld2) <= 1e-3 * abs(ld2) + 1e-6 and lg2 != lg
        print("OK " if ok else "BAD", dict(S=S, A=A, N=N, B=B, maxq=maxq, double_q=double_q, groups_used=len(used)),
              "loss %.6f / %.6f, grads %.1e, step 2 %.6f / %.6f" % (lg, ld, max(rels), lg2, ld2))
        bad += 0 if ok else 1
    except Exception:
        bad += 1
        print("BAD (exception)", dict(S=S, A=A, N=N, B=B, maxq=maxq, double_q=double_q))
        traceback.print_exc(limit=4)
print("bad cases:", bad)
sys.exit(1 if bad else 0)
