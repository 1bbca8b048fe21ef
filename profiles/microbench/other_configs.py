"""BASELINE configurations C3 (QR-DQN, N=200) and C4 (SAC, S=256 A=32 H=3x512) at B=65536 on the GPU box,
bf16 mode, native step, with the per-entry-point breakdown: python profiles/microbench/other_configs.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
import reagent_amd._lib as L  # noqa: E402
from reagent_amd import ops, synthetic  # noqa: E402
from reagent_amd.core.parameters import EvaluationParameters, RLParameters  # noqa: E402
from reagent_amd.models import (FullyConnectedCritic, FullyConnectedDQN, GaussianFullyConnectedActor,  # noqa: E402
                                set_default_precision)
from reagent_amd.optimizer import Optimizer__Union  # noqa: E402
from reagent_amd.training import QRDQNTrainer, SACTrainer  # noqa: E402

dev = torch.device("cuda")
set_default_precision(L.PREC_BF16)
B = 65536


def run(name, step, n=5, w=2):
    for _ in range(w):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"{name}: {dt * 1e3:.2f} ms/step  {B / dt:.3e} transitions/s")
    with ops.profile() as prof:
        for _ in range(n):
            step()
    for r in prof.summary()[:12]:
        print(f"    {r['name']}{tuple(r['meta'].values())}: {r['ms'] / n:.3f} ms/step in {r['calls'] / n:.0f} calls")


torch.manual_seed(0)
q = FullyConnectedDQN(128, 16, [512, 512, 512], ["relu"] * 3, num_atoms=200).to(dev)
tr = QRDQNTrainer(q, q.get_target_network(), actions=[str(i) for i in range(16)], rl=RLParameters(gamma=0.99),
                  num_atoms=200, optimizer=Optimizer__Union.default(lr=1e-3),
                  evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)
b = synthetic.to_dqn_input(synthetic.dqn_batch(B, 128, 16, seed=1), dev)
run("C3 QR-DQN N=200", lambda: tr.train_step_native(b), n=3, w=1)
del tr, q, b
torch.cuda.empty_cache()

torch.manual_seed(0)
actor = GaussianFullyConnectedActor(256, 32, [512, 512, 512], ["relu"] * 3).to(dev)
q1 = FullyConnectedCritic(256, 32, [512, 512, 512], ["relu"] * 3).to(dev)
q2 = FullyConnectedCritic(256, 32, [512, 512, 512], ["relu"] * 3).to(dev)
tr = SACTrainer(actor, q1, q2, rl=RLParameters(gamma=0.99)).to(dev)
b = synthetic.to_policy_input(synthetic.policy_batch(B, 256, 32, seed=1), dev)
n1, n2 = torch.randn(B, 32, device=dev), torch.randn(B, 32, device=dev)
run("C4 SAC", lambda: tr.train_step_native(b, n1, n2))
