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

# ---- widened rows at BASELINE-like sizes -----------------------------------------------
del tr, actor, q1, q2, b
torch.cuda.empty_cache()
from reagent_amd.models import CategoricalDQN, FullyConnectedActor  # noqa: E402
from reagent_amd.training import C51Trainer, DiscreteCRRTrainer, DQNTrainer, TD3Trainer  # noqa: E402

S, A, H = 128, 16, [512, 512, 512]
acts = ["relu"] * 3
names = [str(i) for i in range(A)]
adam = lambda: Optimizer__Union.default(lr=1e-3)  # noqa: E731
bd = synthetic.to_dqn_input(synthetic.dqn_batch(B, S, A, seed=1, with_propensity=True), dev)

torch.manual_seed(0)
q = FullyConnectedDQN(S, A, H, acts).to(dev)
rn, qc = FullyConnectedDQN(S, A, H, acts).to(dev), FullyConnectedDQN(S, A, H, acts).to(dev)
tr = DQNTrainer(q, q.get_target_network(), rn, q_network_cpe=qc, q_network_cpe_target=qc.get_target_network(),
                metrics_to_score=[], actions=names, rl=RLParameters(gamma=0.99, q_network_loss="huber"),
                optimizer=adam(), evaluation=EvaluationParameters(calc_cpe_in_training=True)).to(dev)
run("C2 DQN + CPE heads (reward net, CPE q-net + target)", lambda: tr.train_step_native(bd))
del tr, q, rn, qc
torch.cuda.empty_cache()

torch.manual_seed(0)
dist = FullyConnectedDQN(S, A, H, acts, num_atoms=51).to(dev)
cq = CategoricalDQN(dist, qmin=-10, qmax=10, num_atoms=51)
tr = C51Trainer(cq, cq.get_target_network(), actions=names, rl=RLParameters(gamma=0.99), num_atoms=51, qmin=-10,
                qmax=10, optimizer=adam()).to(dev)
run("C51 (51 atoms, S=128 A=16 3x512)", lambda: tr.train_step_native(bd), n=3, w=1)
del tr, dist, cq
torch.cuda.empty_cache()

torch.manual_seed(0)
actor = FullyConnectedActor(S, A, H, acts).to(dev)
c1, c2 = FullyConnectedDQN(S, A, H, acts).to(dev), FullyConnectedDQN(S, A, H, acts).to(dev)
rn, qc = FullyConnectedDQN(S, A, H, acts).to(dev), FullyConnectedDQN(S, A, H, acts).to(dev)
tr = DiscreteCRRTrainer(actor_network=actor, actor_network_target=actor.get_target_network(), q1_network=c1,
                        q1_network_target=c1.get_target_network(), reward_network=rn, q2_network=c2,
                        q2_network_target=c2.get_target_network(), q_network_cpe=qc,
                        q_network_cpe_target=qc.get_target_network(), metrics_to_score=[],
                        evaluation=EvaluationParameters(calc_cpe_in_training=True), rl=RLParameters(gamma=0.99),
                        q_network_optimizer=adam(), actor_network_optimizer=adam(), actions=names,
                        entropy_coeff=0.01).to(dev)
run("discrete CRR (twin critics + actor + CPE, S=128 A=16 3x512)", lambda: tr.train_step_native(bd))
del tr, actor, c1, c2, rn, qc
torch.cuda.empty_cache()

torch.manual_seed(0)
actor = FullyConnectedActor(256, 32, H, acts).to(dev)
q1 = FullyConnectedCritic(256, 32, H, acts).to(dev)
q2 = FullyConnectedCritic(256, 32, H, acts).to(dev)
tr = TD3Trainer(actor, q1, q2, rl=RLParameters(gamma=0.99), q_network_optimizer=adam(), actor_network_optimizer=adam(),
                delayed_policy_update=1).to(dev)
bp = synthetic.to_policy_input(synthetic.policy_batch(B, 256, 32, seed=1), dev)
run("TD3 (S=256 A=32 3x512, policy updated every step)", lambda: tr.train_step_native(bp))
del tr, actor, q1, q2
torch.cuda.empty_cache()

# prioritized replay on the device sum tree: 65536 stratified draws / priority updates on 2^20 leaves
from reagent_amd.replay_memory import SumTree  # noqa: E402

cap = 1 << 20
tree = SumTree(cap, device=dev)
idx = torch.arange(cap, device=dev)
tree._set_dev(idx, torch.rand(cap, device=dev, dtype=torch.float64) + 0.1)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


upd_idx = torch.randint(cap, (B,), device=dev)
upd_val = torch.rand(B, device=dev, dtype=torch.float64) + 0.1
gen = torch.Generator(device=dev).manual_seed(0)
print(f"sum tree (2^20 leaves): stratified_sample({B}) {timed(lambda: tree.stratified_sample(B, generator=gen)):.1f} us, "
      f"set({B} priorities) {timed(lambda: tree._set_dev(upd_idx, upd_val)):.1f} us, "
      f"get({B}) {timed(lambda: tree.get_many(upd_idx)):.1f} us")
