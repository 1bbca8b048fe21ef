import sys, time, torch
sys.path.insert(0, "/root/repo")
import reagent_amd._lib as L
from reagent_amd import ops, synthetic
from reagent_amd.core.parameters import EvaluationParameters, RLParameters
from reagent_amd.models import FullyConnectedCritic, FullyConnectedDQN, GaussianFullyConnectedActor, set_default_precision
from reagent_amd.optimizer import Optimizer__Union
from reagent_amd.training import QRDQNTrainer, SACTrainer
dev = torch.device("cuda")
set_default_precision(L.PREC_BF16)
def timeit(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
B = 65536
# C3: QR-DQN N=200
torch.manual_seed(0)
q = FullyConnectedDQN(128, 16, [512, 512, 512], ["relu"] * 3, num_atoms=200).to(dev)
tr = QRDQNTrainer(q, q.get_target_network(), actions=[str(i) for i in range(16)], rl=RLParameters(gamma=0.99),
                  num_atoms=200, optimizer=Optimizer__Union.default(lr=1e-3),
                  evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)
b = synthetic.to_dqn_input(synthetic.dqn_batch(B, 128, 16, seed=1), dev)
with ops.profile() as prof:
    dt = timeit(lambda: tr.train_step_native(b), n=3, w=1)
print("C3 QR-DQN: %.2f ms/step  %.3e transitions/s" % (dt * 1e3, B / dt))
torch.cuda.synchronize()
agg = {}
for name, meta, s_, e_ in prof.records:
    key = name + str(tuple(meta.values()))
    agg[key] = agg.get(key, 0.0) + s_.elapsed_time(e_)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:10]:
    print("    %-60s %.3f ms/step" % (k, v / 4))
del tr, q, b; torch.cuda.empty_cache()
# C4: SAC S=256 A=32 H=3x512
torch.manual_seed(0)
actor = GaussianFullyConnectedActor(256, 32, [512, 512, 512], ["relu"] * 3).to(dev)
q1 = FullyConnectedCritic(256, 32, [512, 512, 512], ["relu"] * 3).to(dev)
q2 = FullyConnectedCritic(256, 32, [512, 512, 512], ["relu"] * 3).to(dev)
tr = SACTrainer(actor, q1, q2, rl=RLParameters(gamma=0.99)).to(dev)
b = synthetic.to_policy_input(synthetic.policy_batch(B, 256, 32, seed=1), dev)
n1, n2 = torch.randn(B, 32, device=dev), torch.randn(B, 32, device=dev)
dt = timeit(lambda: tr.train_step_native(b, n1, n2), n=5, w=2)
print("C4 SAC: %.2f ms/step  %.3e transitions/s" % (dt * 1e3, B / dt))
