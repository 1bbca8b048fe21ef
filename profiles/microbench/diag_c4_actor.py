"""Diagnostic (builder tool): C4 SAC, fp32 mode — our actor gradients against the CPU oracle's (fp32 and fp64
autograd) on the golden batch, to tell accumulated fp32 noise from a systematic difference."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import reagent_amd._lib as L
from golden_util import Golden
from oracle import restated as R
from reagent_amd import synthetic
from reagent_amd.core.parameters import RLParameters
from reagent_amd.models import FullyConnectedCritic, GaussianFullyConnectedActor
from reagent_amd.optimizer import Optimizer__Union
from reagent_amd.training import SACTrainer

g = Golden("baseline_c4"); c = g.cfg
S, A, B = c["state_dim"], c["action_dim"], c["batch"]
acts = c["activations"] + ["linear"]
dev = "cuda"
inits = [synthetic.fc_init([S] + c["sizes"] + [2 * A], acts, c["init_seed"]),
         synthetic.fc_init([S + A] + c["sizes"] + [1], acts, c["init_seed"] + 1),
         synthetic.fc_init([S + A] + c["sizes"] + [1], acts, c["init_seed"] + 2)]
nets = [GaussianFullyConnectedActor(S, A, c["sizes"], c["activations"]), FullyConnectedCritic(S, A, c["sizes"], c["activations"]),
        FullyConnectedCritic(S, A, c["sizes"], c["activations"])]
for n, w in zip(nets, inits):
    with torch.no_grad():
        for p, x in zip(n.parameters(), w): p.copy_(x)
adam = lambda: Optimizer__Union.default(lr=c["lr"])
tr = SACTrainer(nets[0].to(dev), nets[1].to(dev), nets[2].to(dev), rl=RLParameters(**c["rl"]), q_network_optimizer=adam(),
                actor_network_optimizer=adam(), alpha_optimizer=adam()).to(dev)
b = synthetic.policy_batch(B, S, A, seed=800)
torch.manual_seed(3000)
nn_, nc = torch.randn(B, A), torch.randn(B, A)
batch = synthetic.to_policy_input(b, dev)
tr.train_step_native(batch, nn_, nc)
torch.cuda.synchronize()
ours = [gr.cpu().clone() for gr in tr._e["actor"]["slab"].grad_views()]
res = {}
for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
    cast = lambda ws: [w.to(dt) for w in ws]
    o = R.SACOracle(cast(inits[0]), cast(inits[1]), cast(inits[2]), acts, acts, A, gamma=c["rl"]["gamma"], tau=c["rl"]["target_update_rate"], lr=c["lr"])
    bb = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in b.items()}
    out = o.step(bb, nn_.to(dt), nc.to(dt))
    res[name] = [x.double() for x in out["actor_grads"]]
    res[name + "_minq"] = out
for i, gr in enumerate(ours):
    r32, r64 = res["f32"][i], res["f64"][i]
    typ = r64.abs().mean().item()
    print(f"param {i} {tuple(gr.shape)} mean|g| {typ:.3e} max|g| {r64.abs().max():.3e} | max|ours-f64| {(gr.double()-r64).abs().max():.3e} "
          f"max|cpu32-f64| {(r32-r64).abs().max():.3e} max|ours-cpu32| {(gr.double()-r32).abs().max():.3e} "
          f"| frac |g64|<1e-7: {(r64.abs()<1e-7).double().mean():.4f}")
# the min(q1, q2) selection
q1a, q2a = tr._q1a.cpu().double().reshape(-1), tr._q2a.cpu().double().reshape(-1)
print("min |q1a - q2a| over the batch:", (q1a - q2a).abs().min().item(), " q scale", q1a.abs().mean().item())
