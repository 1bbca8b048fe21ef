"""multi-step drift of each mode against the oracle: C2 shapes, B = 2048, 30 steps on fresh batches"""
import sys, torch
sys.path.insert(0, '/root/repo')
import reagent_amd._lib as L
from oracle import restated as R
from reagent_amd import synthetic
from reagent_amd.core.parameters import EvaluationParameters, RLParameters
from reagent_amd.models import FullyConnectedDQN, set_default_precision
from reagent_amd.optimizer import Optimizer__Union
from reagent_amd.training import DQNTrainer
S, A, H, B, STEPS = 128, 16, [512, 512, 512], 2048, 30
dev = torch.device("cuda")
acts = ["relu"] * 3 + ["linear"]
init = synthetic.fc_init([S] + H + [A], acts, seed=40)
for mode in ("f32", "bf16x3", "bf16"):
    set_default_precision({"f32": L.PREC_F32, "bf16x3": L.PREC_BF16X3, "bf16": L.PREC_BF16}[mode])
    try:
        q = FullyConnectedDQN(S, A, H, ["relu"] * 3)
    finally:
        set_default_precision(L.PREC_F32)
    with torch.no_grad():
        for p, w in zip(q.parameters(), init): p.copy_(w)
    q = q.to(dev)
    tr = DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(A)],
                    rl=RLParameters(gamma=0.99, target_update_rate=0.001, q_network_loss="huber"),
                    optimizer=Optimizer__Union.default(lr=1e-3), evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)
    o = R.DQNOracle(init, init, acts, gamma=0.99, tau=0.001, loss="huber", lr=1e-3)
    probe = synthetic.dqn_batch(B, S, A, seed=999)
    out = []
    for s in range(STEPS):
        b = synthetic.dqn_batch(B, S, A, seed=100 + s, p_impossible=0.1)
        loss = tr.train_step_native(synthetic.to_dqn_input(b, dev))
        ref = o.step(b)
        if s in (0, 1, 4, 9, 19, 29):
            with torch.no_grad():
                qp = tr.q_network(synthetic.to_dqn_input(probe, dev).state).float().cpu()
                qr = R.fc_forward(o.params, acts, probe["state"])
            dq = (qp - qr).abs().max().item()
            dl = abs(loss.item() - ref["loss"].item()) / abs(ref["loss"].item())
            dws = [(p.detach().cpu() - r.detach()).abs() for p, r in zip(tr.q_network.parameters(), o.params)]
            rms = (sum((d.double() ** 2).sum() for d in dws) / sum(d.numel() for d in dws)).sqrt().item()
            out.append(f"step {s + 1}: dQ(probe) {dq:.2e} rel dloss {dl:.2e} rms dW {rms:.2e} max dW {max(d.max().item() for d in dws):.2e}")
    print(mode, "planes", __import__('os').environ.get("RG_X3_DZ_PLANES", "2"), "|", " | ".join(out))
