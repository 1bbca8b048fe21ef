"""TEST INFRASTRUCTURE — on the MI355X:  python tests/fuzz/determinism_gpu.py [steps]

Every BASELINE configuration's whole training loop (bench.build: sampler, forwards, heads, backward, weight gradients on their side
streams, the fused update) at B = 65 536 in both precisions, built and stepped TWICE from the same seeds: the parameters of every
network must be bit-identical after `steps` steps.  A race between the two streams of the QR-DQN engine, a read of a buffer another
launch still writes, an atomic on a value path — each would show here as a difference."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
bad = 0
for cfg in ("c2", "c3", "c4"):
    for prec in ("bf16", "bf16x3"):
        sys.argv = ["bench.py", "--config", cfg, "--precision", prec]
        args = bench.parse()

        def run():
            torch.manual_seed(1234)
            torch.cuda.manual_seed_all(1234)
            loop, trainer, init, cols, norm = bench.build(args, dev, 0)
            if args.algo == "qrdqn":
                from reagent_amd.qr_engine import GroupedQR

                args.grouped_head = bool(trainer.use_grouped_head and GroupedQR.eligible(trainer))
            last = None
            for _ in range(steps):
                last = loop.step()
            loop.flush()
            torch.cuda.synchronize()
            if isinstance(last, dict):
                last = last["q1_loss"]
            params = [p.detach().clone() for p in trainer.parameters()]
            del loop, trainer, cols
            torch.cuda.empty_cache()
            return float(last.item()), params

        l1, p1 = run()
        l2, p2 = run()
        same = l1 == l2 and len(p1) == len(p2) and all(torch.equal(a, b) for a, b in zip(p1, p2))
        finite = all(bool(torch.isfinite(p).all()) for p in p1)
        bad += 0 if (same and finite) else 1
        print(("OK " if same and finite else "BAD"), cfg, prec, f"{steps} steps twice: {len(p1)} parameter tensors",
              "bit-identical" if same else "DIFFER", f"loss {l1:.6f} / {l2:.6f}", flush=True)
print("bad cases:", bad)
sys.exit(1 if bad else 0)
