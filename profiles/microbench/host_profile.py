"""Where does the host spend its time enqueueing one eager step?  usage (GPU box): python profiles/microbench/host_profile.py c2 bf16"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

cfg, prec = (sys.argv[1] if len(sys.argv) > 1 else "c2"), (sys.argv[2] if len(sys.argv) > 2 else "bf16")
sys.argv = ["bench.py", "--config", cfg, "--precision", prec, "--no-cpu-baseline", "--no-parity"]
args = bench.parse()
dev = torch.device("cuda:0")
loop, trainer, *_ = bench.build(args, dev, 0)
for _ in range(5):
    loop.step()
loop.flush()
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    loop.step()
t1 = time.perf_counter()
loop.flush()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{cfg} {prec}: host enqueue {(t1 - t0) / N * 1e3:.3f} ms/step, wall {(t2 - t0) / N * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    loop.step()
pr.disable()
loop.flush()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
st.print_stats(28)
