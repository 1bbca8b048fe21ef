"""Which torch operators (not rg_* launches) still run inside a native step, and from which line?
usage (GPU box): python profiles/microbench/glue_trace.py c4 bf16"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

sys.argv = ["bench.py", "--config", sys.argv[1] if len(sys.argv) > 1 else "c4", "--precision",
            sys.argv[2] if len(sys.argv) > 2 else "bf16", "--no-cpu-baseline", "--no-parity"]
args = bench.parse()
dev = torch.device("cuda:0")
loop, trainer, *_ = bench.build(args, dev, 0)
for _ in range(3):
    loop._eager_step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        loop._eager_step()
    torch.cuda.synchronize()
rows = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    if any(c.name.startswith("aten::") for c in ev.cpu_children):
        continue  # count the innermost operator only
    t = sum(k.duration for k in ev.kernels)
    if not ev.kernels:
        continue
    frames = [f for f in (ev.stack or []) if "reagent_amd" in f or "bench.py" in f]
    key = (ev.name, frames[0].split("/root/repo/")[-1] if frames else "?")
    rows[key] += len(ev.kernels)
    dur[key] += t
tot = 0.0
for key, n in sorted(rows.items(), key=lambda kv: -dur[kv[0]]):
    print(f"{n / 2:5.1f} launches per step  {dur[key] / 2:8.1f} us  {key[0]:28s} {key[1]}")
    tot += dur[key] / 2
print(f"torch operators: {tot:.1f} us per step")
