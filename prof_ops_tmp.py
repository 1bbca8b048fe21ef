import sys, torch
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-kernel-profile"]
import bench
from torch.profiler import profile, ProfilerActivity
args = bench.parse()
dev = torch.device("cuda", 0)
loop, trainer, init, cols, norm = bench.build(args, dev, 0)
for _ in range(5): loop.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3): loop.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
evs = [e for e in prof.events() if any(s in e.name for s in ("copy_", "fill_", "index", "randint", "ones", "clone", "contiguous", "to_copy"))]
seen = set()
for e in evs:
    st = [s for s in (e.stack or []) if "reagent_amd" in s or "bench.py" in s][:3]
    key = (e.name, tuple(st))
    if key in seen: continue
    seen.add(key)
    print(e.name, "|", " <- ".join(st))
