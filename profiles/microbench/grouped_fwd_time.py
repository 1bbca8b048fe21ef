"""Round 6: the grouped forward launch ALONE (C3 shapes, B = 65536, 200 quantiles) — in the step the two grouped forwards run
side by side on two streams, so their rocprofv3 durations overlap and say little about the kernel.  Times, with HIP events on
the launch stream, N back-to-back launches of: the plain C2-shaped forward (16 outputs), the target net's grouped forward
(scatter, non-saving), the online net's grouped forward (saving).   python profiles/microbench/grouped_fwd_time.py [bf16|bf16x3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from reagent_amd.engine import fused_forward_grouped  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sys.argv = ["bench.py", "--config", "c3", "--precision", prec]
args = bench.parse()
dev = torch.device("cuda", 0)
loop, trainer, init, cols, norm = bench.build(args, dev, 0)
for _ in range(3):
    loop.step()
loop.flush()
torch.cuda.synchronize()
gq = trainer._grouped()
state = gq._state
N = 30


def timed(fn, label):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(N):
        fn()
    z.record()
    torch.cuda.synchronize()
    print(f"{prec} {label:54s} {a.elapsed_time(z) / N * 1e3:8.1f} us / launch", flush=True)


timed(lambda: gq.online.st.forward(state, gq.qbar_next, save=False), "plain forward, 16-wide mean layer, save=0")
timed(lambda: fused_forward_grouped(gq.target.st, gq.target.gh, state, gq.sp_next, gq.zt, scatter=True, save=False),
      "grouped forward (target: scatter, save=0)")
timed(lambda: fused_forward_grouped(gq.online.st, gq.online.gh, state, gq.sp_cur, gq.z, scatter=False, save=True),
      "grouped forward (online: grouped order, save=1)")
