"""rg_fc_forward (bf16) on the GPU box for three large shapes, aligned operands (the dispatch picks the
256x256 DMA kernel for K >= 1024, the 128x128 kernel otherwise, swapped accumulators when no transposed
copy is asked for) and a leading dimension that is not a multiple of 8 (scalar-load path of the 128x128
kernel), with the bf16 / bf16 + transposed / fp32 outputs:
python profiles/microbench/gemm_shapes.py"""
import sys

import torch

sys.path.insert(0, ".")
import reagent_amd._lib as L  # noqa: E402
from reagent_amd import ops  # noqa: E402

dev = torch.device("cuda")


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for M, N, K in ((65536, 512, 512), (65536, 3200, 512), (65536, 512, 3200)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    xw = torch.zeros(M, K + 4, device=dev, dtype=torch.bfloat16)
    xw[:, :K] = x
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    yt = torch.empty(N, M, dtype=torch.bfloat16, device=dev)
    y32 = torch.empty(M, N, device=dev)
    gf = 2.0 * M * N * K / 1e9
    for name, xin in (("aligned", x), ("ld % 8 != 0", xw[:, :K])):  # aligned: 256x256 DMA kernel for K >= 1024, else 128x128
        for outs in (dict(y=y), dict(y=y, yt=yt), dict(y32=y32)):
            us = timed(lambda: ops.fc_forward(xin, w, b, L.ACT["relu"], L.PREC_BF16, **outs))
            print(f"M={M} N={N} K={K} {name:12s} outputs={'+'.join(outs):8s} {us:8.1f} us  {gf / us * 1e3:7.1f} TFLOP/s")
