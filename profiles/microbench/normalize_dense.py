"""rg_normalize_dense (Preprocessor.forward, the stand-alone launch) on [65536, F] rows: python profiles/microbench/normalize_dense.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reagent_amd.core.parameters import NormalizationParameters as NP  # noqa: E402
from reagent_amd.preprocessing import Preprocessor  # noqa: E402

dev = torch.device("cuda:0")
B = 65536
for F, kind in ((128, "CONTINUOUS"), (256, "CONTINUOUS"), (128, "mixed")):
    g = torch.Generator().manual_seed(0)
    norm = {}
    for i in range(F):
        if kind == "CONTINUOUS" or i % 4 == 0:
            norm[i] = NP(feature_type="CONTINUOUS", mean=float(torch.randn(1, generator=g)), stddev=float(0.5 + torch.rand(1, generator=g)))
        elif i % 4 == 1:
            norm[i] = NP(feature_type="BOXCOX", boxcox_lambda=0.5, boxcox_shift=1.0, mean=0.3, stddev=1.2)
        elif i % 4 == 2:
            norm[i] = NP(feature_type="PROBABILITY")
        else:
            norm[i] = NP(feature_type="ENUM", possible_values=[0, 1, 2])
    pre = Preprocessor(norm, device=dev)
    x = torch.rand(B, F, device=dev) * 2
    pres = torch.ones(B, F, dtype=torch.uint8, device=dev)
    for _ in range(5):
        y = pre(x, pres)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        y = pre(x, pres)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    nbytes = B * (F * 5 + y.shape[1] * 4)
    print(f"F={F} {kind}: out {tuple(y.shape)} {us:.1f} us/launch = {nbytes / us * 1e-6:.2f} TB/s algorithmic")
