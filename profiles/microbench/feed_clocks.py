"""clock and power of the GPU while one variant of mfma_feed_dma runs for a few seconds (bench.GpuTelemetry: sysfs of THIS GPU's card)"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from bench import GpuTelemetry
torch.zeros(1, device="cuda")
here = os.path.dirname(os.path.abspath(__file__))
for k, reps in ((2, 12000), (1, 11000), (0, 9000), (3, 8000)):
    with GpuTelemetry(0) as t:
        out = subprocess.run([os.path.join(here, "mfma_feed_dma"), str(k), str(reps)], capture_output=True, text=True).stdout.strip()
    s = t.summary() or {}
    print(out[:150])
    print("    sclk MHz", s.get("sclk_mhz"), " power W", s.get("power_w"))
