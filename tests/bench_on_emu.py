"""TEST INFRASTRUCTURE ONLY: bench.py's `main()` — launcher, ranks, secondary regions, report file, the one printed line —
on a box without GPUs.  The C ABI is patched to the host-compiled kernel sources (tests/emu_backend.py), device
synchronisation and events become host clocks, the process group is gloo; then bench.main runs unchanged.  Started by
tests/test_bench_cli.py, directly or as the ranks of `python -m torch.distributed.run`."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch  # noqa: E402

import emu_backend  # noqa: E402

emu_backend.install()
torch.cuda.synchronize = lambda *a, **k: None


class HostEvent:
    def __init__(self, enable_timing=True):
        self.t = None

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


torch.cuda.Event = HostEvent

import bench  # noqa: E402

if __name__ == "__main__":
    bench.main(device=torch.device("cpu"), backend="gloo")
