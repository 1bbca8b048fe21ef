"""Helpers to read the committed golden fixtures (tests/golden/*.npz, made by oracle/make_golden.py
from the unmodified reference)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.cfg = json.loads(str(self.z["config_json"]))

    def t(self, key):
        return torch.from_numpy(self.z[key])

    def a(self, key):
        return self.z[key]

    def has(self, key):
        return key in self.z.files

    def seq(self, prefix):
        out, i = [], 0
        while f"{prefix}{i}" in self.z.files:
            out.append(self.t(f"{prefix}{i}"))
            i += 1
        return out

    def batch(self, step):
        pre = f"step{step}_batch_"
        return {k[len(pre):]: self.t(k) for k in self.z.files if k.startswith(pre)}


def check_reported(g, s, seen, tol=1e-4):
    """the tensors a step handed its reporter against the reference's (`step{s}_report_*`, recorded by
    oracle/make_golden.py::_record_reporter): same keys, integers exact, floats within tol of the largest magnitude"""
    pre = f"step{s}_report_"
    want = {k[len(pre):]: g.t(k) for k in g.z.files if k.startswith(pre)}
    got = {k: v for k, v in seen.items() if isinstance(v, torch.Tensor)}
    assert set(got) == set(want), (s, sorted(got), sorted(want))
    for k, ref in want.items():
        v = got[k].detach().cpu().reshape(ref.shape)
        if ref.dtype.is_floating_point:
            assert (v.to(ref.dtype) - ref).abs().max() <= tol * max(1.0, ref.abs().max().item()), (s, k)
        else:
            assert torch.equal(v, ref), (s, k)
    return len(want)
