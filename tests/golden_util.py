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
