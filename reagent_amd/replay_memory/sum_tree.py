"""Device-resident SumTree with the surface of reagent/replay_memory/sum_tree.py:30-189.

The tree lives in HBM as one fp64 array in heap order (the reference's list of power-of-two numpy
levels laid end to end); `set`, `get`, `sample` and `stratified_sample` keep the reference's
signatures and error behaviour, and batched variants (`set_many`, `get_many`, `sample_many`) do
the per-index Python loops of PrioritizedReplayBuffer as single launches.  Randomness stays where
the reference has it — Python's `random` module on the host — so a seeded run draws the same
query values; `stratified_sample(..., generator=...)` is the device-RNG throughput variant.
"""
import random
from typing import List, Optional

import numpy as np
import torch

from .. import _lib as L
from .. import ops


class SumTree:
    def __init__(self, capacity: int, device: Optional[torch.device] = None) -> None:
        assert isinstance(capacity, int)
        if capacity <= 0:
            raise ValueError("Sum tree capacity should be positive. Got: {}".format(capacity))
        self.capacity = capacity
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.depth = L.lib().rg_sumtree_depth(capacity)
        self._tree = torch.zeros(L.lib().rg_sumtree_nodes(capacity), dtype=torch.float64, device=self.device)
        self._claim = None  # int32 [2^depth] scratch of the batched set, allocated on first use
        self.max_recorded_priority = 1.0

    # ---- reference surface -------------------------------------------------------------------
    @property
    def nodes(self) -> List[np.ndarray]:
        """The reference's `self.nodes` (a list of levels), copied to the host for inspection."""
        flat = self._tree.cpu().numpy()
        return [flat[(1 << d) - 1 : (1 << (d + 1)) - 1] for d in range(self.depth + 1)]

    def _total_priority(self) -> float:
        return float(self._tree[0].item())

    def sample(self, query_value: Optional[float] = None) -> int:
        if self._total_priority() == 0.0:
            raise Exception("Cannot sample from an empty sum tree.")
        if query_value and (query_value < 0.0 or query_value > 1.0):
            raise ValueError("query_value must be in [0, 1].")
        query_value = random.random() if query_value is None else query_value
        return int(self.sample_many([query_value])[0].item())

    def stratified_sample(self, batch_size: int, generator: Optional[torch.Generator] = None,
                          as_tensor: bool = False):
        """sum_tree.py:133-153.  Default: the reference's host draws (`np.linspace` segments,
        `random.uniform` per segment) and a python list of ints.  `generator` (a device
        torch.Generator): u_i = (i + U[0,1)) / batch_size drawn on the device, no host sync, returns
        an int64 device tensor (also with as_tensor=True)."""
        if generator is not None:
            u = torch.rand(batch_size, dtype=torch.float64, device=self.device, generator=generator)
            q = (torch.arange(batch_size, dtype=torch.float64, device=self.device) + u) / batch_size
            return self.sample_many(q)
        if self._total_priority() == 0.0:
            raise Exception("Cannot sample from an empty sum tree.")
        bounds = np.linspace(0.0, 1.0, batch_size + 1)
        assert len(bounds) == batch_size + 1
        query_values = [random.uniform(bounds[i], bounds[i + 1]) for i in range(batch_size)]
        out = self.sample_many(query_values)
        return out if as_tensor else out.cpu().tolist()

    def get(self, node_index: int) -> float:
        return float(self.get_many(torch.tensor([int(node_index)]), dtype=torch.float64)[0].item())

    def set(self, node_index: int, value: float) -> None:
        if value < 0.0:
            raise ValueError("Sum tree values should be nonnegative. Got {}".format(value))
        self.max_recorded_priority = max(value, self.max_recorded_priority)
        self._set_dev(torch.tensor([int(node_index)], dtype=torch.int64),
                      torch.tensor([float(value)], dtype=torch.float64))

    # ---- batched forms -----------------------------------------------------------------------
    def _check_indices(self, idx: torch.Tensor) -> None:
        """IndexError for a leaf outside [0, capacity) like the reference's list indexing (which, unlike
        here, lets negative indices wrap) — on host data only; indices that already live on the device
        are not synchronised for and the kernels skip out-of-range ones"""
        if idx.device.type == "cpu" and idx.numel() > 0:
            lo, hi = int(idx.min()), int(idx.max())
            if lo < 0 or hi >= self.capacity:
                raise IndexError(f"sum tree index out of range: [{lo}, {hi}] for capacity {self.capacity}")

    def _set_dev(self, indices: torch.Tensor, values: torch.Tensor, sequential: bool = False) -> None:
        self._check_indices(indices)
        indices = indices.to(device=self.device, dtype=torch.int64).contiguous()
        values = values.to(device=self.device, dtype=torch.float64).contiguous()
        claim = None
        if indices.numel() > 32 and not sequential:
            if self._claim is None:
                self._claim = torch.full((1 << self.depth,), -1, dtype=torch.int32, device=self.device)
            claim = self._claim
        ops.sumtree_set(self._tree, self.depth, self.capacity, indices, values, claim)

    def set_many(self, indices, values, sequential: bool = False) -> None:
        """`for i, v in zip(indices, values): self.set(i, v)` as one update (later pairs win).
        More than 32 pairs take the parallel path (leaves written, levels rebuilt as left + right);
        sequential=True forces the in-order single-thread walk with the reference's delta arithmetic
        (bit-identical to the Python loop for any values; O(n log C) serial)."""
        values_t = torch.as_tensor(np.asarray(values, dtype=np.float64)) if not isinstance(values, torch.Tensor) else values
        indices_t = torch.as_tensor(np.asarray(indices, dtype=np.int64)) if not isinstance(indices, torch.Tensor) else indices
        if values_t.numel() == 0:
            return
        vmin, vmax = float(values_t.min().item()), float(values_t.max().item())
        if vmin < 0.0:
            raise ValueError("Sum tree values should be nonnegative. Got {}".format(vmin))
        self.max_recorded_priority = max(vmax, self.max_recorded_priority)
        self._set_dev(indices_t, values_t, sequential=sequential)

    def get_many(self, indices, dtype=torch.float32) -> torch.Tensor:
        idx = indices if isinstance(indices, torch.Tensor) else torch.as_tensor(np.asarray(indices, dtype=np.int64))
        self._check_indices(idx)
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty(idx.numel(), dtype=dtype, device=self.device)
        if dtype == torch.float32:
            ops.sumtree_get(self._tree, self.depth, self.capacity, idx, out32=out)
        else:
            assert dtype == torch.float64
            ops.sumtree_get(self._tree, self.depth, self.capacity, idx, out64=out)
        return out

    def sample_many(self, query_values) -> torch.Tensor:
        """Leaf index for every query value in [0, 1]; int64 device tensor."""
        q = query_values if isinstance(query_values, torch.Tensor) else torch.tensor(list(query_values), dtype=torch.float64)
        q = q.to(device=self.device, dtype=torch.float64).contiguous()
        out = torch.empty(q.numel(), dtype=torch.int64, device=self.device)
        ops.sumtree_sample(self._tree, self.depth, q, out)
        return out
