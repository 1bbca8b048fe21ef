"""PrioritizedReplayBuffer with the surface of reagent/replay_memory/prioritized_replay_buffer.py:30-185
on the device-resident ReplayBuffer and SumTree.

Same control flow as the reference (Schaul et al. 2015): new elements enter with the priority the
caller passes, indices are drawn by stratified sampling over the sum tree, invalid draws are
replaced by unstratified re-draws within a shared attempt budget, and `sampling_probabilities` is the
float32 leaf value of every sampled index.  The per-index Python loops (`set_priority`,
`get_priority`, the tree descents) are single launches; the host is consulted only where the
reference's semantics are sequential (the retry loop, reached only when a draw was invalid).
"""
from typing import Optional

import numpy as np
import torch

from .circular_replay_buffer import ReplayBuffer
from .sum_tree import SumTree


class PrioritizedReplayBuffer(ReplayBuffer):
    def __init__(
        self,
        stack_size: int,
        replay_capacity: int,
        batch_size: int,
        update_horizon: int = 1,
        gamma: float = 0.99,
        max_sample_attempts: int = 1000,
        device: Optional[torch.device] = None,
    ) -> None:
        super().__init__(stack_size=stack_size, replay_capacity=replay_capacity, batch_size=batch_size,
                         update_horizon=update_horizon, gamma=gamma, device=device)
        self._max_sample_attempts = max_sample_attempts
        self.sum_tree = SumTree(replay_capacity, device=self.device)

    # ---- add ---------------------------------------------------------------------------------
    def _add(self, **kwargs) -> None:
        """:60-83 — `priority` goes to the sum tree (at the cursor), everything else to storage."""
        self._check_args_length(**kwargs)
        self.sum_tree.set(self.cursor(), kwargs["priority"])
        super()._add(**kwargs)

    def load_columns(self, columns, mark_all_valid: bool = False, priorities=None):
        """Bulk ingestion (see ReplayBuffer.load_columns); `priorities` (n,) default to 1.0, the
        initial max_recorded_priority a stream of `add(priority=max)` calls would use."""
        n = columns["observation"].shape[0]
        cols = dict(columns)
        if "priority" not in cols:
            cols["priority"] = torch.zeros(n, dtype=torch.float32)  # storage column the reference never reads
        super().load_columns(cols, mark_all_valid=mark_all_valid)
        self._valid_mask_cache = None
        pr = torch.ones(n, dtype=torch.float64) if priorities is None else torch.as_tensor(priorities, dtype=torch.float64)
        self.sum_tree.set_many(torch.arange(n, dtype=torch.int64), pr)

    # ---- sampling ----------------------------------------------------------------------------
    def sample_index_batch(self, batch_size: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """:85-115.  Stratified draws (host `random` like the reference, or a device generator),
        validity looked up on the device; the sequential retry loop runs only if some draw is invalid."""
        indices = self.sum_tree.stratified_sample(batch_size, generator=generator, as_tensor=True)
        valid = self._valid_mask_dev()[indices]
        if bool(valid.all().item()):
            return indices
        idx = indices.cpu().tolist()
        ok = valid.cpu().tolist()
        allowed_attempts = self._max_sample_attempts
        for i in range(len(idx)):
            if not ok[i]:
                if allowed_attempts == 0:
                    raise RuntimeError(
                        "Max sample attempts: Tried {} times but only sampled {}"
                        " valid indices. Batch size is {}".format(self._max_sample_attempts, i, batch_size)
                    )
                index = idx[i]
                while not self.is_valid_transition(index) and allowed_attempts > 0:
                    index = self.sum_tree.sample()  # not stratified, like the reference
                    allowed_attempts -= 1
                idx[i] = index
        return torch.tensor(idx, dtype=torch.int64, device=self.device)

    def _valid_mask_dev(self) -> torch.Tensor:
        """bool [capacity] copy of the host validity mask, rebuilt after the mask changed"""
        if getattr(self, "_valid_mask_cache", None) is None:
            self._valid_mask_cache = torch.from_numpy(self._valid_host.copy()).to(self.device)
        return self._valid_mask_cache

    def set_index_valid_status(self, idx: int, is_valid: bool):
        super().set_index_valid_status(idx, is_valid)
        self._valid_mask_cache = None

    def sample_transition_batch(self, batch_size=None, indices=None, **kwargs):
        """:117-144 — the parent's batch plus `sampling_probabilities` (batch_size, 1) float32."""
        if batch_size is None:
            batch_size = self._batch_size
        transition = super().sample_transition_batch(batch_size, indices, **kwargs)
        probs = self.sum_tree.get_many(transition.indices.reshape(-1), dtype=torch.float32).view(batch_size, 1)
        return transition._replace(sampling_probabilities=probs)

    # ---- priorities --------------------------------------------------------------------------
    def set_priority(self, indices, priorities) -> None:
        """:146-157 (one launch instead of a Python loop of tree walks).  numpy int32 indices as in
        the reference, or int64 device tensors."""
        if isinstance(indices, np.ndarray):
            assert indices.dtype == np.int32, "Indices must be integers, given: {}".format(indices.dtype)
        self.sum_tree.set_many(indices, priorities)

    def get_priority(self, indices):
        """:159-180 — float32 priorities; numpy in -> numpy out, device tensor in -> device tensor out."""
        if isinstance(indices, torch.Tensor):
            return self.sum_tree.get_many(indices, dtype=torch.float32)
        assert getattr(indices, "shape", ()), "Indices must be an array."
        assert indices.dtype == np.int32, "Indices must be int32s, given: {}".format(indices.dtype)
        return self.sum_tree.get_many(indices.astype(np.int64), dtype=torch.float32).cpu().numpy()

    def get_transition_elements(self):
        return super().get_transition_elements() + ["sampling_probabilities"]
