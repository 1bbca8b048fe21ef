"""Sparse replay elements resident in HBM: IDListMetadata / IDScoreListMetadata of the reference
(reagent/replay_memory/circular_replay_buffer.py:144-274).

The reference keeps one python dict per transition in a numpy object array and builds (offsets, ids[, scores]) with
per-element python loops at sample time.  Here a feature's lists live in padded device slots — ids [capacity, W]
(+ scores [capacity, W]) and lens [capacity]; W doubles when a longer list arrives — and a sampled batch is two
launches (rg_ragged_offsets, rg_ragged_copy) plus one host read of the total length (the output's shape is data
dependent).  Same outputs: Dict[feature -> (offsets int32 [B], ids int64 [n])] / (..., scores float32 [n]).
"""
from typing import Dict, List

import numpy as np
import torch

from .. import ops


class _Column:
    def __init__(self, capacity: int, device, with_scores: bool):
        self.capacity, self.device, self.with_scores = capacity, device, with_scores
        self.W = 4
        self.ids = torch.zeros(capacity, self.W, dtype=torch.int64, device=device)
        self.scores = torch.zeros(capacity, self.W, dtype=torch.float32, device=device) if with_scores else None
        self.lens = torch.zeros(capacity, dtype=torch.int32, device=device)

    def _grow(self, n: int):
        W = self.W
        while W < n:
            W *= 2
        ids = torch.zeros(self.capacity, W, dtype=torch.int64, device=self.device)
        ids[:, : self.W] = self.ids
        self.ids = ids
        if self.with_scores:
            sc = torch.zeros(self.capacity, W, dtype=torch.float32, device=self.device)
            sc[:, : self.W] = self.scores
            self.scores = sc
        self.W = W

    def set(self, cursor: int, ids, scores=None):
        n = len(ids)
        if n > self.W:
            self._grow(n)
        if n:
            self.ids[cursor, :n] = torch.from_numpy(np.asarray(ids, dtype=np.int64))
            if self.with_scores:
                self.scores[cursor, :n] = torch.from_numpy(np.asarray(scores, dtype=np.float32))
        self.lens[cursor] = n

    def gather(self, indices: torch.Tensor):
        offsets, ids, scores = ops.ragged_gather(self.ids, self.scores, self.lens, indices)
        return (offsets, ids, scores) if self.with_scores else (offsets, ids)


class RaggedElement:
    """One sparse element of the buffer (e.g. `id_list`): a dict of features, each a ragged column.
    kind: "id_list" (values: lists of int64) or "id_score_list" (values: (ids, scores) tuples)."""

    def __init__(self, name: str, example: dict, capacity: int, device):
        self.name, self.keys = name, list(example.keys())
        self.kind = self._kind_of(name, example)
        self.columns: Dict[str, _Column] = {k: _Column(capacity, device, self.kind == "id_score_list") for k in self.keys}
        self.validate(example)

    @staticmethod
    def _kind_of(name: str, example) -> str:
        assert isinstance(example, dict), f"{name}: {type(example)} isn't dict"
        if all(isinstance(v, tuple) and len(v) == 2 for v in example.values()) and len(example) > 0:
            return "id_score_list"
        return "id_list"

    def zero_example(self):
        """:160-161, :217-218"""
        return {k: ([], []) for k in self.keys} if self.kind == "id_score_list" else {k: [] for k in self.keys}

    def validate(self, value):
        """IDListMetadata.validate :163-172 / IDScoreListMetadata.validate :220-239"""
        name = self.name
        assert isinstance(value, dict), f"{name}: {type(value)} isn't dict"
        for k, v in value.items():
            assert isinstance(k, str), f"{name}: {k} ({type(k)}) is not str"
            assert k in self.keys, f"{name}: {k} not in {self.keys}"
            if self.kind == "id_list":
                arr = np.array(v)
                if len(arr) > 0:
                    assert arr.dtype == np.int64, f"{name}: {v} arr has dtype {arr.dtype}, not np.int64"
            else:
                assert isinstance(v, tuple) and len(v) == 2, f"{name}: {v} ({type(v)}) is not len 2 tuple"
                ids, scores = np.array(v[0]), np.array(v[1])
                assert len(ids) == len(scores), f"{name}: {len(ids)} != {len(scores)}"
                if len(ids) > 0:
                    assert ids.dtype == np.int64, f"{name}: ids dtype {ids.dtype} isn't np.int64"
                    assert scores.dtype in (np.float32, np.float64), f"{name}: scores dtype {scores.dtype} isn't np.float32/64"

    def set(self, cursor: int, value: dict):
        for k in self.keys:
            v = value.get(k, ([], []) if self.kind == "id_score_list" else [])
            if self.kind == "id_score_list":
                self.columns[k].set(cursor, v[0], v[1])
            else:
                self.columns[k].set(cursor, v)

    def sample_to_output(self, indices: torch.Tensor):
        """:178-195 / :247-274 for the rows at `indices` (int64, on the buffer's device)"""
        return {k: self.columns[k].gather(indices) for k in self.keys}


def sparse_keys(kwargs) -> List[str]:
    return [k for k, v in kwargs.items() if isinstance(v, dict)]
