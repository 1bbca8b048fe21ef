"""The offline (post-timeline) dataset as a column store in HBM.

The reference streams this table from parquet through a petastorm reader on the CPU, one dict of
tensors per batch, then moves the batch to the device (`batch_to_device`) and runs
`DiscreteDqnBatchPreprocessor.forward` on it (reagent/data/oss_data_fetcher.py:293-336 for the column
schema, reagent/preprocessing/batch_preprocessor.py:18-66).  With 288 GB of HBM the whole table lives
on the device: it is read from parquet once (pyarrow), converted to the storage types below and kept
as one array per column; a training batch is then `rg_table_dqn_batch` over a batch of row indices
(see `DiscreteDqnBatchPreprocessor.from_table`) — no CPU row assembly and no H2D copy per batch.
"""
import ctypes
from typing import Dict, Iterator, Optional

import numpy as np
import torch

from .. import _lib as L

# column -> (storage dtype, 2-D?)   (select_relevant_columns, oss_data_fetcher.py:293-336; masks and
# presence are stored as bytes instead of int64 / bool)
SCHEMA = {
    "state_features": (torch.float32, True),
    "state_features_presence": (torch.uint8, True),
    "next_state_features": (torch.float32, True),
    "next_state_features_presence": (torch.uint8, True),
    "action": (torch.int64, False),
    "next_action": (torch.int64, False),
    "reward": (torch.float32, False),
    "action_probability": (torch.float32, False),
    "time_diff": (torch.int64, False),
    "step": (torch.int64, False),
    "mdp_id": (torch.int64, False),
    "sequence_number": (torch.int64, False),
    "possible_actions_mask": (torch.uint8, True),
    "possible_next_actions_mask": (torch.uint8, True),
}
REQUIRED = ("state_features", "next_state_features", "action", "next_action", "reward", "possible_next_actions_mask")
assert tuple(SCHEMA) == L.TABLE_COLUMNS


class OfflineTable:
    def __init__(self, columns: Dict[str, object], num_actions: int, device=None, validate: bool = True):
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.num_actions = int(num_actions)
        self.columns: Dict[str, torch.Tensor] = {}
        for name in REQUIRED:
            if name not in columns or columns[name] is None:
                raise KeyError(f"offline table is missing the column {name!r}")
        n = None
        for name, (dtype, two_d) in SCHEMA.items():
            v = columns.get(name)
            if v is None:
                continue
            t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.require(v, requirements=["C", "W"]))
            if t.dtype == torch.bool and dtype == torch.uint8:
                t = t.view(torch.uint8) if t.is_contiguous() else t.to(torch.uint8)
            t = t.to(device=self.device, dtype=dtype).contiguous()
            if t.dim() != (2 if two_d else 1):
                raise ValueError(f"column {name!r}: expected {2 if two_d else 1} dimensions, got shape {tuple(t.shape)}")
            n = t.shape[0] if n is None else n
            if t.shape[0] != n:
                raise ValueError(f"column {name!r} has {t.shape[0]} rows, expected {n}")
            self.columns[name] = t
        self.num_features = self.columns["state_features"].shape[1]
        for name in ("state_features_presence", "next_state_features", "next_state_features_presence"):
            if name in self.columns and self.columns[name].shape[1] != self.num_features:
                raise ValueError(f"column {name!r} has {self.columns[name].shape[1]} features, expected {self.num_features}")
        for name in ("possible_actions_mask", "possible_next_actions_mask"):
            if name in self.columns and self.columns[name].shape[1] != self.num_actions:
                raise ValueError(f"column {name!r} has {self.columns[name].shape[1]} actions, expected {self.num_actions}")
        self._desc = None
        if validate and len(self):
            self.validate()

    def __len__(self) -> int:
        return self.columns["reward"].shape[0]

    @property
    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.columns.values())

    def desc(self) -> "L.DqnTable":
        if self._desc is None:
            d = L.DqnTable()
            for name in L.TABLE_COLUMNS:
                t = self.columns.get(name)
                setattr(d, name, t.data_ptr() if t is not None else None)
            d.n_rows, d.n_features, d.n_actions = len(self), self.num_features, self.num_actions
            self._desc = d
        return self._desc

    def validate(self, indices: Optional[torch.Tensor] = None):
        """F.one_hot of the reference raises on an action outside its classes; here the whole table is
        checked once at ingestion (one launch, one host read) instead of every batch."""
        from .. import ops

        idx = indices if indices is not None else torch.arange(len(self), device=self.device)
        bad = ops.table_check_actions(self, idx)
        if bad == 2:
            raise IndexError("row index outside the table")
        if bad:
            raise RuntimeError("Class values must be smaller than num_classes.")  # F.one_hot's message

    # ---- parquet (pyarrow; the reference reads the same files with petastorm) ------------------
    @classmethod
    def from_parquet(cls, path: str, num_actions: int, device=None, validate: bool = True) -> "OfflineTable":
        import pyarrow.parquet as pq

        tab = pq.read_table(path)
        cols = {}
        for name, (dtype, two_d) in SCHEMA.items():
            if name not in tab.column_names:
                continue
            arr = tab.column(name).combine_chunks()
            if two_d:
                flat = arr.flatten().to_numpy(zero_copy_only=False)
                n = len(arr)
                if n and len(flat) % n:
                    raise ValueError(f"column {name!r}: ragged rows are not supported")
                cols[name] = np.asarray(flat).reshape(n, -1)
            else:
                cols[name] = np.asarray(arr.to_numpy(zero_copy_only=False))
        return cls(cols, num_actions, device=device, validate=validate)

    def to_parquet(self, path: str):
        import pyarrow as pa
        import pyarrow.parquet as pq

        data = {}
        for name, t in self.columns.items():
            a = t.cpu().numpy()
            if name.endswith("_presence"):
                a = a.astype(bool)
            elif name.endswith("_mask"):
                a = a.astype(np.int64)  # ArrayType(LongType()), oss_data_fetcher.py:331-334
            data[name] = pa.array(list(a)) if a.ndim == 2 else pa.array(a)
        pq.write_table(pa.table(data), path)

    # ---- batches --------------------------------------------------------------------------------
    def rows(self, indices: torch.Tensor) -> Dict[str, torch.Tensor]:
        """the dict the reference's reader would yield for these rows (device tensors; for
        inspection and tests — training goes through DiscreteDqnBatchPreprocessor.from_table)"""
        idx = indices.to(self.device)
        return {k: t.index_select(0, idx) for k, t in self.columns.items()}

    def epoch(self, batch_size: int, shuffle: bool = True, generator: Optional[torch.Generator] = None,
              drop_last: bool = True) -> Iterator[torch.Tensor]:
        """row-index batches covering the table once; the permutation is drawn on the device"""
        n = len(self)
        if shuffle:
            order = torch.randperm(n, device=self.device, generator=generator)
        else:
            order = torch.arange(n, device=self.device)
        stop = n - (n % batch_size) if drop_last else n
        for i in range(0, stop, batch_size):
            yield order[i:i + batch_size]
