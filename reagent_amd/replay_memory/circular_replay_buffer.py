"""Device-resident circular replay buffer with the surface of
reagent/replay_memory/circular_replay_buffer.py:310-890 (dense columns).

The column store lives in HBM (one tensor ``(capacity, *shape)`` per key, dtype inferred from the
first ``add`` exactly as DenseMetadata.create_from_example :98-106 does); sampling is two kernel
launches — rg_replay_nstep (steps / next index / terminal / n-step reward) and rg_replay_gather
(every output column in one launch) — instead of ~17 advanced-indexing ops, and the result is
bit-identical to the reference's namedtuple (same field names, order, dtypes and (B,1) shapes).
Validity bookkeeping (add rules :468-522) is host-side integer logic, mirrored to the device lazily.

Sparse (id-list / id-score-list) elements live in padded device slots (ragged.py); ``return_everything_as_stack`` and
``return_as_timeline_format`` (next_* elements and rewards as per-transition lists: one gather over all rows, split into
views on the host) follow the reference's shapes.
"""
import collections
import gzip
import logging
import os
import pickle
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import _lib as L
from .. import ops

logger = logging.getLogger(__name__)

REQUIRED_KEYS = ["observation", "action", "reward", "terminal"]

_NP2TORCH = {
    np.dtype("float32"): torch.float32,
    np.dtype("float64"): torch.float32,
    np.dtype("int64"): torch.int64,
    np.dtype("int32"): torch.int32,
    np.dtype("int16"): torch.int16,
    np.dtype("int8"): torch.int8,
    np.dtype("uint8"): torch.uint8,
    np.dtype("bool"): torch.bool,
    np.dtype("float16"): torch.float16,
}


STORE_FILENAME_PREFIX = "$store$_"  # circular_replay_buffer.py:302
_RG_FILENAME_PREFIX = "$rg$_"
CHECKPOINT_DURATION = 4  # :305
_NOT_CHECKPOINTED = ("device",)  # where the columns live is a property of the loading process


class ReplayBuffer:
    def __init__(
        self,
        stack_size: int = 1,
        replay_capacity: int = 10000,
        batch_size: int = 1,
        return_everything_as_stack: bool = False,
        return_as_timeline_format: bool = False,
        update_horizon: int = 1,
        gamma: float = 0.99,
        device: Optional[torch.device] = None,
    ) -> None:
        if replay_capacity < update_horizon + stack_size:
            raise ValueError("There is not enough capacity to cover update_horizon and stack_size.")
        self._initialized_buffer = False
        self._stack_size = stack_size
        self._return_everything_as_stack = return_everything_as_stack
        self._return_as_timeline_format = return_as_timeline_format
        self._replay_capacity = replay_capacity
        self._batch_size = batch_size
        self._update_horizon = update_horizon
        self._gamma = gamma
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        if self.device.type == "cuda" and self.device.index is None and torch.cuda.is_available():
            # the device tensors report "cuda:N": compare like with like (an index-less handle made the cached
            # device copies look stale, i.e. one host-to-device copy per sampled batch)
            self.device = torch.device("cuda", torch.cuda.current_device())

        self.add_count = np.array(0)
        # `_decays` is computed by torch exactly as the reference does (:373) so that the n-step
        # reward product is bit-identical
        self._decays = (self._gamma ** torch.arange(self._update_horizon)).unsqueeze(0)
        self._decays_dev = None
        self._valid_host = np.zeros(self._replay_capacity, dtype=bool)
        self._terminal_host = np.zeros(self._replay_capacity, dtype=bool)
        self._valid_dirty = True
        self._valid_indices_dev = None
        self._num_valid_indices = 0
        self._num_transitions_in_current_episode = 0

        self._store: Dict[str, torch.Tensor] = {}
        self._shapes: Dict[str, tuple] = {}
        self._np_dtypes: Dict[str, np.dtype] = {}
        self._extra_keys: List[str] = []
        self._transition_elements: List[str] = []
        self._batch_type = collections.namedtuple("filler", [])
        self._zero_transition = {}
        self._sparse = {}  # sparse (id-list / id-score-list) elements: replay_memory/ragged.py

    # ---- storage -----------------------------------------------------------------------------
    def initialize_buffer(self, **kwargs):
        kwarg_keys = set(kwargs.keys())
        assert set(REQUIRED_KEYS).issubset(kwarg_keys), f"{kwarg_keys} doesn't contain all of {REQUIRED_KEYS}"
        # deterministic order (the reference iterates a set, :394; consumers address fields by name)
        self._extra_keys = sorted(kwarg_keys - set(REQUIRED_KEYS))
        self._sparse = {}
        for k in REQUIRED_KEYS + self._extra_keys:
            ex = kwargs[k]
            if isinstance(ex, torch.Tensor):
                raise ValueError(f"Unable to deduce type for {k}: Input shouldn't be tensor")
            if isinstance(ex, dict):  # IDListMetadata / IDScoreListMetadata (:144-274): ragged columns in HBM
                if k in REQUIRED_KEYS:
                    raise ValueError(f"Unable to deduce a dense type for {k}")
                if self._stack_size != 1:
                    raise NotImplementedError("sparse replay elements with stack_size > 1 (the reference leaves it TODO, :150)")
                from .ragged import RaggedElement

                self._sparse[k] = RaggedElement(k, ex, self._replay_capacity, self.device)
                continue
            arr = np.array(ex)
            dtype = np.dtype("float32") if arr.dtype == np.dtype("float64") else arr.dtype
            if dtype not in _NP2TORCH:
                raise ValueError(f"Unable to deduce type for {k}: {ex}")
            self._shapes[k], self._np_dtypes[k] = arr.shape, dtype
            # `terminal` is only ever consumed through `.to(torch.bool)` (:678, :765): keep it as
            # one byte per transition whatever integer type the caller passes
            tdtype = torch.bool if k == "terminal" else _NP2TORCH[dtype]
            self._store[k] = torch.zeros((self._replay_capacity, *arr.shape), dtype=tdtype,
                                         device=self.device)
        self._transition_elements = self.get_transition_elements()
        self._batch_type = collections.namedtuple("batch_type", self._transition_elements)
        self._zero_transition = {k: np.zeros(self._shapes[k], dtype=self._np_dtypes[k]) for k in self._store}
        self._zero_transition.update({k: el.zero_example() for k, el in self._sparse.items()})
        self._initialized_buffer = True

    @property
    def size(self) -> int:
        return self._num_valid_indices

    def set_index_valid_status(self, idx: int, is_valid: bool):
        old_valid = self._valid_host[idx]
        if not old_valid and is_valid:
            self._num_valid_indices += 1
        elif old_valid and not is_valid:
            self._num_valid_indices -= 1
        assert self._num_valid_indices >= 0, f"{self._num_valid_indices} is negative"
        self._valid_host[idx] = is_valid
        self._valid_dirty = True

    def get_add_args_signature(self):
        return list(self._store.keys()) + list(self._sparse.keys())

    def _check_args_length(self, **kwargs):
        if len(kwargs) != len(self._store) + len(self._sparse):
            raise ValueError(f"Add expects: {self.get_add_args_signature()}; received {kwargs}")

    def _check_add_types(self, **kwargs):
        self._check_args_length(**kwargs)
        for k, el in self._sparse.items():
            el.validate(kwargs[k])
        for k in self._store:
            v = kwargs[k]
            assert not isinstance(v, (dict, torch.Tensor)), f"{k}: {type(v)} is dict or torch.Tensor"
            arr = np.array(v)
            dtype = np.dtype("float32") if arr.dtype == np.dtype("float64") else arr.dtype
            assert arr.shape == self._shapes[k] and dtype == self._np_dtypes[k], (
                f"{k}: Expected {self._shapes[k]} {self._np_dtypes[k]}, got {arr.shape} {dtype}"
            )

    def add(self, **kwargs):
        """circular_replay_buffer.py:468-522 (validity rules) — one transition per call."""
        if not self._initialized_buffer:
            self.initialize_buffer(**kwargs)
        self._check_add_types(**kwargs)
        last_idx = (self.cursor() - 1) % self._replay_capacity
        if self.is_empty() or self._terminal_host[last_idx]:
            self._num_transitions_in_current_episode = 0
            for _ in range(self._stack_size - 1):
                self._add(**self._zero_transition)
        cur_idx = self.cursor()
        self.set_index_valid_status(idx=cur_idx, is_valid=False)
        if self._num_transitions_in_current_episode >= self._update_horizon:
            idx = (cur_idx - self._update_horizon) % self._replay_capacity
            self.set_index_valid_status(idx=idx, is_valid=True)
        self._add(**kwargs)
        self._num_transitions_in_current_episode += 1
        for i in range(self._stack_size - 1):
            idx = (self.cursor() + i) % self._replay_capacity
            self.set_index_valid_status(idx=idx, is_valid=False)
        if kwargs["terminal"]:
            num_back = min(self._num_transitions_in_current_episode, self._update_horizon)
            for i in range(0, num_back):
                idx = (cur_idx - i) % self._replay_capacity
                self.set_index_valid_status(idx=idx, is_valid=True)

    def _add(self, **kwargs):
        self._check_args_length(**kwargs)
        cursor = self.cursor()
        for k, v in kwargs.items():
            if k in self._sparse:
                self._sparse[k].set(int(cursor), v)
                continue
            arr = np.array(v, dtype=self._np_dtypes[k])
            if k == "terminal":
                arr = arr.astype(bool)
            self._store[k][cursor] = torch.from_numpy(arr)
        self._terminal_host[cursor] = bool(kwargs["terminal"])
        self.add_count += 1

    def add_many(self, **columns):
        """T consecutive transitions in ONE call: `columns[key]` holds T rows (tensor or array, host or device).  Equivalent
        to T `add` calls (circular_replay_buffer.py:468-547) — same storage contents, cursor, validity and episode state,
        also across the ring's wrap and over several calls — but the rows travel as one indexed device copy per column and
        the validity rules are applied in closed form, instead of 143 us of per-key python validation per transition.
        stack_size 1 (no zero-padding transitions), dense elements, T <= capacity."""
        assert self._stack_size == 1, "add_many: stacked-frame buffers pad episode starts (use add)"
        term = np.asarray(torch.as_tensor(columns["terminal"]).cpu()).astype(bool).reshape(-1)
        T = int(term.shape[0])
        if T == 0:
            return
        assert T <= self._replay_capacity, "add_many: more rows than the buffer holds"
        if not self._initialized_buffer:
            first = {}
            for k, v in columns.items():
                x = torch.as_tensor(v)[0].cpu()
                first[k] = bool(x) if k == "terminal" else (x.numpy() if x.ndim else x.numpy()[()])
            self.initialize_buffer(**first)
        if set(columns) != set(self._store) | set(self._sparse):
            raise ValueError("Add expects: {}; received {}".format(sorted(self._store), sorted(columns)))
        assert not self._sparse, "add_many: sparse (id-list) elements are added one transition at a time"
        C, h = self._replay_capacity, self._update_horizon
        c0 = self.cursor()
        last_idx = (c0 - 1) % C
        n0 = 0 if (self.is_empty() or self._terminal_host[last_idx]) else self._num_transitions_in_current_episode
        pos = (c0 + np.arange(T)) % C
        # ---- rows: one indexed copy per column
        # every column is checked BEFORE the first copy: a bad column late in the dict must not leave earlier columns
        # half-written over valid transitions.  Types follow `add` (np.array(v, dtype=...) there: same-kind conversions
        # only — a float column into an integer element would truncate silently).
        staged = {}
        for k, v in columns.items():
            col = self._store[k]
            src = torch.as_tensor(v)
            if tuple(src.shape) != (T, *col.shape[1:]):
                raise ValueError(f"add_many: {k} has shape {tuple(src.shape)}, expected {(T, *col.shape[1:])}")
            if k != "terminal" and src.dtype != col.dtype:
                float_to_int = src.dtype.is_floating_point and not col.dtype.is_floating_point
                if float_to_int or src.dtype == torch.bool or col.dtype == torch.bool:
                    raise ValueError(f"add_many: {k} arrives as {src.dtype}, the buffer stores {col.dtype}")
            staged[k] = src
        pos_dev = torch.from_numpy(pos).to(self.device)
        for k, src in staged.items():
            col = self._store[k]
            col.index_copy_(0, pos_dev, src.to(device=col.device, dtype=col.dtype))
        # ---- validity (the rules of add, :491-522, for stack_size 1)
        # k[t]: transitions of t's episode before t;  at add t: valid[pos t] = False; if k[t] >= h: valid[pos t - h] = True;
        # at a terminal t: the last min(k[t] + 1, h) indices of the episode become valid.  An index ends up valid iff one
        # of those events follows its own add:  s + h is added in the same episode, or its episode ends within h - 1 steps.
        ends = np.flatnonzero(term)
        start_of_ep = np.zeros(T, dtype=np.int64)  # index (in 0..T) of the first new transition of t's episode; -n0 for the open one
        prev_end = np.concatenate([[-1], ends])[np.searchsorted(ends, np.arange(T), side="left")]
        start_of_ep = np.where(prev_end < 0, -n0, prev_end + 1)
        k = np.arange(T) - start_of_ep
        next_end = np.concatenate([ends, [np.iinfo(np.int64).max]])[np.searchsorted(ends, np.arange(T), side="left")]
        s = np.arange(T)
        by_successor = np.zeros(T, dtype=bool)
        ok = s + h < T
        by_successor[ok] = k[s[ok] + h] >= h
        by_terminal = (next_end - s) < h
        valid = self._valid_host
        valid[pos] = by_successor | by_terminal
        # indices of the episode that was open before this call (the n0 newest old transitions): the same two events
        first_end = int(ends[0]) if len(ends) else None
        for j in range(1, min(n0, h, C - T) + 1):  # (an old index this call overwrote is a new index now)
            idx = (c0 - j) % C
            t_succ = h - j  # the new transition that is h steps after it
            ok_succ = t_succ < T and (first_end is None or first_end >= t_succ)
            ok_term = first_end is not None and first_end + j < h
            if ok_succ or ok_term:
                valid[idx] = True
        self._terminal_host[pos] = term
        self.add_count = self.add_count + T
        self._num_transitions_in_current_episode = int(k[T - 1]) + 1  # (a terminal leaves its episode's count: add resets on the NEXT call)
        self._num_valid_indices = int(valid.sum())
        self._valid_dirty = True

    def load_columns(self, columns: Dict[str, torch.Tensor], mark_all_valid: bool = False):
        """Bulk ingestion of an offline dataset: `columns[key]` holds
        n <= capacity consecutive transitions; equivalent to n `add` calls with stack_size == 1
        but done as whole-column device copies.  Validity follows the add rules unless
        mark_all_valid (throughput runs)."""
        assert self._stack_size == 1 and self.is_empty(), "bulk load needs an empty, unstacked buffer"
        n = columns["observation"].shape[0]
        assert n <= self._replay_capacity
        if not self._initialized_buffer:
            first = {k: v[0].cpu().numpy() if v.dim() > 1 else v[0].cpu().numpy()[()] for k, v in columns.items()}
            first["terminal"] = bool(first["terminal"])
            self.initialize_buffer(**first)
        for k, v in columns.items():
            self._store[k][:n].copy_(v.to(self._store[k].dtype))
        term = columns["terminal"].cpu().numpy().astype(bool)
        self._terminal_host[:n] = term
        self.add_count = np.array(n)
        valid = np.zeros(self._replay_capacity, dtype=bool)
        if mark_all_valid:
            valid[:n] = True
        else:
            # add rules (:491-522) in closed form for stack_size == 1: every index of a finished
            # episode is valid; in the still-open last episode index i is valid once i + h exists
            h = self._update_horizon
            ends = np.flatnonzero(term)
            last_end = int(ends[-1]) if len(ends) else -1
            ar = np.arange(n)
            valid[:n] = (ar <= last_end) | (ar + h <= n - 1)
        self._valid_host = valid
        self._num_valid_indices = int(valid.sum())
        self._valid_dirty = True
        ep_start = 0 if not len(np.flatnonzero(term)) else int(np.flatnonzero(term)[-1]) + 1
        self._num_transitions_in_current_episode = n - ep_start

    def is_empty(self) -> bool:
        return self.add_count == 0

    def is_full(self) -> bool:
        return self.add_count >= self._replay_capacity

    def cursor(self) -> int:
        return int(self.add_count % self._replay_capacity)

    def is_valid_transition(self, index):
        return self._valid_host[index]

    @property
    def _is_index_valid(self) -> torch.Tensor:
        return torch.from_numpy(self._valid_host.copy())

    # ---- sampling ----------------------------------------------------------------------------
    def _valid_indices(self) -> torch.Tensor:
        if self._valid_dirty or self._valid_indices_dev is None:
            self._valid_indices_dev = torch.from_numpy(np.flatnonzero(self._valid_host)).to(self.device)
            self._valid_dirty = False
        return self._valid_indices_dev

    def sample_index_batch(self, batch_size: int) -> torch.Tensor:
        if self._num_valid_indices == 0:
            raise RuntimeError(f"Cannot sample {batch_size} since there are no valid indices so far.")
        if self._num_valid_indices == self._replay_capacity:
            # every slot valid: valid_indices is arange(capacity), so valid_indices[pick] == pick
            return torch.randint(self._replay_capacity, (batch_size,), device=self.device)
        valid_indices = self._valid_indices()
        pick = torch.randint(valid_indices.shape[0], (batch_size,), device=self.device)
        return valid_indices[pick]

    def sample_all_valid_transitions(self):
        valid_indices = self._valid_indices()
        return self.sample_transition_batch(batch_size=len(valid_indices), indices=valid_indices)

    def sample_transition_batch(self, batch_size=None, indices=None, state_preprocessor=None,
                                state_dtype=None):
        """circular_replay_buffer.py:614-706.

        state_preprocessor (optional, extension): a reagent_amd Preprocessor whose table is 1:1
        (no ENUM expansion) — `state` / `next_state` are then returned already normalized
        (Preprocessor.forward with all features present fused into the gather), in `state_dtype`
        (torch.float32 default, or torch.bfloat16 = the network-ready layout of the bf16 path)."""
        if batch_size is None:
            batch_size = self._batch_size
        if indices is None:
            indices = self.sample_index_batch(batch_size)
        else:
            assert isinstance(indices, torch.Tensor), (
                f"Indices {indices} have type {type(indices)} instead of torch.Tensor"
            )
            indices = indices.to(device=self.device, dtype=torch.int64)
        assert len(indices) == batch_size
        indices = indices.contiguous()
        B, dev, S = batch_size, self.device, self._stack_size
        if self._decays_dev is None or self._decays_dev.device != dev:
            self._decays_dev = self._decays.reshape(-1).to(device=dev, dtype=torch.float32)

        steps = torch.empty(B, dtype=torch.int64, device=dev)
        next_indices = torch.empty(B, dtype=torch.int64, device=dev)
        terminal = torch.empty(B, dtype=torch.bool, device=dev)
        reward = torch.empty(B, dtype=torch.float32, device=dev)
        reward_col = self._store["reward"]
        if reward_col.dtype != torch.float32:  # e.g. integer rewards: promoted like `store * decays`
            reward_col = reward_col.float()
        ops.replay_nstep(indices, self._store["terminal"], reward_col, self._decays_dev,
                         self._replay_capacity, self._update_horizon, steps, next_indices, terminal, reward)

        norm = None
        if state_preprocessor is not None:
            assert S == 1 and state_preprocessor.elementwise, "normalize-on-gather needs a 1:1 column table"
            assert self._store["observation"].dtype == torch.float32
            norm = (state_preprocessor._col_table, state_preprocessor._quantiles)

        def out_for(key, normalized=False):
            shape = self._shapes[key]
            full = (B, *shape, S) if S > 1 else (B, *shape)
            dt = (state_dtype or torch.float32) if normalized else self._store[key].dtype
            return torch.empty(full, dtype=dt, device=dev)

        # return_as_timeline_format (:659-664, :716-741): next_* elements and `reward` are python lists, entry i holding
        # the steps[i] stored rows that follow transition i (its own rows for `reward`).  One gather over all
        # sum(steps) rows per element, split into per-transition views on the host (one read of `steps`: the lists'
        # shapes are data dependent).
        timeline = self._return_as_timeline_format
        cols_t, lists = [], {}
        if timeline:
            steps_host = steps.tolist()
            T = sum(steps_host)
            sample = torch.repeat_interleave(torch.arange(B, device=dev), steps, output_size=T)
            within = torch.arange(T, device=dev) - (torch.cumsum(steps, 0) - steps)[sample]
            rows_cur = ((indices[sample] + within) % self._replay_capacity).contiguous()
            rows_next = ((indices[sample] + 1 + within) % self._replay_capacity).contiguous()

            def list_for(name, key, rows):
                shape = self._shapes[key]
                dst = torch.empty((T, *shape, S) if S > 1 else (T, *shape), dtype=self._store[key].dtype, device=dev)
                cols_t.append((self._store[key], dst, rows))
                lists[name] = dst

        results, cols = {}, []
        for name in self._transition_elements:
            if name == "state":
                key, idx = "observation", indices
            elif timeline and (name == "reward" or name == "next_state" or (name.startswith("next_") and name[5:] in self._store)):
                list_for(name, "observation" if name == "next_state" else name[5:] if name != "reward" else "reward",
                         rows_cur if name == "reward" else rows_next)
                continue
            elif timeline and name.startswith("next_") and name[5:] in self._sparse:
                raise NotImplementedError("sparse replay elements in the timeline format")
            elif name == "next_state":
                key, idx = "observation", next_indices
            elif name == "reward" and self._return_everything_as_stack:
                key, idx = "reward", indices  # the stored rewards as a stack at `indices`, not the n-step sum (:680-683)
            elif name in ("indices", "terminal", "reward", "step"):
                continue
            elif name in self._sparse or (name.startswith("next_") and name[len("next_"):] in self._sparse):
                # Dict[feature -> (offsets, ids[, scores])], circular_replay_buffer.py:178-195, :247-274
                el = self._sparse.get(name) or self._sparse[name[len("next_"):]]
                results[name] = el.sample_to_output(indices if name in self._sparse else next_indices)
                continue
            elif name in self._store:
                key, idx = name, indices
            elif name.startswith("next_"):
                key, idx = name[len("next_"):], next_indices
                assert key in self._store, f"{key} is not in {self._store.keys()}"
            else:
                results[name] = None
                continue
            is_state = norm is not None and name in ("state", "next_state")
            dst = out_for(key, normalized=is_state)
            cols.append((self._store[key], dst, idx, norm) if is_state else (self._store[key], dst, idx))
            results[name] = dst
        ops.replay_gather(cols, self._replay_capacity, S, B)
        if cols_t and T > 0:
            ops.replay_gather(cols_t, self._replay_capacity, S, T)
        for name, flat in lists.items():
            results[name] = list(torch.split(flat, steps_host))
        results.update(indices=indices, terminal=terminal, step=steps)
        results.setdefault("reward", reward)

        batch_arrays = []
        for name in self._transition_elements:
            batch = results[name]
            if isinstance(batch, torch.Tensor) and batch.ndim == 1:
                batch = batch.unsqueeze(1)
            batch_arrays.append(batch)
        return self._batch_type(*batch_arrays)

    def sample_dqn_input(self, num_actions: int, batch_size=None, indices=None, state_preprocessor=None,
                         state_dtype=None):
        """sample_transition_batch + DiscreteDqnInputMaker (trainer_preprocessor.py:100-158) [+ the state
        Preprocessor] as ONE launch (rg_replay_dqn_batch): the n-step bookkeeping, both state gathers and
        the one-hot / not_terminal / exp(log_prob) work, bit-identical to the three-launch path.  Returns
        None when the store is not of the shape the fused kernel serves (stacked frames, non-fp32 or
        non-vector observations, an ENUM-expanding preprocessor, ...): callers then use the generic path."""
        from ..core import types as rlt

        st = self._store
        obs, act = st.get("observation"), st.get("action")
        if (self._stack_size != 1 or self._return_everything_as_stack or self._return_as_timeline_format or obs is None
                or obs.dtype != torch.float32
                or obs.dim() != 2 or act is None
                or act.dtype != torch.int64 or act.dim() != 1 or "log_prob" not in st
                or st["reward"].dtype != torch.float32):
            return None
        if state_preprocessor is not None and not state_preprocessor.elementwise:
            return None
        if state_preprocessor is None and state_dtype not in (None, torch.float32):
            return None
        mask = st.get("possible_actions_mask")
        if mask is not None and (mask.dtype != torch.float32 or mask.shape[1:] != (num_actions,)):
            return None
        if batch_size is None:
            batch_size = self._batch_size
        if isinstance(indices, ops.PooledIndices):  # a replayed step: the row of an index pool a device cursor points at
            assert indices.numel() == batch_size and self._num_valid_indices == self._replay_capacity
        else:
            if indices is None:
                indices = self.sample_index_batch(batch_size)
            else:
                indices = indices.to(device=self.device, dtype=torch.int64)
            assert len(indices) == batch_size
            indices = indices.contiguous()
        B, dev, F, A = batch_size, self.device, obs.shape[1], num_actions
        if self._decays_dev is None or self._decays_dev.device != dev:
            self._decays_dev = self._decays.reshape(-1).to(device=dev, dtype=torch.float32)
        view = L.ReplayView()
        view.observation, view.action, view.reward = obs.data_ptr(), act.data_ptr(), st["reward"].data_ptr()
        view.terminal, view.log_prob = st["terminal"].data_ptr(), st["log_prob"].data_ptr()
        view.possible_actions_mask = mask.data_ptr() if mask is not None else None
        view.mdp_id, view.sequence_number = None, None
        view.decays = self._decays_dev.data_ptr()
        view.capacity, view.n_features, view.n_actions = self._replay_capacity, F, A
        view.update_horizon = self._update_horizon
        f32 = dict(dtype=torch.float32, device=dev)
        sdt = state_dtype or torch.float32
        out = dict(state=torch.empty(B, F, dtype=sdt, device=dev), next_state=torch.empty(B, F, dtype=sdt, device=dev),
                   action=torch.empty(B, A, **f32), next_action=torch.empty(B, A, **f32), reward=torch.empty(B, 1, **f32),
                   not_terminal=torch.empty(B, 1, **f32), possible_actions_mask=torch.empty(B, A, **f32),
                   possible_next_actions_mask=torch.empty(B, A, **f32), action_probability=torch.empty(B, 1, **f32))
        pre = state_preprocessor
        if not ops.replay_dqn_batch(view, indices, pre._col_table if pre is not None else None,
                                    pre._quantiles if pre is not None else None, out):
            return None
        return rlt.DiscreteDqnInput(
            state=rlt.FeatureData(float_features=out["state"]), action=out["action"],
            next_state=rlt.FeatureData(float_features=out["next_state"]), next_action=out["next_action"],
            possible_actions_mask=out["possible_actions_mask"],
            possible_next_actions_mask=out["possible_next_actions_mask"], reward=out["reward"],
            not_terminal=out["not_terminal"], step=None, time_diff=None,
            extras=rlt.ExtraData(mdp_id=None, sequence_number=None, action_probability=out["action_probability"],
                                 max_num_actions=None, metrics=None))

    def sample_policy_input(self, input_maker, batch_size=None, indices=None, state_preprocessor=None, state_dtype=None):
        """sample_transition_batch + PolicyNetworkInputMaker (trainer_preprocessor.py:175-227) [+ the state Preprocessor] as
        ONE launch (rg_replay_policy_batch, round 6): the n-step bookkeeping, both state gathers, the two rescaled action rows,
        not_terminal and exp(log_prob) — bit-identical to rg_replay_nstep + rg_replay_gather + rg_make_policy_input.  Returns
        None when the store is not of the shape the fused kernel serves (stacked frames, non-fp32 or non-vector
        observations / actions, an ENUM-expanding preprocessor, ...): callers then use the generic path."""
        from ..core import types as rlt

        st = self._store
        obs, act = st.get("observation"), st.get("action")
        if (self._stack_size != 1 or self._return_everything_as_stack or self._return_as_timeline_format or obs is None
                or obs.dtype != torch.float32 or obs.dim() != 2 or act is None or act.dtype != torch.float32
                or act.dim() != 2 or "log_prob" not in st or st["log_prob"].dtype != torch.float32
                or st["reward"].dtype != torch.float32 or not hasattr(input_maker, "_ranges")):
            return None
        if state_preprocessor is not None and not state_preprocessor.elementwise:
            return None
        if state_preprocessor is None and state_dtype not in (None, torch.float32):
            return None
        if batch_size is None:
            batch_size = self._batch_size
        if indices is None:
            indices = self.sample_index_batch(batch_size)
        else:
            indices = indices.to(device=self.device, dtype=torch.int64)
        assert len(indices) == batch_size
        indices = indices.contiguous()
        B, dev, F, A = batch_size, self.device, obs.shape[1], act.shape[1]
        if self._decays_dev is None or self._decays_dev.device != dev:
            self._decays_dev = self._decays.reshape(-1).to(device=dev, dtype=torch.float32)
        ranges = input_maker._ranges(dev, A)
        view = L.PolicyReplayView()
        view.observation, view.action, view.reward = obs.data_ptr(), act.data_ptr(), st["reward"].data_ptr()
        view.terminal, view.log_prob = st["terminal"].data_ptr(), st["log_prob"].data_ptr()
        view.decays, view.ranges = self._decays_dev.data_ptr(), ranges.data_ptr()
        view.capacity, view.n_features, view.action_dim = self._replay_capacity, F, A
        view.update_horizon = self._update_horizon
        f32 = dict(dtype=torch.float32, device=dev)
        sdt = state_dtype or torch.float32
        out = dict(state=torch.empty(B, F, dtype=sdt, device=dev), next_state=torch.empty(B, F, dtype=sdt, device=dev),
                   action=torch.empty(B, A, **f32), next_action=torch.empty(B, A, **f32), reward=torch.empty(B, 1, **f32),
                   not_terminal=torch.empty(B, 1, **f32), action_probability=torch.empty(B, 1, **f32))
        pre = state_preprocessor
        if not ops.replay_policy_batch(view, indices, pre._col_table if pre is not None else None,
                                       pre._quantiles if pre is not None else None, out):
            return None
        return rlt.PolicyNetworkInput(
            state=rlt.FeatureData(out["state"]), next_state=rlt.FeatureData(out["next_state"]),
            action=rlt.FeatureData(out["action"]), next_action=rlt.FeatureData(out["next_action"]), reward=out["reward"],
            not_terminal=out["not_terminal"], step=None, time_diff=None,
            extras=rlt.ExtraData(mdp_id=None, sequence_number=None, action_probability=out["action_probability"],
                                 max_num_actions=None, metrics=None))

    def get_transition_elements(self):
        extra_names = []
        for name in self._extra_keys:
            for prefix in ["", "next_"]:
                extra_names.append(f"{prefix}{name}")
        return ["state", "action", "reward", "next_state", "next_action", "next_reward", "terminal",
                "indices", "step", *extra_names]

    # ---- checkpointing (circular_replay_buffer.py:795-890; same files, same names) ------------
    def _generate_filename(self, checkpoint_dir, name, suffix):
        return os.path.join(checkpoint_dir, "{}_ckpt.{}.gz".format(name, suffix))

    def _return_checkpointable_elements(self):
        """:798-811 — every public attribute plus one `$store$_<key>` entry per storage column."""
        elements = {}
        for member_name, member in self.__dict__.items():
            if member_name == "_store":
                for array_name, array in self._store.items():
                    elements[STORE_FILENAME_PREFIX + array_name] = array
            elif not member_name.startswith("_") and member_name not in _NOT_CHECKPOINTED:
                elements[member_name] = member
        return elements

    # private bookkeeping the reference does not checkpoint (a buffer restored by it forgets which
    # indices are valid); written as extra `$rg$_*` files the reference ignores
    _EXTRA_STATE = ("_valid_host", "_terminal_host", "_num_transitions_in_current_episode")

    def save(self, checkpoint_dir, iteration_number):
        """:813-861.  One gzip file per element: storage columns and numpy attributes with
        np.save(allow_pickle=False), anything else pickled; the checkpoint CHECKPOINT_DURATION
        iterations back is deleted.  Device columns are copied to the host for writing."""
        if not os.path.exists(checkpoint_dir):
            return
        if self._sparse:  # (the reference's np.save(allow_pickle=False) of their object arrays raises as well)
            raise ValueError("sparse (id-list) replay elements cannot be checkpointed")
        elements = self._return_checkpointable_elements()
        for name in self._EXTRA_STATE:
            elements[_RG_FILENAME_PREFIX + name] = np.asarray(getattr(self, name))
        for attr, value in elements.items():
            with open(self._generate_filename(checkpoint_dir, attr, iteration_number), "wb") as f:
                with gzip.GzipFile(fileobj=f, mode="wb") as outfile:
                    if attr.startswith(STORE_FILENAME_PREFIX):
                        np.save(outfile, value.cpu().numpy(), allow_pickle=False)
                    elif isinstance(value, np.ndarray):
                        np.save(outfile, value, allow_pickle=False)
                    else:
                        pickle.dump(value, outfile)
            stale = iteration_number - CHECKPOINT_DURATION
            if stale >= 0:
                try:
                    os.remove(self._generate_filename(checkpoint_dir, attr, stale))
                except FileNotFoundError:
                    pass

    def load(self, checkpoint_dir, suffix):
        """:863-890.  Nothing is loaded unless every expected file exists (FileNotFoundError).  A
        checkpoint written by the reference has no `$rg$_*` files: validity is then rebuilt from the
        `terminal` column with the add rules (possible while the ring has not wrapped, stack_size 1)."""
        elements = self._return_checkpointable_elements()
        for attr in elements:
            filename = self._generate_filename(checkpoint_dir, attr, suffix)
            if not os.path.exists(filename):
                raise FileNotFoundError(None, None, "Missing file: {}".format(filename))
        for attr, current in elements.items():
            with open(self._generate_filename(checkpoint_dir, attr, suffix), "rb") as f:
                with gzip.GzipFile(fileobj=f) as infile:
                    if attr.startswith(STORE_FILENAME_PREFIX):
                        key = attr[len(STORE_FILENAME_PREFIX):]
                        arr = torch.from_numpy(np.load(infile, allow_pickle=False))
                        self._store[key] = arr.to(device=self.device, dtype=self._store[key].dtype)
                    elif isinstance(current, np.ndarray):
                        self.__dict__[attr] = np.load(infile, allow_pickle=False)
                    else:
                        self.__dict__[attr] = pickle.load(infile)
        n = int(min(int(self.add_count), self._replay_capacity))
        extras = {name: self._generate_filename(checkpoint_dir, _RG_FILENAME_PREFIX + name, suffix)
                  for name in self._EXTRA_STATE}
        if all(os.path.exists(f) for f in extras.values()):
            for name, filename in extras.items():
                with open(filename, "rb") as f, gzip.GzipFile(fileobj=f) as infile:
                    value = np.load(infile, allow_pickle=False)
                setattr(self, name, int(value) if value.ndim == 0 else value.astype(bool))
        else:
            if int(self.add_count) > self._replay_capacity or self._stack_size != 1:
                raise ValueError("checkpoint without validity state ($rg$_* files): it can only be rebuilt "
                                 "for an unwrapped, unstacked buffer")
            term = self._store["terminal"][:n].cpu().numpy().astype(bool)
            self._terminal_host = np.zeros(self._replay_capacity, dtype=bool)
            self._terminal_host[:n] = term
            ends = np.flatnonzero(term)
            last_end = int(ends[-1]) if len(ends) else -1
            ar = np.arange(n)
            valid = np.zeros(self._replay_capacity, dtype=bool)
            valid[:n] = (ar <= last_end) | (ar + self._update_horizon <= n - 1)
            self._valid_host = valid
            self._num_transitions_in_current_episode = n - (last_end + 1)
        self._num_valid_indices = int(self._valid_host.sum())
        self._valid_dirty = True
        self._valid_indices_dev = None
