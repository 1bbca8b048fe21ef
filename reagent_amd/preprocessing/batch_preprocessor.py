"""DiscreteDqnBatchPreprocessor of reagent/preprocessing/batch_preprocessor.py:18-66 on the HIP path.

`forward(batch)` takes the dict of column tensors the reference's data loader yields and returns the
same rlt.DiscreteDqnInput; `from_table(table, indices)` builds it straight from an OfflineTable in
HBM.  Both are ONE rg_table_dqn_batch launch: Preprocessor.forward on state and next_state, the two
one-hots, not_terminal = max(possible_next_actions_mask) and the pass-through columns.

Differences from the reference's tensors, none in value: action / next_action one-hots and the masks
are float32 (the reference hands on int64; every trainer promotes them with `.float()`), time_diff and
step are float32 (they are exponents of gamma).  `state_dtype=torch.bfloat16` emits the network-ready
bf16 rows of the fused MLP path instead of fp32.
"""
from typing import Dict, Optional

import torch

from .. import ops
from ..core import types as rlt
from ..data.offline_table import OfflineTable
from .preprocessor import Preprocessor


def batch_to_device(batch: Dict[str, torch.Tensor], device: torch.device):
    return {k: batch[k].to(device) for k in batch}


class BatchPreprocessor(torch.nn.Module):
    pass


class DiscreteDqnBatchPreprocessor(BatchPreprocessor):
    def __init__(self, num_actions: int, state_preprocessor: Preprocessor, use_gpu: bool = True,
                 state_dtype: torch.dtype = torch.float32, device: Optional[torch.device] = None) -> None:
        super().__init__()
        self.num_actions = num_actions
        self.state_preprocessor = state_preprocessor
        # the reference's use_gpu=False (CPU tensors) has no counterpart: the product path is the GPU
        self.device = torch.device(device) if device is not None else state_preprocessor.device
        assert state_dtype in (torch.float32, torch.bfloat16)
        self.state_dtype = state_dtype

    @torch.no_grad()
    def forward(self, batch: Dict[str, torch.Tensor]) -> rlt.DiscreteDqnInput:
        table = OfflineTable(batch, self.num_actions, device=self.device, validate=True)
        return self.from_table(table, torch.arange(len(table), device=self.device))

    @torch.no_grad()
    def from_table(self, table: OfflineTable, indices: torch.Tensor) -> rlt.DiscreteDqnInput:
        pre = self.state_preprocessor
        assert table.num_actions == self.num_actions
        assert table.num_features == len(pre.sorted_features), (
            f"table has {table.num_features} state features, the preprocessor {len(pre.sorted_features)}")
        idx = indices.to(device=table.device, dtype=torch.int64).contiguous()
        B, A, dev = idx.numel(), self.num_actions, table.device
        f32 = dict(dtype=torch.float32, device=dev)
        i64 = dict(dtype=torch.int64, device=dev)
        out = dict(
            state=torch.empty(B, pre.num_output_features, dtype=self.state_dtype, device=dev),
            next_state=torch.empty(B, pre.num_output_features, dtype=self.state_dtype, device=dev),
            action=torch.empty(B, A, **f32), next_action=torch.empty(B, A, **f32),
            reward=torch.empty(B, 1, **f32), time_diff=torch.empty(B, 1, **f32), step=torch.empty(B, 1, **f32),
            not_terminal=torch.empty(B, 1, **f32), possible_actions_mask=torch.empty(B, A, **f32),
            possible_next_actions_mask=torch.empty(B, A, **f32), action_probability=torch.empty(B, 1, **f32),
            mdp_id=torch.empty(B, 1, **i64), sequence_number=torch.empty(B, 1, **i64),
        )
        ops.table_dqn_batch(table, idx, pre._col_table, pre.num_output_features, pre._quantiles, out)
        return rlt.DiscreteDqnInput(
            state=rlt.FeatureData(out["state"]), next_state=rlt.FeatureData(out["next_state"]),
            action=out["action"], next_action=out["next_action"], reward=out["reward"], time_diff=out["time_diff"],
            step=out["step"], not_terminal=out["not_terminal"], possible_actions_mask=out["possible_actions_mask"],
            possible_next_actions_mask=out["possible_next_actions_mask"],
            extras=rlt.ExtraData(mdp_id=out["mdp_id"], sequence_number=out["sequence_number"],
                                 action_probability=out["action_probability"]),
        )


class PolicyNetworkBatchPreprocessor(BatchPreprocessor):
    """batch_preprocessor.py:110-166 (continuous actions: SAC / TD3 on an offline table): the state and
    action preprocessors (rg_normalize_dense, four launches) over the reader's dict; the scalar columns
    are handed on as the reference does (`unsqueeze(1)`), moved to the device."""

    def __init__(self, state_preprocessor: Preprocessor, action_preprocessor: Preprocessor, use_gpu: bool = True,
                 device: Optional[torch.device] = None) -> None:
        super().__init__()
        self.state_preprocessor = state_preprocessor
        self.action_preprocessor = action_preprocessor
        self.device = torch.device(device) if device is not None else state_preprocessor.device

    @torch.no_grad()
    def forward(self, batch: Dict[str, torch.Tensor]) -> rlt.PolicyNetworkInput:
        batch = batch_to_device(batch, self.device)
        sp, ap = self.state_preprocessor, self.action_preprocessor
        return rlt.PolicyNetworkInput(
            state=rlt.FeatureData(sp(batch["state_features"], batch["state_features_presence"])),
            next_state=rlt.FeatureData(sp(batch["next_state_features"], batch["next_state_features_presence"])),
            action=rlt.FeatureData(ap(batch["action"], batch["action_presence"])),
            next_action=rlt.FeatureData(ap(batch["next_action"], batch["next_action_presence"])),
            reward=batch["reward"].unsqueeze(1), time_diff=batch["time_diff"].unsqueeze(1),
            step=batch["step"].unsqueeze(1), not_terminal=batch["not_terminal"].unsqueeze(1),
            extras=rlt.ExtraData(mdp_id=batch["mdp_id"].unsqueeze(1),
                                 sequence_number=batch["sequence_number"].unsqueeze(1),
                                 action_probability=batch["action_probability"].unsqueeze(1)),
        )
