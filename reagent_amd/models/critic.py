"""FullyConnectedCritic (reagent/models/critic.py:37-92): q = fc(cat(state, action))."""
from typing import List

import torch

from ..core import types as rlt
from .base import ModelBase
from .fully_connected_network import FullyConnectedNetwork


class FullyConnectedCritic(ModelBase):
    def __init__(
        self,
        state_dim: int,
        action_dim: int,
        sizes: List[int],
        activations: List[str],
        use_batch_norm: bool = False,
        use_layer_norm: bool = False,
        output_dim: int = 1,
        final_activation: str = "linear",
    ) -> None:
        super().__init__()
        assert state_dim > 0, "state_dim must be > 0, got {}".format(state_dim)
        assert action_dim > 0, "action_dim must be > 0, got {}".format(action_dim)
        self.state_dim = state_dim
        self.action_dim = action_dim
        assert len(sizes) == len(activations), (
            "The numbers of sizes and activations must match; got {} vs {}".format(len(sizes), len(activations))
        )
        self.fc = FullyConnectedNetwork(
            [state_dim + action_dim] + list(sizes) + [output_dim],
            list(activations) + [final_activation],
            use_batch_norm=use_batch_norm,
            use_layer_norm=use_layer_norm,
        )

    def input_prototype(self):
        return (
            rlt.FeatureData(torch.randn(1, self.state_dim)),
            rlt.FeatureData(torch.randn(1, self.action_dim)),
        )

    def forward(self, state, action):
        s, a = state.float_features, action.float_features
        assert s.dim() == 2, f"Expected state to have 2 dimensions (batch, features), but got {s.dim()}"
        assert a.dim() == 2, f"Expected action to have 2 dimensions (batch, features), but got {a.dim()}"
        assert s.size(0) == a.size(0), "Batch sizes of state and action mismatch"
        return self.fc(torch.cat((s, a), dim=-1))
