"""Serving path of a trained discrete DQN, with the surface of
reagent/prediction/predictor_wrapper.py:94-152:

    DiscreteDqnWithPreprocessor : (float features, presence) -> Preprocessor -> q-network
    DiscreteDqnPredictorWrapper : ... -> (action_names, q_values)

The reference freezes these with TorchScript tracing; here the two launches of a request
(rg_normalize_dense, then the whole network in one fused-MLP launch) are replayed from a HIP graph
captured per batch size (`capture`), so a request costs one graph launch and no Python dispatch
in between.  Sparse id-list features are outside the dense hot path and are rejected.
"""
from typing import Dict, List, NamedTuple, Optional, Tuple

import torch

from ..core import types as rlt
from ..models.base import ModelBase
from ..preprocessing import Preprocessor


class ServingFeatureData(NamedTuple):
    """reagent/core/types.py:434-437"""

    float_features_with_presence: Tuple[torch.Tensor, torch.Tensor]
    id_list_features: dict = {}
    id_score_list_features: dict = {}


def _dense_only(state: ServingFeatureData):
    if state.id_list_features or state.id_score_list_features:
        raise NotImplementedError("sparse id-list serving features are not on the MI355X dense path")
    return state.float_features_with_presence


class DiscreteDqnWithPreprocessor(ModelBase):
    """predictor_wrapper.py:94-127"""

    def __init__(self, model: ModelBase, state_preprocessor: Preprocessor, state_feature_config=None):
        super().__init__()
        self.model = model
        self.state_preprocessor = state_preprocessor
        self.state_feature_config = state_feature_config

    @torch.no_grad()
    def forward(self, state: ServingFeatureData) -> torch.Tensor:
        x, presence = _dense_only(state)
        feats = self.state_preprocessor(x, presence)
        return self.model(rlt.FeatureData(float_features=feats))

    def input_prototype(self):
        return (ServingFeatureData(float_features_with_presence=self.state_preprocessor.input_prototype()),)


class DiscreteDqnPredictorWrapper(torch.nn.Module):
    """predictor_wrapper.py:130-152: forward(state) -> (action_names, q_values)."""

    def __init__(self, dqn_with_preprocessor: DiscreteDqnWithPreprocessor, action_names: List[str],
                 state_feature_config=None) -> None:
        super().__init__()
        self.dqn_with_preprocessor = dqn_with_preprocessor
        self.action_names = list(action_names)
        self._graphs: Dict[int, tuple] = {}

    @torch.no_grad()
    def capture(self, batch_size: int) -> None:
        """Record the request for `batch_size` rows as a HIP graph (static input / output buffers).
        The weights are baked in as staged at capture time: capture again after loading new ones."""
        x, presence = self.dqn_with_preprocessor.state_preprocessor.input_prototype()
        dev = x.device
        F = x.shape[1]
        sx = torch.zeros(batch_size, F, dtype=torch.float32, device=dev)
        sp = torch.ones(batch_size, F, dtype=torch.uint8, device=dev)
        state = ServingFeatureData(float_features_with_presence=(sx, sp))
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up: weight staging and workspace allocation stay out of the graph
            for _ in range(2):
                self.dqn_with_preprocessor(state)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.dqn_with_preprocessor(state)
        self._graphs[batch_size] = (graph, sx, sp, out)

    @torch.no_grad()
    def forward(self, state: ServingFeatureData) -> Tuple[List[str], torch.Tensor]:
        x, presence = _dense_only(state)
        entry = self._graphs.get(x.shape[0])
        if entry is None:
            return self.action_names, self.dqn_with_preprocessor(state)
        graph, sx, sp, out = entry
        sx.copy_(x)
        sp.copy_(presence.view(torch.uint8) if presence.dtype == torch.bool else presence)
        graph.replay()
        return self.action_names, out.clone()
