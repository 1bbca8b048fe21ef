"""CategoricalDQN (reagent/models/categorical_dqn.py:12-38): a distributional network whose logits
(B, A, N) become per-action categorical distributions over a fixed support; `forward` returns the
expected values.  The training step evaluates log_dist inside rg_c51_head; the methods here are the
model surface used by policies and evaluation code."""
import torch
import torch.nn.functional as F

from ..core import types as rlt
from .base import ModelBase


class CategoricalDQN(ModelBase):
    def __init__(self, distributional_network: ModelBase, *, qmin: float, qmax: float, num_atoms: int) -> None:
        super().__init__()
        self.distributional_network = distributional_network
        self.support = torch.linspace(qmin, qmax, num_atoms)

    @property
    def fc(self):
        """the FullyConnectedNetwork that the native trainers drive"""
        return self.distributional_network.fc

    def input_prototype(self):
        return self.distributional_network.input_prototype()

    def forward(self, state: rlt.FeatureData):
        dist = self.log_dist(state).exp()
        return (dist * self.support.to(dist.device)).sum(2)

    def log_dist(self, state: rlt.FeatureData) -> torch.Tensor:
        return F.log_softmax(self.distributional_network(state), -1)
