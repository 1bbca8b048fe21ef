"""Net builders for the networks on the hot path: the configuration objects a ReAgent model manager holds and asks
for its networks (reagent/net_builder/**; §8(b)'s "called by" column).  A builder is its hyper-parameters (the reference's
field names and defaults) plus one `build_*` method with the reference's arguments; what it returns is this package's
model class — an FC stack staged for the HIP kernels — sized from the normalization data exactly as the reference sizes
it (`get_num_output_features`: an ENUM feature widens the input by its number of possible values).

All builders live here, once: they differ only in their fields and in the model they construct.  The reference's module
paths exist as namespaces made at the bottom of this file, not as files of their own
(`reagent_amd.net_builder.discrete_dqn.FullyConnected`, `.quantile_dqn.Quantile`, ...; `import reagent_amd.net_builder.value` works).
Sparse-feature embeddings (`embedding_dim`, `FullyConnectedWithEmbedding`) are not on the path and are
rejected loudly.  `state_feature_config` is accepted where the reference takes it and only forwarded to the serving
wrapper (dense features).
"""
import sys
import types
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from ..core.parameters import NormalizationData
from ..models import (
    CategoricalDQN,
    DuelingQNetwork,
    FullyConnectedActor,
    FullyConnectedCritic,
    FullyConnectedDQN,
    GaussianFullyConnectedActor,
)
from ..models.fully_connected_network import FloatFeatureFullyConnected

CONTINUOUS_ACTION = "CONTINUOUS_ACTION"  # reagent/preprocessing/identify_types.py


def get_num_output_features(normalization_parameters: Dict[int, object]) -> int:
    """width of Preprocessor's output for these features (reagent/preprocessing/normalization.py:188-198)"""
    return sum(len(p.possible_values) if p.feature_type == "ENUM" else 1 for p in normalization_parameters.values())


def _dim(data: NormalizationData) -> int:
    return get_num_output_features(data.dense_normalization_parameters)


def _two(a, b):
    return field(default_factory=lambda: [a, b])


@dataclass
class _Stack:
    """hidden sizes + activations of an FC stack, one activation per layer"""

    sizes: List[int] = _two(256, 128)
    activations: List[str] = _two("relu", "relu")

    def __post_init__(self):
        assert len(self.sizes) == len(self.activations), (
            f"Must have the same numbers of sizes and activations; got: {self.sizes}, {self.activations}")

    def _get_input_dim(self, state_normalization_data: NormalizationData) -> int:
        return _dim(state_normalization_data)


# ---- discrete DQN (reagent/net_builder/discrete_dqn_net_builder.py, discrete_dqn/{fully_connected,dueling}.py) ---------
class DiscreteDQNNetBuilder:
    def build_serving_module(self, q_network, state_normalization_data: NormalizationData, action_names: List[str],
                             state_feature_config=None, predictor_wrapper_type=None) -> torch.nn.Module:
        """discrete_dqn_net_builder.py:52-72: Preprocessor -> Q-network -> (action names, Q-values); here the wrapper stays
        on the network's device and can capture the call as one HIP graph (prediction/predictor_wrapper.py)"""
        from ..prediction.predictor_wrapper import DiscreteDqnPredictorWrapper, DiscreteDqnWithPreprocessor
        from ..preprocessing import Preprocessor

        device = next(q_network.parameters()).device
        pre = Preprocessor(state_normalization_data.dense_normalization_parameters, device=device)
        with_pre = DiscreteDqnWithPreprocessor(q_network.eval(), pre, state_feature_config)
        return (predictor_wrapper_type or DiscreteDqnPredictorWrapper)(with_pre, action_names, state_feature_config)


@dataclass
class DiscreteDqnFullyConnected(_Stack, DiscreteDQNNetBuilder):
    dropout_ratio: float = 0.0
    use_batch_norm: bool = False

    def build_q_network(self, state_feature_config, state_normalization_data: NormalizationData, output_dim: int):
        return FullyConnectedDQN(state_dim=self._get_input_dim(state_normalization_data), action_dim=output_dim,
                                 sizes=self.sizes, activations=self.activations, dropout_ratio=self.dropout_ratio,
                                 use_batch_norm=self.use_batch_norm)


@dataclass
class DiscreteDqnDueling(_Stack, DiscreteDQNNetBuilder):
    def build_q_network(self, state_feature_config, state_normalization_data: NormalizationData, output_dim: int):
        return DuelingQNetwork.make_fully_connected(self._get_input_dim(state_normalization_data), output_dim, self.sizes,
                                                    self.activations)


# ---- quantile-regression DQN (quantile_dqn_net_builder.py, quantile_dqn/{quantile,dueling_quantile}.py) -----------------
class QRDQNNetBuilder(DiscreteDQNNetBuilder):
    pass


@dataclass
class Quantile(_Stack, QRDQNNetBuilder):
    dropout_ratio: float = 0.0

    def build_q_network(self, state_normalization_data: NormalizationData, output_dim: int, num_atoms: int):
        return FullyConnectedDQN(state_dim=self._get_input_dim(state_normalization_data), action_dim=output_dim,
                                 sizes=self.sizes, num_atoms=num_atoms, activations=self.activations,
                                 dropout_ratio=self.dropout_ratio)


@dataclass
class DuelingQuantile(_Stack, QRDQNNetBuilder):
    def build_q_network(self, state_normalization_data: NormalizationData, output_dim: int, num_atoms: int):
        return DuelingQNetwork.make_fully_connected(self._get_input_dim(state_normalization_data), output_dim,
                                                    layers=self.sizes, activations=self.activations, num_atoms=num_atoms)


# ---- C51 (categorical_dqn_net_builder.py, categorical_dqn/categorical.py) ------------------------------------------------
@dataclass
class Categorical(_Stack, DiscreteDQNNetBuilder):
    def build_q_network(self, state_normalization_data: NormalizationData, output_dim: int, num_atoms: int, qmin: int,
                        qmax: int):
        dist = FullyConnectedDQN(state_dim=self._get_input_dim(state_normalization_data), action_dim=output_dim,
                                 num_atoms=num_atoms, sizes=self.sizes, activations=self.activations, use_batch_norm=False,
                                 dropout_ratio=0.0)
        return CategoricalDQN(dist, qmin=qmin, qmax=qmax, num_atoms=num_atoms)


# ---- actors (continuous_actor_net_builder.py, continuous_actor/{gaussian_fully_connected,fully_connected}.py,
#      discrete_actor/fully_connected.py) -----------------------------------------------------------------------------------
@dataclass
class _ActorStack(_Stack):
    sizes: List[int] = _two(128, 64)
    use_batch_norm: bool = False
    use_layer_norm: bool = False


@dataclass
class GaussianFullyConnected(_ActorStack):
    use_l2_normalization: bool = False
    embedding_dim: Optional[int] = None

    def __post_init__(self):
        super().__post_init__()
        if self.embedding_dim is not None:
            raise NotImplementedError("sparse-feature embeddings in front of the actor are not on the path (SURVEY.md §8)")

    @property
    def default_action_preprocessing(self) -> str:
        return CONTINUOUS_ACTION

    def build_actor(self, state_feature_config, state_normalization_data: NormalizationData,
                    action_normalization_data: NormalizationData):
        return GaussianFullyConnectedActor(
            state_dim=_dim(state_normalization_data), action_dim=_dim(action_normalization_data), sizes=self.sizes,
            activations=self.activations, use_batch_norm=self.use_batch_norm, use_layer_norm=self.use_layer_norm,
            use_l2_normalization=self.use_l2_normalization)


@dataclass
class ContinuousActorFullyConnected(_ActorStack):
    action_activation: str = "tanh"
    exploration_variance: Optional[float] = None

    @property
    def default_action_preprocessing(self) -> str:
        return CONTINUOUS_ACTION

    def build_actor(self, state_feature_config, state_normalization_data: NormalizationData,
                    action_normalization_data: NormalizationData):
        return FullyConnectedActor(
            state_dim=_dim(state_normalization_data), action_dim=_dim(action_normalization_data), sizes=self.sizes,
            activations=self.activations, use_batch_norm=self.use_batch_norm, action_activation=self.action_activation,
            exploration_variance=self.exploration_variance)


@dataclass
class DiscreteActorFullyConnected(_ActorStack):
    action_activation: str = "tanh"
    exploration_variance: Optional[float] = None

    def build_actor(self, state_normalization_data: NormalizationData, num_actions: int):
        return FullyConnectedActor(
            state_dim=_dim(state_normalization_data), action_dim=num_actions, sizes=self.sizes, activations=self.activations,
            use_batch_norm=self.use_batch_norm, action_activation=self.action_activation,
            exploration_variance=self.exploration_variance)


# ---- critics and value networks (parametric_dqn/fully_connected.py, value/fully_connected.py) ---------------------------
@dataclass
class ParametricDqnFullyConnected(_ActorStack):
    final_activation: str = "linear"

    def build_q_network(self, state_normalization_data: NormalizationData, action_normalization_data: NormalizationData,
                        output_dim: int = 1):
        return FullyConnectedCritic(
            state_dim=_dim(state_normalization_data), action_dim=_dim(action_normalization_data), sizes=self.sizes,
            activations=self.activations, use_batch_norm=self.use_batch_norm, use_layer_norm=self.use_layer_norm,
            output_dim=output_dim, final_activation=self.final_activation)


@dataclass
class ValueFullyConnected(_Stack):
    use_layer_norm: bool = False

    def build_value_network(self, state_normalization_data: NormalizationData, output_dim: int = 1) -> torch.nn.Module:
        return FloatFeatureFullyConnected(state_dim=_dim(state_normalization_data), output_dim=output_dim, sizes=self.sizes,
                                          activations=self.activations, use_layer_norm=self.use_layer_norm)


# the reference's module paths -> these classes, under the reference's names
BUILDERS = {
    "discrete_dqn": dict(FullyConnected=DiscreteDqnFullyConnected, Dueling=DiscreteDqnDueling),
    "quantile_dqn": dict(Quantile=Quantile, DuelingQuantile=DuelingQuantile),
    "categorical_dqn": dict(Categorical=Categorical),
    "continuous_actor": dict(GaussianFullyConnected=GaussianFullyConnected, FullyConnected=ContinuousActorFullyConnected),
    "discrete_actor": dict(FullyConnected=DiscreteActorFullyConnected),
    "parametric_dqn": dict(FullyConnected=ParametricDqnFullyConnected),
    "value": dict(FullyConnected=ValueFullyConnected),
}

for _name, _classes in BUILDERS.items():  # reagent/net_builder/<family>/*: one importable namespace per family
    _ns = types.ModuleType(f"{__name__}.{_name}", f"reagent/net_builder/{_name}/*: the builders of this family")
    _ns.__dict__.update(_classes, __all__=sorted(_classes))
    sys.modules[_ns.__name__] = globals()[_name] = _ns
