"""Parameter dataclasses with the field names / defaults of reagent/core/parameters.py
(RLParameters :46-67, EvaluationParameters :118-120, NormalizationParameters :138-152).
Objects of the reference's own classes are accepted anywhere these are (duck typing)."""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

CONTINUOUS_TRAINING_ACTION_RANGE = (-1.0, 1.0)


@dataclass(frozen=True)
class RLParameters:
    gamma: float = 0.9
    epsilon: float = 0.1
    target_update_rate: float = 0.001
    maxq_learning: bool = True
    reward_boost: Optional[Dict[str, float]] = None
    temperature: float = 0.01
    softmax_policy: bool = False
    use_seq_num_diff_as_time_diff: bool = False
    q_network_loss: str = "mse"
    set_missing_value_to_zero: bool = False
    tensorboard_logging_freq: int = 0
    predictor_atol_check: float = 0.0
    predictor_rtol_check: float = 5e-5
    time_diff_unit_length: float = 1.0
    multi_steps: Optional[int] = None
    ratio_different_predictions_tolerance: float = 0


@dataclass(frozen=True)
class EvaluationParameters:
    calc_cpe_in_training: bool = True


@dataclass(frozen=True)
class NormalizationParameters:
    feature_type: str
    boxcox_lambda: Optional[float] = None
    boxcox_shift: Optional[float] = None
    mean: Optional[float] = None
    stddev: Optional[float] = None
    possible_values: Optional[List[int]] = None
    quantiles: Optional[List[float]] = None
    min_value: Optional[float] = None
    max_value: Optional[float] = None


class NormalizationKey:
    """keys of a normalization_data_map (reagent/core/parameters.py:155-161)"""

    STATE = "state"
    ACTION = "action"
    ITEM = "item"
    CANDIDATE = "candidate"


@dataclass(frozen=True)
class NormalizationData:
    """reagent/core/parameters.py:164-167: what the net builders size their networks from"""

    dense_normalization_parameters: Dict[int, NormalizationParameters]
