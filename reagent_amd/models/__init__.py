from .actor import FullyConnectedActor, GaussianFullyConnectedActor  # noqa: F401
from .base import ModelBase  # noqa: F401
from .critic import FullyConnectedCritic  # noqa: F401
from .dqn import FullyConnectedDQN  # noqa: F401
from .dueling_q_network import DuelingQNetwork  # noqa: F401
from .fully_connected_network import (  # noqa: F401
    FloatFeatureFullyConnected,
    FullyConnectedNetwork,
    get_default_precision,
    set_default_precision,
)
from .categorical_dqn import CategoricalDQN  # noqa: F401
