from .dqn_trainer import BCQConfig, DQNTrainer  # noqa: F401
from .qrdqn_trainer import QRDQNTrainer  # noqa: F401
from .reagent_lightning_module import ReAgentLightningModule  # noqa: F401
from .sac_trainer import CRRWeightFn, SACTrainer  # noqa: F401
from .td3_trainer import TD3Trainer  # noqa: F401
from .c51_trainer import C51Trainer  # noqa: F401
from .discrete_crr_trainer import DiscreteCRRTrainer  # noqa: F401
from .parameters import (  # noqa: F401
    C51TrainerParameters,
    CRRTrainerParameters,
    DQNTrainerParameters,
    QRDQNTrainerParameters,
    SACTrainerParameters,
    TD3TrainerParameters,
)
