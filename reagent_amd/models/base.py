"""ModelBase with the reference's surface (reagent/models/base.py:14-62)."""
from copy import deepcopy
from typing import Any

import torch.nn as nn


class ModelBase(nn.Module):
    def input_prototype(self) -> Any:
        raise NotImplementedError

    def feature_config(self):
        return None

    def get_target_network(self) -> "ModelBase":
        return deepcopy(self)

    def get_distributed_data_parallel_model(self):
        raise NotImplementedError

    def cpu_model(self) -> "ModelBase":
        return deepcopy(self).cpu()

    def requires_model_parallel(self) -> bool:
        return False
