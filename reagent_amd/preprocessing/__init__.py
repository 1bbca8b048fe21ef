from .preprocessor import Preprocessor  # noqa: F401
from .trainer_preprocessor import DiscreteDqnInputMaker, PolicyNetworkInputMaker  # noqa: F401
from .batch_preprocessor import (BatchPreprocessor, DiscreteDqnBatchPreprocessor,  # noqa: F401
                                 PolicyNetworkBatchPreprocessor, batch_to_device)
