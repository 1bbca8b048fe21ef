from .preprocessor import Preprocessor  # noqa: F401
from .trainer_preprocessor import DiscreteDqnInputMaker, PolicyNetworkInputMaker  # noqa: F401
from .batch_preprocessor import BatchPreprocessor, DiscreteDqnBatchPreprocessor, batch_to_device  # noqa: F401
