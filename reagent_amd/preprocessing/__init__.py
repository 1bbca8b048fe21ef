from .preprocessor import Preprocessor  # noqa: F401
from .trainer_preprocessor import DiscreteDqnInputMaker, PolicyNetworkInputMaker  # noqa: F401
