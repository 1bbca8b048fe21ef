from .replay_buffer_inserters import BasicReplayBufferInserter, make_replay_buffer_inserter  # noqa: F401
from .trainer_preprocessor import (  # noqa: F401
    REPLAY_BUFFER_MAKER_MAP,
    make_replay_buffer_trainer_preprocessor,
    make_trainer_preprocessor,
)
