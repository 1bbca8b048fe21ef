from .replay_buffer_dataset import OfflineReplayBufferDataset, ReplayBufferDataset  # noqa: F401
