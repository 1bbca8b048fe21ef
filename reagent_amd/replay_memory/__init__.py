from .circular_replay_buffer import ReplayBuffer  # noqa: F401
