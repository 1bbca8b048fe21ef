"""Iterable datasets that feed a trainer from a device-resident replay buffer
(reagent/gym/datasets/replay_buffer_dataset.py:22-206).

ReplayBufferDataset interleaves environment steps with training batches (one `add` per step, one sampled batch
every `training_frequency` steps once the buffer holds a batch); OfflineReplayBufferDataset only samples.  Per
batch the reference runs sample_transition_batch on the host, the input maker, and a host-to-device copy; here the
buffer's columns live in HBM, and for a discrete-action trainer sampling + n-step bookkeeping + the maker's one-hot /
not_terminal / exp(log_prob) work is ONE launch (ReplayBuffer.sample_dqn_input, bit-identical to the separate steps),
so what is yielded is already on the training device.  Keep `pin_memory=False` in a DataLoader around these datasets
(device tensors cannot be pinned); `collate_fn` is the identity as in the reference's tests (test_gym.py:249-252).
"""
import logging
from typing import Callable, Optional

import torch

from ...preprocessing.trainer_preprocessor import DiscreteDqnInputMaker
from ...replay_memory.circular_replay_buffer import ReplayBuffer
from ..preprocessors import make_replay_buffer_inserter, make_replay_buffer_trainer_preprocessor
from ..types import Trajectory, Transition

logger = logging.getLogger(__name__)


def _sample_for_trainer(replay_buffer: ReplayBuffer, batch_size: int, trainer_preprocessor):
    """one training batch: the fused sampler when the preprocessor is this package's plain discrete maker and the
    buffer has the shape the kernel serves, the reference's two steps (sample, then preprocess) otherwise"""
    maker = getattr(trainer_preprocessor, "maker", None)
    indices = None
    if type(maker) is DiscreteDqnInputMaker and maker.trainer_preprocessor is None:
        indices = replay_buffer.sample_index_batch(batch_size)  # drawn once: a declined fused launch reuses them
        fused = replay_buffer.sample_dqn_input(maker.num_actions, batch_size=batch_size, indices=indices)
        if fused is not None:
            return fused
    train_batch = replay_buffer.sample_transition_batch(batch_size=batch_size, indices=indices)
    if trainer_preprocessor:
        train_batch = trainer_preprocessor(train_batch)
    return train_batch


class ReplayBufferDataset(torch.utils.data.IterableDataset):
    def __init__(self, env, agent, replay_buffer: ReplayBuffer, batch_size: int, training_frequency: int = 1,
                 num_episodes: Optional[int] = None, max_steps: Optional[int] = None,
                 post_episode_callback: Optional[Callable] = None, trainer_preprocessor=None, replay_buffer_inserter=None):
        super().__init__()
        assert replay_buffer_inserter is not None
        self._env, self._agent, self._replay_buffer = env, agent, replay_buffer
        self._batch_size, self._training_frequency = batch_size, training_frequency
        self._num_episodes, self._max_steps = num_episodes, max_steps
        self._post_episode_callback = post_episode_callback
        self._trainer_preprocessor, self._replay_buffer_inserter = trainer_preprocessor, replay_buffer_inserter

    @classmethod
    def create_for_trainer(cls, trainer, env, agent, replay_buffer: ReplayBuffer, batch_size: int, training_frequency: int = 1,
                           num_episodes: Optional[int] = None, max_steps: Optional[int] = None,
                           post_episode_callback: Optional[Callable] = None, trainer_preprocessor=None,
                           replay_buffer_inserter=None, device=None):
        """replay_buffer_dataset.py:50-86; `device` defaults to where the buffer's columns live (the reference: cpu)"""
        device = device or getattr(replay_buffer, "device", None) or torch.device("cpu")
        return cls(env=env, agent=agent, replay_buffer=replay_buffer, batch_size=batch_size,
                   training_frequency=training_frequency, num_episodes=num_episodes, max_steps=max_steps,
                   post_episode_callback=post_episode_callback,
                   trainer_preprocessor=trainer_preprocessor or make_replay_buffer_trainer_preprocessor(trainer, device, env),
                   replay_buffer_inserter=replay_buffer_inserter or make_replay_buffer_inserter(env))

    def _episode(self, mdp_id: int, steps_so_far: int):
        """one episode (replay_buffer_dataset.py:96-137): yields training batches, returns (steps, reward sum)"""
        env, agent, rb = self._env, self._agent, self._replay_buffer
        obs = env.reset()
        mask = env.possible_actions_mask
        trajectory, info = Trajectory(), None
        num_steps, reward_sum, terminal = 0, 0, False
        while not terminal:
            action, log_prob = agent.act(obs, mask)
            next_obs, reward, terminal, info = env.step(action)
            next_mask = env.possible_actions_mask
            if self._max_steps is not None and num_steps >= self._max_steps:
                terminal = True
            # partially filled: an agent's post_step may add to it
            transition = Transition(mdp_id=mdp_id, sequence_number=num_steps, observation=obs, action=action,
                                    reward=float(reward), terminal=bool(terminal), log_prob=log_prob,
                                    possible_actions_mask=mask)
            trajectory.add_transition(transition)
            self._replay_buffer_inserter(rb, transition)
            reward_sum += reward
            if (steps_so_far + num_steps) % self._training_frequency == 0 and rb.size >= self._batch_size:
                yield _sample_for_trainer(rb, self._batch_size, self._trainer_preprocessor)
            obs, mask = next_obs, next_mask
            num_steps += 1
            if agent.post_step:
                agent.post_step(transition)
        if self._post_episode_callback:
            self._post_episode_callback(trajectory, info)
        return num_steps, reward_sum

    def __iter__(self):
        mdp_id, global_num_steps, rewards = 0, 0, []
        while self._num_episodes is None or mdp_id < self._num_episodes:
            steps, reward_sum = yield from self._episode(mdp_id, global_num_steps)
            global_num_steps += steps
            rewards.append(reward_sum)
            mdp_id += 1
            logger.info(f"Training episode: {mdp_id}, total episode reward = {reward_sum}")
        logger.info(f"Episode rewards during training: {rewards}")


class OfflineReplayBufferDataset(torch.utils.data.IterableDataset):
    """num_batches batches sampled from a filled buffer (replay_buffer_dataset.py:153-206)"""

    def __init__(self, env, replay_buffer: ReplayBuffer, batch_size: int, num_batches: int, trainer_preprocessor=None):
        super().__init__()
        self._env = env
        self._replay_buffer = replay_buffer
        self._batch_size = batch_size
        self._num_batches = num_batches
        self._trainer_preprocessor = trainer_preprocessor

    @classmethod
    def create_for_trainer(cls, trainer, env, replay_buffer: ReplayBuffer, batch_size: int, num_batches: int,
                           trainer_preprocessor=None, device=None):
        device = device or getattr(replay_buffer, "device", None) or torch.device("cpu")
        if trainer_preprocessor is None:
            trainer_preprocessor = make_replay_buffer_trainer_preprocessor(trainer, device, env)
        return cls(env=env, replay_buffer=replay_buffer, batch_size=batch_size, num_batches=num_batches,
                   trainer_preprocessor=trainer_preprocessor)

    def __iter__(self):
        for _ in range(self._num_batches):
            yield _sample_for_trainer(self._replay_buffer, self._batch_size, self._trainer_preprocessor)
