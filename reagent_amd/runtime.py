"""The offline batch-RL step loop on one GPU: what OfflineReplayBufferDataset + pl.Trainer.fit do in
the reference (reagent/gym/datasets/replay_buffer_dataset.py:153-206), without a
dataloader, Lightning or host synchronisation:

    indices = sample_index_batch(B)                       (device RNG)
    batch   = ReplayBuffer.sample_transition_batch         rg_replay_nstep + rg_replay_gather
    input   = DiscreteDqnInputMaker(batch)                 rg_make_dqn_input
    state, next_state = Preprocessor(...)                  rg_normalize_dense x2
    trainer.train_step_native(input)                       FC fwd x3, head, bwd, Adam, soft update
Everything is enqueued on torch's current stream; the loss stays on the device.
"""
from typing import Optional

import torch

from .core import types as rlt
from .preprocessing import DiscreteDqnInputMaker, Preprocessor
from .replay_memory import ReplayBuffer


class OfflineDqnLoop:
    def __init__(self, replay_buffer: ReplayBuffer, trainer, batch_size: int,
                 state_preprocessor: Optional[Preprocessor] = None, state_dtype=None):
        self.rb = replay_buffer
        self.trainer = trainer
        self.batch_size = batch_size
        self.pre = state_preprocessor
        self.maker = DiscreteDqnInputMaker(trainer.num_actions)
        self._presence = None
        self.state_dtype = state_dtype
        # 1:1 normalization tables ride along with the gather (no separate normalize pass)
        self.fuse_norm = state_preprocessor is not None and state_preprocessor.elementwise

    def make_batch(self, indices: Optional[torch.Tensor] = None) -> rlt.DiscreteDqnInput:
        if self.fuse_norm:
            tup = self.rb.sample_transition_batch(self.batch_size, indices=indices, state_preprocessor=self.pre,
                                                  state_dtype=self.state_dtype)
            return self.maker(tup)
        tup = self.rb.sample_transition_batch(self.batch_size, indices=indices)
        inp = self.maker(tup)
        if self.pre is not None:
            s, ns = inp.state.float_features, inp.next_state.float_features
            if self._presence is None or self._presence.shape != s.shape or self._presence.device != s.device:
                self._presence = torch.ones(s.shape, dtype=torch.uint8, device=s.device)
            inp.state = rlt.FeatureData(self.pre(s, self._presence))
            inp.next_state = rlt.FeatureData(self.pre(ns, self._presence))
        return inp

    def step(self, indices: Optional[torch.Tensor] = None) -> torch.Tensor:
        # data parallel: the previous step's gradient all-reduce is still in flight here, and the
        # gather below does not depend on it — the trainer joins it right before Adam
        batch = self.make_batch(indices)
        return self.trainer.train_step_native(batch, defer_update=True)

    def flush(self):
        """apply an update left pending by the last step (call before reading parameters)"""
        self.trainer.apply_pending_update()
