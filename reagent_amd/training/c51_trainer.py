"""C51Trainer with the constructor / generator surface of reagent/training/c51_trainer.py:18-212,
executed on the HIP kernels.

Step (:98-187): target(next_state) and — with double Q — online(next_state) forwards, online(state)
forward, then rg_c51_head: softmax over atoms, masked arg-max of the expected values, the categorical
projection onto the fixed support (the two scatter_adds, in order), the cross-entropy loss and its
gradient w.r.t. the logits; backward, Adam, soft update.
"""
from typing import List, Optional

import torch

from .. import ops
from ..core import types as rlt
from ..core.parameters import EvaluationParameters, RLParameters
from ..optimizer import Optimizer__Union
from .dqn_trainer import QStepCore


class C51Trainer(QStepCore):
    _post_step_forward = False  # c51_trainer.py:100-190 evaluates the network on next_state and state only

    def __init__(
        self,
        q_network,
        q_network_target,
        actions: Optional[List[str]] = None,
        rl: Optional[RLParameters] = None,
        double_q_learning: bool = True,
        minibatch_size: int = 1024,
        minibatches_per_step: int = 1,
        num_atoms: int = 51,
        qmin: float = -100,
        qmax: float = 200,
        optimizer: Optional[Optimizer__Union] = None,
    ) -> None:
        rl = rl if rl is not None else RLParameters()
        actions = actions if actions is not None else []
        super().__init__(rl_parameters=rl, metrics_to_score=None, actions=actions,
                         evaluation_parameters=EvaluationParameters(calc_cpe_in_training=False))
        self.double_q_learning = double_q_learning
        self.minibatch_size = minibatch_size
        self.minibatches_per_step = minibatches_per_step
        self.q_network = q_network
        self.q_network_target = q_network_target
        self.q_network_optimizer = optimizer if optimizer is not None else Optimizer__Union.default()
        self.qmin = qmin
        self.qmax = qmax
        self.num_atoms = num_atoms
        self.register_buffer("support", None)
        self.support = torch.linspace(self.qmin, self.qmax, self.num_atoms)
        self.scale_support = (self.qmax - self.qmin) / (self.num_atoms - 1.0)

    def _out_cols(self) -> int:
        return self.num_actions * self.num_atoms

    def _needs_online_next(self) -> bool:
        return bool(self.maxq_learning and self.double_q_learning)

    def _alloc_head(self, batch, device):
        f32 = dict(dtype=torch.float32, device=device)
        self._loss_partials = torch.empty(batch, **f32)
        self._all_q = torch.empty(batch, self.num_actions, **f32)

    def _run_head(self, b, B, action, next_mask, boosts, gamma_exp):
        if self.support.device != self._q.device:
            self.support = self.support.to(self._q.device)
        ops.c51_head(self._q, self._qn_online if self._needs_online_next() else None, self._qn_target, action,
                     next_mask, self._f32c(b.reward).reshape(-1), boosts, self._f32c(b.not_terminal).reshape(-1),
                     self.gamma, gamma_exp, self.support, self.qmin, self.qmax, self.num_atoms, self.maxq_learning,
                     self._dq, self._loss_partials, self._all_q)
        ops.reduce_sum(self._loss_partials, B, 1.0, self._loss)
        self.all_q_values = self._all_q

    def train_step_gen(self, training_batch: rlt.DiscreteDqnInput, batch_idx: int):
        loss = self._hip_loss(training_batch)
        # c51_trainer.py:178: reported every `log_every_n_steps` batches of the driving trainer (50 without one)
        if batch_idx % getattr(getattr(self, "trainer", None), "log_every_n_steps", 50) == 0:
            self._log(loss, training_batch)
        yield loss
        yield self.soft_update_result()

    def _log(self, loss, training_batch):
        from .reagent_lightning_module import _NoOpReporter

        if isinstance(self._reporter, _NoOpReporter):
            return
        mask = training_batch.possible_actions_mask.float() if self.maxq_learning else training_batch.action
        self.reporter.log(td_loss=loss.detach(), logged_actions=torch.argmax(training_batch.action, dim=1, keepdim=True),
                          logged_propensities=training_batch.extras.action_probability,
                          logged_rewards=self.boost_rewards(training_batch.reward, training_batch.action),
                          model_values=self.all_q_values,
                          model_action_idxs=self.argmax_with_mask(self.all_q_values, mask))

    def argmax_with_mask(self, q_values, possible_actions_mask):
        """c51_trainer.py:203-211"""
        q_values = q_values.reshape(possible_actions_mask.shape)
        q_values = q_values + self.ACTION_NOT_POSSIBLE_VAL * (1 - possible_actions_mask)
        return q_values.argmax(1)
