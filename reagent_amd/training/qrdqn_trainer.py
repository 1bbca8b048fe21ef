"""QRDQNTrainer with the constructor / generator surface of
reagent/training/qrdqn_trainer.py:22-227, executed on the HIP kernels.

Step (reference :108-194): target(next_state) and — when double_q_learning — online(next_state)
forwards, online(state) forward, the quantile-Huber head (rg_qr_head: per-transition N x N pairs in
LDS/registers, the (N, B, N) tensor of the reference is never built), backward, Adam, soft update.
The 4th forward (:162-164) feeds only the CPE heads and runs only with calc_cpe_in_training.
"""
from typing import List, Optional, Tuple

import torch

from .. import ops
from ..core import types as rlt
from ..core.parameters import EvaluationParameters, RLParameters
from ..optimizer import Optimizer__Union
from .dqn_trainer import QStepCore


class QRDQNTrainer(QStepCore):
    def __init__(
        self,
        q_network,
        q_network_target,
        metrics_to_score=None,
        reward_network=None,
        q_network_cpe=None,
        q_network_cpe_target=None,
        actions: Optional[List[str]] = None,
        rl: Optional[RLParameters] = None,
        double_q_learning: bool = True,
        num_atoms: int = 51,
        minibatch_size: int = 1024,
        minibatches_per_step: int = 1,
        optimizer: Optional[Optimizer__Union] = None,
        cpe_optimizer: Optional[Optimizer__Union] = None,
        evaluation: Optional[EvaluationParameters] = None,
    ) -> None:
        evaluation = evaluation if evaluation is not None else EvaluationParameters()
        rl = rl if rl is not None else RLParameters()
        actions = actions if actions is not None else []
        optimizer = optimizer if optimizer is not None else Optimizer__Union.default()
        super().__init__(rl_parameters=rl, metrics_to_score=metrics_to_score, actions=actions,
                         evaluation_parameters=evaluation)
        self.double_q_learning = double_q_learning
        self.minibatch_size = minibatch_size
        self.minibatches_per_step = minibatches_per_step
        self._actions = actions
        self.q_network = q_network
        self.q_network_target = q_network_target
        self.q_network_optimizer = optimizer
        self.num_atoms = num_atoms
        self.register_buffer("quantiles", None)
        self.quantiles = ((0.5 + torch.arange(self.num_atoms).float()) / float(self.num_atoms)).view(1, -1)
        self._initialize_cpe(reward_network, q_network_cpe, q_network_cpe_target,
                             optimizer=cpe_optimizer if cpe_optimizer is not None else Optimizer__Union.default())
        if self.calc_cpe_in_training:
            from .dqn_trainer import _CpeEngine

            self._cpe = _CpeEngine(self)

    def _out_cols(self) -> int:
        return self.num_actions * self.num_atoms

    # ---- grouped wide layer (qr_engine.py): bf16 fused stacks, no CPE ---------------------------------------
    use_grouped_head = True  # set False to force the dense [B, A * N] path

    def _grouped(self):
        from ..qr_engine import GroupedQR

        if not self.use_grouped_head or not GroupedQR.eligible(self):
            return None
        g = getattr(self, "_gq", None)
        if g is None or g.online.lin[-1].weight is not self.q_network.fc.linears()[-1].weight:
            g = self._gq = GroupedQR(self)
        return g

    def _engine(self, batch: int, device):
        g = self._grouped()
        if g is None:
            self._gq_active = None
            return super()._engine(batch, device)
        from ..engine import ensure_slab, grad_views

        self._hip_params = list(self.q_network.parameters())
        self._slab = ensure_slab(self._hip_params)
        self._qs, self._ts = g, g.target
        if getattr(self, "_loss", None) is None or self._loss.device != device:
            self._loss = torch.empty(1, dtype=torch.float32, device=device)
        self._q = self._loss  # (device marker of the dense path's buffers)
        self._ws_batch = -1
        self._dw, self._db = grad_views(self.q_network.fc, self._slab, self._hip_params)
        self._xs_t = None
        self._gq_active = g

    def _hip_forward(self, b):
        state = b.state.float_features
        self._engine(state.shape[0], state.device)
        g = self._gq_active
        if g is None:
            return super()._hip_forward(b)
        loss = g.forward(b)
        from .reagent_lightning_module import _NoOpReporter

        # q_network(state).mean(dim=2) is logging output only (a local of the reference's step, qrdqn_trainer.py:147,189):
        # with a reporter attached it is evaluated here, with the step's weights like the reference; otherwise on first
        # read, with the network as it is then — after the step's update (one extra forward with the mean layer)
        self._all_q_values = None if isinstance(self._reporter, _NoOpReporter) else g.all_q_values()
        return loss

    @property
    def all_q_values(self):
        if getattr(self, "_all_q_values", None) is None and getattr(self, "_gq_active", None) is not None:
            self._all_q_values = self._gq_active.all_q_values()
        return getattr(self, "_all_q_values", None)

    @all_q_values.setter
    def all_q_values(self, v):
        self._all_q_values = v

    def _needs_online_next(self) -> bool:
        return bool(self.maxq_learning and self.double_q_learning)

    def _alloc_head(self, batch, device):
        f32 = dict(dtype=torch.float32, device=device)
        self._loss_partials = torch.empty(batch, **f32)
        self._all_q = torch.empty(batch, self.num_actions, **f32)

    def _run_head(self, b, B, action, next_mask, boosts, gamma_exp):
        if self.quantiles.device != self._q.device:
            self.quantiles = self.quantiles.to(self._q.device)
        ops.qr_head(self._q, self._qn_online if self._needs_online_next() else None, self._qn_target, action,
                    next_mask, self._f32c(b.reward).reshape(-1), boosts, self._f32c(b.not_terminal).reshape(-1),
                    self.gamma, gamma_exp, self.quantiles.reshape(-1), self.num_atoms, self.maxq_learning,
                    self._dq, self._loss_partials, self._all_q)
        ops.reduce_sum(self._loss_partials, B, 1.0, self._loss)
        self.all_q_values = self._all_q

    def _cpe_next_scores(self, xn, out):
        """qrdqn_trainer.py:162-164: q_network(next_state).mean(dim=2) with the just-updated weights"""
        B = out.shape[0]
        self._qs.forward(xn, self._qn_online, save=False)  # the buffer is free once the head has run
        torch.mean(self._qn_online.view(B, self.num_actions, self.num_atoms), dim=2, out=out)

    def _cpe_scores_for_logging(self):
        return self.all_q_values

    def train_step_gen(self, training_batch: rlt.DiscreteDqnInput, batch_idx: int):
        self._check_input(training_batch)
        loss = self._hip_loss(training_batch)
        yield loss
        self.loss = loss.detach()
        yield from self._cpe_segment(training_batch)
        self._log(training_batch)
        yield self.soft_update_result()

    def _log(self, training_batch):
        from .reagent_lightning_module import _NoOpReporter

        if isinstance(self._reporter, _NoOpReporter):
            return
        rewards = self.boost_rewards(training_batch.reward, training_batch.action)
        logged_action_idxs = torch.argmax(training_batch.action, dim=1, keepdim=True)
        mask = training_batch.possible_actions_mask.float() if self.maxq_learning else training_batch.action
        model_action_idxs = self.argmax_with_mask(self.all_q_values, mask)
        self.reporter.log(td_loss=self.loss, logged_actions=logged_action_idxs,
                          logged_propensities=training_batch.extras.action_probability, logged_rewards=rewards,
                          logged_values=None, model_values=self.all_q_values,
                          model_values_on_logged_actions=None, model_action_idxs=model_action_idxs)

    def argmax_with_mask(self, q_values, possible_actions_mask):
        """qrdqn_trainer.py:210-214"""
        q_values = q_values.reshape(possible_actions_mask.shape)
        q_values = q_values + self.ACTION_NOT_POSSIBLE_VAL * (1 - possible_actions_mask)
        return q_values.argmax(1)

    def huber(self, x):
        """qrdqn_trainer.py:217-218"""
        return torch.where(x.abs() < 1, 0.5 * x.pow(2), x.abs() - 0.5)

    @torch.no_grad()
    def get_detached_model_outputs(self, state) -> Tuple[torch.Tensor, torch.Tensor]:
        """qrdqn_trainer.py:220-227"""
        return self.q_network(state).mean(dim=2), self.q_network_target(state).mean(dim=2)
