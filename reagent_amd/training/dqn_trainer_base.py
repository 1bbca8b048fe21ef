"""DQNTrainerMixin / DQNTrainerBaseLightning surface of
reagent/training/dqn_trainer_base.py:23-241 for the natively executed trainers.

CPE (reward network / q_network_cpe heads, :243-452) is executed by the
DQN trainer (`_CpeEngine` in dqn_trainer.py); the trainers that do not run it yet reject
calc_cpe_in_training=True at construction rather than silently skipping it.  The Evaluator /
EvaluationDataPage side (:454-509) is control plane and stays out.
"""
from typing import Dict, List, Optional

import torch

from .. import _lib as L
from ..core.parameters import EvaluationParameters, RLParameters
from .reagent_lightning_module import ReAgentLightningModule
from .rl_trainer_pytorch import RLTrainerMixin


class DQNTrainerMixin:
    ACTION_NOT_POSSIBLE_VAL = -1e9

    def get_max_q_values(self, q_values, possible_actions_mask):
        return self.get_max_q_values_with_target(q_values, q_values, possible_actions_mask)

    def get_max_q_values_with_target(self, q_values, q_values_target, possible_actions_mask):
        """dqn_trainer_base.py:33-77 — utility entry point (evaluation pages, tests).  The training
        step itself evaluates the same rule inside rg_dqn_head."""
        q_values = q_values.reshape(possible_actions_mask.shape)
        q_values_target = q_values_target.reshape(possible_actions_mask.shape)
        inverse_pna = 1 - possible_actions_mask
        impossible_action_penalty = self.ACTION_NOT_POSSIBLE_VAL * inverse_pna
        q_values = q_values + impossible_action_penalty
        q_values_target = q_values_target + impossible_action_penalty
        if self.double_q_learning:
            _, max_indicies = torch.max(q_values, dim=1, keepdim=True)
            max_q_values_target = torch.gather(q_values_target, 1, max_indicies)
        else:
            max_q_values_target, max_indicies = torch.max(q_values_target, dim=1, keepdim=True)
        return max_q_values_target, max_indicies


class DQNTrainerBaseLightning(DQNTrainerMixin, RLTrainerMixin, ReAgentLightningModule):
    def __init__(
        self,
        rl_parameters: RLParameters,
        metrics_to_score=None,
        actions: Optional[List[str]] = None,
        evaluation_parameters: Optional[EvaluationParameters] = None,
    ):
        super().__init__()
        self.rl_parameters = rl_parameters
        self.time_diff_unit_length = rl_parameters.time_diff_unit_length
        self.tensorboard_logging_freq = rl_parameters.tensorboard_logging_freq
        self.calc_cpe_in_training = bool(evaluation_parameters and evaluation_parameters.calc_cpe_in_training)
        assert actions is not None
        self._actions: List[str] = actions

        if rl_parameters.q_network_loss not in L.LOSS:
            raise Exception("Q-Network loss type {} not valid loss.".format(rl_parameters.q_network_loss))
        self._loss_type = L.LOSS[rl_parameters.q_network_loss]

        if metrics_to_score:
            self.metrics_to_score = metrics_to_score + ["reward"]
        else:
            self.metrics_to_score = ["reward"]
        self._init_reward_boosts(rl_parameters.reward_boost)

    def _init_reward_boosts(self, rl_reward_boost: Optional[Dict[str, float]]) -> None:
        reward_boosts = torch.zeros([1, len(self._actions)])
        if rl_reward_boost is not None:
            for k in rl_reward_boost.keys():
                i = self._actions.index(k)
                reward_boosts[0, i] = rl_reward_boost[k]
        self.register_buffer("reward_boosts", reward_boosts)
        self._has_reward_boost = bool(rl_reward_boost)

    def _check_input(self, training_batch):
        """dqn_trainer_base.py:188-208 (same asserts and the same ValueError)."""
        assert hasattr(training_batch, "possible_next_actions_mask") and hasattr(training_batch, "not_terminal")
        assert training_batch.not_terminal.dim() == training_batch.reward.dim() == 2
        assert training_batch.not_terminal.shape[1] == training_batch.reward.shape[1] == 1
        assert training_batch.action.dim() == training_batch.next_action.dim() == 2
        assert training_batch.action.shape[1] == training_batch.next_action.shape[1] == self.num_actions
        if torch.logical_and(
            training_batch.possible_next_actions_mask.float().sum(dim=1) == 0,
            training_batch.not_terminal.squeeze().bool(),
        ).any():
            raise ValueError("No possible next actions. Should the environment have terminated?")

    @property
    def num_actions(self) -> int:
        assert self._actions is not None, "Not a discrete action DQN"
        return len(self._actions)

    @torch.no_grad()
    def boost_rewards(self, rewards: torch.Tensor, actions: torch.Tensor) -> torch.Tensor:
        """dqn_trainer_base.py:216-241 — utility entry point; the step fuses it into rg_dqn_head."""
        reward_boosts = torch.sum(actions.float() * self.reward_boosts, dim=1, keepdim=True)
        return rewards + reward_boosts

    def _initialize_cpe(self, reward_network, q_network_cpe, q_network_cpe_target, optimizer) -> None:
        """dqn_trainer_base.py:243-311 without the Evaluator object (evaluation pages are control plane)."""
        if not self.calc_cpe_in_training:
            self.reward_network = None
            return
        assert reward_network is not None, "reward_network is required for CPE"
        self.reward_network = reward_network
        self.reward_network_optimizer = optimizer
        assert q_network_cpe is not None and q_network_cpe_target is not None, (
            "q_network_cpe and q_network_cpe_target are required for CPE"
        )
        self.q_network_cpe = q_network_cpe
        self.q_network_cpe_target = q_network_cpe_target
        self.q_network_cpe_optimizer = optimizer
        num_output_nodes = len(self.metrics_to_score) * self.num_actions
        self.register_buffer("reward_idx_offsets",
                             torch.arange(0, num_output_nodes, self.num_actions, dtype=torch.long))

    def _configure_cpe_optimizers(self):
        """dqn_trainer_base.py:313-336"""
        target_params = list(self.q_network_cpe_target.parameters())
        source_params = list(self.q_network_cpe.parameters())
        optimizers = [
            self.reward_network_optimizer.make_optimizer_scheduler(self.reward_network.parameters()),
            self.q_network_cpe_optimizer.make_optimizer_scheduler(self.q_network_cpe.parameters()),
        ]
        return target_params, source_params, optimizers

    def _reject_cpe(self, reward_network, q_network_cpe, q_network_cpe_target):
        if self.calc_cpe_in_training:
            raise NotImplementedError(
                "calc_cpe_in_training=True (reward_network / q_network_cpe heads) is not built yet: "
                "SURVEY.md §8(f) rank 1.  Pass evaluation=EvaluationParameters(calc_cpe_in_training=False)."
            )
        # dqn_trainer_base.py:271-275: with CPE off only `reward_network = None` is set
        self.reward_network = None
