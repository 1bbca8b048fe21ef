"""DiscreteCRRTrainer with the constructor / generator surface of
reagent/training/discrete_crr_trainer.py:25-440, executed on the HIP kernels.

Step (:287-385): target critics and the (target) actor on next_state, the online critics on state,
then rg_crr_critic_head — V(s') = sum_a Q_target(s', a) * softmax(actor(s'))_a, min over the twin
critics, y = r + gamma * not_terminal * V, MSE and its gradient for q1 / q2; after the q1 step the
actor segment: q1(state) with the NEW weights and actor(state), then rg_crr_actor_head — the
exp-advantage weight (clamped, detached), -log pi(logged action) * weight, the optional entropy term and
d loss / d scores; backward through the actor's output activation and stack; the CPE heads (shared with
the DQN step, the next-state scores being the target critic's); soft update of every target.
"""
from typing import List, Optional, Tuple

import torch

from .. import _lib as L
from .. import ops
from ..core import types as rlt
from ..core.parameters import EvaluationParameters, RLParameters
from ..engine import ensure_slab
from ..optimizer import Optimizer__Union, SoftUpdate
from .dqn_trainer import _CpeEngine, _SegmentLoss, dp_reduce, held_gradients, native_step, publish_gradients
from .dqn_trainer_base import DQNTrainerBaseLightning


class DiscreteCRRTrainer(DQNTrainerBaseLightning):
    """Critic Regularized Regression (https://arxiv.org/abs/2006.15134), discrete actions."""

    def __init__(
        self,
        actor_network,
        actor_network_target,
        q1_network,
        q1_network_target,
        reward_network,
        q2_network=None,
        q2_network_target=None,
        q_network_cpe=None,
        q_network_cpe_target=None,
        metrics_to_score=None,
        evaluation: Optional[EvaluationParameters] = None,
        rl: Optional[RLParameters] = None,
        double_q_learning: bool = True,
        q_network_optimizer: Optional[Optimizer__Union] = None,
        actor_network_optimizer: Optional[Optimizer__Union] = None,
        use_target_actor: bool = False,
        actions: Optional[List[str]] = None,
        delayed_policy_update: int = 1,
        beta: float = 1.0,
        entropy_coeff: float = 0.0,
        clip_limit: float = 10.0,
        max_weight: float = 20.0,
    ) -> None:
        rl = rl if rl is not None else RLParameters()
        evaluation = evaluation if evaluation is not None else EvaluationParameters()
        actions = actions if actions is not None else []
        super().__init__(rl, metrics_to_score=metrics_to_score, actions=actions, evaluation_parameters=evaluation)
        assert self._actions is not None, "Discrete-action CRR needs action names"
        d = Optimizer__Union.default
        self.double_q_learning = double_q_learning
        self.use_target_actor = use_target_actor
        self.q1_network = q1_network
        self.q1_network_target = q1_network_target
        self.q_network_optimizer = q_network_optimizer if q_network_optimizer is not None else d()
        self.q2_network = q2_network
        if self.q2_network is not None:
            assert q2_network_target is not None, "q2_network provided without a target network"
            self.q2_network_target = q2_network_target
        self.actor_network = actor_network
        self.actor_network_target = actor_network_target
        self.actor_network_optimizer = actor_network_optimizer if actor_network_optimizer is not None else d()
        self.delayed_policy_update = delayed_policy_update
        self._initialize_cpe(reward_network, q_network_cpe, q_network_cpe_target, optimizer=self.q_network_optimizer)
        self.beta = beta
        self.entropy_coeff = entropy_coeff
        self.clip_limit = clip_limit
        self.max_weight = max_weight
        self._ws_batch = -1
        self._dp_group, self._dp_world = None, 1
        self._native_idx = 0
        self._cpe = _CpeEngine(self) if self.calc_cpe_in_training else None

    @property
    def q_network(self):
        return self.q1_network

    @torch.no_grad()
    def get_detached_model_outputs(self, state) -> Tuple[torch.Tensor, None]:
        """:141-149 — the actor's scores, and None in the place of the target network's"""
        return self.actor_network(state).action, None

    # ---- optimizers (:151-189) -------------------------------------------------------------------
    def configure_optimizers(self):
        optimizers = []
        target_params = list(self.q1_network_target.parameters())
        source_params = list(self.q1_network.parameters())
        optimizers.append(self.q_network_optimizer.make_optimizer_scheduler(self.q1_network.parameters()))
        if self.q2_network:
            target_params += list(self.q2_network_target.parameters())
            source_params += list(self.q2_network.parameters())
            optimizers.append(self.q_network_optimizer.make_optimizer_scheduler(self.q2_network.parameters()))
        target_params += list(self.actor_network_target.parameters())
        source_params += list(self.actor_network.parameters())
        optimizers.append(self.actor_network_optimizer.make_optimizer_scheduler(self.actor_network.parameters()))
        if self.calc_cpe_in_training:
            cpe_target_params, cpe_source_params, cpe_optimizers = self._configure_cpe_optimizers()
            target_params += cpe_target_params
            source_params += cpe_source_params
            optimizers += cpe_optimizers
        optimizers.append(SoftUpdate.make_optimizer_scheduler(target_params, source_params, tau=self.tau))
        return optimizers

    # ---- engine ----------------------------------------------------------------------------------
    @staticmethod
    def _f32c(t: torch.Tensor) -> torch.Tensor:
        t = t if t.dtype == torch.float32 else t.float()
        return t if t.is_contiguous() else t.contiguous()

    @staticmethod
    def _net_in(t: torch.Tensor) -> torch.Tensor:
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float()
        return t if t.stride(-1) == 1 and t.is_contiguous() else t.contiguous()

    def _engine(self, B, dev):
        A = self.num_actions
        nets = dict(actor=self.actor_network, q1=self.q1_network, q2=self.q2_network)
        self._e = {k: _CpeEngine._net_engine(n) for k, n in nets.items() if n is not None}
        tg = dict(actor=self.actor_network_target, q1=self.q1_network_target,
                  q2=getattr(self, "q2_network_target", None))
        self._t = {k: n.fc.stack() for k, n in tg.items() if n is not None}
        if self._ws_batch != B or self._q1v.device != dev:
            f = dict(dtype=torch.float32, device=dev)
            P = ops.crr_partials(B)
            for n in ("q1v", "q2v", "q1n", "q2n", "next_scores", "dq1", "dq2", "q1_new", "raw_scores", "dscores"):
                setattr(self, "_" + n, torch.empty(B, A, **f))
            self._target = torch.empty(B, **f)
            self._parts = {n: torch.empty(P, **f) for n in ("q1", "q2", "plain", "entropy")}
            self._losses = {n: torch.empty(1, **f) for n in ("q1", "q2", "plain", "entropy", "actor")}
            self._ws_batch = B

    def _publish(self, e, held=()):
        slab = e["slab"]
        if self._dp_group is not None:
            dp_reduce(self, slab)
        publish_gradients(slab, e["params"], held)

    # ---- segments --------------------------------------------------------------------------------
    def _critic_forward(self, b):
        state, next_state = self._net_in(b.state.float_features), self._net_in(b.next_state.float_features)
        L.require_cuda(state, "training_batch.state")
        B, dev = state.shape[0], state.device
        self._engine(B, dev)
        e, t = self._e, self._t
        has_q2 = "q2" in e
        for k in ("q1", "q2"):
            if k in e:
                e[k]["stack"].stage_weights(need_transposed=True)
                t[k].stage_weights(need_transposed=False)
        # next_q_values = q1_network_target(next_state) (:307); compute_target_q_values (:191-206)
        xn, _ = t["q1"].stage_input(next_state, need_transposed=False)
        t["q1"].forward(xn, self._q1n, save=False)
        if has_q2:
            xn2, _ = t["q2"].stage_input(next_state, need_transposed=False)
            t["q2"].forward(xn2, self._q2n, save=False)
        next_actor = t["actor"] if self.use_target_actor else e["actor"]["stack"]
        next_actor.stage_weights(need_transposed=not self.use_target_actor)
        xa, _ = next_actor.stage_input(next_state, need_transposed=False)
        next_actor.forward(xa, self._next_scores, save=False)
        next_module = self.actor_network_target if self.use_target_actor else self.actor_network
        if next_module.exploration_variance is not None:  # `.action` of an exploring actor carries its noise
            self._next_scores.copy_(next_module.explore(self._next_scores)[0])
        # compute_td_loss for both critics (:208-212)
        q1s = e["q1"]["stack"]
        x1, self._x1_t = q1s.stage_input(state, need_transposed=True)
        q1s.forward(x1, self._q1v, save=True)
        if has_q2:
            q2s = e["q2"]["stack"]
            x2, self._x2_t = q2s.stage_input(state, need_transposed=True)
            q2s.forward(x2, self._q2v, save=True)
        boosts = self.reward_boosts.reshape(-1) if self._has_reward_boost else None
        if boosts is not None and boosts.device != dev:
            self.reward_boosts = self.reward_boosts.to(dev)
            boosts = self.reward_boosts.reshape(-1)
        self._action = self._f32c(b.action)
        ops.crr_critic_head(self._q1v, self._q2v if has_q2 else None, self._q1n, self._q2n if has_q2 else None,
                            self._next_scores, self._action, self._f32c(b.reward).reshape(-1), boosts,
                            self._f32c(b.not_terminal).reshape(-1), self.gamma, self._target, self._dq1,
                            self._dq2 if has_q2 else None, self._parts["q1"], self._parts["q2"] if has_q2 else None)
        P = self._parts["q1"].numel()
        ops.reduce_sum(self._parts["q1"], P, 1.0 / B, self._losses["q1"])
        if has_q2:
            ops.reduce_sum(self._parts["q2"], P, 1.0 / B, self._losses["q2"])

    def _critic_backward(self, which, grad_out=None):
        e = self._e[which]
        dq = self._dq1 if which == "q1" else self._dq2
        if grad_out is not None:
            dq = dq * grad_out
        held = held_gradients(e["slab"], e["params"])
        e["stack"].backward(dq, self._x1_t if which == "q1" else self._x2_t, e["dw"], e["db"])
        self._publish(e, held)

    def _actor_forward(self, b, with_loss: bool):
        """all_q_values = q1_network(state) with the weights its optimizer just produced (:327), the
        actor's scores, and — on the batches that update the policy — compute_actor_loss (:214-285)"""
        state = self._net_in(b.state.float_features)
        B = state.shape[0]
        e = self._e
        act = e["actor"]["stack"]
        act.stage_weights(need_transposed=True)
        xs, self._xs_t = act.stage_input(state, need_transposed=True)
        act.forward(xs, self._raw_scores, save=with_loss)
        self._scores, self._clamp_passes = self._raw_scores, None
        if self.actor_network.exploration_variance is not None:
            noisy, self._clamp_passes, _ = self.actor_network.explore(self._raw_scores)
            self._scores = noisy.contiguous()
        self.all_action_scores = self._scores
        if not with_loss:
            return
        q1s = e["q1"]["stack"]
        q1s.stage_weights(need_transposed=True)
        x1, _ = q1s.stage_input(state, need_transposed=False)
        q1s.forward(x1, self._q1_new, save=False)
        pi_b = None
        if self.entropy_coeff > 0:
            pi_b = self._f32c(b.extras.action_probability).reshape(-1)
            assert pi_b.numel() == B
            assert torch.min(pi_b) > 0, "Logged action probability <= 0"
        ops.crr_actor_head(self._q1_new, self._scores, self._action, pi_b, self.beta, self.max_weight,
                           self.entropy_coeff, self.clip_limit, self._dscores, self._parts["plain"],
                           self._parts["entropy"] if self.entropy_coeff > 0 else None)
        P = self._parts["plain"].numel()
        ops.reduce_sum(self._parts["plain"], P, 1.0 / B, self._losses["plain"])
        if self.entropy_coeff > 0:
            ops.reduce_sum(self._parts["entropy"], P, 1.0 / B, self._losses["entropy"])
            torch.add(self._losses["plain"], self._losses["entropy"], alpha=self.entropy_coeff,
                      out=self._losses["actor"])
        else:
            self._losses["actor"].copy_(self._losses["plain"])

    def _actor_backward(self, grad_out=None):
        a = self._e["actor"]
        d = self._dscores if grad_out is None else self._dscores * grad_out
        if self._clamp_passes is not None:  # backward of the exploration clamp
            d = d * self._clamp_passes
        held = held_gradients(a["slab"], a["params"])
        a["stack"].backward(d, self._xs_t, a["dw"], a["db"], out32=self._raw_scores)
        self._publish(a, held)

    # ---- CPE hooks (dqn_trainer_base.py:338-452 as called at :354-363) ---------------------------
    def _cpe_next_action_scores(self, next_state, out):
        out.copy_(self._q1n)  # next_q_values.detach()

    def _cpe_gamma_exponent(self, b):
        return None  # discount_tensor = full_like(rewards, gamma) (:305)

    # ---- reference surface -----------------------------------------------------------------------
    def train_step_gen(self, training_batch: rlt.DiscreteDqnInput, batch_idx: int):
        """IMPORTANT: the action of a DiscreteDqnInput is one-hot (DiscreteDqnInputMaker)."""
        self._check_input(training_batch)
        b = training_batch
        self._critic_forward(b)
        q1 = self._e["q1"]
        q1_loss = _SegmentLoss.apply(lambda g: self._critic_backward("q1", g), self._losses["q1"], *q1["params"])
        self.log("td_loss", q1_loss.detach(), prog_bar=True, batch_size=b.batch_size())
        yield q1_loss
        if self.q2_network:
            q2 = self._e["q2"]
            yield _SegmentLoss.apply(lambda g: self._critic_backward("q2", g), self._losses["q2"], *q2["params"])
        # only update the actor after a fixed number of Q updates (:218-222)
        update_actor = batch_idx % self.delayed_policy_update == 0
        self._actor_forward(b, with_loss=update_actor)
        if update_actor:
            self.actor_loss_without_reg = self._losses["plain"]
            yield _SegmentLoss.apply(self._actor_backward, self._losses["actor"], *self._e["actor"]["params"])
        else:
            self.actor_loss_without_reg = None
            yield None  # None keeps the actor network from updating
        if self._cpe is not None:
            self._cpe.forward(b)
            reward_loss = self._cpe.loss("reward")
            yield reward_loss
            from .reagent_lightning_module import _NoOpReporter

            if not isinstance(self._reporter, _NoOpReporter):  # dqn_trainer_base.py:430-450 (inside _calculate_cpes)
                from ..core.torch_utils import masked_softmax

                mask = b.possible_actions_mask if self.maxq_learning else b.action
                self.reporter.log(reward_loss=reward_loss.detach(),
                                  model_propensities=masked_softmax(self.all_action_scores, mask.float(), self.rl_temperature),
                                  model_rewards=self._cpe.reward_est[:, : self.num_actions])
            yield self._cpe.loss("cpe")
        self._log_crr(q1_loss, b)
        yield self.soft_update_result()

    def _log_crr(self, q1_loss, b):
        from .reagent_lightning_module import _NoOpReporter

        if isinstance(self._reporter, _NoOpReporter):
            return
        mask = b.possible_actions_mask if self.maxq_learning else b.action
        self.reporter.log(logged_actions=torch.argmax(b.action, dim=1, keepdim=True), td_loss=q1_loss.detach(),
                          logged_propensities=b.extras.action_probability,
                          logged_rewards=self.boost_rewards(b.reward, b.action), model_values=self.all_action_scores,
                          model_action_idxs=self.get_max_q_values(self.all_action_scores, mask.float())[1])

    # ---- fused native step -----------------------------------------------------------------------
    def native_optimizers(self):
        if getattr(self, "_native_opts", None) is None:
            self._native_opts = [o["optimizer"] for o in self.configure_optimizers()]
        return self._native_opts

    def enable_data_parallel(self, process_group=None):
        import torch.distributed as dist

        self._dp_group = process_group if process_group is not None else dist.group.WORLD
        self._dp_world = dist.get_world_size(self._dp_group)
        from .dqn_trainer import require_grad_scaling_optimizers

        require_grad_scaling_optimizers(self)  # the 1/world of the summed gradients is folded into the Adam launches
        return self

    @torch.no_grad()
    @native_step
    def train_step_native(self, training_batch, batch_idx: Optional[int] = None):
        """the same segments and optimizer order with no autograd graph / generator / host sync"""
        opts = iter(self.native_optimizers())
        b = training_batch
        idx = self._native_idx if batch_idx is None else batch_idx
        self._native_idx = idx + 1
        gs = 1.0 / self._dp_world

        def step(params, backward):
            for p in params:
                p.grad = None
            backward()
            o = next(opts)
            o.grad_scale = gs
            o.step()

        self._critic_forward(b)
        for k in ("q1", "q2"):
            if k in self._e:
                step(self._e[k]["params"], lambda k=k: self._critic_backward(k))
        out = dict(q1_loss=self._losses["q1"], q2_loss=self._losses["q2"] if "q2" in self._e else None,
                   actor_loss=None, reward_loss=None, cpe_loss=None)
        update_actor = idx % self.delayed_policy_update == 0
        self._actor_forward(b, with_loss=update_actor)
        if update_actor:
            step(self._e["actor"]["params"], self._actor_backward)
            out["actor_loss"] = self._losses["actor"]
        else:
            next(opts)
        if self._cpe is not None:
            self._cpe.forward(b)
            for k in ("reward", "cpe"):
                step(self._cpe.e[k]["params"], lambda k=k: self._cpe.backward(k))
                out[k + "_loss"] = self._cpe.losses[k]
        next(opts).step()  # soft update of every target
        self.all_batches_processed += 1
        return out
