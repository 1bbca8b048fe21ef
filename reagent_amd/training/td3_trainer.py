"""TD3Trainer with the constructor / generator surface of reagent/training/td3_trainer.py:26-199,
executed on the HIP kernels.

Step: actor_target(s') -> target-policy smoothing (rg_td3_target_action) -> min of the target
critics -> y = r + gamma * not_terminal * min; MSE for q1 / q2 (the SAC critic head with
temperature 0); every `delayed_policy_update`-th batch the actor ascends q1(s, actor(s)) — backward
through the frozen critic into the action columns, through the tanh head (rg_act_backward) and the
actor stack — followed by the soft update of all three targets.
"""
import copy
from typing import Optional

import torch

from .. import _lib as L
from .. import ops
from ..core import types as rlt
from ..core.parameters import RLParameters
from ..engine import dx_save, ensure_slab, grad_views
from ..optimizer import Optimizer__Union, SoftUpdate
from .reagent_lightning_module import ReAgentLightningModule
from .rl_trainer_pytorch import RLTrainerMixin
from .dqn_trainer import dp_reduce, held_gradients, native_step, publish_gradients
from .sac_trainer import _SegmentLoss

CONTINUOUS_TRAINING_ACTION_RANGE = (-1.0, 1.0)  # reagent/core/parameters.py:20


class TD3Trainer(RLTrainerMixin, ReAgentLightningModule):
    def __init__(
        self,
        actor_network,
        q1_network,
        q2_network=None,
        rl: Optional[RLParameters] = None,
        q_network_optimizer: Optional[Optimizer__Union] = None,
        actor_network_optimizer: Optional[Optimizer__Union] = None,
        minibatch_size: int = 64,
        noise_variance: float = 0.2,
        noise_clip: float = 0.5,
        delayed_policy_update: int = 2,
        minibatches_per_step: int = 1,
    ) -> None:
        super().__init__()
        self.rl_parameters = rl if rl is not None else RLParameters()
        self.minibatch_size = minibatch_size
        self.minibatches_per_step = minibatches_per_step or 1
        d = Optimizer__Union.default
        self.q1_network = q1_network
        self.q1_network_target = copy.deepcopy(self.q1_network)
        self.q_network_optimizer = q_network_optimizer if q_network_optimizer is not None else d()
        self.q2_network = q2_network
        if self.q2_network is not None:
            self.q2_network_target = copy.deepcopy(self.q2_network)
        self.actor_network = actor_network
        self.actor_network_target = copy.deepcopy(self.actor_network)
        self.actor_network_optimizer = actor_network_optimizer if actor_network_optimizer is not None else d()
        self.noise_variance = noise_variance
        self.noise_clip_range = (-noise_clip, noise_clip)
        self.delayed_policy_update = delayed_policy_update
        self._ws_batch = -1
        self._dp_group, self._dp_world = None, 1
        self._native_idx = 0

    # ---- optimizers (td3_trainer.py:89-122) ------------------------------------------------------
    def configure_optimizers(self):
        optimizers = [self.q_network_optimizer.make_optimizer_scheduler(self.q1_network.parameters())]
        if self.q2_network:
            optimizers.append(self.q_network_optimizer.make_optimizer_scheduler(self.q2_network.parameters()))
        optimizers.append(self.actor_network_optimizer.make_optimizer_scheduler(self.actor_network.parameters()))
        target_params = list(self.q1_network_target.parameters())
        source_params = list(self.q1_network.parameters())
        if self.q2_network:
            target_params += list(self.q2_network_target.parameters())
            source_params += list(self.q2_network.parameters())
        target_params += list(self.actor_network_target.parameters())
        source_params += list(self.actor_network.parameters())
        optimizers.append(SoftUpdate.make_optimizer_scheduler(target_params, source_params, tau=self.tau))
        return optimizers

    # ---- engine ----------------------------------------------------------------------------------
    @staticmethod
    def _net_engine(net):
        params = list(net.parameters())
        slab = ensure_slab(params)
        dw, db = grad_views(net.fc, slab, params)
        return dict(params=params, slab=slab, stack=net.fc.stack(), dw=dw, db=db)

    def _engine(self, B, dev, S, A):
        nets = dict(actor=self.actor_network, q1=self.q1_network, q2=self.q2_network)
        self._e = {k: self._net_engine(n) for k, n in nets.items() if n is not None}
        tg = dict(actor=self.actor_network_target, q1=self.q1_network_target,
                  q2=getattr(self, "q2_network_target", None))
        self._t = {k: n.fc.stack() for k, n in tg.items() if n is not None}
        self._e["q1"]["stack"].set_need_input_grad(True)
        if self._ws_batch != B or self._x.device != dev:
            f = dict(dtype=torch.float32, device=dev)
            P = ops.sac_partials(B)
            self._x, self._xn, self._xa = (torch.empty(B, S + A, **f) for _ in range(3))
            self._next_actor, self._a_out = torch.empty(B, A, **f), torch.empty(B, A, **f)
            for n in ("q1v", "q2v", "q1t", "q2t", "q1a", "dq1", "dq2", "y"):
                setattr(self, "_" + n, torch.empty(B, 1, **f))
            self._dq1a = torch.full((B, 1), -1.0 / B, **f)  # d(-mean q)/dq
            self._zero_lp = torch.zeros(B, **f)
            self._zero_alpha = torch.zeros(1, dtype=torch.float64, device=dev)
            self._dx1 = torch.empty(B, S + A, **f)
            self._parts = {n: torch.empty(P, **f) for n in ("l1", "l2")}
            self._losses = {n: torch.empty(1, **f) for n in ("q1", "q2", "actor")}
            self._ws_batch = B

    @staticmethod
    def _f32c(t):
        t = t if t.dtype == torch.float32 else t.float()
        return t if t.is_contiguous() else t.contiguous()

    def _publish(self, e, held=()):
        slab = e["slab"]
        if self._dp_group is not None:
            dp_reduce(self, slab)
        publish_gradients(slab, e["params"], held)

    # ---- segments --------------------------------------------------------------------------------
    def _critic_forward(self, b, noise):
        state, action = self._f32c(b.state.float_features), self._f32c(b.action.float_features)
        next_state = self._f32c(b.next_state.float_features)
        L.require_cuda(state, "training_batch.state")
        B, S, A, dev = state.shape[0], state.shape[1], action.shape[1], state.device
        self._engine(B, dev, S, A)
        self._S, self._A, self._B = S, A, B
        e, t = self._e, self._t
        for k in ("q1", "q2"):
            if k in e:
                e[k]["stack"].stage_weights(need_transposed=True)
        for k in t:
            t[k].stage_weights(need_transposed=False)
        # a' = clamp(actor_target(s') + clamp(noise * variance, +-clip), training range)  (:141-146)
        at = t["actor"]
        xn_s, _ = at.stage_input(next_state, need_transposed=False)
        at.forward(xn_s, self._next_actor, save=False)
        if self.actor_network_target.exploration_variance is not None:  # `.action` carries the exploration noise
            self._next_actor.copy_(self.actor_network_target.explore(self._next_actor)[0])
        self._xn[:, :S].copy_(next_state)
        ops.td3_target_action(self._next_actor, noise, self.noise_variance, self.noise_clip_range,
                              CONTINUOUS_TRAINING_ACTION_RANGE[0], CONTINUOUS_TRAINING_ACTION_RANGE[1],
                              self._xn[:, S:])
        xn_c, _ = t["q1"].stage_input(self._xn, need_transposed=False)
        t["q1"].forward(xn_c, self._q1t, save=False)
        has_q2 = "q2" in e
        if has_q2:
            t["q2"].forward(xn_c, self._q2t, save=False)
        self._x[:, :S].copy_(state)
        self._x[:, S:].copy_(action)
        q1s = e["q1"]["stack"]
        x_c, self._x_t = q1s.stage_input(self._x, need_transposed=True)
        q1s.forward(x_c, self._q1v, save=True)
        if has_q2:
            e["q2"]["stack"].forward(x_c, self._q2v, save=True)
        # y = r + gamma * not_terminal * min(q1', q2'); mse for both critics (:147-167) — the SAC
        # critic head with temperature 0
        ops.sac_critic_head(self._q1v, self._q2v if has_q2 else None, self._q1t, self._q2t if has_q2 else None,
                            self._zero_lp, self._f32c(b.reward).reshape(-1), self._f32c(b.not_terminal).reshape(-1),
                            self.gamma, self._zero_alpha, self._y, self._dq1, self._dq2 if has_q2 else None,
                            self._parts["l1"], self._parts["l2"] if has_q2 else None)
        P = self._parts["l1"].numel()
        ops.reduce_sum(self._parts["l1"], P, 1.0 / B, self._losses["q1"])
        if has_q2:
            ops.reduce_sum(self._parts["l2"], P, 1.0 / B, self._losses["q2"])

    def _critic_backward(self, which, grad_out=None):
        e = self._e[which]
        dq = self._dq1 if which == "q1" else self._dq2
        if grad_out is not None:
            dq = dq * grad_out
        held = held_gradients(e["slab"], e["params"])
        e["stack"].backward(dq, self._x_t, e["dw"], e["db"])
        self._publish(e, held)

    def _actor_forward(self, b):
        state = self._f32c(b.state.float_features)
        S, B = self._S, self._B
        e = self._e
        e["q1"]["stack"].stage_weights(need_transposed=True)  # q1 was just updated by its Adam step
        act = e["actor"]["stack"]
        act.stage_weights(need_transposed=True)
        xs_c, self._xs_t = act.stage_input(state, need_transposed=True)
        act.forward(xs_c, self._a_out, save=True)
        self._xa[:, :S].copy_(state)
        self._clamp_passes = None
        if self.actor_network.exploration_variance is not None:
            noisy, self._clamp_passes, _ = self.actor_network.explore(self._a_out)
            self._xa[:, S:].copy_(noisy)
        else:
            self._xa[:, S:].copy_(self._a_out)
        q1s = e["q1"]["stack"]
        xa_c, _ = q1s.stage_input(self._xa, need_transposed=False)
        q1s.forward(xa_c, self._q1a, save=dx_save(q1s))  # q1 is frozen here: only d q / d action comes back
        ops.reduce_sum(self._q1a.reshape(-1), B, -1.0 / B, self._losses["actor"])  # -(q1(s, actor(s)).mean())

    def _actor_backward(self, grad_out=None):
        e, S = self._e, self._S
        dq = self._dq1a if grad_out is None else self._dq1a * grad_out
        e["q1"]["stack"].backward(dq, None, None, None, dx32=self._dx1, skip_wgrad=True)
        a = e["actor"]
        da = self._dx1[:, S:]
        if self._clamp_passes is not None:  # backward of the exploration clamp
            da = da * self._clamp_passes
        held = held_gradients(a["slab"], a["params"])
        a["stack"].backward(da, self._xs_t, a["dw"], a["db"], out32=self._a_out)
        self._publish(a, held)

    def _noise(self, B, A, dev, given):
        if given is not None:
            return given.to(device=dev, dtype=torch.float32).contiguous()
        return torch.randn(B, A, device=dev)

    # ---- reference surface -----------------------------------------------------------------------
    def set_noise(self, noise: torch.Tensor):
        """inject this step's N(0,1) draw (the reference's torch.randn_like(next_actor), :142) — parity runs"""
        self._injected = noise

    def train_step_gen(self, training_batch: rlt.PolicyNetworkInput, batch_idx: int):
        """IMPORTANT: the input action here is assumed to match the range of the output of the actor."""
        assert hasattr(training_batch, "action") and hasattr(training_batch.action, "float_features")
        b = training_batch
        B, A = b.action.float_features.shape
        inj = getattr(self, "_injected", None)
        self._injected = None
        self._critic_forward(b, self._noise(B, A, b.action.float_features.device, inj))
        # td3_trainer.py:158-164, :173-177, :185-189: tensors handed to the reporter every `log_every_n_steps` batches
        # (no-op reporter: nothing is evaluated)
        from .reagent_lightning_module import _NoOpReporter

        every = getattr(getattr(self, "trainer", None), "log_every_n_steps", 50)
        report = not isinstance(self._reporter, _NoOpReporter) and batch_idx % every == 0
        has_q2 = "q2" in self._e
        if report:
            next_q = torch.minimum(self._q1t, self._q2t) if has_q2 else self._q1t
            self.reporter.log(q1_loss=self._losses["q1"].reshape(()).clone(), q1_value=self._q1v.reshape(-1, 1).clone(),
                              next_q_value=next_q.reshape(-1, 1), target_q_value=self._y.reshape(-1, 1).clone())
        q1 = self._e["q1"]
        yield _SegmentLoss.apply(lambda g: self._critic_backward("q1", g), self._losses["q1"], *q1["params"])
        if self.q2_network:
            if report:
                self.reporter.log(q2_loss=self._losses["q2"].reshape(()).clone(), q2_value=self._q2v.reshape(-1, 1).clone())
            q2 = self._e["q2"]
            yield _SegmentLoss.apply(lambda g: self._critic_backward("q2", g), self._losses["q2"], *q2["params"])
        # only update actor and target networks after a fixed number of Q updates (:176-199)
        if batch_idx % self.delayed_policy_update == 0:
            self._actor_forward(b)
            if report:
                self.reporter.log(actor_loss=self._losses["actor"].reshape(()).clone(),
                                  actor_q1_value=self._q1a.reshape(-1, 1).clone())
            yield _SegmentLoss.apply(self._actor_backward, self._losses["actor"], *self._e["actor"]["params"])
            yield self.soft_update_result()
        else:
            yield None  # None keeps the actor and the target networks from updating
            yield None

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
    def train_step_native(self, training_batch, noise=None, batch_idx: Optional[int] = None):
        """critic segments, and on every delayed_policy_update-th call the actor segment and the soft
        update, with no autograd graph / generator / host sync"""
        opts = self.native_optimizers()
        b = training_batch
        B, A = b.action.float_features.shape
        idx = self._native_idx if batch_idx is None else batch_idx
        self._native_idx = idx + 1
        gs = 1.0 / self._dp_world
        it = iter(opts)
        self._critic_forward(b, self._noise(B, A, b.action.float_features.device, noise))
        for k in ("q1", "q2"):
            if k in self._e:
                for p in self._e[k]["params"]:
                    p.grad = None
                self._critic_backward(k)
                o = next(it)
                o.grad_scale = gs
                o.step()
        actor_opt, soft = next(it), next(it)
        out = dict(q1_loss=self._losses["q1"], q2_loss=self._losses["q2"] if "q2" in self._e else None,
                   actor_loss=None)
        if idx % self.delayed_policy_update == 0:
            self._actor_forward(b)
            for p in self._e["actor"]["params"]:
                p.grad = None
            self._actor_backward()
            actor_opt.grad_scale = gs
            actor_opt.step()
            soft.step()
            out["actor_loss"] = self._losses["actor"]
        self.all_batches_processed += 1
        return out
