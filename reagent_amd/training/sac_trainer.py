"""SACTrainer with the constructor / generator surface of reagent/training/sac_trainer.py:50-385
(twin or single critic, optional value network, CRR actor weights, detached log-prob, temperature optimizer
optional; only the action-embedding KLD term is not built), executed on the HIP kernels.

One step, in the reference's segment order (each segment = one optimizer under the
Lightning-1.6 toggle, so other networks are constants inside it):
  seg q1/q2 : a' = actor(s'), log_prob' ; y = r + g*(min(q1_t,q2_t)(s',a') - alpha*clamp(log_prob'))*not_done
              q_i = q_i(s, a); loss_i = mse(q_i, y)                       -> Adam(q1), Adam(q2)
  seg actor : (a_pi, log_prob) = actor(s); loss = mean(alpha*clamp(log_prob) - min(q1,q2)(s, a_pi)) with the
              UPDATED critics; the gradient reaches the actor through the critics' action input -> Adam(actor)
  seg alpha : loss = -mean(log_alpha * (clamp(log_prob) + target_entropy)); alpha = exp(log_alpha)   (fp64)
  seg value : (value_network only) loss = mse(V(s), min(q1,q2)(s, a_pi) [- alpha*clamp(log_prob) unless
              logged_action_uniform_prior]) -> Adam(value)                                            (:325-340)
  soft update of the target critics (of the target value network when there is one, :177-191).
With a value network the critics regress r + g*V_target(s')*not_done (:214-215) and there are no target critics;
crr_config replaces the actor loss by -(clamp(log_prob) * w(min q - V(s))) (:265-273, CRRWeightFn :23-47);
backprop_through_log_prob=False detaches log_prob in the actor loss (:262-263).
Algorithmic FC work: 2 actor forwards + 1 actor backward, 6 critic forwards, 2 full critic backwards,
2 input-gradient-only critic backwards (the reference additionally re-evaluates the actor FC inside
get_log_prob: identical values, not repeated here).
"""
import copy
import math
import os
from typing import List, Optional

import numpy as np
import torch

from .. import _lib as L
from .. import ops
from ..core import types as rlt
from ..core.parameters import RLParameters
from ..engine import dx_save, ensure_slab, grad_views
from ..models.actor import LOG_PROB_MAX, LOG_PROB_MIN
from ..optimizer import Optimizer__Union, SoftUpdate
from .dqn_trainer import dp_reduce, held_gradients, native_step, publish_gradients
from .reagent_lightning_module import ReAgentLightningModule
from .rl_trainer_pytorch import RLTrainerMixin


class CRRWeightFn:
    """sac_trainer.py:23-47: indicator (advantage >= threshold) or exp(advantage / beta) clamped to [0, exponent_clamp]"""

    def __init__(self, indicator_fn_threshold: Optional[float] = None, exponent_beta: Optional[float] = None,
                 exponent_clamp: Optional[float] = None):
        assert exponent_beta or indicator_fn_threshold
        assert not (exponent_beta and indicator_fn_threshold)
        if exponent_beta:
            assert exponent_beta > 1e-6
        if exponent_clamp:
            assert exponent_clamp > 1e-6
        self.indicator_fn_threshold, self.exponent_beta, self.exponent_clamp = indicator_fn_threshold, exponent_beta, exponent_clamp

    def get_weight_from_advantage(self, advantage):
        if self.indicator_fn_threshold:
            return (advantage >= self.indicator_fn_threshold).float()
        exp = torch.exp(advantage / self.exponent_beta)
        if self.exponent_clamp:
            exp = torch.clamp(exp, 0.0, self.exponent_clamp)
        return exp

    def kernel_args(self):
        if self.indicator_fn_threshold:
            return 1, float(self.indicator_fn_threshold), 0.0
        return 2, float(self.exponent_beta), float(self.exponent_clamp or 0.0)


class _SegmentLoss(torch.autograd.Function):
    """Scalar loss whose backward runs a HIP backward closure (writes ``.grad`` in place)."""

    @staticmethod
    def forward(ctx, closure, loss_buf, *params):
        ctx.closure = closure
        ctx.n = len(params)
        return loss_buf.detach().clone().reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        ctx.closure(grad_out)
        return (None, None) + (None,) * ctx.n


class AdamF64(torch.optim.Optimizer):
    """torch.optim.Adam arithmetic for the single fp64 temperature parameter (rg_adam_step_f64);
    also publishes alpha = exp(log_alpha) to the device scalar the loss heads read."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, alpha_out=None):
        if weight_decay != 0.0:
            # torch.optim.Adam on log_alpha would add weight_decay * log_alpha to the gradient; rg_adam_step_f64 has no
            # such term, so a non-zero value must not pass silently
            raise NotImplementedError("alpha_optimizer weight_decay != 0 is not supported by the fp64 temperature step")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.alpha_out = alpha_out
        self._sched = None

    # graph-safe stepping: same protocol as FusedAdam (optimizer/__init__.py)
    def enable_device_schedule(self):
        if self._sched is None:
            from ..optimizer import AdamSchedule

            group = self.param_groups[0]
            p = group["params"][0]
            st = self.state[p]
            if len(st) == 0:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            self._sched = AdamSchedule(group["lr"], group["betas"], int(st["step"]), p.device)
        return self

    def disable_device_schedule(self):
        self.materialize_steps()
        self._sched = None
        return self

    def note_device_steps(self, n: int):
        if self._sched is not None:
            self._sched.pending += n

    def materialize_steps(self):
        if self._sched is not None and self._sched.pending:
            for p in self.param_groups[0]["params"]:
                self.state[p]["step"] += self._sched.pending
            self._sched.pending = 0

    def state_dict(self):
        self.materialize_steps()
        return super().state_dict()

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                out = self.alpha_out() if callable(self.alpha_out) else self.alpha_out
                if self._sched is not None:
                    from ..optimizer import capturing

                    if not capturing():
                        self._sched.set_lr(group["lr"])
                        self._sched.pending += 1
                    ops.adam_step_f64_sched(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], b1, b2, group["eps"],
                                            self._sched.buf, out)
                    ops.sched_tick(self._sched.buf)
                    continue
                st["step"] += 1
                t = st["step"]
                ops.adam_step_f64(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2,
                                  group["eps"], 1.0 - b1**t, math.sqrt(1.0 - b2**t), out)
        return None


class SACTrainer(RLTrainerMixin, ReAgentLightningModule):
    def __init__(
        self,
        actor_network,
        q1_network,
        q2_network=None,
        value_network=None,
        rl: Optional[RLParameters] = None,
        q_network_optimizer: Optional[Optimizer__Union] = None,
        value_network_optimizer: Optional[Optimizer__Union] = None,
        actor_network_optimizer: Optional[Optimizer__Union] = None,
        alpha_optimizer: Optional[Optimizer__Union] = "default",
        minibatch_size: int = 1024,
        entropy_temperature: float = 0.01,
        logged_action_uniform_prior: bool = True,
        target_entropy: float = -1.0,
        action_embedding_kld_weight: Optional[float] = None,
        apply_kld_on_mean: bool = False,
        action_embedding_mean: Optional[List[float]] = None,
        action_embedding_variance: Optional[List[float]] = None,
        crr_config: Optional[CRRWeightFn] = None,
        backprop_through_log_prob: bool = True,
    ) -> None:
        super().__init__()
        self.rl_parameters = rl if rl is not None else RLParameters()
        d = Optimizer__Union.default
        self.q1_network = q1_network
        self.q2_network = q2_network
        self.q_network_optimizer = q_network_optimizer if q_network_optimizer is not None else d()
        self.value_network = value_network
        self.value_network_optimizer = value_network_optimizer if value_network_optimizer is not None else d()
        if self.value_network is not None:  # sac_trainer.py:108-112: a target value network INSTEAD of target critics
            self.value_network_target = copy.deepcopy(self.value_network)
            self.q1_network_target = self.q2_network_target = None
        else:
            self.q1_network_target = copy.deepcopy(self.q1_network)
            self.q2_network_target = copy.deepcopy(self.q2_network)
        self.actor_network = actor_network
        self.actor_network_optimizer = actor_network_optimizer if actor_network_optimizer is not None else d()
        self.entropy_temperature = entropy_temperature
        self.alpha_optimizer = d() if isinstance(alpha_optimizer, str) else alpha_optimizer
        if self.alpha_optimizer is not None:
            self.target_entropy = target_entropy
            self.log_alpha = torch.nn.Parameter(torch.tensor([np.log(self.entropy_temperature)]))  # fp64 (:124-126)
        else:
            self.target_entropy = target_entropy
        self.logged_action_uniform_prior = logged_action_uniform_prior
        # action-embedding KLD term of the actor loss (sac_trainer.py:130-140, 282-306)
        self.add_kld_to_loss = bool(action_embedding_kld_weight)
        self.apply_kld_on_mean = apply_kld_on_mean
        if self.add_kld_to_loss:
            self.kld_weight = action_embedding_kld_weight
            self.register_buffer("action_emb_mean", None)
            self.register_buffer("action_emb_variance", None)
            self.action_emb_mean = torch.tensor(action_embedding_mean)
            self.action_emb_variance = torch.tensor(action_embedding_variance)
        self.crr_config = crr_config
        if crr_config:
            assert self.value_network is not None
        self.backprop_through_log_prob = backprop_through_log_prob
        self.minibatch_size = minibatch_size
        self.use_fused_update = os.environ.get("RG_SAC_FUSED_UPDATE", "1") != "0"  # engine.FusedUpdate in the native step
        self._ws_batch = -1
        self._alpha_dev = None
        self._dp_group, self._dp_world = None, 1

    # ---- optimizers (sac_trainer.py:148-193) ---------------------------------------------------
    def configure_optimizers(self):
        optimizers = [self.q_network_optimizer.make_optimizer_scheduler(self.q1_network.parameters())]
        if self.q2_network:
            optimizers.append(self.q_network_optimizer.make_optimizer_scheduler(self.q2_network.parameters()))
        optimizers.append(self.actor_network_optimizer.make_optimizer_scheduler(self.actor_network.parameters()))
        if self.alpha_optimizer is not None:
            cfg = self.alpha_optimizer.value
            optimizers.append({"optimizer": AdamF64([self.log_alpha], lr=cfg.lr, betas=tuple(cfg.betas), eps=cfg.eps,
                                                    weight_decay=cfg.weight_decay,
                                                    alpha_out=lambda: self._alpha(self.log_alpha.device))})
        if self.value_network is not None:
            optimizers.append(self.value_network_optimizer.make_optimizer_scheduler(self.value_network.parameters()))
            target_params = list(self.value_network_target.parameters())
            source_params = list(self.value_network.parameters())
        else:
            target_params = list(self.q1_network_target.parameters())
            source_params = list(self.q1_network.parameters())
            if self.q2_network:
                target_params += list(self.q2_network_target.parameters())
                source_params += list(self.q2_network.parameters())
        optimizers.append(SoftUpdate.make_optimizer_scheduler(target_params, source_params, tau=self.tau))
        return optimizers

    # ---- checkpoint extras (runtime loops) ---------------------------------------------------------
    def checkpoint_extras(self) -> dict:
        """alpha = exp(log_alpha) as the loss heads read it: like the reference's `entropy_temperature` attribute
        (sac_trainer.py:322) it is not part of the state_dict, and is the constructor's value until the first
        temperature step — a resumed run must continue from the current one"""
        return {"alpha": self._alpha(self._extras_device()).detach().cpu().clone()}

    def load_checkpoint_extras(self, extras: dict):
        self._alpha(self._extras_device()).copy_(extras["alpha"])

    def _extras_device(self):
        # `log_alpha` exists only with an alpha optimizer (a fixed temperature is a supported configuration)
        return next(self.actor_network.parameters()).device

    # ---- engine ----------------------------------------------------------------------------------
    def _alpha(self, device):
        if self._alpha_dev is None or self._alpha_dev.device != device:
            et = self.entropy_temperature
            val = float(et.item()) if isinstance(et, torch.Tensor) else float(et)
            self._alpha_dev = torch.tensor([val], dtype=torch.float64, device=device)
        return self._alpha_dev

    def _net_engine(self, net):
        params = list(net.parameters())
        slab = ensure_slab(params)
        dw, db = grad_views(net.fc, slab, params)
        return dict(params=params, slab=slab, stack=net.fc.stack(), dw=dw, db=db)

    def _engine(self, B, dev, S, A):
        self._e = {k: self._net_engine(n) for k, n in dict(actor=self.actor_network, q1=self.q1_network,
                                                           q2=self.q2_network, value=self.value_network).items()
                   if n is not None}
        self._t = {k: n.fc.stack() for k, n in dict(q1=self.q1_network_target, q2=self.q2_network_target,
                                                    value=getattr(self, "value_network_target", None)).items()
                   if n is not None}
        for k in ("q1", "q2"):
            if k in self._e:
                self._e[k]["stack"].set_need_input_grad(True)
        from ..engine import FusedMLP

        # cat(state, action) is read in place by the fused kernels as two K-panels (critic.py:79-92) when every
        # critic stack runs on them; the per-layer GEMM engine takes the assembled [B, S + A] matrix
        self._panels = S % 32 == 0 and all(isinstance(st, FusedMLP) for st in
                                          [self._e[k]["stack"] for k in ("q1", "q2") if k in self._e]
                                          + [st for k, st in self._t.items() if k != "value"])
        if self._ws_batch != B or self._x.device != dev:
            f = dict(dtype=torch.float32, device=dev)
            P = ops.sac_partials(B)
            cat_shape = (0, S + A) if self._panels else (B, S + A)  # assembled inputs: per-layer GEMM engine only
            self._x, self._xn, self._xa = (torch.empty(*cat_shape, **f) for _ in range(3))
            self._ls, self._lsn, self._dls = (torch.empty(B, 2 * A, **f) for _ in range(3))
            if getattr(self.actor_network, "use_layer_norm", False):  # raw FC outputs / their gradient + LN statistics
                self._ls_raw, self._dls_raw = torch.empty(B, 2 * A, **f), torch.empty(B, 2 * A, **f)
                self._ln_stats = tuple((torch.empty(B, **f), torch.empty(B, **f)) for _ in range(2))
                nb = L.lib().rg_layer_norm_backward_workspace_bytes(B, A)
                self._ln_ws = torch.empty((nb + 3) // 4, **f)
            self._lp, self._lpn = torch.empty(B, **f), torch.empty(B, **f)
            names = ["q1v", "q2v", "q1t", "q2t", "q1a", "q2a", "dq1", "dq2", "dq1a", "dq2a", "y", "glp", "vcur", "vval", "dv", "yv"]
            for n in names:
                setattr(self, "_" + n, torch.empty(B, 1, **f))
            self._dx1, self._dx2 = torch.empty(*cat_shape, **f), torch.empty(*cat_shape, **f)
            self._ga = torch.empty(B, A, **f)
            # two-panel critic input (fused kernels): the actor's actions and the action part of dQ/dx on their own
            self._an, self._api = torch.empty(B, A, **f), torch.empty(B, A, **f)
            self._dxa1, self._dxa2 = torch.empty(B, A, **f), torch.empty(B, A, **f)
            self._parts = {n: torch.empty(P, **f) for n in ("l1", "l2", "la", "ent", "lv")}
            self._losses = {n: torch.empty(1, **f) for n in ("q1", "q2", "actor", "value")}
            self._zeros, self._ones = torch.zeros(B, **f), torch.ones(B, **f)
            self._alpha_grad = torch.zeros(1, dtype=torch.float64, device=dev)
            self._alpha_loss = torch.zeros(1, dtype=torch.float64, device=dev)
            self._ws_batch = B

    @staticmethod
    def _f32c(t):
        t = t if t.dtype == torch.float32 else t.float()
        return t if t.is_contiguous() else t.contiguous()

    @staticmethod
    def _state_in(t):
        """state rows as a network input: fp32, or the sampler's network-ready bf16 rows (bf16 engine: the fused
        kernels would round the fp32 rows to the same bf16 values on load)"""
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float()
        return t if t.is_contiguous() else t.contiguous()

    def _actor_out(self, act, x, out, save: bool):
        """actor FC stack -> [loc | scale_log] in `out`; a layer-normed actor passes both halves through their
        LayerNorm (actor.py:194-196), keeping the raw outputs and the statistics of a saving call"""
        if hasattr(act, "stat_updates"):
            # batch-norm running statistics move once per evaluation of the stack in the reference: forward() evaluates it
            # twice (actor.py:216,246), and the next-state branch calls get_log_prob() once more (sac_trainer.py:228)
            act.stat_updates = 2 if save else 3
        if not getattr(self.actor_network, "use_layer_norm", False):
            return act.forward(x, out, save=save)
        raw = self._ls_raw if save else self._dls_raw  # (the gradient buffer is free during a forward)
        act.forward(x, raw, save=save)
        return self.actor_network.head_norm(raw, out, self._ln_stats if save else None)

    def _publish(self, e, held=(), reduce=True):
        slab = e["slab"]
        if self._dp_group is not None and reduce:
            rider = self._alpha_rider if e is self._e.get("actor") else None
            if rider is not None:
                # native data-parallel step: the temperature's gradient rides behind the actor's slab (_dp_actor_bucket) — ONE
                # collective for both (the 1 / world is folded into the Adam launches)
                with ops.profile_span("all_reduce", dict(bytes=rider.numel() * 4, world=self._dp_world)):
                    torch.distributed.all_reduce(rider, group=self._dp_group)
            else:
                with ops.profile_span("all_reduce", dict(bytes=slab.grad.numel() * 4, world=self._dp_world)):
                    dp_reduce(self, slab)
        publish_gradients(slab, e["params"], held)

    _alpha_rider = None  # set by the native data-parallel step around the actor's backward pass

    def _dp_actor_bucket(self):
        """Data parallel: [the actor's gradient slab | one float] — the temperature's gradient (the mean of log_prob + target
        entropy over the rank's rows, a float64 scalar) travels as the float behind the actor's gradients, so a native SAC step
        is TWO collectives (critics, actor + temperature), not three (few, large collectives).  The fp32 sum over
        the ranks rounds the temperature's gradient at 6e-8 relative; its Adam step is float64 as before."""
        e = self._e["actor"]
        slab = e["slab"]
        b = getattr(self, "_dp_bucket_actor", None)
        ok = (b is not None and b.device == slab.grad.device and b.numel() == slab.total + 1
              and slab.grad.data_ptr() == b.data_ptr())
        if not ok:
            b = torch.zeros(slab.total + 1, dtype=torch.float32, device=slab.grad.device)
            view = b[:slab.total]
            view.copy_(slab.grad)
            old = slab.grad.data_ptr()
            for i, p in enumerate(slab.params):  # gradients published through the old slab follow it
                if p.grad is not None and p.grad.data_ptr() == old + 4 * slab.offsets[i]:
                    p.grad = slab.view(view, i)
            slab.grad = view
            e["dw"], e["db"] = grad_views(self.actor_network.fc, slab, e["params"])
            self._dp_bucket_actor = b
        return b

    def _dp_bucket(self):
        """Data parallel: the gradient slabs of the two critics become the two halves of ONE buffer, so that the native step
        sums both with a single all-reduce (few, large collectives — at C4 2 x 2.7 MB are latency-bound either
        way, and a step is a chain of them).  Everything that reads a slab's gradients goes through `slab.grad`."""
        if "q2" not in self._e:
            return None
        s1, s2 = self._e["q1"]["slab"], self._e["q2"]["slab"]
        b = getattr(self, "_dp_bucket_q", None)
        ok = (b is not None and b.device == s1.grad.device and b.numel() == s1.total + s2.total
              and s1.grad.data_ptr() == b.data_ptr() and s2.grad.data_ptr() == b.data_ptr() + 4 * s1.total)
        if not ok:
            b = torch.zeros(s1.total + s2.total, dtype=torch.float32, device=s1.grad.device)
            for slab, off in ((s1, 0), (s2, s1.total)):
                view = b[off:off + slab.total]
                view.copy_(slab.grad)
                old = slab.grad.data_ptr()
                for i, p in enumerate(slab.params):  # gradients published through the old slab follow it
                    if p.grad is not None and p.grad.data_ptr() == old + 4 * slab.offsets[i]:
                        p.grad = slab.view(view, i)
                slab.grad = view
            for k in ("q1", "q2"):
                e = self._e[k]
                e["dw"], e["db"] = grad_views(getattr(self, f"{k}_network").fc, e["slab"], e["params"])
            self._dp_bucket_q = b
        return b

    # ---- segments ----------------------------------------------------------------------------------
    def _critic_forward(self, b, noise_next):
        state, action = self._state_in(b.state.float_features), self._f32c(b.action.float_features)
        next_state = self._state_in(b.next_state.float_features)
        L.require_cuda(state, "training_batch.state")
        B, S, A, dev = state.shape[0], state.shape[1], action.shape[1], state.device
        self._engine(B, dev, S, A)
        self._S, self._A, self._B = S, A, B
        e, t = self._e, self._t
        for k in e:
            e[k]["stack"].stage_weights(need_transposed=True)
        for k in t:
            t[k].stage_weights(need_transposed=False)
        alpha = self._alpha(dev)
        act = e["actor"]["stack"]
        q1s = e["q1"]["stack"]
        has_q2 = "q2" in e
        if "value" in t:  # next_state_value = value_network_target(next_state), no entropy term (:214-215)
            xn_v, _ = t["value"].stage_input(next_state, need_transposed=False)
            t["value"].forward(xn_v, self._q1t, save=False)
            if self._panels:
                self._x_t = None
                q1s.forward(state, self._q1v, save=True, x2=action)
                if has_q2:
                    e["q2"]["stack"].forward(state, self._q2v, save=True, x2=action)
            else:
                self._x[:, :S].copy_(state)
                self._x[:, S:].copy_(action)
                x_c, self._x_t = q1s.stage_input(self._x, need_transposed=True)
                q1s.forward(x_c, self._q1v, save=True)
                if has_q2:
                    e["q2"]["stack"].forward(x_c, self._q2v, save=True)
            ops.sac_critic_head(self._q1v, self._q2v if has_q2 else None, self._q1t, None, self._zeros,
                                self._f32c(b.reward).reshape(-1), self._f32c(b.not_terminal).reshape(-1), self.gamma, alpha,
                                self._y, self._dq1, self._dq2 if has_q2 else None, self._parts["l1"],
                                self._parts["l2"] if has_q2 else None)
            self._loss_mean("q1", self._parts["l1"], 1.0 / B, self._losses["q1"])
            if has_q2:
                self._loss_mean("q2", self._parts["l2"], 1.0 / B, self._losses["q2"])
            self._critic_metrics(b, alpha)
            return
        # a' = actor(s'), log_prob'  (actor frozen in this segment)
        xn_s, _ = act.stage_input(next_state, need_transposed=False)
        self._actor_out(act, xn_s, self._lsn, save=False)
        if self._panels:
            ops.gaussian_head_forward(self._lsn, noise_next, self._an, self._lpn, None)
            t["q1"].forward(next_state, self._q1t, save=False, x2=self._an)
            if "q2" in t:
                t["q2"].forward(next_state, self._q2t, save=False, x2=self._an)
            self._x_t = None
            q1s.forward(state, self._q1v, save=True, x2=action)  # q_i(s, a)
            if has_q2:
                e["q2"]["stack"].forward(state, self._q2v, save=True, x2=action)
        else:
            self._xn[:, :S].copy_(next_state)
            a_next = self._xn[:, S:]
            ops.gaussian_head_forward(self._lsn, noise_next, a_next, self._lpn, None)
            xn_c, _ = t["q1"].stage_input(self._xn, need_transposed=False)
            t["q1"].forward(xn_c, self._q1t, save=False)
            if "q2" in t:
                t["q2"].forward(xn_c, self._q2t, save=False)
            # q_i(s, a)
            self._x[:, :S].copy_(state)
            self._x[:, S:].copy_(action)
            x_c, self._x_t = q1s.stage_input(self._x, need_transposed=True)
            q1s.forward(x_c, self._q1v, save=True)
            if has_q2:
                e["q2"]["stack"].forward(x_c, self._q2v, save=True)
        ops.sac_critic_head(self._q1v, self._q2v if has_q2 else None, self._q1t, self._q2t if has_q2 else None,
                            self._lpn, self._f32c(b.reward).reshape(-1), self._f32c(b.not_terminal).reshape(-1),
                            self.gamma, alpha, self._y, self._dq1, self._dq2 if has_q2 else None, self._parts["l1"],
                            self._parts["l2"] if has_q2 else None)
        self._loss_mean("q1", self._parts["l1"], 1.0 / B, self._losses["q1"])
        if has_q2:
            self._loss_mean("q2", self._parts["l2"], 1.0 / B, self._losses["q2"])
        self._critic_metrics(b, alpha)

    # The native step evaluates a segment's mean loss inside the reduce launch of that network's weight gradient
    # (rg_mlp_wgrad_fused's sum_in, the arithmetic of rg_reduce_sum: same bits) instead of its own launch — nobody reads the
    # loss before the backward of the same segment has been enqueued there.  The generator path (train_step_gen yields the
    # loss before its backward) and stacks off the fused kernels keep the separate launch.
    _fold_losses = False
    _tails: dict = {}  # (replaced by a fresh dict in every native step; empty outside)

    def _loss_mean(self, key, part, scale, out):
        from ..engine import FusedMLP

        st = self._e[key]["stack"]
        if self._fold_losses and isinstance(st, FusedMLP) and st.fold_tails and not (key == "actor" and self.add_kld_to_loss):
            self._tails[key] = (part, scale, out)
        else:
            ops.reduce_sum(part, part.numel(), scale, out)

    def _take_tail(self, key) -> dict:
        tail = self._tails.pop(key, None)
        return {"tail_sum": tail} if tail is not None else {}

    def _critic_metrics(self, b, alpha):
        """critic-segment means of sac_trainer.py:343-364 for the logger, taken while alpha is still the value the
        targets were built with (the temperature step of this batch has not run yet)"""
        if not self.logger:
            return
        t, has_q2 = self._t, "q2" in self._e
        m = self._metrics = {"logged_rewards": b.reward.float().mean(), "q1_value": self._q1v.mean(),
                             "target_q_value": self._y.mean()}
        if has_q2:
            m["q2_value"] = self._q2v.mean()
        if "value" in t:
            m["next_state_value"] = self._q1t.mean()
        else:
            lpa = self._lpn.clamp(LOG_PROB_MIN, LOG_PROB_MAX)
            nsv = torch.minimum(self._q1t, self._q2t) if "q2" in t else self._q1t
            m["log_prob_a"] = lpa.mean()
            m["next_state_value"] = (nsv.reshape(-1) - alpha.float() * lpa.reshape(-1)).mean()

    def _critic_backward(self, which, grad_out=None, reduce=True):
        e = self._e[which]
        dq = self._dq1 if which == "q1" else self._dq2
        if grad_out is not None:
            dq = dq * grad_out
        held = held_gradients(e["slab"], e["params"])
        e["stack"].backward(dq, self._x_t, e["dw"], e["db"], **self._take_tail(which))
        self._publish(e, held, reduce=reduce)

    def _actor_forward(self, b, noise_cur):
        state = self._state_in(b.state.float_features)
        S, B, dev = self._S, self._B, state.device
        e = self._e
        for k in ("q1", "q2"):  # critics were just updated by their Adam steps
            if k in e:
                e[k]["stack"].stage_weights(need_transposed=True)
        act = e["actor"]["stack"]
        xs_c, self._xs_t = act.stage_input(state, need_transposed=True)
        self._actor_out(act, xs_c, self._ls, save=True)
        self._noise_cur = noise_cur
        q1s = e["q1"]["stack"]
        has_q2 = "q2" in e
        if self._panels:
            ops.gaussian_head_forward(self._ls, noise_cur, self._api, self._lp, None)
            # the critics are frozen in this segment: only d q / d action comes back (dx_save)
            q1s.forward(state, self._q1a, save=dx_save(q1s), x2=self._api)
            if has_q2:
                e["q2"]["stack"].forward(state, self._q2a, save=dx_save(q1s), x2=self._api)
        else:
            self._xa[:, :S].copy_(state)
            ops.gaussian_head_forward(self._ls, noise_cur, self._xa[:, S:], self._lp, None)
            xa_c, _ = q1s.stage_input(self._xa, need_transposed=False)
            q1s.forward(xa_c, self._q1a, save=dx_save(q1s))
            if has_q2:
                e["q2"]["stack"].forward(xa_c, self._q2a, save=dx_save(q1s))
        crr_mode, crr_p0, crr_clamp, v_cur = 0, 0.0, 0.0, None
        if self.crr_config is not None:  # advantage = min q - V(state), both detached (:265-268)
            vs = e["value"]["stack"]
            vs.stage_weights(need_transposed=True)
            xv, _ = vs.stage_input(state, need_transposed=False)
            vs.forward(xv, self._vcur, save=False)
            crr_mode, crr_p0, crr_clamp = self.crr_config.kernel_args()
            v_cur = self._vcur
        ops.sac_actor_head(self._lp, self._q1a, self._q2a if has_q2 else None, self._alpha(dev),
                           self.target_entropy, self._glp, self._dq1a, self._dq2a if has_q2 else None,
                           self._parts["la"], self._parts["ent"], v_cur=v_cur, crr_mode=crr_mode, crr_p0=crr_p0,
                           crr_clamp=crr_clamp, backprop_log_prob=self.backprop_through_log_prob)
        self._loss_mean("actor", self._parts["la"], 1.0 / B, self._losses["actor"])
        if self.add_kld_to_loss:  # + kld_weight * KLD(batch statistics of the action || embedding prior), :282-306
            A = self.action_emb_mean.numel()
            if getattr(self, "_kld_coef", None) is None or self._kld_coef.device != dev:
                f32 = dict(dtype=torch.float32, device=dev)
                self._kld_coef, self._kld_terms, self._kld = torch.empty(2 * A, **f32), torch.empty(A, **f32), torch.empty(1, **f32)
            if self.apply_kld_on_mean:  # statistics of squashed_mean = clamp(tanh(loc))
                x, squash = self._ls[:, :A], True
            else:
                x, squash = (self._api if self._panels else self._xa[:, S:]), False
            ops.sac_kld(x, squash, self.action_emb_mean.to(dev), self.action_emb_variance.to(dev), self.kld_weight,
                        self._kld_coef, self._kld_terms, self._kld, self._losses["actor"])

    def _actor_backward(self, grad_out=None):
        e, S = self._e, self._S
        has_q2 = "q2" in e
        if self._panels:  # only the action columns of dQ/d(input) are produced
            e["q1"]["stack"].backward(self._dq1a, None, None, None, dx32=self._dxa1, skip_wgrad=True, dx_col0=S)
            if has_q2:
                e["q2"]["stack"].backward(self._dq2a, None, None, None, dx32=self._dxa2, skip_wgrad=True, dx_col0=S)
            ops.add_cols(self._dxa1, self._dxa2 if has_q2 else None, self._ga)
        else:
            e["q1"]["stack"].backward(self._dq1a, None, None, None, dx32=self._dx1, skip_wgrad=True)
            if has_q2:
                e["q2"]["stack"].backward(self._dq2a, None, None, None, dx32=self._dx2, skip_wgrad=True)
            ops.add_cols(self._dx1[:, S:], self._dx2[:, S:] if has_q2 else None, self._ga)
        ops.gaussian_head_backward(self._ls, self._noise_cur, self._ga, self._glp.reshape(-1), self._dls,
                                   kld_coef=self._kld_coef if self.add_kld_to_loss else None,
                                   kld_on_mean=self.add_kld_to_loss and self.apply_kld_on_mean)
        dls = self._dls if grad_out is None else self._dls * grad_out
        a = e["actor"]
        held = held_gradients(a["slab"], a["params"])
        if getattr(self.actor_network, "use_layer_norm", False):  # through loc_layer_norm / scale_layer_norm first
            an, A = self.actor_network, self._A
            index = {id(p): i for i, p in enumerate(a["params"])}
            view = lambda p: a["slab"].view(a["slab"].grad, index[id(p)])  # noqa: E731
            for h, ln in enumerate((an.loc_layer_norm, an.scale_layer_norm)):
                sl = slice(h * A, (h + 1) * A)
                m, r = self._ln_stats[h]
                ops.layer_norm_backward(dls[:, sl], self._ls_raw[:, sl], m, r, ln.weight.detach(), view(ln.weight),
                                        view(ln.bias), self._ln_ws, dz32=self._dls_raw[:, sl])
            dls = self._dls_raw
        a["stack"].backward(dls, self._xs_t, a["dw"], a["db"], **self._take_tail("actor"))
        self._publish(a, held)

    def _alpha_backward(self, grad_out=None, alias=False):
        ops.sac_alpha_grad(self._parts["ent"], self._B, self.log_alpha.data, self._alpha_grad, self._alpha_loss)
        g = self._alpha_grad if grad_out is None else self._alpha_grad * grad_out.double()
        if self.log_alpha.grad is None:
            # native step (alias): the kernel's output buffer IS the gradient — it is rewritten before its next use — instead
            # of a one-element clone per step (the last torch operator of the C4 step, 5 us)
            self.log_alpha.grad = g if (alias and grad_out is None) else g.clone()
        else:
            self.log_alpha.grad.add_(g)

    # value segment (:325-340): V(s) regressed on min q (s, a_pi) [- alpha * clamp(log_prob)] — the critic head's
    # arithmetic with reward 0, discount 1, not_terminal 1 and (q1a, q2a) in the place of the target critics
    def _value_forward(self, b):
        state = self._state_in(b.state.float_features)
        vs = self._e["value"]["stack"]
        vs.stage_weights(need_transposed=True)
        xv, self._xv_t = vs.stage_input(state, need_transposed=True)
        vs.forward(xv, self._vval, save=True)
        has_q2 = "q2" in self._e
        lp = self._zeros if self.logged_action_uniform_prior else self._lp
        ops.sac_critic_head(self._vval, None, self._q1a, self._q2a if has_q2 else None, lp, self._zeros, self._ones, 1.0,
                            self._alpha(state.device), self._yv, self._dv, None, self._parts["lv"], None)
        ops.reduce_sum(self._parts["lv"], self._parts["lv"].numel(), 1.0 / self._B, self._losses["value"])

    def _value_backward(self, grad_out=None):
        e = self._e["value"]
        dv = self._dv if grad_out is None else self._dv * grad_out
        held = held_gradients(e["slab"], e["params"])
        e["stack"].backward(dv, self._xv_t, e["dw"], e["db"])
        self._publish(e, held)

    def _noise(self, B, A, dev, given):
        if given is not None:
            return given.to(device=dev, dtype=torch.float32).contiguous()
        return torch.randn(B, A, device=dev)

    # ---- reference surface -------------------------------------------------------------------------
    def set_noise(self, noise_next: torch.Tensor, noise_cur: torch.Tensor):
        """Inject the two N(0,1) draws of this step (the reference's torch.randn_like calls in
        actor(next_state) and actor(state), actor.py:217) — parity runs only."""
        self._injected = (noise_next, noise_cur)

    def train_step_gen(self, training_batch: rlt.PolicyNetworkInput, batch_idx: int):
        """IMPORTANT: the input action here is assumed to match the range of the output of the actor."""
        assert hasattr(training_batch, "action") and hasattr(training_batch.action, "float_features")
        b = training_batch
        B, A = b.action.float_features.shape
        dev = b.action.float_features.device
        inj = getattr(self, "_injected", None) or (None, None)
        self._injected = None
        self._critic_forward(b, self._noise(B, A, dev, inj[0]))
        q1 = self._e["q1"]
        yield _SegmentLoss.apply(lambda g: self._critic_backward("q1", g), self._losses["q1"], *q1["params"])
        if self.q2_network:
            q2 = self._e["q2"]
            yield _SegmentLoss.apply(lambda g: self._critic_backward("q2", g), self._losses["q2"], *q2["params"])
        self._actor_forward(b, self._noise(B, A, dev, inj[1]))
        yield _SegmentLoss.apply(self._actor_backward, self._losses["actor"], *self._e["actor"]["params"])
        if self.alpha_optimizer is not None:
            ops.sac_alpha_grad(self._parts["ent"], self._B, self.log_alpha.data, self._alpha_grad, self._alpha_loss)
            yield _SegmentLoss.apply(self._alpha_backward, self._alpha_loss, self.log_alpha)
            self.entropy_temperature = self._alpha(dev)  # = exp(log_alpha), written by AdamF64.step (:322)
        if self.value_network is not None:
            self._value_forward(b)
            yield _SegmentLoss.apply(self._value_backward, self._losses["value"], *self._e["value"]["params"])
        self._log_metrics()
        yield self.soft_update_result()

    def _log_metrics(self):
        """sac_trainer.py:343-380: the step's means handed to `self.logger.log_metrics` (device scalars: no host sync
        here; a TensorBoard logger reads them out itself).  Evaluated only when a logger is attached."""
        if not self.logger:
            return
        m, has_q2 = self._metrics, "q2" in self._e
        min_q = torch.minimum(self._q1a, self._q2a) if has_q2 else self._q1a
        et = self.entropy_temperature
        step = self.all_batches_processed
        out = {"td_loss": self._losses["q1"].reshape(()), "logged_rewards": m["logged_rewards"],
               "model_values_on_logged_actions": m["q1_value"], "q1_value": m["q1_value"],
               "entropy_temperature": et.reshape(()) if isinstance(et, torch.Tensor) else et}
        if self.value_network is not None:  # log_prob_a is the value segment's (:331-336)
            out["log_prob_a"] = (torch.zeros((), device=min_q.device) if self.logged_action_uniform_prior
                                 else self._lp.clamp(LOG_PROB_MIN, LOG_PROB_MAX).mean())
        else:
            out["log_prob_a"] = m["log_prob_a"]
        actor_loss = self._losses["actor"].reshape(())
        if self.add_kld_to_loss:  # the logged actor_loss is the mean before the KLD term is added (:280, :308, :357)
            actor_loss = actor_loss - self.kld_weight * self._kld.reshape(())
        out.update(next_state_value=m["next_state_value"], target_q_value=m["target_q_value"], min_q_actor_value=min_q.mean(),
                   actor_output_log_prob=self._lp.mean(), actor_loss=actor_loss)
        self.logger.log_metrics(out, step=step)
        if has_q2:
            self.logger.log_metrics({"q2_value": m["q2_value"]}, step=step)
        if self.value_network is not None:
            self.logger.log_metrics({"target_state_value": self._yv.mean()}, step=step)
        if self.add_kld_to_loss:  # :366-376
            A, S = self.action_emb_mean.numel(), self._S
            if self.apply_kld_on_mean:
                x = torch.tanh(self._ls[:, :A]).clamp(-1.0 + 1e-6, 1.0 - 1e-6)
            else:
                x = self._api if self._panels else self._xa[:, S:]
            self.logger.log_metrics({"action_batch_mean": x.mean(0).mean(), "action_batch_var": x.var(0).mean(),
                                     "kld": self._kld.reshape(())}, step=step)

    # ---- fused native step ---------------------------------------------------------------------------
    def _fused_updates(self, opts):
        """{network: engine.FusedUpdate} when every network of the plain step (twin critics + actor on the fused bf16
        kernels, one Adam per network, targets = the critics' copies) qualifies, else None.  All or nothing: the soft
        update of the targets rides in the critics' launches, so a step never mixes the two forms."""
        plan = getattr(self, "_fused_plan", None)
        if plan is None:
            from ..engine import FusedUpdate

            plan = False
            e = self._e
            if (self.value_network is None and "q2" in e and self.q1_network_target is not None and len(opts) >= 4
                    and self.use_fused_update):
                t = {"q1": self.q1_network_target, "q2": self.q2_network_target}
                made = {}
                for k, adam in (("q1", opts[0]), ("q2", opts[1]), ("actor", opts[2])):
                    net = dict(q1=self.q1_network, q2=self.q2_network, actor=self.actor_network)[k]
                    tgt = t.get(k)
                    if not hasattr(adam, "moments_for"):
                        made = None
                        break
                    fu = FusedUpdate.make(adam, e[k]["params"], net.fc.linears(), e[k]["stack"],
                                          target_params=list(tgt.parameters()) if tgt is not None else None,
                                          target_stack=self._t[k] if tgt is not None else None, tau=self.tau)
                    if fu is None:
                        made = None
                        break
                    made[k] = fu
                if made:
                    plan = made
            self._fused_plan = plan
        if plan is False:
            return None
        # (first step: fragments not staged yet -> the separate launches, for every network of the step)
        return plan if all(fu.staged() for fu in plan.values()) else None

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
    def train_step_native(self, training_batch, noise_next=None, noise_cur=None):
        """All five segments with no autograd graph / generator / host sync.  Returns the dict of
        device-resident loss scalars."""
        opts = self.native_optimizers()
        b = training_batch
        B, A = b.action.float_features.shape
        dev = b.action.float_features.device
        gs = 1.0 / self._dp_world
        it = iter(opts)
        self._fold_losses, self._tails = True, {}
        try:
            return self._native_segments(b, B, A, dev, gs, it, opts, noise_next, noise_cur)
        finally:
            self._fold_losses = False
            for part, scale, out in self._tails.values():  # (a segment whose backward did not take its tail)
                ops.reduce_sum(part, part.numel(), scale, out)
            self._tails = {}

    def _native_segments(self, b, B, A, dev, gs, it, opts, noise_next, noise_cur):
        self._critic_forward(b, self._noise(B, A, dev, noise_next))
        fused = self._fused_updates(opts)  # Adam (+ soft update) + re-staging per network in one launch, or None
        dp = self._dp_group is not None
        bucket = self._dp_bucket() if dp else None
        critics = [k for k in ("q1", "q2") if k in self._e]
        if bucket is not None:
            # data parallel, twin critics: both backward passes first (q2's reads nothing q1's update writes), ONE all-reduce
            # of the two slabs, then the two updates in the reference's order
            for k in critics:
                for p in self._e[k]["params"]:
                    p.grad = None
                self._critic_backward(k, reduce=False)
            with ops.profile_span("all_reduce", dict(bytes=bucket.numel() * 4, world=self._dp_world)):
                torch.distributed.all_reduce(bucket, group=self._dp_group)
        for k in critics:
            if bucket is None:
                for p in self._e[k]["params"]:
                    p.grad = None
                self._critic_backward(k)
            o = next(it)
            if fused is not None:
                fused[k].step(gs)
            else:
                o.grad_scale = gs
                o.step()
        self._actor_forward(b, self._noise(B, A, dev, noise_cur))
        alpha_reduce = None
        if self.alpha_optimizer is not None and dp:
            # the temperature's gradient needs the actor FORWARD only (mean of log_prob + target entropy): under data
            # parallelism it is taken here and summed over the ranks by the actor's own all-reduce, as one more float behind
            # the actor's gradient slab (round 6; rounds 4-5: an asynchronous 8-byte all-reduce of its own, a third collective)
            self.log_alpha.grad = None
            self._alpha_backward(alias=True)
            alpha_reduce = self._dp_actor_bucket()
            alpha_reduce[-1:].copy_(self.log_alpha.grad)
        for p in self._e["actor"]["params"]:
            p.grad = None
        self._alpha_rider = alpha_reduce
        try:
            self._actor_backward()
        finally:
            self._alpha_rider = None
        o = next(it)
        alpha_opt = next(it) if self.alpha_optimizer is not None else None

        def actor_and_temperature():
            if fused is not None:
                fused["actor"].step(gs)
            else:
                o.grad_scale = gs
                o.step()
            if alpha_opt is not None:
                if alpha_reduce is not None:
                    self.log_alpha.grad.copy_(alpha_reduce[-1:])  # the sum over the ranks (float64 again for its Adam step)
                    self.log_alpha.grad.mul_(gs)
                else:
                    self.log_alpha.grad = None
                    self._alpha_backward(alias=True)
                alpha_opt.step()
                # the aliased gradient IS the kernel's output buffer: never leave it attached — a later generator-path backward
                # with grads kept (zero_grad(set_to_none=False)) would add the buffer to itself after the kernel rewrote it
                self.log_alpha.grad = None
                self.entropy_temperature = self._alpha(dev)

        actor_and_temperature()
        if self.value_network is not None:
            self._value_forward(b)
            for p in self._e["value"]["params"]:
                p.grad = None
            self._value_backward()
            o = next(it)
            o.grad_scale = gs
            o.step()
        soft = next(it)
        if fused is None:
            soft.step()  # (the fused critic updates already moved their targets)
        self.all_batches_processed += 1
        out = dict(q1_loss=self._losses["q1"], q2_loss=self._losses["q2"], actor_loss=self._losses["actor"],
                   alpha_loss=self._alpha_loss)
        if self.value_network is not None:
            out["value_loss"] = self._losses["value"]
        return out
