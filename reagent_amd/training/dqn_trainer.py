"""DQNTrainer with the constructor / generator / attribute surface of
reagent/training/dqn_trainer.py:27-379, executing the step on the HIP kernels.

One training step (reference :241-304, algorithmic form of SURVEY.md §8d: 3 forwards + 1 backward):
    q'_online = q_network(next_state)            rg_fc_forward x L        (no saved activations)
    q'_target = q_network_target(next_state)     rg_fc_forward x L
    q         = q_network(state)                 rg_fc_forward x L        (saves transposed acts)
    loss, dq  = TD head                          rg_dqn_head + rg_reduce_sum
    grads     = backward                         rg_fc_wgrad / rg_fc_dgrad x L
    Adam, soft update                            rg_adam_step, rg_soft_update (via the optimizers)
The CPE-only 4th forward of the reference (:268) is dead when CPE is off and is not executed.
"""
import logging
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import os

import torch

from .. import _lib as L
from .. import ops
from ..core import types as rlt
from ..core.parameters import EvaluationParameters, RLParameters
from ..engine import ensure_slab, grad_views
from ..optimizer import Optimizer__Union, SoftUpdate
from .dqn_trainer_base import DQNTrainerBaseLightning

logger = logging.getLogger(__name__)


@dataclass(frozen=True)
class BCQConfig:
    drop_threshold: float = 0.1


class _HipLoss(torch.autograd.Function):
    """Scalar loss whose backward runs the HIP backward pass and writes ``.grad`` in place."""

    @staticmethod
    def forward(ctx, owner, loss_buf, *params):
        ctx.owner = owner
        return loss_buf.detach().clone().reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        ctx.owner._hip_backward(grad_out)
        return (None, None) + (None,) * len(ctx.owner._hip_params)


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


def dp_reduce(tr, slab):
    """Sum the gradient slab over the data-parallel group (RCCL).  Inside a native step the 1/world factor
    is folded into the Adam launch (grad_scale); on the generator / Lightning path the optimizers are
    stepped by the caller, so the sum is turned into the mean here and Adam, weight decay and any gradient
    clipping see what a single rank would on the concatenated batch."""
    torch.distributed.all_reduce(slab.grad, group=tr._dp_group)
    if not getattr(tr, "_native_active", False):
        slab.grad.mul_(1.0 / tr._dp_world)


class _NativeStep:
    """marks the trainer as inside a native step (see dp_reduce)"""

    def __init__(self, tr):
        self.tr = tr

    def __enter__(self):
        self.prev = getattr(self.tr, "_native_active", False)
        self.tr._native_active = True

    def __exit__(self, *exc):
        self.tr._native_active = self.prev


def held_gradients(slab, params):
    """A backward without a preceding zero_grad() accumulates in PyTorch, but the HIP backward OVERWRITES
    the gradient slab.  Returns copies of the gradients still published through p.grad (aliases of the
    slab) so that `publish_gradients` can add them back; empty when the gradients were cleared
    (`zero_grad(set_to_none=True)`, every native step)."""
    base = slab.grad.data_ptr()
    return [(i, slab.view(slab.grad, i).clone()) for i, p in enumerate(params)
            if p.grad is not None and p.grad.data_ptr() == base + 4 * slab.offsets[i]]


def publish_gradients(slab, params, held=()):
    """p.grad aliases the slab the backward wrote (gradients held over a missing zero_grad() are added
    back; foreign .grad tensors are accumulated into)"""
    for i, g in held:
        slab.view(slab.grad, i).add_(g)
    base = slab.grad.data_ptr()
    for i, p in enumerate(params):
        gv = slab.view(slab.grad, i)
        if p.grad is None or p.grad.data_ptr() == base + 4 * slab.offsets[i]:
            p.grad = gv
        else:
            p.grad.add_(gv)


def enable_graph_mode(tr):
    """Switch every Adam of the trainer's native step to device-scheduled stepping (optimizer.AdamSchedule): what
    a HIP-graph capture of the step needs.  Eager steps keep working (and produce the same bits)."""
    if any(isinstance(m, torch.nn.Dropout) and m.p > 0.0 for m in tr.modules()):
        # rg_dropout's Philox offset is a host-side launch argument: a replayed graph would repeat one mask forever
        raise NotImplementedError("networks with dropout layers are not captured into a HIP graph: run the native step eagerly")
    for o in tr.native_optimizers():
        if type(o).__module__.startswith("torch.optim"):
            # torch's own optimizers (Optimizer__Union's other members) count their steps on the host
            raise NotImplementedError(f"{type(o).__name__}: only Adam steps are captured into a HIP graph; run the native step eagerly")
    for o in tr.native_optimizers():
        if hasattr(o, "enable_device_schedule"):
            o.enable_device_schedule()
    tr._graph_mode = True


def require_grad_scaling_optimizers(tr):
    """data parallel on the trainers that fold 1/world into their Adam launches (SAC, TD3, discrete CRR): one of torch's own
    optimizers (Optimizer__Union's other members) has no such argument — refuse rather than step on summed gradients"""
    if getattr(tr, "_dp_world", 1) == 1:
        return
    for o in tr.native_optimizers():
        if type(o).__module__.startswith("torch.optim"):  # (this package's own optimizer classes all scale in their launches)
            raise NotImplementedError(f"{type(tr).__name__}: data parallel needs Adam optimizers (got torch.optim.{type(o).__name__})")


def disable_graph_mode(tr):
    """back to scalar-argument Adam launches (host step counters brought up to date first)"""
    for o in tr.native_optimizers():
        if hasattr(o, "disable_device_schedule"):
            o.disable_device_schedule()
    tr._graph_mode = False


def note_graph_replays(tr, n: int):
    """host-side bookkeeping for n steps that ran as graph replays (no Python in between)"""
    if n <= 0:
        return
    for o in tr.native_optimizers():
        if hasattr(o, "note_device_steps"):
            o.note_device_steps(n)
        for g in o.param_groups:  # the compute-type weight copies of an eager call after the replays are re-staged
            for p in g["params"]:
                p._rg_version = getattr(p, "_rg_version", 0) + 1
    tr.all_batches_processed += n


def native_step(fn):
    """decorator for train_step_native of the trainers: the whole call runs as a native step"""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        # (ops.deferred_ticks: the schedule ticks of the step's device-scheduled updates leave as ONE launch at its end)
        with _NativeStep(self), ops.deferred_ticks():
            return fn(self, *a, **k)

    return wrapper


class _CpeEngine:
    """The CPE part of the DQN step (reagent/training/dqn_trainer_base.py:338-452, `_calculate_cpes`):
    reward network and CPE q-network on `state`, CPE target network on `next_state`, the 4th forward
    `q_network(next_state)` with the weights the q step just produced, all losses and output
    gradients in one rg_cpe_head launch, then the two FC backward passes."""

    def __init__(self, trainer):
        self.tr = trainer
        self._ws_batch = -1

    @staticmethod
    def _net_engine(net):
        params = list(net.parameters())
        slab = ensure_slab(params)
        dw, db = grad_views(net.fc, slab, params)
        return dict(params=params, slab=slab, stack=net.fc.stack(), dw=dw, db=db)

    def _engine(self, B, dev):
        tr = self.tr
        self.e = dict(reward=self._net_engine(tr.reward_network), cpe=self._net_engine(tr.q_network_cpe))
        self.t = tr.q_network_cpe_target.fc.stack()
        if self._ws_batch != B or self.reward_est.device != dev:
            A, M = tr.num_actions, len(tr.metrics_to_score)
            f32 = dict(dtype=torch.float32, device=dev)
            self.reward_est, self.q_cpe, self.q_cpe_tgt, self.d_reward, self.d_cpe = (
                torch.empty(B, M * A, **f32) for _ in range(5))
            self.next_scores = torch.empty(B, A, **f32)
            P = ops.dqn_head_partials(B)
            self.parts = dict(reward=torch.empty(P, **f32), cpe=torch.empty(P, **f32))
            self.losses = dict(reward=torch.empty(1, **f32), cpe=torch.empty(1, **f32))
            self._ws_batch = B

    def forward(self, b, need_propensities: bool = False):
        tr = self.tr
        state, next_state = tr._net_in(b.state.float_features), tr._net_in(b.next_state.float_features)
        B, dev = state.shape[0], state.device
        self._engine(B, dev)
        A, M = tr.num_actions, len(tr.metrics_to_score)
        # all_next_action_scores: q_network(next_state) AFTER the q-network step for DQN / QR-DQN
        # (dqn_trainer.py:268), the target critic's next-state values for CRR
        tr._cpe_next_action_scores(next_state, self.next_scores)
        for k in ("reward", "cpe"):
            self.e[k]["stack"].stage_weights(need_transposed=True)
        self.t.stage_weights(need_transposed=False)
        rs, cs = self.e["reward"]["stack"], self.e["cpe"]["stack"]
        xs_r, self._xs_t_r = rs.stage_input(state, need_transposed=True)
        rs.forward(xs_r, self.reward_est, save=True)
        xs_c, self._xs_t_c = cs.stage_input(state, need_transposed=True)
        cs.forward(xs_c, self.q_cpe, save=True)
        xn_t, _ = self.t.stage_input(next_state, need_transposed=False)
        self.t.forward(xn_t, self.q_cpe_tgt, save=False)
        extras = getattr(b, "extras", None)
        metrics = getattr(extras, "metrics", None) if extras is not None else None
        if M > 1:
            assert metrics is not None and metrics.shape == (B, M - 1), "extras.metrics must hold the extra CPE metrics"
            metrics = tr._f32c(metrics)
        else:
            metrics = None
        gamma_exp = tr._cpe_gamma_exponent(b)
        next_mask = tr._f32c(b.possible_next_actions_mask if tr.maxq_learning else b.next_action)
        self.propensities = torch.empty(B, A, dtype=torch.float32, device=dev) if need_propensities else None
        ops.cpe_head(self.reward_est, self.q_cpe, self.q_cpe_tgt, self.next_scores, next_mask, tr._f32c(b.action),
                     tr._f32c(b.reward).reshape(-1), metrics, tr._f32c(b.not_terminal).reshape(-1), tr.gamma,
                     gamma_exp, tr.rl_temperature, M, tr._loss_type, self.d_reward, self.d_cpe,
                     self.parts["reward"], self.parts["cpe"], self.propensities)
        for k in ("reward", "cpe"):
            ops.reduce_sum(self.parts[k], self.parts[k].numel(), 1.0 / (B * M), self.losses[k])

    def backward(self, which, grad_out=None):
        e = self.e[which]
        d = self.d_reward if which == "reward" else self.d_cpe
        if grad_out is not None:
            d = d * grad_out
        held = held_gradients(e["slab"], e["params"])
        e["stack"].backward(d, self._xs_t_r if which == "reward" else self._xs_t_c, e["dw"], e["db"])
        slab = e["slab"]
        if self.tr._dp_group is not None:
            dp_reduce(self.tr, slab)
        publish_gradients(slab, e["params"], held)

    def loss(self, which):
        e = self.e[which]
        return _SegmentLoss.apply(lambda g: self.backward(which, g), self.losses[which], *e["params"])


class QStepCore(DQNTrainerBaseLightning):
    """Engine shared by DQNTrainer and QRDQNTrainer: flat parameter slab, FC stacks, the
    3-forward / head / backward sequence, the autograd bridge and the fused native step.
    Subclasses provide the head (``_alloc_head`` / ``_run_head``) and ``_out_cols``."""

    _ws_batch = -1
    _dp_group = None
    _dp_world = 1

    def _out_cols(self) -> int:
        return self.num_actions

    # ---- optimizers (dqn_trainer.py:119-155 / qrdqn_trainer.py:81-106) ------------------------
    def configure_optimizers(self):
        """[q_network, (reward_network, q_network_cpe,) soft update of the target(s)]"""
        optimizers = []
        target_params = list(self.q_network_target.parameters())
        source_params = list(self.q_network.parameters())
        optimizers.append(self.q_network_optimizer.make_optimizer_scheduler(self.q_network.parameters()))
        if self.calc_cpe_in_training:
            cpe_target_params, cpe_source_params, cpe_optimizers = self._configure_cpe_optimizers()
            target_params += cpe_target_params
            source_params += cpe_source_params
            optimizers += cpe_optimizers
        optimizers.append(SoftUpdate.make_optimizer_scheduler(target_params, source_params, tau=self.tau))
        return optimizers

    # ---- CPE (dqn_trainer_base.py:338-452) ----------------------------------------------------
    _cpe = None

    def _cpe_gamma_exponent(self, b):
        """exponent of gamma in the CPE discount tensor (dqn_trainer.py:240-254), None = 1"""
        gamma_exp = None
        if self.use_seq_num_diff_as_time_diff:
            gamma_exp = self._f32c(b.time_diff).reshape(-1)
        if self.multi_steps is not None:
            gamma_exp = self._f32c(b.step).reshape(-1)
        return gamma_exp

    def _cpe_next_action_scores(self, next_state, out):
        qs = self._qs
        qs.stage_weights(need_transposed=True)
        xn, _ = qs.stage_input(next_state, need_transposed=False)
        self._cpe_next_scores(xn, out)

    def _cpe_next_scores(self, xn, out):
        """all_next_action_scores = q_network(next_state) with the just-updated weights -> out [B, A]"""
        self._qs.forward(xn, out, save=False)

    def _cpe_scores_for_logging(self):
        return self.all_action_scores

    def _cpe_segment(self, training_batch):
        """the two CPE losses of train_step_gen, yielded after the q-network loss: by then the caller
        (Lightning's optimizer loop) has stepped the q-network, and the CPE targets use
        q_network(next_state) with the NEW weights (dqn_trainer.py:267-281, qrdqn_trainer.py:161-177)"""
        if self._cpe is None:
            self._post_step_stats_forward(training_batch)
            return
        from .reagent_lightning_module import _NoOpReporter

        self._cpe.forward(training_batch)
        reward_loss = self._cpe.loss("reward")
        yield reward_loss
        if not isinstance(self._reporter, _NoOpReporter):  # dqn_trainer_base.py:430-450
            from ..core.torch_utils import masked_softmax

            mask = training_batch.possible_actions_mask if self.maxq_learning else training_batch.action
            self.reporter.log(reward_loss=reward_loss.detach(),
                              model_propensities=masked_softmax(self._cpe_scores_for_logging(), mask.float(),
                                                                self.rl_temperature),
                              model_rewards=self._cpe.reward_est[:, : self.num_actions])
        yield self._cpe.loss("cpe")

    # ---- engine -------------------------------------------------------------------------------
    def _engine(self, batch: int, device):
        """Slabs, FC stacks and step buffers (re)built lazily for the current batch size/device."""
        self._hip_params = list(self.q_network.parameters())
        self._slab = ensure_slab(self._hip_params)
        self._qs = self.q_network.fc.stack()
        self._ts = self.q_network_target.fc.stack()
        if self._ws_batch != batch or self._q.device != device:
            n = self._out_cols()
            f32 = dict(dtype=torch.float32, device=device)
            self._q = torch.empty(batch, n, **f32)
            self._qn_online = torch.empty(batch, n, **f32)
            self._qn_target = torch.empty(batch, n, **f32)
            self._dq = torch.empty(batch, n, **f32)
            self._loss = torch.empty(1, **f32)
            self._alloc_head(batch, device)
            self._ws_batch = batch
        # weight/bias gradient destinations = views of the flat gradient slab, in layer order
        self._dw, self._db = grad_views(self.q_network.fc, self._slab, self._hip_params)

    @staticmethod
    def _f32c(t: torch.Tensor) -> torch.Tensor:
        t = t if t.dtype == torch.float32 else t.float()
        return t if t.is_contiguous() else t.contiguous()

    def _bcq_mask(self, next_mask, next_state):
        """possible_next_actions_mask * get_valid_actions_from_imitator(imitator, next_state, threshold)
        (dqn_trainer.py:209-215) on a copy — the batch's own mask is left alone"""
        st = self.bcq_imitator.stack()
        st.stage_weights(need_transposed=False)
        xi, _ = st.stage_input(next_state, need_transposed=False)
        logits = torch.empty(next_state.shape[0], self.num_actions, dtype=torch.float32, device=next_state.device)
        st.forward(xi, logits, save=False)
        mask = next_mask.clone()
        ops.bcq_filter(logits, self.bcq_drop_threshold, mask)
        return mask

    def _needs_online_next(self) -> bool:
        return True

    # the reference's train_step_gen evaluates q_network(next_state) again after the optimizer step (DQN, QR-DQN: yes;
    # C51: no)
    _post_step_forward = True

    def _q_has_batch_norm(self) -> bool:
        if getattr(self, "_q_bn", None) is None:
            self._q_bn = self._post_step_forward and any(isinstance(m, torch.nn.BatchNorm1d) for m in self.q_network.modules())
        return self._q_bn

    def _post_step_stats_forward(self, training_batch):
        """dqn_trainer.py:267-268 / qrdqn_trainer.py:161-163 evaluate q_network(next_state) once more after the optimizer
        step; the values feed only the CPE heads, but batch-norm layers in training mode move their running statistics
        in that forward too — so a batch-normed network runs it (anything else skips the dead forward)"""
        if not self._q_has_batch_norm() or not self.q_network.training:
            return
        qs = self._qs
        qs.stage_weights(need_transposed=True)
        xn, _ = qs.stage_input(self._net_in(training_batch.next_state.float_features), need_transposed=False)
        qs.forward(xn, self._qn_online, save=False)

    @staticmethod
    def _net_in(t: torch.Tensor) -> torch.Tensor:
        """network input: fp32, or bf16 (the normalize-on-gather output of the bf16 path)"""
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float()
        return t if t.stride(-1) == 1 and t.is_contiguous() else t.contiguous()

    def _hip_forward(self, b) -> torch.Tensor:
        state, next_state = self._net_in(b.state.float_features), self._net_in(b.next_state.float_features)
        L.require_cuda(state, "training_batch.state")
        B, dev = state.shape[0], state.device
        self._engine(B, dev)
        qs, ts = self._qs, self._ts
        qs.stage_weights(need_transposed=True)
        ts.stage_weights(need_transposed=False)
        xs, self._xs_t = qs.stage_input(state, need_transposed=True)
        xn, _ = qs.stage_input(next_state, need_transposed=False)
        if self._needs_online_next():
            qs.forward(xn, self._qn_online, save=False)
        ts.forward(xn, self._qn_target, save=False)
        qs.forward(xs, self._q, save=True)
        gamma_exp = None
        if self.use_seq_num_diff_as_time_diff:
            assert self.multi_steps is None
            gamma_exp = self._f32c(b.time_diff).reshape(-1)
        if self.multi_steps is not None:
            assert b.step is not None
            gamma_exp = self._f32c(b.step).reshape(-1)
        boosts = self.reward_boosts.reshape(-1).to(dev) if self._has_reward_boost else None
        if self.maxq_learning:
            next_mask = self._f32c(b.possible_next_actions_mask)
            if getattr(self, "bcq", False):
                next_mask = self._bcq_mask(next_mask, next_state)
        else:  # SARSA: the taken next action is the only "possible" one (dqn_trainer.py:218-224)
            next_mask = self._f32c(b.next_action)
        self._run_head(b, B, self._f32c(b.action), next_mask, boosts, gamma_exp)
        return self._loss

    def _hip_backward(self, grad_out=None, async_reduce: bool = False):
        if grad_out is not None:
            self._dq.mul_(grad_out)
        held = held_gradients(self._slab, self._hip_params)
        self._qs.backward(self._dq, self._xs_t, self._dw, self._db, **self._take_loss_tail())
        if self._dp_group is not None:
            if async_reduce and ops.profiling():
                # instrumented pass (bench.py): the collective joined at once, between two events on the compute stream
                # (it runs on RCCL's stream; the compute stream waits for it), so its duration is visible per rank
                with ops.profile_span("all_reduce", dict(bytes=self._slab.grad.numel() * 4, world=self._dp_world)):
                    dp_reduce(self, self._slab)
            elif async_reduce:  # runs on the collective's own stream; joined by apply_pending_update()
                self._pending_reduce = torch.distributed.all_reduce(self._slab.grad, group=self._dp_group,
                                                                   async_op=True)
            else:
                dp_reduce(self, self._slab)
        publish_gradients(self._slab, self._hip_params, held)

    # the step's mean loss evaluated by the weight gradient's reduce launch instead of its own (native steps on the
    # fused stacks only: there the loss is not read before the backward has been enqueued)
    _loss_tail_wanted = False
    _loss_tail = None

    _loss_side_event = None  # recorded after a loss sum that ran on the engine's side stream (qr_engine.py)

    def _join_loss_side(self):
        """the current stream waits for a loss sum enqueued on the side stream — whether or not the backward that
        normally joins that stream ran"""
        ev, self._loss_side_event = self._loss_side_event, None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _take_loss_tail(self) -> dict:
        tail, self._loss_tail = self._loss_tail, None
        return {"tail_sum": tail} if tail is not None else {}

    # ---- data parallel -------------------------------------------------------
    def enable_data_parallel(self, process_group=None):
        """All-reduce(sum) the flat fp32 gradient slab over RCCL after every backward; the 1/world
        factor is folded into the Adam kernel (FusedAdam.grad_scale)."""
        import torch.distributed as dist

        self._dp_group = process_group if process_group is not None else dist.group.WORLD
        self._dp_world = dist.get_world_size(self._dp_group)
        return self

    def _hip_loss(self, batch):
        self.apply_pending_update()  # a deferred native update must land before the next forward
        loss_buf = self._hip_forward(batch)
        return _HipLoss.apply(self, loss_buf, *self._hip_params)

    # ---- fused native step (what bench.py and the native loop drive) --------------------------
    def native_optimizers(self):
        if getattr(self, "_native_opts", None) is None:
            made = self.configure_optimizers()
            self._native_opts = [o["optimizer"] for o in made]
            # lr schedulers of the optimizer configs (None where there is none): with Lightning its loop
            # steps them per epoch; a caller of the native loop does `for s in native_schedulers(): s.step()`
            self._native_scheds = [o.get("lr_scheduler") for o in made]
        return self._native_opts

    def native_schedulers(self):
        self.native_optimizers()
        return [s for s in self._native_scheds if s is not None]

    @torch.no_grad()
    def train_step_native(self, training_batch, defer_update: bool = False) -> torch.Tensor:
        """forward + head + backward + Adam + soft update with no autograd graph, no generator and
        no host synchronisation.  Returns the device-resident loss scalar (shape [1]).

        defer_update (data parallel only): the gradient all-reduce is launched asynchronously and
        Adam + soft update are left pending until `apply_pending_update()` — which the next
        `train_step_native` calls first.  The step sequence is unchanged (update k always precedes
        forward k+1); what the caller gains is that whatever it enqueues between two steps (the next
        batch's replay gather) runs under the all-reduce instead of after it."""
        self.apply_pending_update()
        self._loss_tail_wanted = True  # the loss mean may ride in the weight gradient's reduce launch (_run_head decides)
        try:
            loss = self._hip_forward(training_batch)
        finally:
            self._loss_tail_wanted = False
        for p in self._hip_params:
            p.grad = None
        deferred = defer_update and self._dp_group is not None
        try:
            with _NativeStep(self):
                self._hip_backward(None, async_reduce=deferred)
        finally:
            self._join_loss_side()
        self._update_pending = True
        self._pending_batch = (training_batch if getattr(self, "_cpe", None) is not None or self._q_has_batch_norm()
                               else None)
        if not deferred:
            self.apply_pending_update()
        return loss

    # ---- the native step in two halves (data-parallel HIP-graph replay, runtime._GraphedLoop) -------------
    @torch.no_grad()
    def native_forward_backward(self, training_batch) -> torch.Tensor:
        """forwards + head + backward + wgrad into the gradient slab; no collective, no update"""
        if getattr(self, "_cpe", None) is not None:
            raise NotImplementedError("the two-halves form of the native step does not cover the CPE heads")
        if self._q_has_batch_norm():
            # the reference evaluates q_network(next_state) once more after the optimizer step (dqn_trainer.py:267-268),
            # which moves a batch-normed network's running statistics; the two-halves form has no such forward
            raise NotImplementedError("the two-halves form of the native step does not cover batch-normed Q-networks")
        self._loss_tail_wanted = True
        try:
            loss = self._hip_forward(training_batch)
        finally:
            self._loss_tail_wanted = False
        for p in self._hip_params:
            p.grad = None
        try:
            with _NativeStep(self):
                self._qs.backward(self._dq, self._xs_t, self._dw, self._db, **self._take_loss_tail())
                publish_gradients(self._slab, self._hip_params)
        finally:
            self._join_loss_side()
        return loss

    @torch.no_grad()
    def native_update(self):
        """Adam (1/world folded in) + soft update + re-staging on the gradient slab as it stands"""
        self._update_pending = True
        self._pending_reduce = None
        with _NativeStep(self):
            self._apply_pending_update()

    _update_pending = False
    _pending_reduce = None
    _fused_plan = None
    _graph_tick = None

    def _fused_update(self, adam, soft) -> bool:
        """Adam + soft update + bf16 re-staging of both networks in ONE launch (rg_mlp_update_fused)
        when the step has the plain shape: both stacks on the fused kernels, one Adam group over the
        q-network's slab, a soft update pairing exactly the target's parameters with it.  Arithmetic per
        element is that of the separate launches; returns False (nothing done) otherwise."""
        import math

        from ..engine import FusedMLP, ensure_slab
        from ..qr_engine import GroupedQR

        qs, ts = self._qs, self._ts
        # QR-DQN's grouped engine (qr_engine.py): trunk stacks + the wide layer as a grouped last layer of the same launch
        grouped = qs if isinstance(qs, GroupedQR) else None
        if grouped is not None:
            qs, ts = grouped.online.st, grouped.target.st
            if self._fused_plan is None and os.environ.get("RG_QR_FUSED_UPDATE", "1") == "0":  # same-box A/B switch
                self._fused_plan = False
        plan = self._fused_plan
        if plan is None:
            from ..optimizer import FusedAdam

            ok = (isinstance(adam, FusedAdam) and isinstance(qs, FusedMLP) and isinstance(ts, FusedMLP) and qs.x3 == ts.x3
                  and len(adam.param_groups) == 1
                  and len(soft.param_groups) == 1)
            if ok:
                sp = soft.param_groups[0]["params"]
                n = len(sp) // 2
                tgt, src = sp[:n], sp[n:]
                ok = (len(src) == len(self._hip_params) and all(a is b for a, b in zip(src, self._hip_params))
                      and all(a is b for a, b in zip(adam.param_groups[0]["params"], self._hip_params))
                      and all(t is not s_ for t, s_ in zip(tgt, src)))
            if ok:
                tslab = ensure_slab(tgt)
                ok = tslab.offsets == self._slab.offsets and tslab.total == self._slab.total
            if not ok:
                self._fused_plan = plan = False
            else:
                index = {id(p): i for i, p in enumerate(self._hip_params)}
                lin = self.q_network.fc.linears()
                d = L.MlpUpdateDesc()
                d.n_layers = len(lin)
                d.x3 = int(qs.x3)  # split-bf16: both planes of every fragment set are re-staged
                for i, v in enumerate(qs.dims):
                    d.dims[i] = v
                if grouped is not None:  # the trunk stack ends in the per-action mean layer; the parameters end in the wide one
                    d.dims[len(lin)] = lin[-1].weight.shape[0]
                    d.group_rows[len(lin) - 1] = grouped.N
                for l, layer in enumerate(lin):
                    d.w_off[l] = self._slab.offsets[index[id(layer.weight)]]
                    d.b_off[l] = self._slab.offsets[index[id(layer.bias)]]
                self._fused_plan = plan = dict(desc=d, tgt=tgt, tslab=tslab)
        if plan is False:
            return False
        for p in self._hip_params:  # needs dense gradients in the slab (what the HIP backward wrote)
            if p.grad is None:
                return False
        if any(w is None for w in qs._wf) or any(w is None for w in qs._wb) or any(w is None for w in ts._wf):
            return False  # fragments not staged yet (their padding is written by the first staging)
        if grouped is not None and (grouped.online._staged is None or grouped.target._staged is None):
            return False  # the grouped head's fragments likewise
        (slab, exp_avg, exp_avg_sq), tslab, d = adam.moments_for(0), plan["tslab"], plan["desc"]
        if slab is not self._slab or not tslab.is_bound():
            return False
        group = adam.param_groups[0]
        beta1, beta2 = group["betas"]
        sched = adam.schedule_for(0)
        if sched is None:
            steps = {adam.advance(0, i) for i in range(len(slab.params))}
            if len(steps) != 1:
                raise RuntimeError("fused update needs every parameter at the same Adam step")
            step = steps.pop()
        d.param, d.grad = slab.data.data_ptr(), slab.grad.data_ptr()
        d.exp_avg, d.exp_avg_sq = exp_avg.data_ptr(), exp_avg_sq.data_ptr()
        d.target = tslab.data.data_ptr()
        for l in range(d.n_layers):
            d.wfrag_fwd[l], d.wfrag_bwd[l] = qs._wf[l].data_ptr(), qs._wb[l].data_ptr()
            d.target_wfrag_fwd[l] = ts._wf[l].data_ptr()
        if grouped is not None:  # last parameter layer = the wide layer: per-action fragment blocks of the grouped heads
            l = d.n_layers - 1
            d.wfrag_fwd[l], d.wfrag_bwd[l] = grouped.online.gh.wf.data_ptr(), grouped.online.gh.wb.data_ptr()
            d.target_wfrag_fwd[l] = grouped.target.gh.wf.data_ptr()
        tau = soft.param_groups[0]["tau"]
        gt = self._graph_tick  # a step being captured for replay with a device-side index cursor (runtime._GraphedLoop)
        d.sched_pre_ticked, d.post_tick_mod, d.post_tick = 0, 0, None
        if gt is not None and sched is None:
            raise RuntimeError("a replayed step counts its Adam steps on the device: enable_graph_mode first")
        if sched is not None:  # graph-safe: lr and the bias corrections come from HBM, the step is counted there
            from ..optimizer import capturing

            if not capturing():
                sched.set_lr(group["lr"])
                sched.pending += 1
            if gt is not None:  # the step's sampler launch has counted it; this launch advances the index cursor
                assert gt["sched"] is sched.buf
                d.sched_pre_ticked, d.post_tick_mod, d.post_tick = 1, int(gt["mod"]), gt["cursor"].data_ptr()
            ops.tick_fence(sched.buf)
            ops._run("rg_mlp_update_fused", dict(P=slab.total),
                     lambda: L.lib().rg_mlp_update_fused_sched(d, beta1, beta2, group["eps"], group["weight_decay"],
                                                               1.0 / self._dp_world, tau, sched.buf.data_ptr(),
                                                               L.stream_ptr()))
            if gt is None:
                ops.sched_tick(sched.buf)
            else:
                gt["used"] = True
        else:
            ops._run("rg_mlp_update_fused", dict(P=slab.total),
                     lambda: L.lib().rg_mlp_update_fused(d, group["lr"], beta1, beta2, group["eps"], group["weight_decay"],
                                                         1.0 - beta1**step, math.sqrt(1.0 - beta2**step),
                                                         1.0 / self._dp_world, tau, L.stream_ptr()))
        from ..optimizer import _bump

        _bump(slab.params)
        _bump(plan["tgt"])
        # both networks' fragments are current for the bumped versions: no separate staging launch
        for st_, need_t in ((qs, True), (ts, False)):
            st_._staged_versions = tuple((w._version, getattr(w, "_rg_version", 0)) for w in st_.weights) + (need_t,)
            st_._wsrc_ptrs = [w.data_ptr() for w in st_.weights]
        if grouped is not None:
            grouped.after_fused_update()  # the per-action mean layer(s) of the updated wide layer
        return True

    @torch.no_grad()
    def apply_pending_update(self):
        """Adam + soft update of the last backward, after joining its gradient all-reduce."""
        if not self._update_pending:
            return
        with _NativeStep(self):
            self._apply_pending_update()

    def _step_optimizer(self, opt):
        """FusedAdam folds the data-parallel 1/world into its launch; one of torch's own optimizers (Optimizer__Union's other
        members) gets the gradients scaled first — the all-reduce summed them"""
        from ..optimizer import FusedAdam

        if isinstance(opt, FusedAdam):
            opt.grad_scale = 1.0 / self._dp_world
        elif self._dp_world != 1:
            for g in opt.param_groups:
                for p in g["params"]:
                    if p.grad is not None:
                        p.grad.mul_(1.0 / self._dp_world)
        opt.step()

    def _apply_pending_update(self):
        if self._pending_reduce is not None:
            self._pending_reduce.wait()  # the compute stream waits for the collective; the host does not
            self._pending_reduce = None
        opts = self.native_optimizers()
        adam, soft = opts[0], opts[-1]
        cpe = getattr(self, "_cpe", None)
        if cpe is None and self._fused_update(adam, soft):
            self.all_batches_processed += 1
            self._update_pending = False
            return
        if self._graph_tick is not None:  # the sampler launch has already counted the step for the one-launch update
            raise RuntimeError("a step captured with a device-side index cursor needs the one-launch update")
        self._step_optimizer(adam)
        if cpe is not None:  # reward network and CPE q-network, in the reference's optimizer order
            cpe.forward(self._pending_batch)
            for which, opt in (("reward", opts[1]), ("cpe", opts[2])):
                for p in cpe.e[which]["params"]:
                    p.grad = None
                cpe.backward(which)
                self._step_optimizer(opt)
        elif self._pending_batch is not None:
            self._post_step_stats_forward(self._pending_batch)
        self._pending_batch = None
        soft.step()
        self.all_batches_processed += 1
        self._update_pending = False

    def validation_step(self, batch, batch_idx):
        raise NotImplementedError("CPE / EvaluationDataPage is outside the hot path (SURVEY.md §3.4)")


class DQNTrainer(QStepCore):
    def __init__(
        self,
        q_network,
        q_network_target,
        reward_network,
        q_network_cpe=None,
        q_network_cpe_target=None,
        metrics_to_score=None,
        evaluation: Optional[EvaluationParameters] = None,
        imitator=None,
        actions: Optional[List[str]] = None,
        rl: Optional[RLParameters] = None,
        double_q_learning: bool = True,
        bcq: Optional[BCQConfig] = None,
        minibatch_size: int = 1024,
        minibatches_per_step: int = 1,
        optimizer: Optional[Optimizer__Union] = None,
    ) -> None:
        # @resolve_defaults of the reference: default_factory values materialised here
        evaluation = evaluation if evaluation is not None else EvaluationParameters()
        rl = rl if rl is not None else RLParameters()
        actions = actions if actions is not None else []
        optimizer = optimizer if optimizer is not None else Optimizer__Union.default()
        super().__init__(rl, metrics_to_score=metrics_to_score, actions=actions,
                         evaluation_parameters=evaluation)
        assert self._actions is not None, "Discrete-action DQN needs action names"
        self.double_q_learning = double_q_learning
        self.minibatch_size = minibatch_size
        self.minibatches_per_step = minibatches_per_step or 1

        self.q_network = q_network
        self.q_network_target = q_network_target
        self.q_network_optimizer = optimizer
        self._initialize_cpe(reward_network, q_network_cpe, q_network_cpe_target, optimizer=optimizer)
        self._cpe = _CpeEngine(self) if self.calc_cpe_in_training else None

        # batch-constrained q-learning (:113-117): next actions the behaviour policy (the imitator) finds
        # unlikely are masked out of the max (:209-215)
        self.bcq = bcq is not None
        if self.bcq:
            self.bcq_drop_threshold = bcq.drop_threshold
            if imitator is None or not hasattr(imitator, "stack"):
                raise NotImplementedError(
                    "bcq needs `imitator` as a reagent_amd FullyConnectedNetwork (state features -> action logits); "
                    "scikit-learn imitators (a CPU call per batch) are not served")
            self.bcq_imitator = imitator

    # ---- head ---------------------------------------------------------------------------------
    def _alloc_head(self, batch, device):
        f32 = dict(dtype=torch.float32, device=device)
        self._loss_partials = torch.empty(ops.dqn_head_partials(batch), **f32)
        self._next_q = torch.empty(batch, **f32)
        self._next_idx = torch.empty(batch, dtype=torch.int64, device=device)
        self._q_sel = torch.empty(batch, **f32)

    def _run_head(self, b, B, action, next_mask, boosts, gamma_exp):
        ops.dqn_head(self._q, self._qn_online, self._qn_target, action, next_mask,
                     self._f32c(b.reward).reshape(-1), boosts, self._f32c(b.not_terminal).reshape(-1),
                     self.gamma, gamma_exp, self.double_q_learning, self._loss_type, self._dq,
                     self._loss_partials, self._next_q, self._next_idx, self._q_sel)
        from ..engine import FusedMLP

        if self._loss_tail_wanted and isinstance(self._qs, FusedMLP):
            self._loss_tail = (self._loss_partials, 1.0 / B, self._loss)  # summed in rg_mlp_wgrad_fused's reduce launch
        else:
            ops.reduce_sum(self._loss_partials, self._loss_partials.numel(), 1.0 / B, self._loss)
        self.all_action_scores = self._q

    # ---- reference surface --------------------------------------------------------------------
    @torch.no_grad()
    def get_detached_model_outputs(self, state) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """dqn_trainer.py:157-164"""
        return self.q_network(state), self.q_network_target(state)

    def compute_discount_tensor(self, batch, boosted_rewards: torch.Tensor):
        """dqn_trainer.py:166-177 — utility entry point; the step evaluates it inside rg_dqn_head."""
        discount_tensor = torch.full_like(boosted_rewards, self.gamma)
        if self.use_seq_num_diff_as_time_diff:
            assert self.multi_steps is None
            discount_tensor = torch.pow(self.gamma, batch.time_diff.float())
        if self.multi_steps is not None:
            assert batch.step is not None
            discount_tensor = torch.pow(self.gamma, batch.step.float())
        return discount_tensor

    def compute_td_loss(self, batch, boosted_rewards=None, discount_tensor=None):
        """dqn_trainer.py:179-239.  Reward boosting and discounting are recomputed inside the fused
        head from the batch itself; the two extra arguments are accepted for signature parity."""
        return self._hip_loss(batch)

    def train_step_gen(self, training_batch: rlt.DiscreteDqnInput, batch_idx: int):
        self._check_input(training_batch)
        td_loss = self.compute_td_loss(training_batch)
        yield td_loss
        td_loss = td_loss.detach()
        yield from self._cpe_segment(training_batch)
        self._log_dqn(td_loss, training_batch)
        yield self.soft_update_result()

    def _log_dqn(self, td_loss, training_batch):
        """dqn_trainer.py:306-347; evaluated only when a reporter / logger is attached so the
        default (no-op reporter) step has no logging kernels and no host syncs."""
        from .reagent_lightning_module import _NoOpReporter

        if isinstance(self._reporter, _NoOpReporter) and not self.logger:
            return
        rewards = self.boost_rewards(training_batch.reward, training_batch.action)
        logged_action_idxs = torch.argmax(training_batch.action, dim=1, keepdim=True)
        mask = training_batch.possible_actions_mask if self.maxq_learning else training_batch.action
        model_action_idxs = self.get_max_q_values(self.all_action_scores, mask.float())[1]
        self.reporter.log(
            td_loss=td_loss,
            logged_actions=logged_action_idxs,
            logged_propensities=training_batch.extras.action_probability,
            logged_rewards=rewards,
            logged_values=None,
            model_values=self.all_action_scores,
            model_values_on_logged_actions=None,
            model_action_idxs=model_action_idxs,
        )
        if self.logger:  # :320-347: per-action means as {action name: device scalar} (a TensorBoard logger reads them out)
            ap = training_batch.extras.action_probability
            greedy = torch.nn.functional.one_hot(model_action_idxs.reshape(-1), num_classes=self.num_actions).float().mean(dim=0)
            self.logger.log_metrics(
                {
                    "td_loss": td_loss,
                    "logged_actions": self._dense_to_action_dict(training_batch.action.float().mean(dim=0)),
                    "logged_propensities": None if ap is None else ap.mean(dim=0),
                    "logged_rewards": rewards.mean(),
                    "model_values": self._dense_to_action_dict(self.all_action_scores.mean(dim=0)),
                    "model_action_idxs": self._dense_to_action_dict(greedy),
                },
                step=self.all_batches_processed,
            )

    def _dense_to_action_dict(self, dense: torch.Tensor):
        """dqn_trainer.py:349-360: tensor([1.0, 0.0, 1.0]) -> {"1": 1.0, "2": 0.0, "3": 1.0} over the action names"""
        assert dense.size() == (self.num_actions,), f"Invalid dense size {dense.size()} != {(self.num_actions,)}"
        return {a: dense[i] for i, a in enumerate(self._actions)}
