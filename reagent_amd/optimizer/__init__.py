"""Optimizers of the hot path: fused Adam and the SoftUpdate pseudo-optimizer, both as
``torch.optim.Optimizer`` subclasses so ``configure_optimizers()`` keeps the reference contract
(reagent/optimizer/union.py:52-64, reagent/optimizer/soft_update.py:9-71)."""
import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch

from .. import _lib as _L
from .. import ops
from ..engine import ParamSlab, ensure_slab


def _bump(params):
    for p in params:
        p._rg_version = getattr(p, "_rg_version", 0) + 1


class AdamSchedule:
    """Device-resident Adam schedule read by the rg_*_sched entry points (include/reagent_hip.h): the number of
    steps applied, lr, and the table of (1 - beta1^t, sqrt(1 - beta2^t)) — in double, computed with the same
    Python expressions the scalar path passes per launch, up to the step where both have reached 1.0.  With it a
    step captured once in a HIP graph replays as step t, t+1, ... (launch arguments are frozen in a graph; HBM
    is not).  `pending` counts device steps the host-side `state[p]["step"]` has not been told about yet."""

    MAX_ENTRIES = 1 << 20

    def __init__(self, lr: float, betas, step: int, device):
        b1, b2 = betas
        rows = []
        t = 0
        while True:
            t += 1
            bc1, bc2 = 1.0 - b1**t, math.sqrt(1.0 - b2**t)
            rows += [bc1, bc2]
            if bc1 == 1.0 and bc2 == 1.0:
                break
            if t >= self.MAX_ENTRIES:
                raise NotImplementedError(f"betas {betas}: the bias corrections do not reach 1.0 within "
                                          f"{self.MAX_ENTRIES} steps; use the per-launch scalar path")
        self.n, self.lr, self.betas = t, float(lr), (b1, b2)
        self.buf = torch.tensor([float(step), float(lr), float(t), 0.0] + rows, dtype=torch.float64).to(device)
        self.pending = 0

    def set_lr(self, lr: float):
        if float(lr) != self.lr:
            self.lr = float(lr)
            self.buf[1:2].copy_(torch.tensor([self.lr], dtype=torch.float64))

    def set_step(self, step: int):
        self.buf[0:1].copy_(torch.tensor([float(step)], dtype=torch.float64))
        self.pending = 0


def capturing() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam arithmetic (torch/optim/adam.py::_single_tensor_adam) executed by ONE
    rg_adam_step launch over the network's flat parameter slab.

    The parameters of a group are re-homed into a ``ParamSlab`` on first use; ``state[p]`` exposes
    ``step`` / ``exp_avg`` / ``exp_avg_sq`` as views of the flat moment slabs, so ``state_dict()``
    has the layout of ``torch.optim.Adam``.  ``grad_scale`` (e.g. 1/world_size after a summed
    all-reduce) is folded into the kernel.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 amsgrad=False, maximize=False):
        if amsgrad or maximize:
            raise NotImplementedError("amsgrad/maximize are not on the ReAgent hot path")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._moments = {}  # group index -> (slab, exp_avg, exp_avg_sq): THIS optimizer's flat moment buffers
        self._scheds = {}   # group index -> AdamSchedule (graph-safe stepping, enable_device_schedule)
        self.grad_scale = 1.0

    # ---- graph-safe stepping ------------------------------------------------------------------
    def enable_device_schedule(self):
        """From now on step() reads lr and the bias corrections from HBM (rg_adam_step_sched) and counts steps on
        the device: a step captured in a HIP graph stays correct on every replay.  Same bits as the scalar path."""
        for gi, group in enumerate(self.param_groups):
            if gi in self._scheds:
                continue
            slab, _, _ = self.moments_for(gi)
            steps = {int(self.state[p]["step"]) if len(self.state.get(p, {})) else 0 for p in slab.params}
            if len(steps) != 1:
                raise RuntimeError("a device schedule needs every parameter of the group at the same Adam step")
            self._scheds[gi] = AdamSchedule(group["lr"], group["betas"], steps.pop(), slab.data.device)
        return self

    def disable_device_schedule(self):
        """back to per-launch scalar coefficients (no rg_sched_tick launch per step)"""
        self.materialize_steps()
        self._scheds = {}
        return self

    def schedule_for(self, gi: int):
        return self._scheds.get(gi)

    def note_device_steps(self, n: int):
        """n more steps were applied on the device without step() being called (graph replays)"""
        for s in self._scheds.values():
            s.pending += n

    def materialize_steps(self):
        """bring state[p]["step"] up to the device-side step counters"""
        for gi, s in self._scheds.items():
            if s.pending:
                slab = self.slab_for(gi)
                for i in range(len(slab.params)):
                    st = self.state[slab.params[i]]
                    if len(st) == 0:
                        self.advance(gi, i)
                        st["step"] += s.pending - 1
                    else:
                        st["step"] += s.pending
                s.pending = 0

    def state_dict(self):
        self.materialize_steps()
        return super().state_dict()

    def slab_for(self, gi: int) -> ParamSlab:
        return self.moments_for(gi)[0]

    def moments_for(self, gi: int):
        """(parameter slab, exp_avg, exp_avg_sq) of group gi.  The moment buffers belong to the optimizer
        instance (a second FusedAdam over the same network starts from zero moments like a second
        torch.optim.Adam would); state[p]["exp_avg"/"exp_avg_sq"] are views into them."""
        group = self.param_groups[gi]
        slab = ensure_slab(group["params"])
        rec = self._moments.get(gi)
        if rec is None or rec[0] is not slab or rec[1].numel() != slab.total:
            m, v = torch.zeros_like(slab.data), torch.zeros_like(slab.data)
            for i, p in enumerate(slab.params):  # moments that already exist (loaded state, re-homed slab)
                st = self.state.get(p)
                if st:
                    slab.view(m, i).copy_(st["exp_avg"])
                    slab.view(v, i).copy_(st["exp_avg_sq"])
                    st["exp_avg"], st["exp_avg_sq"] = slab.view(m, i), slab.view(v, i)
            rec = self._moments[gi] = (slab, m, v)
        if rec[1].device != slab.data.device:
            m, v = rec[1].to(slab.data.device), rec[2].to(slab.data.device)
            rec = self._moments[gi] = (slab, m, v)
            for i, p in enumerate(slab.params):
                st = self.state.get(p)
                if st:
                    st["exp_avg"], st["exp_avg_sq"] = slab.view(m, i), slab.view(v, i)
        return rec

    def advance(self, gi: int, i: int) -> int:
        """count one more step of parameter i of group gi; returns the new step number"""
        slab, m, v = self.moments_for(gi)
        p = slab.params[i]
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = slab.view(m, i)
            st["exp_avg_sq"] = slab.view(v, i)
        st["step"] += 1
        return int(st["step"])

    def load_state_dict(self, state_dict):
        """torch.optim.Adam's state_dict layout (step / exp_avg / exp_avg_sq per parameter): the loaded
        moments are copied INTO the flat buffers the kernel reads and state[p] re-bound to views of them."""
        super().load_state_dict(state_dict)
        self._moments = {}
        had_sched = bool(self._scheds)
        self._scheds = {}
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    st.pop("_step_int", None)
                    st["step"] = torch.tensor(float(st["step"]))  # own copy: torch hands the saved tensor through
            self.moments_for(gi)  # adopts the loaded tensors
        if had_sched:
            self.enable_device_schedule()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            slab, exp_avg, exp_avg_sq = self.moments_for(gi)
            beta1, beta2 = group["betas"]
            sched = self._scheds.get(gi)
            if sched is not None:
                gbase = slab.grad.data_ptr()
                for i, p in enumerate(slab.params):
                    if p.grad is None:
                        raise RuntimeError("device-scheduled Adam steps every parameter of the group: a gradient is missing")
                    if p.grad.data_ptr() != gbase + 4 * slab.offsets[i]:
                        slab.view(slab.grad, i).copy_(p.grad)
                if not capturing():
                    sched.set_lr(group["lr"])
                    sched.pending += 1
                ops.adam_step_sched(slab.data, slab.grad, exp_avg, exp_avg_sq, slab.total, beta1, beta2, group["eps"],
                                    group["weight_decay"], sched.buf, self.grad_scale)
                ops.sched_tick(sched.buf)
                _bump(slab.params)
                continue
            # collect gradients into the flat slab (zero-copy when backward already wrote there)
            runs: List[Tuple[int, int, int]] = []  # (offset, n, step)
            gbase = slab.grad.data_ptr()
            for i, p in enumerate(slab.params):
                if p.grad is None:
                    continue
                off, n = slab.offsets[i], p.numel()
                if p.grad.data_ptr() != gbase + 4 * off:
                    slab.view(slab.grad, i).copy_(p.grad)
                step = self.advance(gi, i)
                padded = (n + ParamSlab.ALIGN - 1) // ParamSlab.ALIGN * ParamSlab.ALIGN
                if runs and runs[-1][0] + runs[-1][1] == off and runs[-1][2] == step:
                    runs[-1] = (runs[-1][0], runs[-1][1] + padded, step)
                else:
                    runs.append((off, padded, step))
            for off, n, step in runs:
                bc1 = 1.0 - beta1**step
                bc2_sqrt = math.sqrt(1.0 - beta2**step)
                ops.adam_step(slab.data, slab.grad, exp_avg, exp_avg_sq, n, group["lr"],
                              beta1, beta2, group["eps"], group["weight_decay"], bc1, bc2_sqrt,
                              self.grad_scale, offset=off)
            _bump(slab.params)
        return loss

    def zero_grad(self, set_to_none: bool = True):
        # keep p.grad aliased to the gradient slab: the HIP backward overwrites it every step
        if set_to_none:
            for group in self.param_groups:
                for p in group["params"]:
                    p.grad = None
        else:
            super().zero_grad(set_to_none=False)


class SoftUpdate(torch.optim.Optimizer):
    """target = tau * source + (1 - tau) * target  (reagent/optimizer/soft_update.py:9-71)."""

    def __init__(self, target_params, source_params, tau: float = 0.1) -> None:
        target_params = list(target_params)
        source_params = list(source_params)
        if len(target_params) != len(source_params):
            raise ValueError("target and source must have the same number of parameters")
        for t_param, s_param in zip(target_params, source_params):
            if t_param.shape != s_param.shape:
                raise ValueError("The shape of target parameter doesn't match that of the source")
        params = target_params + source_params
        defaults = dict(tau=tau, lr=1.0)
        super().__init__(params, defaults)
        for group in self.param_groups:
            tau = group["tau"]
            if tau > 1.0 or tau < 0.0:
                raise ValueError(f"tau should be in [0.0, 1.0]; got {tau}")

    @classmethod
    def make_optimizer_scheduler(cls, target_params, source_params, tau):
        su = cls(target_params, source_params, tau)
        return {"optimizer": su}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            params = group["params"]
            n = len(params)
            tau = group["tau"]
            tgt, src = params[: n // 2], params[n // 2 :]
            # segments of consecutive pairs whose sources are one whole ParamSlab -> one launch
            segs = []
            for t, s in zip(tgt, src):
                if t is s:
                    continue
                ss = getattr(s, "_rg_slab", None)
                if segs and ss is not None and segs[-1][0] is ss:
                    segs[-1][1].append(t)
                    segs[-1][2].append(s)
                else:
                    segs.append((ss, [t], [s]))
            for ss, ts, srcs in segs:
                whole = ss is not None and len(ss.params) == len(srcs) and all(
                    a is b for a, b in zip(ss.params, srcs)) and ss.is_bound()
                if whole:
                    tslab = ensure_slab(ts)
                    ops.soft_update(tslab.data, ss.data, ss.total, tau)
                else:
                    for t, s in zip(ts, srcs):
                        _L.require_cuda(t)
                        _L.check(_L.lib().rg_soft_update(t.data_ptr(), s.data_ptr(), t.numel(), tau,
                                                         _L.stream_ptr()), "rg_soft_update")
            _bump(tgt)
        return loss


# ---- learning-rate schedulers (reagent/optimizer/scheduler.py:16-41, uninferrable_schedulers.py) -----
class LearningRateSchedulerConfig:
    """A config whose class name is the torch.optim.lr_scheduler class it builds and whose fields are that
    class's arguments (scheduler.py:19-38).  FusedAdam and rg_mlp_update_fused read group["lr"] at every
    step, so the torch schedulers drive the fused optimizer unchanged."""

    def make_from_optimizer(self, optimizer: torch.optim.Optimizer):
        import inspect

        cls = getattr(torch.optim.lr_scheduler, type(self).__name__)
        args = {k: getattr(self, k) for k in inspect.signature(cls).parameters if k != "optimizer" and hasattr(self, k)}
        return cls(optimizer=optimizer, **args)


@dataclass
class StepLR(LearningRateSchedulerConfig):
    step_size: int
    gamma: float = 0.1
    last_epoch: int = -1


@dataclass
class MultiStepLR(LearningRateSchedulerConfig):
    milestones: List[int]
    gamma: float = 0.1
    last_epoch: int = -1


@dataclass
class ExponentialLR(LearningRateSchedulerConfig):
    gamma: float
    last_epoch: int = -1


@dataclass
class CosineAnnealingLR(LearningRateSchedulerConfig):
    T_max: int
    eta_min: float = 0
    last_epoch: int = -1


# ---- config objects (duck-typed stand-ins for reagent.optimizer.Optimizer__Union) -----------
@dataclass
class Adam:
    """reagent/optimizer/uninferrable_optimizers.py:23-33 defaults; `lr_schedulers` as in
    OptimizerConfig (optimizer.py:61-85: at most one)."""

    lr: float = 0.001
    betas: Tuple[float, float] = (0.9, 0.999)
    eps: float = 1e-08
    weight_decay: float = 0
    amsgrad: bool = False
    # accepted for configuration compatibility (uninferrable_optimizers.py:29-32).  foreach / capturable choose among torch's
    # implementations of the same arithmetic (FusedAdam is one launch and capturable as it is); the other two change the
    # arithmetic and are refused when set
    maximize: bool = False
    foreach: Optional[bool] = None
    capturable: bool = False
    differentiable: bool = False
    lr_schedulers: List[LearningRateSchedulerConfig] = field(default_factory=list)

    def make_optimizer_scheduler(self, params):
        assert len(self.lr_schedulers) <= 1, "Multiple schedulers for one optimizer is no longer supported"
        if self.maximize or self.differentiable:
            raise NotImplementedError("Adam(maximize=True / differentiable=True): the fused one-launch Adam minimises and is not "
                                      "differentiable through")
        optimizer = FusedAdam(params, lr=self.lr, betas=tuple(self.betas), eps=self.eps,
                              weight_decay=self.weight_decay, amsgrad=self.amsgrad)
        if len(self.lr_schedulers) == 0:
            return {"optimizer": optimizer}
        return {"optimizer": optimizer, "lr_scheduler": self.lr_schedulers[0].make_from_optimizer(optimizer)}


# ---- the other members of the reference's Optimizer__Union (optimizer/union.py:19-64: every torch.optim class is registered;
# uninferrable_optimizers.py spells out the ones whose defaults cannot be inferred).  Adam is the hot path (FusedAdam, one
# launch); these build torch's OWN optimizer over the same parameters — torch's device kernels, the reference's arithmetic — and
# the native step then takes its separate-launch update path (optimizer.step(), rg_soft_update, re-staging on the version
# counters).  No HIP graph capture with them (their step counts live on the host).
class _TorchOptimizerConfig:
    """a config whose class name is the torch.optim class it builds and whose fields are that class's arguments"""

    lr_schedulers: List[LearningRateSchedulerConfig]

    def make_optimizer_scheduler(self, params):
        import dataclasses
        import inspect

        assert len(self.lr_schedulers) <= 1, "Multiple schedulers for one optimizer is no longer supported"
        cls = getattr(torch.optim, type(self).__name__)
        accepted = inspect.signature(cls).parameters
        kwargs = {}
        for f in dataclasses.fields(self):
            if f.name == "lr_schedulers":
                continue
            if f.name in accepted:
                kwargs[f.name] = getattr(self, f.name)
            elif getattr(self, f.name) != f.default:
                # a field the installed torch.optim class does not take may only be dropped at its default: a dropped
                # `maximize=True` would silently flip the direction of the optimisation
                raise TypeError(f"{type(self).__name__}({f.name}={getattr(self, f.name)!r}): torch.optim.{type(self).__name__} of torch "
                                f"{torch.__version__} has no such argument")
        kwargs = {k: (tuple(v) if isinstance(v, list) else v) for k, v in kwargs.items()}
        optimizer = cls(params, **kwargs)
        if len(self.lr_schedulers) == 0:
            return {"optimizer": optimizer}
        return {"optimizer": optimizer, "lr_scheduler": self.lr_schedulers[0].make_from_optimizer(optimizer)}


def _torch_config(name, **defaults):
    """dataclass `name` with the given fields / defaults (uninferrable_optimizers.py, torch.optim signatures) + lr_schedulers"""
    import dataclasses

    fields = [(k, type(v) if v is not None else Optional[bool], dataclasses.field(default=v)) for k, v in defaults.items()]
    fields.append(("lr_schedulers", List[LearningRateSchedulerConfig], dataclasses.field(default_factory=list)))
    return dataclasses.make_dataclass(name, fields, bases=(_TorchOptimizerConfig,))


# field lists = uninferrable_optimizers.py:36-114 (foreach: None = torch's own choice)
SGD = _torch_config("SGD", lr=0.001, momentum=0.0, weight_decay=0.0, dampening=0.0, nesterov=False, maximize=False, foreach=None,
                    differentiable=False)
AdamW = _torch_config("AdamW", lr=0.001, betas=(0.9, 0.999), eps=1e-08, weight_decay=0.01, amsgrad=False, maximize=False, foreach=None,
                      capturable=False)
NAdam = _torch_config("NAdam", lr=0.001, betas=(0.9, 0.999), eps=1e-08, weight_decay=0.0, momentum_decay=4e-3, maximize=False,
                      foreach=None)
RAdam = _torch_config("RAdam", lr=0.001, betas=(0.9, 0.999), eps=1e-08, weight_decay=0.0, maximize=False, foreach=None)
Adamax = _torch_config("Adamax", lr=0.001, betas=(0.9, 0.999), eps=1e-08, weight_decay=0.0, maximize=False, foreach=None)
Rprop = _torch_config("Rprop", lr=0.01, etas=(0.5, 1.2), step_sizes=(1e-06, 50.0), maximize=False, foreach=None)
RMSprop = _torch_config("RMSprop", lr=0.01, alpha=0.99, eps=1e-08, weight_decay=0.0, momentum=0.0, centered=False)
Adagrad = _torch_config("Adagrad", lr=0.01, lr_decay=0.0, weight_decay=0.0, initial_accumulator_value=0.0, eps=1e-10)
Adadelta = _torch_config("Adadelta", lr=1.0, rho=0.9, eps=1e-06, weight_decay=0.0)
ASGD = _torch_config("ASGD", lr=0.01, lambd=0.0001, alpha=0.75, t0=1000000.0, weight_decay=0.0)
TORCH_OPTIMIZER_CONFIGS = {c.__name__: c for c in (SGD, AdamW, NAdam, RAdam, Adamax, Rprop, RMSprop, Adagrad, Adadelta, ASGD)}
# members of the reference's union that cannot serve this path: LBFGS re-evaluates the loss through a closure (the training
# step has no such re-entry), SparseAdam wants sparse gradients (the networks here are dense)
_UNSERVED = {"LBFGS": "needs a closure that re-evaluates the loss", "SparseAdam": "needs sparse gradients"}


class Optimizer__Union:
    """``Optimizer__Union(Adam=Adam(lr=...))`` / ``Optimizer__Union.default()`` / ``Optimizer__Union(SGD=SGD(lr=...))`` as in
    reagent/optimizer/union.py:52-64: exactly one member set.  Adam is the hot path (FusedAdam, the one-launch update); the
    other torch optimizers run as torch's own (TORCH_OPTIMIZER_CONFIGS) through the native step's separate-launch update."""

    def __init__(self, Adam: Optional["Adam"] = None, **others):
        self._name, self._value = "Adam", None
        for name, cfg in others.items():
            if cfg is None:
                continue
            if name in _UNSERVED:
                raise NotImplementedError(f"Optimizer__Union.{name}: {_UNSERVED[name]}")
            if name not in TORCH_OPTIMIZER_CONFIGS:
                raise ValueError(f"Optimizer__Union has no member {name!r} (members: Adam, {', '.join(sorted(TORCH_OPTIMIZER_CONFIGS))})")
            if Adam is not None or self._value is not None:
                raise ValueError("Optimizer__Union takes exactly one member")
            self._name, self._value = name, cfg
        if self._value is None:
            self._value = Adam if Adam is not None else globals()["Adam"]()
        self.Adam = self._value if self._name == "Adam" else None

    @classmethod
    def default(cls, **kwargs):
        return cls(Adam=globals()["Adam"](**kwargs))

    @property
    def selected_field(self) -> str:
        return self._name

    @property
    def value(self):
        return self._value

    def make_optimizer_scheduler(self, params):
        return self.value.make_optimizer_scheduler(params)
