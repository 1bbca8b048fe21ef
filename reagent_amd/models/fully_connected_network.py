"""FullyConnectedNetwork / FloatFeatureFullyConnected on the HIP FC kernels.

Same constructor arguments, parameter names (``dnn.{i}.0.weight`` / ``.bias``, nn.Linear layout; with
``use_layer_norm`` the layer's nn.LayerNorm sits at ``dnn.{i}.1``) and initialisation (Gaussian with gain, zero bias)
as reagent/models/fully_connected_network.py:67-217.  ``forward`` runs rg_fc_forward launches (inference: no autograd
graph is recorded — training goes through the trainers' fused step, which writes ``.grad`` directly).  Layer-norm
(Linear -> LayerNorm -> activation, :128-130) runs on the per-layer path with rg_layer_norm_*.  Batch-norm on a layer's
input (:107-108), dropout after the activation (:139-141) and the residual wrapper (:144-146) — off in every
configuration on the hot path — run on engine_general.GeneralFCStack (fp32 row sweeps between the
GEMMs: rg_batch_norm_*, rg_dropout, rg_add_cols); the module layout keeps the reference's parameter names
(``dnn.{i}.0.vanilla.*`` for the batch norm, ``dnn.{i}.module.{j}.*`` under a residual wrapper).
"""
import math
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.init as init

from .. import _lib as L
from ..core import types as rlt
from ..engine import FCStack, make_stack
from .base import ModelBase

_DEFAULT_PRECISION = L.PREC_F32


def set_default_precision(precision: int):
    """PREC_F32 (exact-fp32 MFMA, parity mode), PREC_BF16 (bf16 MFMA, throughput mode) or PREC_BF16X3
    (split-bf16 operands on the bf16 MFMA pipe: fp32-class results at a third of the bf16 rate)."""
    global _DEFAULT_PRECISION
    assert precision in (L.PREC_F32, L.PREC_BF16, L.PREC_BF16X3)
    _DEFAULT_PRECISION = precision


def get_default_precision() -> int:
    return _DEFAULT_PRECISION


def gaussian_fill_w_gain(tensor, gain, dim_in, min_std=0.0) -> None:
    """fully_connected_network.py:21-23"""
    init.normal_(tensor, mean=0, std=max(gain * math.sqrt(1 / dim_in), min_std))


class _Linear(nn.Module):
    """Parameter holder with nn.Linear's names/layout (weight [out, in], bias [out])."""

    def __init__(self, in_dim: int, out_dim: int):
        super().__init__()
        self.in_features, self.out_features = in_dim, out_dim
        self.weight = nn.Parameter(torch.empty(out_dim, in_dim))
        self.bias = nn.Parameter(torch.empty(out_dim))


class _Activation(nn.Module):
    def __init__(self, name: str):
        super().__init__()
        self.name = name

    def extra_repr(self):
        return self.name


class _SlateBatchNorm1d(nn.Module):
    """Parameter / running-statistics holder with SlateBatchNorm1d's names (:48-64: ``vanilla`` = nn.BatchNorm1d); the
    arithmetic is rg_batch_norm_* on [batch, features] inputs (the 3-D slate layout is not on this path)"""

    def __init__(self, num_features: int):
        super().__init__()
        self.vanilla = nn.BatchNorm1d(num_features)


class _Residual(nn.Module):
    """residual_wrapper.py:17-22: x + module(x); the add is rg_add_cols"""

    def __init__(self, module: nn.Module):
        super().__init__()
        self.module = module


def _find(layer, kind):
    seq = layer.module if isinstance(layer, _Residual) else layer
    for m in seq:
        if isinstance(m, kind):
            return m
    return None


class FullyConnectedNetwork(ModelBase):
    def __init__(
        self,
        layers,
        activations,
        *,
        use_batch_norm: bool = False,
        min_std: float = 0.0,
        dropout_ratio: float = 0.0,
        use_layer_norm: bool = False,
        normalize_output: bool = False,
        orthogonal_init: bool = False,
        use_skip_connections: bool = False,
    ) -> None:
        super().__init__()
        self.input_dim = layers[0]
        assert len(layers) == len(activations) + 1, (
            f"Invalid number of layers {len(layers)} and activations {len(activations)}. "
            "Number of layers needs to be 1 + number of activations"
        )
        modules: List[nn.Module] = []
        self.activation_names = list(activations)
        for i, (in_dim, out_dim, activation) in enumerate(zip(layers, layers[1:], activations)):
            if activation not in L.ACT:
                raise NotImplementedError(f"activation {activation} has no HIP epilogue")
            components: List[nn.Module] = []
            if use_batch_norm:  # :107-108 on the layer's input
                components.append(_SlateBatchNorm1d(in_dim))
            linear = _Linear(in_dim, out_dim)
            try:
                gain = torch.nn.init.calculate_gain(activation)
            except ValueError:
                gain = 1.0
            if orthogonal_init:
                nn.init.orthogonal_(linear.weight.data, gain=gain)
            else:
                gaussian_fill_w_gain(linear.weight, gain=gain, dim_in=in_dim, min_std=min_std)
            init.constant_(linear.bias, 0)
            # [BatchNorm] -> Linear -> [LayerNorm] -> activation -> [Dropout] (:104-141); the output layer gets the
            # layer norm and the dropout only with normalize_output
            components.append(linear)
            inner = normalize_output or i < len(activations) - 1
            if use_layer_norm and inner:
                components.append(nn.LayerNorm(out_dim))
            components.append(_Activation(activation))
            if dropout_ratio > 0.0 and inner:
                components.append(nn.Dropout(p=dropout_ratio))
            layer: nn.Module = nn.Sequential(*components)
            if use_skip_connections and in_dim == out_dim:  # :142-150 (other layers keep no skip)
                layer = _Residual(layer)
            modules.append(layer)
        self.dnn = nn.Sequential(*modules)
        self.stat_updates = 1  # evaluations of the reference per forward (batch-norm running statistics; the Gaussian actor: 2)
        self.precision = _DEFAULT_PRECISION
        self._stack = None

    # ---- engine plumbing ------------------------------------------------------------------
    def linears(self) -> List[_Linear]:
        return [_find(m, _Linear) for m in self.dnn]

    def layer_norms(self):
        """per layer: its nn.LayerNorm (a parameter holder here: the arithmetic is rg_layer_norm_*) or None"""
        return [_find(m, nn.LayerNorm) for m in self.dnn]

    def batch_norms(self):
        """per layer: the nn.BatchNorm1d holder of its input normalisation, or None"""
        return [b.vanilla if b is not None else None for b in (_find(m, _SlateBatchNorm1d) for m in self.dnn)]

    def dropouts(self) -> List[float]:
        return [d.p if d is not None else 0.0 for d in (_find(m, nn.Dropout) for m in self.dnn)]

    def residuals(self) -> List[bool]:
        return [isinstance(m, _Residual) for m in self.dnn]

    def is_plain(self) -> bool:
        """Linear -> activation layers only (what the fused kernels and the grouped QR-DQN engine serve)"""
        return (not any(self.residuals()) and not any(p > 0.0 for p in self.dropouts())
                and all(b is None for b in self.batch_norms()) and all(n is None for n in self.layer_norms()))

    def stack(self):
        if self._stack is None or getattr(self, "_stack_precision", None) != self.precision:
            lin = self.linears()
            self._stack = make_stack([l.weight for l in lin], [l.bias for l in lin],
                                     [L.ACT[a] for a in self.activation_names], self.precision,
                                     layer_norms=self.layer_norms(), batch_norms=self.batch_norms(),
                                     dropouts=self.dropouts(), residuals=self.residuals(),
                                     training=lambda: self.training)
            self._stack_precision = self.precision  # the engine may run a different one (x3 on odd shapes: fp32)
        if hasattr(self._stack, "stat_updates"):
            self._stack.stat_updates = self.stat_updates
        return self._stack

    def autograd_stack(self):
        """The engine instance `tensor.backward()` runs on: a SECOND stack over the same parameters with its own weight
        fragments and its own saved-activation workspace.  `stack()` belongs to the trainers (their native steps and
        captured HIP graphs hold its buffer addresses), so a grad-mode `q_network(x)` between two training steps — or
        between a yielded loss and its backward — must not replace that workspace."""
        if getattr(self, "_ag_stack", None) is None or getattr(self, "_ag_stack_precision", None) != self.precision:
            lin = self.linears()
            self._ag_stack = make_stack([l.weight for l in lin], [l.bias for l in lin],
                                        [L.ACT[a] for a in self.activation_names], self.precision)
            self._ag_stack_precision = self.precision
        return self._ag_stack

    def __deepcopy__(self, memo):
        stack, self._stack = self._stack, None  # workspaces are not part of the model
        ag, self._ag_stack = getattr(self, "_ag_stack", None), None
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            from copy import deepcopy

            for k, v in self.__dict__.items():
                setattr(new, k, deepcopy(v, memo))
        finally:
            self._stack, self._ag_stack = stack, ag
        # deep-copied parameters own fresh storage -> drop any slab association
        for p in new.parameters():
            if hasattr(p, "_rg_slab"):
                del p._rg_slab
        return new

    def input_prototype(self):
        return torch.randn(1, self.input_dim)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        """fully_connected_network.py:157-163.  Under autograd (grad mode on and the input or a parameter requires grad) a
        Linear -> activation stack records ONE autograd node whose backward is the HIP backward of the stack
        (`q_network(state).sum().backward()` fills `.grad` like the reference nn.Module, models/dqn.py:52-63); stacks
        with norm / dropout / residual layers return a tensor whose backward raises instead of silently yielding no
        gradient (their training path is the trainers' fused step)."""
        if torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters())):
            if self.is_plain():
                lin = self.linears()
                return _StackFunction.apply(self, input, *[p for l in lin for p in (l.weight, l.bias)])
            return no_autograd_guard(self._forward_no_grad(input), self, "a FullyConnectedNetwork with norm / dropout / residual layers")
        return self._forward_no_grad(input)

    @torch.no_grad()
    def _forward_no_grad(self, input: torch.Tensor) -> torch.Tensor:
        L.require_cuda(input, "input")
        st = self.stack()
        st.stage_weights(need_transposed=False)
        x32 = input if input.dtype == torch.float32 else input.float()
        xc, _ = st.stage_input(x32.contiguous() if x32.stride(-1) != 1 else x32, need_transposed=False)
        out = torch.empty(input.shape[0], st.dims[-1], dtype=torch.float32, device=input.device)
        st.forward(xc, out, save=False)
        return out

    _autograd_serial = 0


class _StackFunction(torch.autograd.Function):
    """One autograd node for a plain stack: forward = the saving HIP forward, backward = the stack's HIP backward (input
    gradient, weight and bias gradients) — on the network's `autograd_stack()`, never on the trainers' `stack()`.  That
    stack keeps ONE set of saved activations, so a backward must follow its own forward before the same network runs
    another recorded forward (checked: a stale node raises)."""

    @staticmethod
    def forward(ctx, net, x, *params):
        L.require_cuda(x, "input")
        st = net.autograd_stack()
        need_dx = bool(x.requires_grad)
        st.set_need_input_grad(need_dx or getattr(st, "_need_dx", False))
        st.stage_weights(need_transposed=True)
        x32 = x.detach() if x.dtype == torch.float32 else x.detach().float()
        xc, xt = st.stage_input(x32.contiguous() if x32.stride(-1) != 1 else x32, need_transposed=True)
        out = torch.empty(x.shape[0], st.dims[-1], dtype=torch.float32, device=x.device)
        st.forward(xc, out, save=True)
        net._autograd_serial += 1
        ctx.net, ctx.xt, ctx.need_dx, ctx.serial, ctx.in_dtype = net, xt, need_dx, net._autograd_serial, x.dtype
        ctx.needs = [p.requires_grad for p in params]
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        net = ctx.net
        if net._autograd_serial != ctx.serial:
            raise RuntimeError("reagent_amd: this network ran another recorded forward before this backward — the stack "
                               "keeps one set of saved activations (call backward first, or run the other forward under "
                               "torch.no_grad())")
        (out,) = ctx.saved_tensors
        st = net.autograd_stack()
        lin = net.linears()
        dw = [torch.empty_like(l.weight) for l in lin]
        db = [torch.empty_like(l.bias) for l in lin]
        B = grad_out.shape[0]
        dx = torch.empty(B, st.dims[0], dtype=torch.float32, device=grad_out.device) if ctx.need_dx else None
        g = grad_out.detach()
        g = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()
        with torch.no_grad():
            st.backward(g.clone(), ctx.xt, dw, db, dx32=dx, out32=out)
        grads = [t for pair in zip(dw, db) for t in pair]
        grads = [t if need else None for t, need in zip(grads, ctx.needs)]
        if dx is not None and ctx.in_dtype != torch.float32:
            dx = dx.to(ctx.in_dtype)
        return (None, dx) + tuple(grads)


class _NoAutograd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, what, *params):
        ctx.what = what
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad_out):
        raise NotImplementedError(f"reagent_amd: {ctx.what} was evaluated by HIP kernels outside autograd; gradients of this "
                                  "output are produced by the trainers' steps (train_step_gen / train_step_native), not by "
                                  "tensor.backward()")


def no_autograd_guard(out: torch.Tensor, module: nn.Module, what: str) -> torch.Tensor:
    """`out` was computed without an autograd graph: under grad mode hand back a tensor whose backward RAISES, so that a
    caller differentiating through it does not silently get no gradient (intentional deviations)"""
    if not torch.is_grad_enabled():
        return out
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return out
    return _NoAutograd.apply(out, what, *params)


class FloatFeatureFullyConnected(ModelBase):
    """reagent/models/fully_connected_network.py:166-217"""

    def __init__(
        self,
        state_dim,
        output_dim,
        sizes,
        activations,
        *,
        output_activation: str = "linear",
        num_atoms: Optional[int] = None,
        use_batch_norm: bool = False,
        dropout_ratio: float = 0.0,
        normalized_output: bool = False,
        use_layer_norm: bool = False,
    ):
        super().__init__()
        assert state_dim > 0, "state_dim must be > 0, got {}".format(state_dim)
        assert output_dim > 0, "output_dim must be > 0, got {}".format(output_dim)
        self.state_dim = state_dim
        self.output_dim = output_dim
        assert len(sizes) == len(activations), (
            "The numbers of sizes and activations must match; got {} vs {}".format(len(sizes), len(activations))
        )
        self.num_atoms = num_atoms
        self.fc = FullyConnectedNetwork(
            [state_dim] + list(sizes) + [output_dim * (num_atoms or 1)],
            list(activations) + [output_activation],
            use_batch_norm=use_batch_norm,
            dropout_ratio=dropout_ratio,
            normalize_output=normalized_output,
            use_layer_norm=use_layer_norm,
        )

    def input_prototype(self):
        return rlt.FeatureData(self.fc.input_prototype())

    def forward(self, state) -> torch.Tensor:
        float_features = state.float_features
        x = self.fc(float_features)
        if self.num_atoms is not None:
            x = x.view(float_features.shape[0], self.action_dim, self.num_atoms)
        return x
