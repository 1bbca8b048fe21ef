"""FullyConnectedNetwork / FloatFeatureFullyConnected on the HIP FC kernels.

Same constructor arguments, parameter names (``dnn.{i}.0.weight`` / ``.bias``, nn.Linear layout; with
``use_layer_norm`` the layer's nn.LayerNorm sits at ``dnn.{i}.1``) and initialisation (Gaussian with gain, zero bias)
as reagent/models/fully_connected_network.py:67-217.  ``forward`` runs rg_fc_forward launches (inference: no autograd
graph is recorded — training goes through the trainers' fused step, which writes ``.grad`` directly).  Layer-norm
(Linear -> LayerNorm -> activation, :128-130) runs on the per-layer path with rg_layer_norm_*; batch-norm, dropout and
skip connections are off in every configuration on the hot path and are rejected here instead of
silently falling back to torch.
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
        if use_batch_norm or dropout_ratio > 0.0 or use_skip_connections:
            raise NotImplementedError(
                "batch-norm / dropout / skip connections are not part of the MI355X hot path (off in every BASELINE "
                "configuration); layer-norm is (use_layer_norm)"
            )
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
            # Linear -> [LayerNorm] -> activation (:121-137); the output layer is normalised only with normalize_output
            if use_layer_norm and (normalize_output or i < len(activations) - 1):
                modules.append(nn.Sequential(linear, nn.LayerNorm(out_dim), _Activation(activation)))
            else:
                modules.append(nn.Sequential(linear, _Activation(activation)))
        self.dnn = nn.Sequential(*modules)
        self.precision = _DEFAULT_PRECISION
        self._stack = None

    # ---- engine plumbing ------------------------------------------------------------------
    def linears(self) -> List[_Linear]:
        return [m[0] for m in self.dnn]

    def layer_norms(self):
        """per layer: its nn.LayerNorm (a parameter holder here: the arithmetic is rg_layer_norm_*) or None"""
        return [m[1] if isinstance(m[1], nn.LayerNorm) else None for m in self.dnn]

    def stack(self):
        if self._stack is None or getattr(self, "_stack_precision", None) != self.precision:
            lin = self.linears()
            self._stack = make_stack([l.weight for l in lin], [l.bias for l in lin],
                                     [L.ACT[a] for a in self.activation_names], self.precision,
                                     layer_norms=self.layer_norms())
            self._stack_precision = self.precision  # the engine may run a different one (x3 on odd shapes: fp32)
        return self._stack

    def __deepcopy__(self, memo):
        stack, self._stack = self._stack, None  # workspaces are not part of the model
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            from copy import deepcopy

            for k, v in self.__dict__.items():
                setattr(new, k, deepcopy(v, memo))
        finally:
            self._stack = stack
        # deep-copied parameters own fresh storage -> drop any slab association
        for p in new.parameters():
            if hasattr(p, "_rg_slab"):
                del p._rg_slab
        return new

    def input_prototype(self):
        return torch.randn(1, self.input_dim)

    @torch.no_grad()
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        L.require_cuda(input, "input")
        st = self.stack()
        st.stage_weights(need_transposed=False)
        x32 = input if input.dtype == torch.float32 else input.float()
        xc, _ = st.stage_input(x32.contiguous() if x32.stride(-1) != 1 else x32, need_transposed=False)
        out = torch.empty(input.shape[0], st.dims[-1], dtype=torch.float32, device=input.device)
        st.forward(xc, out, save=False)
        return out


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
