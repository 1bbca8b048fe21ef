"""GaussianFullyConnectedActor (reagent/models/actor.py:113-261): FC stack -> (loc, scale_log),
tanh-squashed reparameterised Gaussian sample and its log-probability.  The FC stack runs on the HIP
FC kernels, the head on rg_gaussian_head_forward / rg_gaussian_log_prob.  One FC evaluation per call
(the reference evaluates it twice in forward() — the values are identical)."""
import math
from typing import List

import torch

from .. import _lib as L
from .. import ops
from ..core import types as rlt
from .base import ModelBase
from .fully_connected_network import no_autograd_guard
from .fully_connected_network import FullyConnectedNetwork

LOG_PROB_MIN: float = -2.0
LOG_PROB_MAX = 2.0


class GaussianFullyConnectedActor(ModelBase):
    def __init__(
        self,
        state_dim: int,
        action_dim: int,
        sizes: List[int],
        activations: List[str],
        scale: float = 0.05,
        use_batch_norm: bool = False,
        use_layer_norm: bool = False,
        use_l2_normalization: bool = False,
    ) -> None:
        super().__init__()
        assert state_dim > 0, "state_dim must be > 0, got {}".format(state_dim)
        assert action_dim > 0, "action_dim must be > 0, got {}".format(action_dim)
        if use_l2_normalization:
            raise NotImplementedError("use_l2_normalization: the reference leaves the log-probability of the normalised "
                                      "action a TODO (actor.py:237-241); not served on the MI355X path")
        self.state_dim = state_dim
        self.action_dim = action_dim
        assert len(sizes) == len(activations), (
            "The numbers of sizes and activations must match; got {} vs {}".format(len(sizes), len(activations))
        )
        self.fc = FullyConnectedNetwork([state_dim] + list(sizes) + [action_dim * 2], list(activations) + ["linear"],
                                        use_layer_norm=use_layer_norm, use_batch_norm=use_batch_norm)
        # the reference's forward() evaluates the stack twice on the same batch (forward :216 and get_log_prob :246): one
        # evaluation here, but batch-norm layers in training mode move their running statistics twice
        self.fc.stat_updates = 2
        self.use_layer_norm = use_layer_norm
        if self.use_layer_norm:  # actor.py:153-155: loc and scale_log are each normalised over the action dimension
            self.loc_layer_norm = torch.nn.LayerNorm(action_dim)
            self.scale_layer_norm = torch.nn.LayerNorm(action_dim)
        self.use_l2_normalization = False
        self.const = math.log(math.sqrt(2 * math.pi))
        self.eps = 1e-6
        self.noise_override = None  # tests / parity runs inject the reference's randn draw here

    def input_prototype(self):
        return rlt.FeatureData(torch.randn(1, self.state_dim))

    def head_norm(self, raw: torch.Tensor, out: torch.Tensor, stats=None):
        """loc_layer_norm / scale_layer_norm (actor.py:194-196) on the two halves of the FC output `raw` [B, 2A] -> `out`;
        stats = ((mean, rstd), (mean, rstd)) buffers kept for the backward (None: inference)"""
        A = self.action_dim
        for h, ln in enumerate((self.loc_layer_norm, self.scale_layer_norm)):
            m, r = stats[h] if stats is not None else (None, None)
            ops.layer_norm_forward(raw[:, h * A:(h + 1) * A], ln.weight.detach(), ln.bias.detach(), ln.eps, L.ACT["linear"],
                                   y32=out[:, h * A:(h + 1) * A], mean=m, rstd=r)
        return out

    def _fc_out(self, state, evaluations: int = 1):
        self.fc.stat_updates = evaluations
        try:
            loc_scale = self.fc(state.float_features)
        finally:
            self.fc.stat_updates = 2
        return self.head_norm(loc_scale, torch.empty_like(loc_scale)) if self.use_layer_norm else loc_scale

    def _get_loc_and_scale_log(self, state):
        loc_scale = self._fc_out(state)
        loc = loc_scale[::, : self.action_dim]
        scale_log = loc_scale[::, self.action_dim :].clamp(LOG_PROB_MIN, LOG_PROB_MAX)
        return loc, scale_log

    def _noise(self, batch: int, device) -> torch.Tensor:
        if self.noise_override is not None:
            n, self.noise_override = self.noise_override, None
            return n.to(device=device, dtype=torch.float32).contiguous()
        return torch.randn(batch, self.action_dim, device=device)

    def forward(self, state):
        """actor.py:215-231.  The sampling head is a HIP kernel outside autograd: under grad mode the outputs carry a
        node whose backward raises (no silent zero gradient); SACTrainer's step produces the actor's gradients."""
        with torch.no_grad():
            loc_scale = self._fc_out(state, evaluations=2)
            B, dev = loc_scale.shape[0], loc_scale.device
            action = torch.empty(B, self.action_dim, device=dev)
            log_prob = torch.empty(B, 1, device=dev)
            squashed_mean = torch.empty(B, self.action_dim, device=dev)
            ops.gaussian_head_forward(loc_scale, self._noise(B, dev), action, log_prob, squashed_mean)
        what = "GaussianFullyConnectedActor.forward"
        return rlt.ActorOutput(action=no_autograd_guard(action, self, what), log_prob=no_autograd_guard(log_prob, self, what),
                               squashed_mean=no_autograd_guard(squashed_mean, self, what))

    def get_log_prob(self, state, squashed_action: torch.Tensor):
        with torch.no_grad():
            loc_scale = self._fc_out(state)
            log_prob = torch.empty(loc_scale.shape[0], 1, device=loc_scale.device)
            a = squashed_action if squashed_action.stride(-1) == 1 else squashed_action.contiguous()
            ops.gaussian_log_prob(loc_scale, a.float(), log_prob)
        return no_autograd_guard(log_prob, self, "GaussianFullyConnectedActor.get_log_prob")


class FullyConnectedActor(ModelBase):
    """Deterministic actor, reagent/models/actor.py:42-110: FC stack whose last activation is
    `action_activation` (tanh) -> ActorOutput(action, log_prob = 0); optional Gaussian exploration
    noise (`exploration_variance`) added and clamped to the training action range (serving path)."""

    def __init__(self, state_dim: int, action_dim: int, sizes: List[int], activations: List[str],
                 use_batch_norm: bool = False, action_activation: str = "tanh",
                 exploration_variance: float = None) -> None:
        super().__init__()
        assert state_dim > 0, "state_dim must be > 0, got {}".format(state_dim)
        assert action_dim > 0, "action_dim must be > 0, got {}".format(action_dim)
        self.state_dim = state_dim
        self.action_dim = action_dim
        assert len(sizes) == len(activations), (
            "The numbers of sizes and activations must match; got {} vs {}".format(len(sizes), len(activations))
        )
        self.action_activation = action_activation
        self.fc = FullyConnectedNetwork([state_dim] + list(sizes) + [action_dim],
                                        list(activations) + [self.action_activation], use_batch_norm=use_batch_norm)
        self.exploration_variance = exploration_variance
        if exploration_variance is not None:
            assert exploration_variance > 0

    def input_prototype(self):
        return rlt.FeatureData(torch.randn(1, self.state_dim))

    def forward(self, state) -> rlt.ActorOutput:
        """actor.py:86-97 (differentiable through the FC stack like the reference module)"""
        action = self.fc(state.float_features)
        batch_size = action.shape[0]
        assert action.shape == (batch_size, self.action_dim), f"{action.shape} != ({batch_size}, {self.action_dim})"
        if self.exploration_variance is None:
            return rlt.ActorOutput(action=action, log_prob=torch.zeros(batch_size, 1, device=action.device))
        action, _, log_prob = self.explore(action)
        return rlt.ActorOutput(action=action, log_prob=log_prob)

    def explore(self, action: torch.Tensor):
        """actor.py:99-110: add N(0, variance) noise (`scale` = variance, as the reference has it) and clamp
        to the training action range.  Returns (noisy action, mask of the entries whose clamp passes the
        gradient, log_prob).  The trainers that evaluate `actor(state).action` inside the step (TD3, CRR)
        call this on the stack's output so that the noise is part of the step as in the reference."""
        batch_size = action.shape[0]
        dist = torch.distributions.Normal(torch.zeros(self.action_dim), torch.ones(self.action_dim) * self.exploration_variance)
        noise = dist.sample((batch_size,))
        log_prob = dist.log_prob(noise).to(action.device).sum(dim=1).view(-1, 1).clamp(LOG_PROB_MIN, LOG_PROB_MAX)
        pre = action + noise.to(action.device)
        passes = (pre >= -1.0) & (pre <= 1.0)  # CONTINUOUS_TRAINING_ACTION_RANGE
        return pre.clamp(-1.0, 1.0), passes, log_prob
