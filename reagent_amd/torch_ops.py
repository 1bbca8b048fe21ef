"""`torch.ops.reagent_amd.*`: the C-ABI entry points a caller is most likely to use on their own, registered as
PyTorch custom ops (torch.library) with tensor signatures — BASELINE.json's "thin C-ABI layer (PyTorch-ROCm custom
ops)".  Each op is a few lines over `reagent_amd.ops` (ctypes on torch's device pointers and current HIP stream);
they are registered for the CUDA (= HIP on ROCm) dispatch key only, so a CPU tensor fails in the dispatcher instead
of reaching a fallback.  Shape functions (register_fake) make them traceable.  The trainers call `ops` directly.

    import reagent_amd.torch_ops            # registers the library
    q = torch.ops.reagent_amd.mlp_forward(x, weights, biases, ["relu", "relu", "linear"], "bf16")
    torch.ops.reagent_amd.adam_step_(p, g, m, v, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=t)
"""
import math
from typing import List, Tuple

import torch

from . import _lib as L
from . import ops

_PREC = {"f32": L.PREC_F32, "bf16": L.PREC_BF16, "bf16x3": L.PREC_BF16X3}
_lib = torch.library.Library("reagent_amd", "DEF")
_impls = {}  # name -> python implementation (the suite's interpreter backend serves them from the test side)


def _define(schema: str, fn, fake=None):
    name = schema.split("(")[0]
    _lib.define(schema)
    _lib.impl(name, fn, "CUDA")
    _impls[name] = fn
    if fake is not None:
        torch.library.register_fake(f"reagent_amd::{name}", fake)


# ---- fully-connected stacks (fully_connected_network.py:157-163) ------------------------------------------------
_stacks = {}
# The package's in-place ops write through kernels: neither torch's version counter nor the trainers' `_rg_version`
# (which lives on Parameter objects, not on the `.detach()` views a caller may pass) sees them.  They therefore record a
# write epoch per STORAGE, and a cached stack whose weights' storages were written since its last staging re-stages.
# Only storages a cached stack reads are tracked (a write to anything else needs no record), the table is dropped with the
# stack cache, and a tracked address whose storage has died is re-armed when a stack is built on it — so the table is
# bounded by the cache (64 stacks) and an address recycled by the allocator cannot hand a stale epoch to an unrelated tensor.
_write_epoch = {}


def _note_write(*tensors: torch.Tensor) -> None:
    for t in tensors:
        k = t.untyped_storage().data_ptr()
        if k in _write_epoch:
            _write_epoch[k] += 1


def _track(tensors) -> None:
    for t in tensors:
        _write_epoch.setdefault(t.untyped_storage().data_ptr(), 0)


def _epochs(tensors) -> tuple:
    return tuple(_write_epoch.get(t.untyped_storage().data_ptr(), 0) for t in tensors)


def _mlp_forward(x: torch.Tensor, weights: List[torch.Tensor], biases: List[torch.Tensor], activations: List[str],
                 precision: str) -> torch.Tensor:
    from .engine import make_stack

    key = (tuple((w.data_ptr(), tuple(w.shape)) for w in weights) + tuple(b.data_ptr() for b in biases)
           + (tuple(activations), precision, str(x.device)))
    st = _stacks.get(key)
    if st is None:
        if len(_stacks) > 64:
            _stacks.clear()
            _write_epoch.clear()
        st = _stacks[key] = make_stack(weights, biases, [L.ACT[a] for a in activations], _PREC[precision])
        st._rg_write_epochs = None
        _track(list(weights) + list(biases))
    ep = _epochs(list(weights) + list(biases))
    # re-staged only when a weight's version counter moved or one of this package's in-place ops wrote its storage
    st.stage_weights(need_transposed=False, force=st._rg_write_epochs != ep)
    st._rg_write_epochs = ep
    xc, _ = st.stage_input(x if x.dtype in (torch.float32, torch.bfloat16) else x.float(), need_transposed=False)
    out = torch.empty(x.shape[0], weights[-1].shape[0], dtype=torch.float32, device=x.device)
    st.forward(xc, out, save=False)
    return out


_define("mlp_forward(Tensor x, Tensor[] weights, Tensor[] biases, str[] activations, str precision) -> Tensor", _mlp_forward,
        lambda x, weights, biases, activations, precision: x.new_empty(x.shape[0], weights[-1].shape[0], dtype=torch.float32))


# ---- optimizer (torch.optim.Adam, reagent/optimizer/soft_update.py:60-70) -----------------------------------------
def _adam_step_(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, lr: float,
                beta1: float, beta2: float, eps: float, weight_decay: float, step: int, grad_scale: float = 1.0) -> None:
    assert param.is_contiguous() and grad.is_contiguous() and exp_avg.is_contiguous() and exp_avg_sq.is_contiguous()
    ops.adam_step(param, grad, exp_avg, exp_avg_sq, param.numel(), lr, beta1, beta2, eps, weight_decay,
                  1.0 - beta1 ** step, math.sqrt(1.0 - beta2 ** step), grad_scale)
    _note_write(param)


_define("adam_step_(Tensor(a!) param, Tensor grad, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, float lr, float beta1, "
        "float beta2, float eps, float weight_decay, int step, float grad_scale=1.0) -> ()", _adam_step_,
        lambda *a, **k: None)


def _soft_update_(target: torch.Tensor, source: torch.Tensor, tau: float) -> None:
    assert target.is_contiguous() and source.is_contiguous() and target.numel() == source.numel()
    ops.soft_update(target, source, target.numel(), tau)
    _note_write(target)


_define("soft_update_(Tensor(a!) target, Tensor source, float tau) -> ()", _soft_update_, lambda *a, **k: None)


# ---- heads ------------------------------------------------------------------------------------------------------
def _gaussian_head(loc_scale: torch.Tensor, noise: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """GaussianFullyConnectedActor.forward after the FC stack (actor.py:215-231): (action, log_prob [B, 1])"""
    B, A = noise.shape
    action = torch.empty(B, A, dtype=torch.float32, device=noise.device)
    log_prob = torch.empty(B, dtype=torch.float32, device=noise.device)
    ops.gaussian_head_forward(loc_scale, noise, action, log_prob, None)
    return action, log_prob.view(B, 1)


_define("gaussian_head(Tensor loc_scale, Tensor noise) -> (Tensor, Tensor)", _gaussian_head,
        lambda loc_scale, noise: (noise.new_empty(noise.shape), noise.new_empty(noise.shape[0], 1)))


def _dueling_combine(value: torch.Tensor, advantage: torch.Tensor, num_actions: int, num_atoms: int) -> torch.Tensor:
    """dueling_q_network.py:93-103: value + advantage - mean(advantage over actions and atoms)"""
    q = torch.empty_like(advantage)
    ops.dueling_combine(value, advantage, num_actions, num_atoms, q)
    return q


_define("dueling_combine(Tensor value, Tensor advantage, int num_actions, int num_atoms) -> Tensor", _dueling_combine,
        lambda value, advantage, num_actions, num_atoms: torch.empty_like(advantage))


def _max_q_values_with_target(q_next_online: torch.Tensor, q_next_target: torch.Tensor, possible_actions_mask: torch.Tensor,
                              double_q: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """DQNTrainerBaseLightning.get_max_q_values_with_target (dqn_trainer_base.py:33-77): (max_q [B, 1], argmax [B, 1]) —
    double-Q: the online net picks among the possible actions, the target net evaluates"""
    B, A = q_next_target.shape
    f = dict(dtype=torch.float32, device=q_next_target.device)
    zq, z1, o1 = torch.zeros(B, A, **f), torch.zeros(B, **f), torch.ones(B, **f)
    dq, parts = torch.empty(B, A, **f), torch.empty(ops.dqn_head_partials(B), **f)
    next_q, next_idx = torch.empty(B, **f), torch.empty(B, dtype=torch.int64, device=q_next_target.device)
    ops.dqn_head(zq, q_next_online.contiguous(), q_next_target.contiguous(), zq, possible_actions_mask.float().contiguous(),
                 z1, None, o1, 1.0, None, double_q, L.LOSS["mse"], dq, parts, next_q=next_q, next_idx=next_idx)
    return next_q.view(B, 1), next_idx.view(B, 1)


_define("max_q_values_with_target(Tensor q_next_online, Tensor q_next_target, Tensor possible_actions_mask, bool double_q) "
        "-> (Tensor, Tensor)", _max_q_values_with_target,
        lambda q_next_online, q_next_target, possible_actions_mask, double_q:
        (q_next_target.new_empty(q_next_target.shape[0], 1),
         q_next_target.new_empty(q_next_target.shape[0], 1, dtype=torch.long)))
