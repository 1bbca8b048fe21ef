"""Tensor utilities of reagent/core/torch_utils.py that the native trainers expose for logging and
evaluation callers (the training step evaluates the same expressions inside its fused heads)."""
import torch


def masked_softmax(x: torch.Tensor, mask: torch.Tensor, temperature: float) -> torch.Tensor:
    """Softmax of x / temperature over the entries `mask` keeps, rows with nothing kept -> zeros
    (the arithmetic, operation for operation, of reagent/core/torch_utils.py:62-73)."""
    logits = x / temperature - (1.0 - mask) * 1e20  # dropped entries far below any kept one
    weights = (logits - logits.max(dim=1, keepdim=True).values).exp() * mask
    probs = weights / weights.sum(dim=1, keepdim=True)
    return torch.where(torch.isnan(probs), torch.zeros_like(probs), probs)  # 0 / 0 of a fully masked row
