"""Tensor utilities of reagent/core/torch_utils.py that the native trainers expose for logging and
evaluation callers (the training step evaluates the same expressions inside its fused heads)."""
import torch


def masked_softmax(x: torch.Tensor, mask: torch.Tensor, temperature: float) -> torch.Tensor:
    """reagent/core/torch_utils.py:62-73"""
    x = x / temperature
    mask_min_x = x - ((1.0 - mask) * 1e20)
    mask_min_x = mask_min_x - torch.max(mask_min_x, dim=1, keepdim=True)[0]
    e_x = torch.exp(mask_min_x) * mask
    out = e_x / e_x.sum(dim=1, keepdim=True)
    out[out != out] = 0  # a fully masked row
    return out
