"""reagent_amd — MI355X (gfx950) native batch-RL training step behind ReAgent's trainer API.

Hot path only: replay gather -> dense normalization -> FullyConnected Q/policy
forward+backward -> TD / quantile / SAC losses -> fused Adam + soft target update, as hand-written
HIP kernels in reagent_amd/csrc behind the C ABI of include/reagent_hip.h.  No CPU fallback.
"""
from ._lib import PREC_BF16, PREC_BF16X3, PREC_F32, ReagentHipError  # noqa: F401

__version__ = "0.1.0"
