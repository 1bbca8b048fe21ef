"""TEST INFRASTRUCTURE ONLY.

Loads tests/emu/libreagent_emu.so — the product kernel sources compiled for the x86 host against
the SIMT interpreter shim (tests/emu/rg_platform.h) — and offers numpy helpers to call the same
C ABI (include/reagent_hip.h) with host pointers.  The reagent_amd package never imports this.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PREC_F32, PREC_BF16 = 0, 1
DT_F32, DT_BF16 = 0, 1


def build():
    subprocess.run(["make", "-s", "-C", HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        build()
        _LIB = ctypes.CDLL(os.path.join(HERE, "libreagent_emu.so"))
        _LIB.rg_fc_wgrad_workspace_bytes.restype = ctypes.c_size_t
    return _LIB


def f32_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))
    return ((u + r) >> np.uint32(16)).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def ptr(a):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"] or a.ndim <= 1 or True
    return ctypes.c_void_p(a.ctypes.data)


def to_compute(a: np.ndarray, prec: int) -> np.ndarray:
    """fp32 array -> array in the compute element type of `prec` (bf16 as uint16 bits)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if prec == PREC_F32 else f32_to_bf16_bits(a)


def from_compute(a: np.ndarray, prec: int) -> np.ndarray:
    return a if prec == PREC_F32 else bf16_bits_to_f32(a)


def empty_compute(shape, prec):
    return np.zeros(shape, dtype=np.float32 if prec == PREC_F32 else np.uint16)


c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_f = ctypes.c_float
c_d = ctypes.c_double
c_sz = ctypes.c_size_t
