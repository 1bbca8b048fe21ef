"""The shipped shared library loads and exports every symbol include/reagent_hip.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import reagent_amd._lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "reagent_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rg_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_and_exports_header_symbols():
    assert os.path.exists(L.LIB_PATH), "run `make -C reagent_amd/csrc` (or __graft_entry__.build())"
    lib = ctypes.CDLL(L.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in reagent_hip.h but not exported"


def test_binding_table_matches_header():
    assert sorted(L.SIGNATURES) == _declared()


def test_strerror_and_version_callable_without_gpu():
    lib = L.load()
    assert lib.rg_abi_version() >= 1
    assert lib.rg_strerror(0) == b"ok"
    assert b"invalid" in lib.rg_strerror(-1)


def test_missing_library_fails_loudly(tmp_path):
    import pytest

    with pytest.raises(L.ReagentHipError, match="no CPU fallback"):
        L.load(str(tmp_path / "nope.so"))


def test_cpu_tensor_is_rejected_not_silently_computed():
    import pytest
    import torch

    from reagent_amd import ops

    x = torch.zeros(4, 4)
    with pytest.raises(L.ReagentHipError, match="no CPU fallback"):
        ops.transpose_cast(x, torch.zeros(4, 4), None)
