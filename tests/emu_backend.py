"""TEST INFRASTRUCTURE ONLY: route reagent_amd's C-ABI calls to the host-compiled kernel sources
(tests/emu/libreagent_emu.so) so the package's HOST logic (engine, optimizers, trainers, replay
buffer bookkeeping) can be exercised by `pytest -m "not gpu"` in a container without a GPU.
The product package has no knowledge of this; the patching happens here, from the test side."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


def install(monkeypatch=None):
    import emulib

    import reagent_amd._lib as L

    cdll = emulib.lib()
    for name, (res, args) in L.SIGNATURES.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    if monkeypatch is not None:
        monkeypatch.setattr(L, "_lib", cdll)
        monkeypatch.setattr(L, "require_cuda", lambda t, name="tensor": None)
        monkeypatch.setattr(L, "stream_ptr", lambda: None)
    else:
        L._lib = cdll
        L.require_cuda = lambda t, name="tensor": None
        L.stream_ptr = lambda: None
    return cdll


_cpu_library = None


def serve_torch_ops_on_cpu():
    """The registered `torch.ops.reagent_amd.*` ops on CPU tensors, for the interpreter backend: a torch.library fragment
    created HERE, on the test side, adds a CPU implementation (the package's own python functions, which then reach the
    patched loader).  The product registers the CUDA (= HIP) key only and carries no dispatch-key switch."""
    global _cpu_library
    if _cpu_library is not None:
        return
    import torch

    import reagent_amd.torch_ops as T

    _cpu_library = torch.library.Library("reagent_amd", "FRAGMENT")
    for name, fn in T._impls.items():
        _cpu_library.impl(name, fn, "CPU")
