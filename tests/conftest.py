import os
import sys
from dataclasses import dataclass

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The class under test is the package's OWN restatement of the rlt containers on every box (build container and GPU box
# alike): core/types.py would otherwise hand out the reference's classes wherever `reagent` happens to be importable.
# The tests of that re-export strip the variable for their subprocesses.
os.environ.setdefault("REAGENT_AMD_OWN_TYPES", "1")


def pytest_sessionstart(session):
    """GPU box only, once per session (not in xdist workers): the device preflight BEFORE this process touches the device.
    A node whose first touch faults under the default runtime settings (seen about once in ten leases on this pool) is tried
    under the alternatives of reagent_amd.device_preflight.ALTERNATIVES; a working one is exported for this process.  A node
    that faults under all of them is named here, so that the tail of a `-x` run blames the node, not the first test."""
    if os.environ.get("PYTEST_XDIST_WORKER") or os.environ.get("RG_SKIP_PREFLIGHT"):
        return
    markexpr = getattr(session.config.option, "markexpr", "") or ""
    if "not gpu" in markexpr:
        return
    if not os.path.exists("/dev/kfd"):  # no AMD GPU driver node: a CPU box.  (NOT torch.cuda.is_available(): that call starts the
        return                          # HSA runtime of THIS process, after which an adopted runtime setting would come too late)
    from reagent_amd.device_preflight import NODE_FAULT, settle

    ok, log, adopted = settle()
    if adopted:
        print(f"\n[conftest] {log.splitlines()[-1]}")
    if not ok and not log.endswith("no GPU visible to torch"):
        pytest.exit(f"{NODE_FAULT}\n{log}", returncode=97)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """A CPU-only run (the kernels on the SIMT interpreter: 8 minutes serially) spreads over four worker processes when
    pytest-xdist is installed and the caller did not pass -n: about 2 minutes.  A box with a GPU runs serially (one
    device); RG_TEST_SERIAL=1 or `-n 0` keep a CPU run serial."""
    try:
        if (os.environ.get("RG_TEST_SERIAL") or os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput")
                or not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None) is not None
                or getattr(config.option, "collectonly", False) or getattr(config.option, "usepdb", False)):
            return None
        if os.path.exists("/dev/kfd"):  # a GPU box (not torch.cuda.is_available(): see pytest_sessionstart)
            return None
        config.option.numprocesses = min(4, os.cpu_count() or 1)
    except Exception:  # never let the convenience break a run
        pass
    return None


def free_port() -> int:
    """a TCP port nobody listens on right now (rendezvous of the world-2 gloo tests: ports derived from the pid collide
    between pytest-xdist workers, whose pids are neighbours)"""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@dataclass
class Backend:
    name: str
    device: str


@pytest.fixture(params=[pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    """Where the C ABI executes.

    hip : the product path — libreagent_hip.so on a real MI355X (`-m gpu`).
    emu : TEST INFRASTRUCTURE — the same kernel sources compiled for the host against the SIMT
          interpreter in tests/emu, patched in from the test side (tests/emu_backend.py) so the
          kernels' index arithmetic and all host logic are checked by `-m "not gpu"` without a GPU.
    """
    if request.param == "emu":
        import emu_backend

        emu_backend.install(monkeypatch)
        return Backend("emu", "cpu")
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import reagent_amd._lib as L

    L.lib()  # fail loudly if the HIP extension is missing
    return Backend("hip", "cuda")


@pytest.fixture
def emu_lib(monkeypatch):
    """The host-compiled kernel library patched into reagent_amd for this test process (CPU only)."""
    import emu_backend

    return emu_backend.install(monkeypatch)
