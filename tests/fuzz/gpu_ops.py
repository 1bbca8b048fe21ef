"""TEST INFRASTRUCTURE: lets the fuzzers that reach the kernels through `reagent_amd.ops` alone run on the REAL library.

`FUZZ_ON_GPU=1 python tests/fuzz/fuzz_heads.py ...` — the fuzzer keeps building its case and its float64 reference on the host;
every `ops.<name>(...)` call moves its tensor arguments to cuda:0, runs the real entry point, and copies every tensor argument back
into the host tensor it came from (outputs are caller-provided tensors in this API; copying inputs back changes nothing)."""
import os

import torch

ON_GPU = bool(os.environ.get("FUZZ_ON_GPU"))


def select():
    """-> the `ops` object a fuzzer should call: the interpreter-backed module (default) or the GPU pass-through"""
    if not ON_GPU:
        import emu_backend

        emu_backend.install()
        from reagent_amd import ops

        return ops
    assert torch.cuda.is_available(), "FUZZ_ON_GPU=1 needs the MI355X"
    from reagent_amd import ops

    dev = torch.device("cuda", 0)

    class GpuOps:
        def __getattr__(self, name):
            fn = getattr(ops, name)
            if not callable(fn):
                return fn

            def call(*args, **kw):
                pairs = []

                def up(a):
                    if isinstance(a, torch.Tensor) and a.device.type == "cpu":
                        d = a.to(dev)
                        pairs.append((a, d))
                        return d
                    return a

                res = fn(*[up(a) for a in args], **{k: up(v) for k, v in kw.items()})
                torch.cuda.synchronize()
                for host, d in pairs:
                    host.copy_(d.cpu())

                def down(r):
                    if isinstance(r, torch.Tensor):
                        return r.cpu()
                    if isinstance(r, (tuple, list)):
                        return type(r)(down(x) for x in r)
                    return r

                return down(res)

            return call

    return GpuOps()


def device():
    """for the fuzzers that work through the host classes (ReplayBuffer, Preprocessor, trainers): the interpreter behind "cpu"
    (default) or, with FUZZ_ON_GPU=1, the real library on "cuda" """
    if not ON_GPU:
        import emu_backend

        emu_backend.install()
        return "cpu"
    assert torch.cuda.is_available(), "FUZZ_ON_GPU=1 needs the MI355X"
    return "cuda"
