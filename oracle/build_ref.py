"""TEST INFRASTRUCTURE (oracle).  Recipe for `oracle/_ref/`: the UNMODIFIED reference, byte-compiled.

The reference on this path is Python.  `python -m oracle.build_ref` imports the reference modules the hot path uses
(through `oracle/stubs.py`, exactly as `reference_harness.py` does), records which files under /root/reference those
imports loaded, and compiles each one FROM WHERE IT LIES with `py_compile` into `oracle/_ref/<same relative path>.pyc`
— sourceless byte code, the Python analogue of compiling a C reference into `oracle/_ref/*.so`.  No reference source is
copied into the repository; `oracle/_ref/` is git-ignored (not gpurun-ignored), so the byte code travels to the GPU
box like the built `.so` files, where `stubs.install()` finds it when /root/reference does not exist and `bench.py`'s
`cpu_baseline` leg times the real reference trainer (`kind: "reference"`).  Same interpreter on both sides (this image).

Run by `__graft_entry__.build()` whenever /root/reference is present.
"""
import importlib
import json
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

# what bench.py's cpu_baseline leg and the harness import (each pulls in its own dependencies)
ENTRY_MODULES = (
    "reagent.core.types", "reagent.core.parameters", "reagent.optimizer.union", "reagent.optimizer.soft_update",
    "reagent.models.dqn", "reagent.models.critic", "reagent.models.actor", "reagent.models.dueling_q_network",
    "reagent.models.fully_connected_network",
    "reagent.training.dqn_trainer", "reagent.training.qrdqn_trainer", "reagent.training.sac_trainer",
    "reagent.replay_memory.circular_replay_buffer", "reagent.preprocessing.preprocessor",
    "reagent.preprocessing.identify_types", "reagent.preprocessing.normalization",
    "reagent.gym.preprocessors.trainer_preprocessor",
)


def build(out: str = OUT, quiet: bool = False) -> int:
    from . import stubs

    src_root = os.path.realpath("/root/reference")
    if not os.path.isdir(os.path.join(src_root, "reagent")):
        raise SystemExit("oracle.build_ref: /root/reference is not here (the recipe runs in the build container only)")
    if os.path.realpath(stubs.REFERENCE_ROOT) != src_root:
        raise SystemExit("oracle.build_ref: the reference is already being imported from a built _ref")
    stubs.install_gym()
    for m in ENTRY_MODULES:
        importlib.import_module(m)
    files = {}
    for name, mod in list(sys.modules.items()):
        f = getattr(mod, "__file__", None)
        if f and f.endswith(".py") and os.path.realpath(f).startswith(src_root + os.sep):
            files[os.path.relpath(os.path.realpath(f), src_root)] = name
    # packages whose __init__ the stubs bypass still need to exist as (empty-path) packages when imported from _ref: the
    # stubs create them as module objects, so nothing is compiled for them
    if os.path.isdir(out):
        shutil.rmtree(out)
    for rel in sorted(files):
        dst = os.path.join(out, rel + "c")  # x.py -> x.pyc next to where the source would be (sourceless import)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        py_compile.compile(os.path.join(src_root, rel), cfile=dst, dfile=os.path.join("/root/reference", rel), doraise=True,
                           optimize=0)
    with open(os.path.join(out, "MANIFEST.json"), "w") as fh:
        json.dump({"built_from": "/root/reference", "python": sys.version.split()[0], "modules": len(files),
                   "files": sorted(files)}, fh, indent=1)
    if not quiet:
        print(f"oracle/_ref: {len(files)} reference modules byte-compiled (python {sys.version.split()[0]})")
    return len(files)


if __name__ == "__main__":
    build()
