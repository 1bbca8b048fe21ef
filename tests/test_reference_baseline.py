"""bench.py's `cpu_baseline` leg (oracle/reference_bench.py): the unmodified reference timed on the host cores — from the
source tree in the build container and from its byte code (oracle/_ref, oracle/build_ref.py) where that is all there is."""
import json
import os
import subprocess
import sys

import pytest

from oracle import stubs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys
sys.path.insert(0, %r)
import torch
from reagent_amd import synthetic
from oracle import reference_bench as RB, stubs
algo = sys.argv[1]
S, A, H, Ly, C, B = 16, 4, 32, 2, 4096, 256
acts = ["relu"] * Ly + ["linear"]
cols = synthetic.replay_contents(C, S, A, seed=100)
if algo == "sac":
    g = torch.Generator().manual_seed(1)
    cols["action"] = torch.rand(C, A, generator=g) * 1.8 - 0.9
    del cols["possible_actions_mask"]
    init = [synthetic.fc_init([S] + [H] * Ly + [2 * A], acts, 40), synthetic.fc_init([S + A] + [H] * Ly + [1], acts, 41),
            synthetic.fc_init([S + A] + [H] * Ly + [1], acts, 42)]
    atoms = None
else:
    atoms = 5 if algo == "qrdqn" else None
    init = [synthetic.fc_init([S] + [H] * Ly + [A * (atoms or 1)], acts, 40)]
best, tried = RB.run(algo, S, A, H, Ly, atoms, C, B, init, cols, synthetic.normalization_table(S, 7), steps=2)
import reagent
print(json.dumps(dict(best=best, root=stubs.runtime_root(), reagent=reagent.__file__)))
""" % ROOT


def _run(algo, env):
    out = subprocess.run([sys.executable, "-c", SCRIPT, algo], capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.skipif(not stubs.reference_available(), reason="needs /root/reference (build container)")
@pytest.mark.parametrize("algo", ["dqn", "qrdqn", "sac"])
def test_reference_loop_runs_from_its_byte_code(algo, tmp_path):
    """the recipe's output is enough to run the whole reference loop where /root/reference does not exist"""
    if not os.path.isfile(os.path.join(stubs.BUILT_ROOT, "MANIFEST.json")):
        subprocess.run([sys.executable, "-m", "oracle.build_ref"], cwd=ROOT, check=True, capture_output=True)
    env = dict(os.environ, REAGENT_REFERENCE_ROOT=str(tmp_path / "absent"))
    r = _run(algo, env)
    assert r["root"] == stubs.BUILT_ROOT and r["reagent"].endswith("__init__.pyc")
    assert r["best"]["ms_per_step"] > 0 and r["best"]["train_ms"] > 0 and r["best"]["sample_ms"] > 0


def test_ref_directory_holds_no_source():
    """oracle/_ref is byte code only (reference sources are never copied into the repository) and git-ignored"""
    if os.path.isdir(stubs.BUILT_ROOT):
        for d, _, files in os.walk(stubs.BUILT_ROOT):
            for f in files:
                assert f.endswith(".pyc") or f == "MANIFEST.json", os.path.join(d, f)
    assert "oracle/_ref/" in open(os.path.join(ROOT, ".gitignore")).read().split()
    ign = os.path.join(ROOT, ".gpurunignore")
    assert not os.path.exists(ign) or "oracle/_ref" not in open(ign).read()
