"""bench.py's launcher and parity object, without a GPU: `--gpus N` must become N ranks (or fail loudly),
never silently run one; the parity object's CPU-side batch (gather + normalize + input maker restated with
torch indexing) must agree with the device path field by field."""
import json
import os
import subprocess
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*flags, timeout=300):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, BENCH, *flags], capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.skipif(torch.cuda.is_available(), reason="the no-GPU behaviour of the launcher")
def test_gpus_flag_fails_loudly_without_gpus():
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0
    assert "needs an MI355X" in (r.stderr + r.stdout)
    r1 = _run("--steps", "1", "--warmup", "0")
    assert r1.returncode != 0 and "needs an MI355X" in (r1.stderr + r1.stdout)


def test_gpus_flag_launches_that_many_ranks():
    r = _run("--gpus", "2", "--rendezvous-only")
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["ranks"] == 2 and out["n_gpus"] == 2 and out["rendezvous"] == "ok"


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--rendezvous-only"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


@pytest.mark.parametrize("config", ["c2", "c4"])
def test_parity_object_on_the_interpreter(emu_lib, config):
    sys.path.insert(0, ROOT)
    import bench

    argv = sys.argv
    sys.argv = ["bench.py", "--config", config, "--capacity", "2048", "--parity-batch", "128", "--hidden", "64",
                "--layers", "2", "--precision", "f32"]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    dev = torch.device("cpu")
    loop, trainer, init, cols, norm = bench.build(args, dev, 0, batch=128)
    p = bench.parity_check(args, dev, init, cols, norm)
    assert "error" not in p, p
    assert p["gather_fields_bit_exact"] and p["max_abs_dstate"] <= 1e-5, p
    assert p["ok"], p
    assert bench.fc_flops(types.SimpleNamespace(algo="dqn", state_dim=128, actions=16, atoms=None, hidden=512, layers=3), 65536) \
        == 2 * 65536 * 2924544  # 5.849 MFLOP / transition
    sac = types.SimpleNamespace(algo="sac", state_dim=256, actions=32, atoms=None, hidden=512, layers=3)
    assert bench.fc_flops(sac, 1) == 2 * 10131456  # 20.26 MFLOP / transition
    qr = types.SimpleNamespace(algo="qrdqn", state_dim=128, actions=16, atoms=200, hidden=512, layers=3)
    assert bench.fc_flops(qr, 1) == 2 * 11075584  # 22.15 MFLOP / transition
