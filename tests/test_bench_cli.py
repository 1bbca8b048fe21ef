"""bench.py's launcher and parity object, without a GPU: `--gpus N` must become N ranks (or fail loudly),
never silently run one; the parity object's CPU-side batch (gather + normalize + input maker restated with
torch indexing) must agree with the device path field by field."""
import json
import os
import subprocess
import sys
import types

import pytest

from conftest import free_port
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


def _measure_worker(rank, world, port, out_dir):
    """one rank of `bench.measure` on the interpreter: the data-parallel control flow of the benchmark itself (shard per rank,
    enable_data_parallel, the eager deferred-update loop, max-over-ranks region time, per-rank report), which otherwise only
    a multi-GPU box runs"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_backend

    emu_backend.install()
    import torch.distributed as dist

    import bench

    torch.cuda.synchronize = lambda *a, **k: None  # (CPU ranks: the barrier's device synchronisation has nothing to wait for)

    class HostEvent:  # the instrumented pass brackets every call with events: host clocks here
        def __init__(self, enable_timing=True):
            self.t = None

        def record(self, stream=None):
            import time

            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    torch.cuda.Event = HostEvent
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    argv = sys.argv
    sys.argv = ["bench.py", "--config", "c2", "--gpus", str(world), "--capacity", "1024", "--batch", "128", "--parity-batch", "128",
                "--hidden", "256", "--layers", "2", "--precision", "bf16", "--steps", "2", "--warmup", "1", "--repeats", "2", "--sustained-steps", "3"]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    dev = torch.device("cpu")
    # a step count a rank derives from its own host timing (the instrumented pass's backlog) is agreed over the ranks before
    # it is used: every step holds an all-reduce, different counts would hang a multi-GPU run
    assert bench.agree_over_ranks(5 + 3 * rank, dev) == 5 + 3 * (world - 1)
    m = bench.measure(args, dev, rank, world, dist)
    # main()'s second pass: the same shard and initial weights in the split-bf16 mode
    import argparse

    a2 = argparse.Namespace(**vars(args))
    a2.precision, a2.repeats = "bf16x3", 1
    ma = bench.measure(a2, dev, rank, world, dist, cols=m["cols"])
    dist.barrier()
    assert m["extra"]["all_reduce_bytes"] > 0 and "roofline" in m["extra"]
    # the long region and its telemetry record (no GPU here: the clock / power fields are None, never an exception)
    assert m["sustained"]["steps"] == 3 and m["sustained"]["ms_per_step"] > 0 and "sclk_mhz" in m["sustained"]
    # the instrumented pass reports what it measured (no rescale to the timed step): off-GPU there is no device-side blocker,
    # the spans are host clocks; the dominant entry point is chosen over ALL its variants, call-weighted
    ip = m["extra"]["instrumented_pass"]
    assert ip["event_ms_per_step_sum"] > 0 and ip["queue_ahead"]["on"] is False and "scale" not in ip
    roof = m["extra"]["roofline"]
    assert roof["kernel"].startswith("rg_mlp_") and len(roof["variants"]) >= 1
    tot = sum(v["launches_per_step"] * v["avg_launch_us"] for v in roof["variants"])
    assert abs(tot / roof["launches_per_step"] - roof["avg_launch_us"]) <= 1e-6 * roof["avg_launch_us"] + 1e-9
    assert m["extra"]["fc_roofline"]["entry_points_us_per_step"]
    # the ONE line main() prints is a digest of the full record: bounded, strict JSON, the contract's keys
    full = {"metric": "transitions/sec at batch=65536 state_dim=128; 1/2/4/8 MI355X scaling", "value": m["value"], "unit": "transitions/s",
            "n_gpus": world, "rccl_ranks": world, "steps": 2, "warmup": 1, "ms_per_step": m["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "w" * 500, "name": "c2", "global_batch": world * 128, "parallelism": f"dp{world}", "launch": m["launch"]},
            "per_rank": m["per_rank"], "sustained": m["sustained"], **m["extra"], "parity": m["parity"],
            "accurate": {"dtype": "bf16x3", "value": ma["value"], "ms_per_step": ma["ms_per_step"], "parity": ma["parity"],
                         "sustained": ma["sustained"], **{k: ma["extra"][k] for k in ("roofline", "fc_roofline") if k in ma["extra"]}},
            "also_measured": {"c3": {"error": "x" * 5000}, "c4": {"dtype": "bf16", "value": 1.0, "ms_per_step": float("nan")}}}
    text = bench.compact_line(full)
    assert len(text) < bench.LINE_LIMIT and "\n" not in text
    line = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))  # no NaN / Infinity tokens
    assert line["value"] == pytest.approx(m["value"], rel=1e-9) and line["n_gpus"] == world and line["config"]["name"] == "c2"
    assert line["roofline"]["frac"] > 0 and line["roofline"]["whole_fc_frac"] > 0 and "traffic" in line["roofline"]
    assert line["compliant"]["dtype"] == "bf16x3" and "executed_frac" in line["compliant"]
    assert len(line["per_rank_ms"]) == world and line["c4"]["ms_per_step"] is None and len(line["c3"]["error"]) <= 160
    if rank == 0:
        assert line["parity"]["gather_fields_bit_exact"] is True and line["compliant"]["parity"]["meets_north_star"] is True
    keep = dict(value=m["value"], ms=m["ms_per_step"], regions=m["region_ms"], per_rank=m["per_rank"], launch=m["launch"],
                loss=m["final_loss"], parity=m["parity"], x3_value=ma["value"], x3_loss=ma["final_loss"],
                x3_parity=ma["parity"], shard=float(m["cols"]["observation"].double().sum()))
    torch.save(keep, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_measure_on_two_ranks(tmp_path, emu_lib):
    import torch.multiprocessing as mp

    port = free_port()
    mp.spawn(_measure_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    # whole-job value: both ranks' transitions over the slowest rank's time — the same number on every rank
    assert r0["value"] == r1["value"] and r0["regions"] == r1["regions"] and len(r0["regions"]) == 2
    assert r0["value"] == pytest.approx(2 * 128 * 2 / (r0["ms"] * 2e-3), rel=1e-6)
    assert [p["rank"] for p in r0["per_rank"]] == [0, 1] and all(p["ms_per_step"] > 0 for p in r0["per_rank"])
    assert all(p["all_reduce_us"] > 0 for p in r0["per_rank"])  # the instrumented pass saw the gradient all-reduce on every rank
    assert "eager" in r0["launch"]  # the replayed graph stays opt-in with more than one rank
    assert r0["shard"] != r1["shard"]  # disjoint shards of the offline data ...
    assert r0["parity"] is not None and r1["parity"] is None and "error" not in r0["parity"], r0["parity"]  # ... rank 0 checks parity
    assert r0["x3_parity"]["meets_north_star"], r0["x3_parity"]
    for r in (r0, r1):
        assert r["loss"] == r["loss"] and r["x3_loss"] == r["x3_loss"] and r["x3_value"] > 0  # finite


@pytest.mark.parametrize("config,precision", [("c3", "bf16x3"), ("c3", "bf16"), ("c4", "bf16x3")])
def test_parity_leg_of_the_other_configurations(emu_lib, monkeypatch, config, precision):
    """bench.parity_check for C3 / C4 at interpreter sizes: C3's compares what the STEP computed on the grouped engine
    (logged-action quantiles of every row, next-state per-action means), in split-bf16 mode within north_star's 1e-4"""
    import bench

    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", config, "--precision", precision, "--capacity", "1024",
                                      "--batch", "128", "--parity-batch", "192", "--hidden", "256", "--layers", "2"])
    args = bench.parse()
    if config == "c3":
        args.atoms = 24
    dev = torch.device("cpu")
    _, _, init, cols, norm = bench.build(args, dev, 0, batch=128)
    out = bench.parity_check(args, dev, init, cols, norm)
    assert out["gather_fields_bit_exact"], out
    if config == "c3":
        assert out["path"].startswith("grouped engine") and out["dq_rows"] == 192, out
    if precision == "bf16x3":
        assert out["meets_north_star"] and out["rel_dloss"] <= 1e-4, out
    else:
        assert out["sane"], out


def test_numa_binding_never_raises_and_reports_why(monkeypatch, tmp_path):
    """bench.bind_rank_to_gpu_numa: a box without GPUs / sysfs topology is left alone with a reason; with a topology the ranks
    whose GPUs share a NUMA node split its cores in local-rank order (fake sysfs + fake device properties)."""
    import types

    sys.path.insert(0, ROOT)
    import bench

    r = bench.bind_rank_to_gpu_numa(0, 8)
    assert r["bound"] is False and "reason" in r
    # fake topology: GPUs 0-3 on node 0 (cpus 0-7), GPUs 4-7 on node 1 (cpus 8-15)
    props = lambda i: types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0x10 + i, pci_device_id=0)  # noqa: E731
    monkeypatch.setattr(bench.torch.cuda, "get_device_properties", props)
    real_open = open

    def fake_open(path, *a, **k):
        if isinstance(path, str) and path.startswith("/sys/bus/pci/devices/0000:"):
            i = int(path.split(":")[1], 16) - 0x10
            text = str(i // 4) if path.endswith("numa_node") else ("0-7" if i < 4 else "8-15")
            f = tmp_path / f"f{i}_{path.split('/')[-1]}"
            f.write_text(text + "\n")
            return real_open(f, *a, **k)
        return real_open(path, *a, **k)

    monkeypatch.setattr("builtins.open", fake_open)
    monkeypatch.setattr(bench.os, "sched_getaffinity", lambda pid: set(range(16)))
    bound = {}
    monkeypatch.setattr(bench.os, "sched_setaffinity", lambda pid, cpus: bound.setdefault("cpus", list(cpus)))
    r = bench.bind_rank_to_gpu_numa(5, 8)
    assert r == {"bound": True, "numa_node": 1, "cpus": 2, "first_cpu": 10, "ranks_on_node": 4} and bound["cpus"] == [10, 11]


REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")


def test_line_is_compact():
    """Round 5's driver record was `parsed: null`: the line had grown to 35.5 KB.  The line is now a digest of the full record
    (which goes to a file): for the largest record the benchmark has produced (round 5's default run, every sub-measurement
    present) it stays under 4 KB, is strict JSON, carries the contract's keys, `roofline` and `cpu_baseline`, and the
    1e-4-compliant mode as a co-headline."""
    sys.path.insert(0, ROOT)
    import bench

    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_report_r05.json")))
    text = bench.compact_line(full)
    assert len(text.encode()) < 4096 and "\n" not in text
    line = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    for k in REQUIRED_KEYS:
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-9) and line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-9)
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us")) <= set(line["roofline"])
    assert line["roofline"]["frac"] == pytest.approx(line["roofline"]["achieved"] / line["roofline"]["peak"], rel=1e-4)
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"]) and "model" not in line["config"]
    assert line["compliant"]["dtype"] == "bf16x3" and line["compliant"]["parity"]["meets_north_star"] is True
    assert line["parity"]["meets_north_star"] is False  # plain bf16 is outside north_star's 1e-4 and the line says so
    assert line["c3"]["full_size_check"].startswith("grouped vs dense fp32") and line["c4"]["compliant"]["parity"]["ok"] is True
    # a record bloated beyond anything real still yields a parseable line: optional digests are shed, the contract's keys stay
    full["also_measured"]["c3"]["workload"] = "x" * 10000
    full["config"]["workload"] = "y" * 10000
    for i in range(64):
        full["also_measured"][f"c{i + 10}"] = full["also_measured"]["c4"]
    text = bench.compact_line(full)
    assert len(text) < 4096
    line = json.loads(text)
    assert all(k in line for k in REQUIRED_KEYS)


def test_report_file_holds_the_full_record(tmp_path):
    sys.path.insert(0, ROOT)
    import bench

    full = {"value": 1.5, "nested": {"nan": float("nan"), "list": [1, float("inf")]}, "cpu_baseline": {"value": 2.0}}
    p = bench.write_report(full, str(tmp_path / "r.json"))
    back = json.load(open(p), parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert back == {"value": 1.5, "nested": {"nan": None, "list": [1, None]}, "cpu_baseline": {"value": 2.0}}


def test_two_ranks_under_torchrun_print_one_compact_line(tmp_path):
    """`bench.py --gpus 2 --config c2` end to end as the driver launches it — `python -m torch.distributed.run --nproc-per-node 2
    ... --gpus 2` — on gloo + the host-compiled kernels (tests/bench_on_emu.py): rank 0 prints exactly one compact line,
    the other rank prints nothing on stdout, the line describes a 2-rank group and names a report file that holds the
    per-rank objects."""
    report = tmp_path / "report.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "bench_on_emu.py"), "--gpus", "2", "--config", "c2",
           "--capacity", "1024", "--batch", "128", "--parity-batch", "128", "--hidden", "128", "--layers", "2", "--steps", "2",
           "--warmup", "1", "--repeats", "2", "--sustained-steps", "2", "--report", str(report)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    out = [l for l in lines if l.startswith("{")]  # (library banners such as gloo's "[Gloo] Rank 0 is connected ..." aside)
    assert len(out) == 1 and lines[-1] == out[0], lines  # ONE JSON line in all, the last line of the job: rank 1 printed none
    assert sum('"metric"' in l for l in lines) == 1, lines
    assert len(out[0].encode()) < 4096
    line = json.loads(out[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["config"]["parallelism"] == "dp2"
    assert line["config"]["global_batch"] == 256 and line["scaling"] == "weak" and len(line["per_rank_ms"]) == 2
    assert line["value"] == pytest.approx(2 * 128 / (line["ms_per_step"] * 1e-3), rel=1e-6)
    assert "cpu_baseline" not in line and "compliant" not in line  # rank 0 at N == 1 only; secondary regions on one rank only
    assert line["all_reduce_us_max"] > 0 and line["roofline"]["frac"] > 0
    full = json.load(open(report))
    assert [p["rank"] for p in full["per_rank"]] == [0, 1] and full["all_reduce_bytes"] > 0
    assert "eager" in full["config"]["launch"]
