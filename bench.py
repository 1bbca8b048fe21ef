#!/usr/bin/env python3
"""bench.py — transitions/sec of the batch-RL hot path on N MI355X of one node.

Default workload = BASELINE.json configs[1] (`--config c2`): Discrete DQN, state_dim=128, |A|=16, 3x512
MLP, batch=65536, bf16.  `--config c3` = QR-DQN with 200 quantiles, `--config c4` = SAC (S=256, A=32,
actor + twin critics, H=3x512), `--config c5` = c2 per rank on 8 ranks (global batch 524288).

One "step" = one pass of the whole hot path over one minibatch, everything already resident in HBM:
replay index sampling -> gather (rg_replay_*) -> input maker -> dense normalization x2 -> FC forwards
+ loss head + FC backward(s) -> [N>1: RCCL all-reduce of the flat fp32 gradient slab] -> fused Adam +
soft target update.

`--gpus N` with N > 1 and no torchrun environment re-launches itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` (one rank per
GPU over RCCL); under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE and checks WORLD_SIZE == --gpus.
Every rank owns a disjoint shard of the offline dataset (its own replay buffer), the per-rank batch
stays 65536 (weak scaling) and the only collective is the gradient all-reduce.

Prints ONE JSON line (rank 0), at most LINE_LIMIT = 4096 bytes: the contract's keys + `roofline` + `cpu_baseline` + digests of
`parity`, of the 1e-4-compliant bf16x3 mode (`compliant`, the co-headline) and of the other single-GPU configurations (`c3`,
`c4`) — see compact_line().  The FULL record (every object below, every digit) goes to bench_report.json (--report).
Objects of the full record:
  roofline     — dominant FC entry point (largest total time, all its variants merged, call-weighted):
                 algorithmic FLOP of its launches / HIP-event time of those launches (events on the launch
                 stream, in a second, instrumented pass so the timed region itself stays un-instrumented;
                 each instrumented step is enqueued behind a device-side blocker — class QueueAhead — so the
                 spans are kernel durations, not host gaps, and agree with rocprofv3's kernel trace in
                 profiles/); `traffic` = HBM bytes per launch from the PMC pass stamped into
                 profiles/traffic.json for exactly this kernel source (null when the stamp does not match).
  fc_roofline  — all FC kernels together against the algorithmic FLOP of the step; in
                 bf16x3 mode `executed_frac` counts the three MFMAs per product that mode issues
  parity       — one extra step on a 4096-row slice of the same workload (fresh trainer, same initial
                 weights), outside the timed region, checked against the CPU oracle (oracle/restated.py,
                 itself pinned to the reference's golden vectors by tests/)
  cpu_baseline — the unmodified reference (oracle/_ref byte code; the restated oracle only when that is absent) timed on
                 this box's host cores, rank 0, N == 1 only
"""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK = {"bf16": 2.5e15, "bf16x3": 2.5e15, "f32": 157.3e12}  # /opt/skills/guides/MI355X_MICROARCH.md:40-42
HBM_PEAK = 8.0e12
LINE_LIMIT = 4096  # bytes of the ONE line rank 0 prints (round 5's 35.5 KB line was not parsed by the driver)
REPORT_FILE = "bench_report.json"

CONFIGS = {
    "c2": dict(algo="dqn", state_dim=128, actions=16, atoms=None,
               name="Discrete DQN state_dim=128 |A|=16 3x512 MLP (BASELINE.json configs[1]; replay gather + "
                    "normalize + 3 fwd + TD/Huber + bwd + Adam + soft update)"),
    "c3": dict(algo="qrdqn", state_dim=128, actions=16, atoms=200,
               name="QR-DQN 200 quantiles state_dim=128 |A|=16 3x512 MLP, 3200-wide head (BASELINE.json configs[2])"),
    "c4": dict(algo="sac", state_dim=256, actions=32, atoms=None,
               name="SAC state_dim=256 action_dim=32, Gaussian actor + twin critics, H=3x512 (BASELINE.json configs[3])"),
    "c5": dict(algo="dqn", state_dim=128, actions=16, atoms=None,
               name="Discrete DQN state_dim=128 |A|=16 3x512 MLP, global batch 524288 over 8 ranks "
                    "(BASELINE.json configs[4])"),
}


class GpuTelemetry:
    """Shader clock and socket power of ONE GPU sampled by a thread while a timed region runs: sysfs (`pp_dpm_sclk`'s
    starred level, hwmon `power1_average` / `power1_input` in microwatt) every 20 ms, else `rocm-smi --showclocks
    --showpower` as fast as it answers.  Reports mean / min / max and the sample count; None when neither is readable."""

    def __init__(self, index=0, period=0.02):
        import glob
        import threading

        self.index, self.period = index, period
        self.sclk, self.power, self.source = [], [], None
        self._stop = threading.Event()
        self._thread = None
        cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))
        # the node shows every GPU's sysfs entry, the container owns one of them: find this device by its PCI address
        self._dev, self.pci = None, None
        try:
            pr = torch.cuda.get_device_properties(index)
            self.pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for d in cards:
                if os.path.basename(os.path.realpath(d)).lower() == self.pci:
                    self._dev = d
        except Exception:
            pass
        if self._dev is None and len(cards) == 1:
            self._dev = cards[0]
        self._pw = None
        if self._dev:
            for name in ("power1_average", "power1_input"):
                hits = glob.glob(os.path.join(self._dev, "hwmon", "hwmon*", name))
                if hits:
                    self._pw = hits[0]
                    break

    def _sysfs(self):
        with open(os.path.join(self._dev, "pp_dpm_sclk")) as fh:
            for line in fh:
                if "*" in line:
                    self.sclk.append(float(line.split(":")[1].lower().split("mhz")[0]))
        if self._pw:
            with open(self._pw) as fh:
                self.power.append(float(fh.read()) * 1e-6)

    def _smi(self):
        import re
        import subprocess

        out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showclocks", "--showpower"], capture_output=True, text=True,
                             timeout=10).stdout
        m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        if m:
            self.sclk.append(float(m.group(1)))
        m = re.search(r"Power \(W\): ([0-9.]+)", out)
        if m:
            self.power.append(float(m.group(1)))

    def _loop(self):
        read = self._sysfs if self._dev else self._smi
        self.source = (f"sysfs {self._dev}/pp_dpm_sclk (starred level) + hwmon power" if self._dev
                       else "rocm-smi --showclocks --showpower (device not found in sysfs by PCI address)")
        while not self._stop.is_set():
            try:
                read()
            except Exception as e:  # telemetry must never take the measurement down
                self.source = f"unreadable: {e!r}"
                return
            self._stop.wait(self.period)

    def __enter__(self):
        import threading

        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=15)

    def summary(self):
        def stat(v):
            return None if not v else {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "samples": len(v)}

        return {"sclk_mhz": stat(self.sclk), "power_w": stat(self.power), "source": self.source, "pci": self.pci}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default 1; 8 for --config c5)")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--batch", type=int, default=65536, help="per-rank batch")
    ap.add_argument("--state-dim", type=int, default=None)
    ap.add_argument("--actions", type=int, default=None)
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--capacity", type=int, default=1 << 20)
    ap.add_argument("--precision", choices=["bf16", "bf16x3", "f32"], default="bf16")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps; the median one is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-accurate", action="store_true",
                    help="skip the second timed region in the 1e-4-compliant bf16x3 mode (default run, --precision bf16)")
    ap.add_argument("--no-also", action="store_true", help="skip the short C3 / C4 regions of the default C2 run")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step as a HIP graph at all")
    ap.add_argument("--launch", choices=["auto", "graph", "eager"], default=None,
                    help="auto: replay the captured HIP graph or enqueue eagerly, whichever a short calibration finds faster")
    ap.add_argument("--prefetch", action="store_true", help="gather the next batch on a second stream (slower, see runtime.py)")
    ap.add_argument("--cpu-steps", type=int, default=None)
    ap.add_argument("--parity-batch", type=int, default=4096)
    ap.add_argument("--graph-steps", type=int, default=1,
                    help="consecutive steps recorded per HIP graph where the loop supports it (must divide --steps); measured round 4 at 1 / 2 / 4 / 8: 0.5125 / 0.5134 / 0.5126 / 0.5159 ms per C2 step — no gain, default 1")
    ap.add_argument("--sustained-steps", type=int, default=4000,
                    help="steps of the one long region reported as `sustained` next to the K-step regions (0: skip)")
    ap.add_argument("--report", default=None, help=f"where the full record goes (default {REPORT_FILE} beside bench.py); the printed "
                                                   "line is a digest of it, at most LINE_LIMIT bytes")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher self-test: form the process group (gloo when there is no GPU), report its size, exit")
    args = ap.parse_args()
    c = CONFIGS[args.config]
    if args.gpus is None:
        args.gpus = 8 if args.config == "c5" else 1
    args.state_dim = args.state_dim or c["state_dim"]
    args.actions = args.actions or c["actions"]
    args.algo, args.atoms = c["algo"], c["atoms"]
    return args


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: become N ranks."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def bind_rank_to_gpu_numa(local_rank, local_world):
    """CPU affinity of this rank = its share of the cores of ITS GPU's NUMA node (sysfs: /sys/bus/pci/devices/<gpu>/numa_node,
    local_cpulist).  Eight ranks of one node otherwise float over all sockets: an eager step costs the host ~0.3 ms of a
    0.5 ms step, and a rank whose enqueue thread sits on the far socket (or shares cores with another rank's) goes host-bound
    first.  The ranks whose GPUs hang off the same node split its cores evenly, in local-rank order.  Returns what was done
    (reported per rank in the line) or a reason; never raises — an unreadable topology leaves the affinity alone."""
    try:
        def parse(cpulist):
            out = []
            for part in cpulist.strip().split(","):
                if not part:
                    continue
                lo, _, hi = part.partition("-")
                out.extend(range(int(lo), int(hi or lo) + 1))
            return out

        def gpu_node(i):
            pr = torch.cuda.get_device_properties(i)
            pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            base = f"/sys/bus/pci/devices/{pci}"
            with open(base + "/numa_node") as fh:
                node = int(fh.read())
            with open(base + "/local_cpulist") as fh:
                return node, parse(fh.read())

        node, cpus = gpu_node(local_rank)
        allowed = os.sched_getaffinity(0)
        cpus = sorted(set(cpus) & allowed)
        if node < 0 or not cpus:
            return {"bound": False, "reason": f"numa_node {node}, {len(cpus)} usable local cpus"}
        sharing = [r for r in range(local_world) if gpu_node(r)[0] == node]
        pos, n = sharing.index(local_rank), len(sharing)
        chunk = max(1, len(cpus) // n)
        mine = cpus[pos * chunk:(pos + 1) * chunk] or cpus
        os.sched_setaffinity(0, mine)
        return {"bound": True, "numa_node": node, "cpus": len(mine), "first_cpu": mine[0], "ranks_on_node": n}
    except Exception as e:  # containers without the sysfs entries, single-socket boxes, ...
        return {"bound": False, "reason": repr(e)}


def layer_dims(args):
    out = args.actions * (args.atoms or 1)
    return [args.state_dim] + [args.hidden] * args.layers + [out]


def mac(dims):
    return sum(a * b for a, b in zip(dims, dims[1:]))


def fc_flops(args, batch):
    """Algorithmic FLOP of one step.  DQN / QR-DQN: 3 forwards + wgrad(all layers) +
    dgrad(all but the first).  SAC: 2 actor fwd + 1 actor bwd + 6 critic fwd + 2 critic full bwd + 2 critic
    dgrad-only passes (down to the action columns of layer 1)."""
    if args.algo == "qrdqn" and getattr(args, "grouped_head", False):
        # grouped wide layer (reagent_amd/qr_engine.py): the per-action mean layer replaces the 3200-wide forward of
        # the a* selection, and only ONE action's [N, H] slice of the wide layer is evaluated / differentiated per row
        H, A, N = args.hidden, args.actions, args.atoms
        trunk = [args.state_dim] + [H] * args.layers
        return 2 * batch * (3 * mac(trunk) + 2 * H * A + 2 * H * N + mac(trunk) + mac(trunk[1:]) + 2 * H * N)
    if args.algo != "sac":
        d = layer_dims(args)
        return 2 * batch * (3 * mac(d) + mac(d) + mac(d[1:]))
    S, A, H = args.state_dim, args.actions, [args.hidden] * args.layers
    da, dc = [S] + H + [2 * A], [S + A] + H + [1]
    actor_fwd, actor_bwd = mac(da), mac(da) + mac(da[1:])
    critic_fwd, critic_bwd = mac(dc), mac(dc) + mac(dc[1:])
    critic_dgrad = mac(dc[1:]) + A * H[0]
    return 2 * batch * (2 * actor_fwd + actor_bwd + 6 * critic_fwd + 2 * critic_bwd + 2 * critic_dgrad)


def precision_code(args):
    import reagent_amd._lib as L

    return {"bf16": L.PREC_BF16, "bf16x3": L.PREC_BF16X3, "f32": L.PREC_F32}[args.precision]


def build(args, device, rank, batch=None, cols=None):
    """this rank's trainer + replay shard + loop for the configured workload"""
    from reagent_amd import synthetic
    from reagent_amd.core.parameters import EvaluationParameters, NormalizationParameters, RLParameters
    from reagent_amd.models import (FullyConnectedCritic, FullyConnectedDQN, GaussianFullyConnectedActor,
                                    set_default_precision)
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.preprocessing import Preprocessor
    from reagent_amd.replay_memory import ReplayBuffer
    from reagent_amd.runtime import OfflineDqnLoop, OfflinePolicyLoop

    batch = batch or args.batch
    S, A, H = args.state_dim, args.actions, [args.hidden] * args.layers
    acts = ["relu"] * args.layers
    adam = lambda: Optimizer__Union.default(lr=1e-3)  # noqa: E731
    import reagent_amd._lib as L

    set_default_precision(precision_code(args))
    try:
        if args.algo == "sac":
            nets = [GaussianFullyConnectedActor(S, A, H, acts), FullyConnectedCritic(S, A, H, acts),
                    FullyConnectedCritic(S, A, H, acts)]
            shapes = [[S] + H + [2 * A], [S + A] + H + [1], [S + A] + H + [1]]
        else:
            nets = [FullyConnectedDQN(S, A, H, acts, num_atoms=args.atoms)]
            shapes = [layer_dims(args)]
    finally:
        set_default_precision(L.PREC_F32)
    init = []
    for k, (net, dims) in enumerate(zip(nets, shapes)):  # identical initial weights on every rank
        w = synthetic.fc_init(dims, acts + ["linear"], seed=40 + k)
        with torch.no_grad():
            for p, x in zip(net.parameters(), w):
                p.copy_(x)
        init.append(w)
        net.to(device)
    if args.algo == "sac":
        from reagent_amd.training import SACTrainer

        trainer = SACTrainer(nets[0], nets[1], nets[2], rl=RLParameters(gamma=0.99, target_update_rate=0.001),
                             q_network_optimizer=adam(), actor_network_optimizer=adam(), alpha_optimizer=adam()).to(device)
    else:
        rl = RLParameters(gamma=0.99, target_update_rate=0.001, maxq_learning=True, q_network_loss="huber")
        common = dict(actions=[str(i) for i in range(A)], rl=rl, double_q_learning=True, optimizer=adam(),
                      evaluation=EvaluationParameters(calc_cpe_in_training=False))
        q = nets[0]
        if args.algo == "qrdqn":
            from reagent_amd.training import QRDQNTrainer

            trainer = QRDQNTrainer(q, q.get_target_network(), num_atoms=args.atoms, **common).to(device)
        else:
            from reagent_amd.training import DQNTrainer

            trainer = DQNTrainer(q, q.get_target_network(), None, **common).to(device)
    # this rank's shard of the offline dataset (different seed per rank), resident in HBM
    own_cols = cols is None
    if own_cols:
        cols = synthetic.replay_contents(args.capacity, S, A, seed=100 + rank)
        if args.algo == "sac":  # continuous actions already in the training range
            g = torch.Generator().manual_seed(200 + rank)
            cols["action"] = torch.rand(args.capacity, A, generator=g) * 1.8 - 0.9
            del cols["possible_actions_mask"]
    rb = ReplayBuffer(replay_capacity=args.capacity, batch_size=batch, device=device)
    rb.load_columns({k: v.to(device) for k, v in cols.items()}, mark_all_valid=True)
    mean, std = synthetic.normalization_table(S, 7)
    norm = {i: NormalizationParameters(feature_type="CONTINUOUS", mean=mean[i].item(), stddev=std[i].item())
            for i in range(S)}
    pre = Preprocessor(norm, device=device)
    if args.algo == "sac":
        import numpy as np

        from reagent_amd.core.parameters import CONTINUOUS_TRAINING_ACTION_RANGE as R
        from reagent_amd.preprocessing import PolicyNetworkInputMaker

        maker = PolicyNetworkInputMaker(np.full(A, R[0], dtype=np.float32), np.full(A, R[1], dtype=np.float32))
        loop = OfflinePolicyLoop(rb, trainer, batch, maker, pre,
                                 state_dtype=torch.bfloat16 if args.precision == "bf16" else None)
    else:
        loop = OfflineDqnLoop(rb, trainer, batch, pre,
                              state_dtype=torch.bfloat16 if args.precision == "bf16" else torch.float32,
                              prefetch=args.prefetch)
    return loop, trainer, init, cols, (mean, std)


def cpu_batch(args, cols, norm, idx):
    """the reference path's batch on the CPU: gather of the same columns, (x - mean) / std normalization
    with the Preprocessor's clamp, input maker"""
    mean, std = norm
    C, A = args.capacity, args.actions
    nxt = (idx + 1) % C
    clamp = lambda x: torch.clamp(x, -11.513, 11.513)  # noqa: E731  (preprocessor.py:156-168; no-op for N(0,1)/0.5 here)
    term = cols["terminal"][idx]
    b = dict(state=clamp((cols["observation"][idx] - mean) / std), next_state=clamp((cols["observation"][nxt] - mean) / std),
             reward=cols["reward"][idx].unsqueeze(1), not_terminal=1.0 - term.float().unsqueeze(1))
    if args.algo == "sac":
        # PolicyNetworkInputMaker (trainer_preprocessor.py:175-227): rescale_actions (training/utils.py:13-29) from
        # the environment range to the training range — here the same interval, the arithmetic still runs
        lo, hi = -1.0, 1.0  # CONTINUOUS_TRAINING_ACTION_RANGE
        rescale = lambda a: ((a - lo) / (hi - lo)) * (hi - lo) + lo  # noqa: E731
        b["action"] = rescale(cols["action"][idx])
        b["next_action"] = rescale(cols["action"][nxt]) * b["not_terminal"]
    else:
        one_hot = torch.nn.functional.one_hot
        b["action"] = one_hot(cols["action"][idx], A).float()
        b["next_action"] = one_hot(cols["action"][nxt], A).float() * b["not_terminal"]
        b["possible_next_actions_mask"] = cols["possible_actions_mask"][nxt]
        b["possible_actions_mask"] = cols["possible_actions_mask"][idx]
    return b


def make_oracle(args, init):
    from oracle import restated as R

    acts = ["relu"] * args.layers + ["linear"]
    if args.algo == "dqn":
        return R.DQNOracle(init[0], init[0], acts, gamma=0.99, tau=0.001, loss="huber", lr=1e-3)
    if args.algo == "qrdqn":
        return R.QRDQNOracle(init[0], init[0], acts, num_actions=args.actions, num_atoms=args.atoms, gamma=0.99,
                             tau=0.001, lr=1e-3)
    return R.SACOracle(init[0], init[1], init[2], acts, acts, args.actions, gamma=0.99, tau=0.001, lr=1e-3)


def cpu_baseline(args, init, cols, norm):
    """The reference's CPU path on this box's host cores, bounded sample.

    kind "reference": the UNMODIFIED reference — ReplayBuffer.sample_transition_batch (its own index draw) ->
    DiscreteDqnInputMaker / PolicyNetworkInputMaker -> Preprocessor.forward x2 -> {DQN,QRDQN,SAC}Trainer.train_step_gen
    through the Lightning-1.6 loop emulation — imported from /root/reference where that exists and from its byte code in
    oracle/_ref (oracle/build_ref.py, built by __graft_entry__.build()) on the GPU box: oracle/reference_bench.py.
    kind "port": only when neither is present — oracle/restated.py (torch-CPU restatement) with an advanced-index gather."""
    B = args.batch if args.algo != "qrdqn" else min(args.batch, 8192)  # the (N, B, N) tensor: 62 GB hosts
    steps = args.cpu_steps or 2
    import logging

    from oracle import reference_bench as RB

    logging.disable(logging.INFO)  # the reference logs every constructed module / replay buffer at INFO: not this run's output
    if RB.available():
        best, tried = RB.run(args.algo, args.state_dim, args.actions, args.hidden, args.layers, args.atoms, args.capacity, B,
                             init, cols, norm, steps=steps)
        return {"value": B / (best["ms_per_step"] * 1e-3), "unit": "transitions/s", "cores": best["threads"],
                "kind": "reference", "ms_per_step": best["ms_per_step"], "sample_ms": best["sample_ms"],
                "train_ms": best["train_ms"], "host_cpus": os.cpu_count(), "thread_settings_tried": tried,
                "reference_from": RB.where(),
                "sample": f"{best['steps']} steps (after 1 warm-up) at B={B} of the same workload, fp32, the unmodified reference: "
                          f"ReplayBuffer.sample_transition_batch (own index draw) + input maker + Preprocessor x2 = "
                          f"{best['sample_ms']:.0f} ms, {args.algo} train_step_gen under the Lightning-loop emulation = "
                          f"{best['train_ms']:.0f} ms; torch intra-op threads = `cores` (fastest of the listed settings)"}
    o = make_oracle(args, init)
    g = torch.Generator().manual_seed(3)

    def one():
        idx = torch.randint(args.capacity, (B,), generator=g)
        b = cpu_batch(args, cols, norm, idx)
        if args.algo == "sac":
            o.step(b, torch.randn(B, args.actions, generator=g), torch.randn(B, args.actions, generator=g))
        else:
            o.step(b)

    default = torch.get_num_threads()
    tried = []
    try:
        for th in [default] + [t for t in (32, 16) if t < default]:
            torch.set_num_threads(th)
            one()  # warm-up
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            tried.append(dict(threads=th, ms_per_step=(time.perf_counter() - t0) / steps * 1e3))
    finally:
        torch.set_num_threads(default)
    best = min(tried, key=lambda r: r["ms_per_step"])
    return {"value": B / (best["ms_per_step"] * 1e-3), "unit": "transitions/s", "cores": best["threads"], "kind": "port",
            "ms_per_step": best["ms_per_step"], "host_cpus": os.cpu_count(), "thread_settings_tried": tried,
            "sample": f"{steps} steps at B={B} of the same workload (advanced-index gather of the same columns + "
                      f"normalize + {args.algo} step, fp32, oracle/restated.py = torch-CPU restatement of the reference "
                      f"trainer — oracle/_ref was not built, so the reference itself could not run here)"}


def parity_check(args, device, init, cols, norm):
    """One step of the SAME workload on a 4096-row slice — fresh trainer with the timed run's initial weights,
    the same replay shard, the same code path (one-launch sampler, native step, fused update) — against the
    CPU oracle on the indices the device drew."""
    B = args.parity_batch
    loop, trainer, _, _, _ = build(args, device, 0, batch=B, cols=cols)
    g = torch.Generator().manual_seed(11)
    idx = torch.randint(args.capacity, (B,), generator=g)
    o = make_oracle(args, init)
    b = cpu_batch(args, cols, norm, idx)
    batch = loop.make_batch(idx.to(device))
    exact = []
    for k in ("action", "next_action", "reward", "not_terminal", "possible_next_actions_mask"):
        if k in b:
            got = getattr(batch, k)
            got = got.float_features if hasattr(got, "float_features") else got
            exact.append(bool(torch.equal(got.float().cpu().reshape(b[k].shape), b[k].float())))
    d_state = max((getattr(batch, k).float_features.float().cpu() - b[k]).abs().max().item() for k in ("state", "next_state"))
    out = {"batch": B, "mode": args.precision, "gather_fields_bit_exact": all(exact), "max_abs_dstate": d_state,
           "oracle": "oracle/restated.py (torch-CPU fp32 restatement, pinned to the reference by tests/golden)"}
    if args.algo == "sac":
        nn_, nc = torch.randn(B, args.actions, generator=g), torch.randn(B, args.actions, generator=g)
        loc, scale_log = trainer.actor_network._get_loc_and_scale_log(batch.state)
        rl, rs = o.pi.loc_scale_log(o.actor, b["state"])
        out["max_abs_dlogits"] = max((loc.cpu() - rl.detach()).abs().max().item(), (scale_log.cpu() - rs.detach()).abs().max().item())
        got = loop.trainer.train_step_native(batch, nn_, nc)
        ref = o.step(b, nn_, nc)
        out["rel_dloss"] = max(abs(float(got[k]) - float(ref[k])) / max(abs(float(ref[k])), 1e-3)
                               for k in ("q1_loss", "q2_loss", "actor_loss"))
        pairs = list(zip(trainer.actor_network.parameters(), o.actor)) + list(zip(trainer.q1_network.parameters(), o.q1))
    else:
        zr_all = None
        if args.algo == "qrdqn":
            with torch.no_grad():  # the oracle's network BEFORE the step: quantiles of every action, state and next state
                zr_all = o.net(o.params, b["state"])
                zrn_mean = o.net(o.params, b["next_state"]).mean(dim=2)
        loss = loop.step(idx.to(device))
        loop.flush()
        if args.algo == "qrdqn":
            gq = getattr(trainer, "_gq_active", None)
            if gq is not None:
                # What the STEP computed, on the grouped engine (qr_engine.py): the logged action's N quantiles of every
                # transition (grouped row r = batch row rowmap[r], action key_cur) and the per-action means of next_state
                # that a* was chosen from (double-Q: the online network's) — every row of the batch, not a slice.
                rowmap, key = gq.sp_cur.rowmap.cpu().long(), gq.key_cur.cpu().long()
                live = rowmap >= 0
                rows_b = rowmap[live]
                got = gq.z.cpu()[live][:, :args.atoms]
                out["max_abs_dquantile"] = (got - zr_all[rows_b, key[rows_b]]).abs().max().item()
                out["max_abs_dq"] = (gq.qbar_next.cpu() - zrn_mean).abs().max().item()
                out["dq_rows"] = int(live.sum())
                out["path"] = "grouped engine, " + ("split-bf16" if gq.x3 else "bf16")
                out["full_size"] = ("the reference cannot run this configuration at B = 65536 on a 62 GB host (its (N, B, N) tensor, "
                                    "qrdqn_trainer.py:152): oracle parity of the full step is tested at B = 8192 and, at B = 65536, the "
                                    "grouped engine against THIS library's dense exact-fp32 head — HIP against HIP (tests/test_full_size.py)")
            else:
                rows = min(B, 512)  # the dense path: its [B, A * N] logits (a bounded slice of the saved forward)
                z = trainer._q.view(B, args.actions, args.atoms)[:rows].cpu()
                out["max_abs_dquantile"] = (z - zr_all[:rows]).abs().max().item()
                out["max_abs_dq"] = (z.mean(dim=2) - zr_all[:rows].mean(dim=2)).abs().max().item()
                out["dq_rows"] = rows
                out["path"] = "dense [B, A * N] logits"
        ref = o.step(b)
        q = trainer.all_action_scores if args.algo == "dqn" else None
        if q is not None:
            out["max_abs_dq"] = (q.cpu() - ref["q"]).abs().max().item()
        out["rel_dloss"] = abs(loss.item() - ref["loss"].item()) / abs(ref["loss"].item())
        pairs = list(zip(trainer.q_network.parameters(), o.params))
    dws = [(p.detach().cpu() - r.detach()).abs() for p, r in pairs]
    out["max_abs_dw"] = max(d.max().item() for d in dws)
    out["frac_dw_beyond_2e-5"] = sum((d > 2e-5).sum().item() for d in dws) / sum(d.numel() for d in dws)
    dq = max(out.get("max_abs_dq", 0.0), out.get("max_abs_dlogits", 0.0), out.get("max_abs_dquantile", 0.0))
    # north_star's floating-point bound ("Q-values / policy logits within 1e-4 fp32"; replay index gather bit-exact)
    out["meets_north_star"] = bool(dq <= 1e-4 and all(exact))
    if args.precision == "f32":
        out["tolerance"] = "Q / logits 1e-4, weights 2e-5 (north_star)"
        if args.algo == "sac":
            out["tolerance"] += ("; actor weights: <= 2 % beyond 2e-5, none beyond 2*lr (its gradient is ill-conditioned: "
                                 "torch-CPU fp32 is itself 5e-6 from fp64 there, tests/test_baseline_shapes.py)")
            out["ok"] = bool(dq <= 1e-4 and out["frac_dw_beyond_2e-5"] <= 0.02 and out["max_abs_dw"] <= 2.1e-3)
        else:
            out["ok"] = bool(dq <= 1e-4 and out["max_abs_dw"] <= 2e-5)
    elif args.precision == "bf16x3":
        out["tolerance"] = ("Q / logits 1e-4 (north_star); weights after one Adam step: <= 2 % beyond 2e-5, none beyond 2*lr "
                            "(Adam moves a weight by lr*g/(|g|+1e-8): the ~1e-5 relative error of split-bf16 gradients "
                            "flips the direction of the few weights whose gradient is below it)")
        out["ok"] = bool(dq <= 1e-4 and out["frac_dw_beyond_2e-5"] <= 0.02 and out["max_abs_dw"] <= 2.1e-3)
    else:
        out["tolerance"] = ("NOT north_star's: plain bf16 operands put Q ~3e-2 from the fp32 reference (the bf16x3 mode in "
                            "`accurate` is the one held to 1e-4); sanity bounds only: Q 6e-2, |dW| <= 2*lr after one Adam step")
        out["sane"] = bool(dq <= 6e-2 and out["max_abs_dw"] <= 2.1e-3 and out["gather_fields_bit_exact"])
        out["ok"] = False  # `ok` means north_star's tolerance; this mode does not meet it by construction
    out["ok"] = bool(out["ok"] and out["gather_fields_bit_exact"] and out["meets_north_star"])
    return out


def source_stamp():
    """sha256 of the kernel sources the dominant kernels are built from (stamps profiles/traffic.json)"""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "reagent_amd", "csrc")
    for f in sorted(os.listdir(src)):  # every kernel source the file's figures describe (FC stacks, sampler, heads, updates)
        if f.endswith((".hip", ".h")):
            with open(os.path.join(src, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


FC_ENTRY_POINTS = ("rg_fc_forward", "rg_fc_dgrad", "rg_fc_wgrad", "rg_fc_wgrad_frag", "rg_mlp_forward_fused",
                   "rg_mlp_backward_fused", "rg_mlp_wgrad_fused", "rg_group_head_forward", "rg_group_head_dgrad",
                   "rg_group_head_wgrad")
# the __global__ functions an entry point launches (what `rocprofv3 --kernel-trace --stats` lists for it: profiles/)
KERNELS_OF = {"rg_mlp_forward_fused": "mlp_fwd_fused_kernel", "rg_mlp_backward_fused": "mlp_bwd_fused_kernel",
              "rg_mlp_wgrad_fused": "wgrad_group_kernel + wgrad_reduce_tail_kernel",
              "rg_replay_dqn_batch": "replay_dqn_batch_kernel", "rg_replay_policy_batch": "replay_policy_batch_kernel"}


class QueueAhead:
    """Calibrations of the instrumented pass (kernel_profile) and its fallback blocker.

    HIP events around a launch that starts from an idle queue time the launch's dispatch latency and whatever the host
    did between its two records (round 4: the same binary summed to 0.57 ms one run and 0.78 ms the next, and a rescale
    hid it).  With the queue kept full, the start event of launch k completes when launch k-1 does and its end event when
    launch k does: the span is the kernel's own duration plus what the two marker packets add, whatever the host's pace —
    reproducible, and within a few percent of what `rocprofv3 --kernel-trace` reports for the same kernels (profiles/).
    kernel_profile keeps the queue full with a backlog of plain steps; this class holds torch's spin kernel
    (torch.cuda._sleep, calibrated with events) as the blocker of last resort for host-bound loops, and the kernel of known
    length the marker pair's addition is measured on.  Nothing of it runs off-GPU (the interpreter tests)."""

    def __init__(self, device, ms=3.0):
        self.on = device.type == "cuda" and hasattr(torch.cuda, "_sleep")
        self.ms, self.cycles_per_ms, self.marker_us = ms, None, None
        if not self.on:
            return
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda._sleep(1_000_000)
        torch.cuda.synchronize()
        ev[0].record()
        torch.cuda._sleep(20_000_000)
        ev[1].record()
        torch.cuda.synchronize()
        self.cycles_per_ms = 20_000_000 / max(ev[0].elapsed_time(ev[1]), 1e-3)
        # what two back-to-back records cost when nothing lies between them (an EMPTY bracket; reported as marker_empty_us)
        self.block()
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(16)]
        for a, b in pairs:
            a.record()
            b.record()
        torch.cuda.synchronize()
        self.marker_us = self.marker_empty_us = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2] * 1e3
        # A kernel of KNOWN length for the in-pass calibration (kernel_profile): a 20 us spin; its length = 64 of them back to
        # back between ONE pair of events (the pair's cost spread over 64 launches), dispatch gap included
        self.probe_cycles = max(1, int(0.02 * self.cycles_per_ms))
        torch.cuda._sleep(self.probe_cycles)
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(64):
            torch.cuda._sleep(self.probe_cycles)
        ev[1].record()
        torch.cuda.synchronize()
        self.probe_us = ev[0].elapsed_time(ev[1]) * 1e3 / 64

    def block(self, ms=None):
        if self.on:
            torch.cuda._sleep(int((ms or self.ms) * self.cycles_per_ms))


def agree_over_ranks(n: int, device) -> int:
    """The largest `n` any rank of the process group holds (n itself without a group).  Every step of a data-parallel loop
    holds a gradient all-reduce, so a count of steps that a rank derives from its OWN host timing — the backlog the
    instrumented pass queues ahead of itself — has to be made the same on every rank before it is used: ranks that enqueue
    different numbers of collectives end in a hang."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return int(n)
    t = torch.tensor([int(n)], dtype=torch.int64, device=device if device.type == "cuda" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def kernel_profile(args, step, steps, device=None, gpu_ms=None):
    """Instrumented pass: HIP events around every C-ABI launch (on the launch stream), all of it enqueued behind a backlog
    of plain steps so that the spans are kernel durations, not host gaps.  Rows of one entry point are
    MERGED over its variants (saving / non-saving forwards, the nets of a step): the dominant entry point is the one with
    the largest total time and its figure the call-weighted average."""
    from reagent_amd import ops

    B = args.batch
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    qa = QueueAhead(device)
    # The host has to be AHEAD of the device for the whole instrumented pass (two warm steps + `steps` instrumented ones), or
    # the spans time host gaps.  The lead is built with the workload itself: plain steps enqueue faster than the device runs
    # them (C2: 0.29 against 0.49 ms), so a burst of them leaves a backlog — and the chip at the clock and power of the
    # timed region.  (The first forms of this pass used torch's spin kernel as the blocker: one wavefront busy, the chip
    # otherwise idle, and its clock sagged under a 10 ms one — spans 7 % above the timed step on the pool's fastest box,
    # profiles/r05_run4.)  Only a host that cannot outrun the device (a host-bound box) falls back to the spin kernel.
    with ops.profile():  # what the host needs per INSTRUMENTED step (records dropped)
        t = time.perf_counter()
        step()
        dry_ms = (time.perf_counter() - t) * 1e3
    if device.type == "cuda":
        torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        step()
    plain_ms = (time.perf_counter() - t) * 1e3 / 3
    lead_needed = 1.3 * dry_ms * (steps + 2) + 0.5
    gap = (gpu_ms - plain_ms) if gpu_ms else 0.0
    backlog_steps = agree_over_ranks(min(400, int(lead_needed / gap) + 1) if (qa.on and gap > 0.03) else 0, device)
    t0 = time.perf_counter()
    if backlog_steps > 0:
        for _ in range(backlog_steps):
            step()
        lead_mode, blocker_ms = f"backlog of {backlog_steps} plain steps", backlog_steps * max(gap, 0.0)
    else:
        backlog_steps = 0
        qa.block(lead_needed)
        lead_mode, blocker_ms = "spin kernel (host-bound loop: the chip's clock may sag under it)", lead_needed
    t1 = time.perf_counter()
    for _ in range(2):
        step()
    pairs = []
    with ops.profile() as prof:
        for _ in range(steps):
            step()
            if qa.on:
                # What a bracket adds to the kernel inside it, measured IN this backed-up queue on a kernel of known length:
                # span(bracketed 20 us spin) - its length.  (An empty bracket — two markers with nothing between them, 4.6-4.7
                # us — overstates it: around a real kernel the second marker is processed while the kernel runs.  With the
                # empty-bracket figure taken off, every launch came out 2-9 % BELOW rocprofv3 of the same box and the spans 5.5 %
                # below the timed step; with nothing taken off 1-3.5 % above: profiles/r05_run5.)
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record()
                torch.cuda._sleep(qa.probe_cycles)
                b_.record()
                pairs.append((a_, b_))
    host_total_ms = (time.perf_counter() - t1) * 1e3
    if pairs:
        torch.cuda.synchronize()
        spans = sorted(x.elapsed_time(y) for x, y in pairs)
        qa.marker_us = max(0.0, spans[len(spans) // 2] * 1e3 - qa.probe_us)
    host_ms = [host_total_ms / (steps + 2)]
    qa.ms = blocker_ms
    rows = prof.summary()
    # A span = the launch(es) of the entry point + what the pair of marker packets around it adds; that addition was measured
    # once per instrumented step in the same backed-up queue, on a spin kernel of known length (QueueAhead.marker_us).  First
    # measurement (profiles/r05_run1): raw spans 93.3 / 87.0 / 41.3 us for forward / backward / gather against rocprofv3's
    # 88.6 / 82.3 / 37.2 us of the same box and binary — each high by exactly that pair.
    marker_ms = (qa.marker_us or 0.0) * 1e-3
    for r in rows:
        r["span_ms"] = r["ms"]
        r["ms"] = max(r["ms"] - r["calls"] * marker_ms, 0.05 * r["ms"])
    rows.sort(key=lambda r: -r["ms"])
    for r in rows:
        m = r["meta"]
        if r["name"] not in FC_ENTRY_POINTS:
            continue
        if r["name"] in ("rg_mlp_forward_fused", "rg_mlp_wgrad_fused"):
            d = m["dims"]
            r["flop_per_launch"] = 2.0 * m["B"] * mac(d)
            r["label"] = f"{r['name']} B={m['B']} dims={list(d)}" + (f" save={m['save']}" if "save" in m else "")
        elif r["name"] == "rg_mlp_backward_fused":
            d = m["dims"]
            r["flop_per_launch"] = 2.0 * m["B"] * (mac(d[1:]) + (d[0] * d[1] if m.get("dx") else 0))
            r["label"] = f"rg_mlp_backward_fused B={m['B']} dims={list(d)}"
        elif r["name"].startswith("rg_group_head"):
            r["flop_per_launch"] = 2.0 * B * m["Ng"] * m["K"]  # one [Ng, K] slice per batch row
            r["label"] = f"{r['name']} rows={B} Ng={m['Ng']} K={m['K']}"
        else:
            r["flop_per_launch"] = 2.0 * m["M"] * m["N"] * m["K"]
            r["label"] = f"{r['name']} M={m['M']} N={m['N']} K={m['K']}"
    fc = [r for r in rows if r["name"] in FC_ENTRY_POINTS]
    peak = MFMA_PEAK[args.precision]
    mfma_per_product = 3 if args.precision == "bf16x3" else 1
    out = {}
    if fc:
        by_name = {}
        for r in fc:  # merge an entry point's variants
            g = by_name.setdefault(r["name"], {"name": r["name"], "ms": 0.0, "span_ms": 0.0, "calls": 0, "flop": 0.0, "variants": []})
            g["ms"] += r["ms"]
            g["span_ms"] += r["span_ms"]
            g["calls"] += r["calls"]
            g["flop"] += r["flop_per_launch"] * r["calls"]
            g["variants"].append({"label": r["label"], "launches_per_step": r["calls"] / steps,
                                  "avg_launch_us": r["ms"] * 1e3 / r["calls"],
                                  "frac": r["flop_per_launch"] / (r["ms"] * 1e-3 / r["calls"]) / peak})
        dom = max(by_name.values(), key=lambda g: g["ms"])
        sec = dom["ms"] * 1e-3 / dom["calls"]
        ach = dom["flop"] / dom["calls"] / sec
        out["roofline"] = {"bound": "mfma", "kernel": f"{dom['name']} ({KERNELS_OF.get(dom['name'], dom['name'])})",
                           "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak,
                           "algorithmic_gflop_per_launch": dom["flop"] / dom["calls"] / 1e9,
                           "avg_launch_us": sec * 1e6, "avg_span_us": dom["span_ms"] * 1e3 / dom["calls"],
                           "marker_pair_us": qa.marker_us, "launches_per_step": dom["calls"] / steps,
                           "averaging": "call-weighted over every launch of the entry point in the instrumented steps "
                                        "(all variants): algorithmic FLOP of those launches / their summed HIP-event time; "
                                        "launch time = event span - what a marker pair adds to a kernel of known length in the same queue "
                                        "(marker_pair_us, measured in the pass)",
                           "variants": dom["variants"], "traffic": None}
        if mfma_per_product != 1:
            out["roofline"]["executed_frac"] = mfma_per_product * ach / peak
        fc_ms = sum(r["ms"] for r in fc) / steps
        alg = fc_flops(args, B)
        out["fc_roofline"] = {"algorithmic_gflop_per_step": alg / 1e9,
                              **({"dense_reference_gflop_per_step": 2 * B * (4 * mac(layer_dims(args)) + mac(layer_dims(args)[1:])) / 1e9,
                                  "note": "grouped wide layer: the step evaluates one action's [N, H] slice per row and a per-action "
                                          "mean layer instead of the dense [B, A*N] logits the reference materialises"}
                                 if getattr(args, "grouped_head", False) else {}), "fc_ms_per_step": fc_ms,
                              "achieved": alg / (fc_ms * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                              "frac": alg / (fc_ms * 1e-3) / peak,
                              "entry_points_us_per_step": {g["name"]: round(g["ms"] * 1e3 / steps, 2) for g in by_name.values()}}
        if mfma_per_product != 1:
            out["fc_roofline"]["executed_frac"] = mfma_per_product * out["fc_roofline"]["frac"]
            out["fc_roofline"]["note"] = ("bf16x3: three bf16 MFMAs per product (hi*hi + hi*lo + lo*hi), fp32 accumulate: the algorithmic "
                                          "fraction is capped at 1/3, `executed_frac` is the figure of merit of this mode")
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            try:
                t = json.load(open(traffic_file))
                if t.get("source_stamp") == source_stamp():
                    ks = t.get("kernels", {})
                    per = {}
                    for v in dom["variants"]:  # the PMC figures are per variant (save=0 / save=1): weight them like the times
                        lab = v["label"]
                        key = lab.split(" ")[0] + (":save=%s" % lab.split("save=")[1] if "save=" in lab else "")
                        per[key] = (ks.get(f"{args.config}:{args.precision}:{key}"), v["launches_per_step"])
                    if per and all(isinstance(tv, (int, float)) for tv, _ in per.values()):
                        out["roofline"]["traffic"] = sum(tv * c for tv, c in per.values()) / sum(c for _, c in per.values())
                        out["roofline"]["traffic_per_variant"] = {k: tv for k, (tv, _) in per.items()}
                        out["roofline"]["algorithmic_bytes_note"] = "HBM bytes per launch, PMC (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), call-weighted like the time"
                    elif per and any(tv is not None for tv, _ in per.values()):
                        out["roofline"]["traffic_per_variant"] = {k: tv for k, (tv, _) in per.items()}
                    out["roofline"]["traffic_source"] = t.get("from")
                else:
                    out["roofline"]["traffic_source"] = "profiles/traffic.json is stamped for other kernel sources: not reported"
            except Exception as e:
                out["roofline"]["traffic_source"] = f"profiles/traffic.json unreadable: {e!r}"
    g = [r for r in rows if r["name"] in ("rg_replay_dqn_batch", "rg_replay_policy_batch", "rg_replay_gather")]
    if g:
        sec = g[0]["ms"] * 1e-3 / g[0]["calls"]
        bytes_ = g[0]["meta"]["bytes_per_row"] * B
        out["gather"] = {"bound": "hbm", "achieved": bytes_ / sec / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": bytes_ / sec / HBM_PEAK, "avg_launch_us": sec * 1e6, "avg_span_us": g[0]["span_ms"] * 1e3 / g[0]["calls"],
                         "algorithmic_bytes_per_transition": g[0]["meta"]["bytes_per_row"], "kernel": g[0]["name"]}
    ar = [r for r in rows if r["name"] == "all_reduce"]
    if ar:
        out["all_reduce_us"] = ar[0]["ms"] * 1e3 / ar[0]["calls"]
        out["all_reduce_bytes"] = ar[0]["meta"]["bytes"]
    out["per_call_ms_per_step"] = {f"{r['name']}{tuple(r['meta'].values())}": round(r["ms"] / steps, 4) for r in rows[:18]}
    out["event_ms_per_step_sum"] = sum(r["ms"] for r in rows if r["name"] != "all_reduce") / steps
    out["event_span_ms_per_step_sum"] = sum(r["span_ms"] for r in rows if r["name"] != "all_reduce") / steps
    host = sorted(host_ms)[len(host_ms) // 2] if host_ms else None
    out["instrumented_pass"] = {
        "steps": steps, "event_ms_per_step_sum": out["event_ms_per_step_sum"], "event_span_ms_per_step_sum": out["event_span_ms_per_step_sum"],
        "queue_ahead": {"on": qa.on, "lead": lead_mode if qa.on else None, "lead_ms": qa.ms if qa.on else None, "host_enqueue_ms_per_step": host,
                        "plain_step_host_ms": plain_ms,
                        "host_enqueue_ms_total": host_total_ms,
                        "queued_behind_blocker": bool(qa.on and host_total_ms < qa.ms),
                        "marker_pair_us": qa.marker_us, "marker_empty_bracket_us": getattr(qa, "marker_empty_us", None),
                        "probe_kernel_us": getattr(qa, "probe_us", None)},
        "note": "HIP events on the launch stream around every C-ABI call; two warm steps and all instrumented ones are enqueued behind a backlog of plain steps, "
                "so a span is the launch's own duration + the marker pair, independent of the host's pace; launch times = span - "
                "marker_pair_us (calibrated in the same pass); nothing is rescaled to the timed step"}
    return out


def measure(args, device, rank, world, dist, cols=None, profile_steps=None):
    """build the workload `args` names on this rank, run the parity step, pick the launch path, time `--repeats` regions
    of exactly `--steps` steps, run the instrumented pass; returns the numbers (all ranks) for rank 0 to print"""
    loop, trainer, init, cols, norm = build(args, device, rank, cols=cols)
    if args.algo == "qrdqn":
        from reagent_amd.qr_engine import GroupedQR

        args.grouped_head = bool(trainer.use_grouped_head and GroupedQR.eligible(trainer))
    if world > 1:
        trainer.enable_data_parallel()
    parity = None
    if rank == 0 and not args.no_parity:
        try:
            parity = parity_check(args, device, init, cols, norm)
        except Exception as e:  # reported, never fatal to the measurement
            parity = {"error": repr(e)}

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step = loop.step
    replayed = False
    steps_per_call = 1  # steps one call of `step` makes (a replayed graph may hold several)
    graph_note = None
    calibration = None
    # data parallel: the eager loop (asynchronous all-reduce, deferred update) is the default — on one GPU the
    # replayed graph measured slower than eager launches for the DQN step, and the three-graph data-parallel form
    # has only been exercised on a one-rank RCCL group (tests/test_graph_replay.py); `--launch graph` / `auto` opt in
    want_graph = not args.no_graph and hasattr(loop, "capture") and not (world > 1 and args.launch is None) and args.launch != "eager"
    launch = args.launch or "auto"
    if (world > 1 and args.launch is None and not args.no_graph and device.type == "cuda" and hasattr(loop, "capture")
            and hasattr(trainer, "native_forward_backward")):
        # Data parallel without an explicit --launch: eager launches (what the world-2 tests cover) UNLESS this node's host
        # cannot keep up with them — seen on one box of the pool: 1.3 ms of host time per 0.54 ms C2 step, a loop that would
        # measure the host.  Then (every rank takes the same decision: the slowest host counts) the three-graph replay.
        for _ in range(3):
            loop.step()
        loop.flush()
        barrier()
        t = time.perf_counter()
        for _ in range(10):
            loop.step()
        host = (time.perf_counter() - t) / 10
        loop.flush()
        barrier()
        tot = (time.perf_counter() - t) / 10
        r = torch.tensor([host / max(tot, 1e-9)], device=device, dtype=torch.float64)
        dist.all_reduce(r, op=dist.ReduceOp.MAX)
        if r.item() >= 0.85:
            want_graph, launch = True, "graph"
            graph_note = f"host-bound eager loop (enqueue {host * 1e3:.3f} of {tot * 1e3:.3f} ms/step on the slowest rank's host)"
    if want_graph:
        # several consecutive steps per graph where the loop supports it (device-side index cursor: the DQN family) and the
        # step counts of this run divide: the ~8 us the queue idles between two graph launches are paid once per replay
        spr = args.graph_steps if (world == 1 and args.graph_steps > 1 and args.steps % args.graph_steps == 0
                                   and args.sustained_steps % args.graph_steps == 0) else 1
        try:
            try:
                replay = loop.capture(warmup=max(2, min(args.warmup, 3)), steps_per_replay=spr)
            except (NotImplementedError, ValueError, TypeError):
                replay = loop.capture(warmup=max(2, min(args.warmup, 3)))
        except Exception as e:
            replay = None
            graph_note = f"graph capture failed, eager launches: {e!r}"
        if dist is not None and agree_over_ranks(0 if replay is not None else 1, device) and replay is not None:
            # a capture that failed on ANOTHER rank: every rank goes on with eager launches (a rank calibrating a replay the
            # others do not have would enqueue collectives they never join)
            replay = None
            loop.release_graph()
            graph_note = "graph capture failed on another rank, eager launches"
        if replay is not None:
            per = int(getattr(loop, "replay_steps", 1))
            # Which launch path is faster depends on the workload: a replayed graph costs the host ~0.02 ms per
            # step but serialises its kernel nodes a little more loosely than back-to-back stream launches, an
            # eager step costs the host 0.3-1.2 ms.  Calibrate outside the timed region (every rank takes the same
            # decision: the slowest rank's times count).
            def timed(fn, k=1, n=max(8, min(args.steps, 30))):
                """seconds per STEP of n steps (fn makes k steps per call)"""
                n = max(k, n // k * k)
                best, host = None, 0.0
                for _ in range(2):  # the better of two regions: a region of a few ms is at the mercy of a clock ramp
                    fn()
                    loop.flush()
                    barrier()
                    t = time.perf_counter()
                    for _ in range(n // k):
                        fn()
                    host = max(host, (time.perf_counter() - t) / n)  # what the host needed to enqueue a step
                    loop.flush()
                    barrier()
                    dt = (time.perf_counter() - t) / n
                    best = dt if best is None else min(best, dt)
                return best, host

            (t_graph, _), (t_eager, host_eager) = timed(replay, per), timed(loop.step)
            if dist is not None:
                tt = torch.tensor([t_graph, t_eager, host_eager], device=device, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                t_graph, t_eager, host_eager = tt.tolist()
            # A tie goes to the graph: the same GPU time for 0.02 ms instead of 0.3-1.2 ms of host time per step.
            # (round 4: 1 % everywhere — where the eager path is close to host-bound the calibration's own eager time already
            # shows it; the 3 % concession round 3 made there could put the slower path in the headline)
            margin = 1.01
            use_graph = launch == "graph" or (launch == "auto" and t_graph <= margin * t_eager)
            what = ((f"one HIP graph per {per} consecutive steps" if per > 1 else "one HIP graph per step") +
                    " (sampler, forwards, head, backward, wgrad, update; indices from the loop's index pool)" if world == 1 else
                    "three HIP graphs per step (sample | update | forward+backward), the RCCL all-reduce of the gradient "
                    "slab launched eagerly between them")
            graph_note = (f"{'graph replay: ' + what if use_graph else 'eager stream launches'}; calibration "
                          f"{t_graph * 1e3:.3f} ms/step replayed vs {t_eager * 1e3:.3f} ms/step eager "
                          f"(eager host enqueue {host_eager * 1e3:.3f} ms/step)")
            calibration = {"graph_ms_per_step": t_graph * 1e3, "eager_ms_per_step": t_eager * 1e3,
                           "eager_host_enqueue_ms_per_step": host_eager * 1e3, "chosen": "graph" if use_graph else "eager",
                           "rule": f"graph unless it is more than {round((margin - 1) * 100)} % slower than eager launches"}
            calibration["steps_per_graph"] = per
            step = replay if use_graph else loop.step
            replayed = use_graph
            steps_per_call = per if use_graph else 1
            if not use_graph:
                loop.release_graph()  # eager steps then pass Adam's coefficients per launch (no tick kernel)
    for _ in range((args.warmup + steps_per_call - 1) // steps_per_call):
        step()
    loop.flush()
    # K steps take ~12 ms at C2: one region is at the mercy of a clock ramp or a stray interrupt.  The region of
    # EXACTLY K steps (barrier + synchronize on both sides, max over ranks) is therefore timed `--repeats` times
    # back to back and the MEDIAN region is reported; every region's time is listed in `region_ms`.
    regions, host_regions, own_regions = [], [], []
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps // steps_per_call):  # EXACTLY args.steps steps (steps_per_call divides it, see capture)
            loss = step()
        loop.flush()  # data parallel: the last step's update joins its all-reduce inside the timed region
        host_regions.append(time.perf_counter() - t0)  # time the host needed to ENQUEUE the steps (diagnostic)
        barrier()
        dt = time.perf_counter() - t0
        own_regions.append(dt)
        if dist is not None:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        regions.append(dt)
    order = sorted(range(len(regions)), key=lambda i: regions[i])
    mid = order[len(order) // 2]
    dt, host_dt = regions[mid], host_regions[mid]
    if isinstance(loss, dict):
        loss = loss["q1_loss"]
    loss_val = float(loss.item())

    # ---- one LONG region (default 4000 steps, seconds not milliseconds) with the clock and the power sampled inside it:
    # the K-step regions above are ~10 ms bursts after an idle gap, and this chip's clock under dense MFMA is power-managed
    sustained = None
    n_sus = int(getattr(args, "sustained_steps", 0) or 0)
    if n_sus > 0:
        barrier()
        with GpuTelemetry(device.index or 0) as tele:
            t0 = time.perf_counter()
            for _ in range(n_sus // steps_per_call):
                step()
            loop.flush()
            barrier()
            sdt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([sdt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sdt = t.item()
        sustained = {"steps": n_sus, "seconds": sdt, "ms_per_step": sdt / n_sus * 1e3, "value": world * args.batch * n_sus / sdt,
                     "unit": "transitions/s", **tele.summary()}

    extra = {}
    if not args.no_kernel_profile:
        # every rank runs the instrumented steps (they contain the gradient all-reduce: a pass on rank 0
        # alone would never return); rank 0 reports.  Eager launches: events bracket each C-ABI call.
        n_prof = profile_steps or min(args.steps, 10)
        extra = kernel_profile(args, loop.step, n_prof, device, gpu_ms=dt / args.steps * 1e3)
        loop.flush()
        gq = trainer._grouped() if (args.algo == "qrdqn" and getattr(args, "grouped_head", False)) else None
        two_streams = gq is not None and (getattr(gq, "two_streams", False) or getattr(gq, "wgrad_streams", False))
        extra["instrumented_pass"]["timed_ms_per_step"] = dt / args.steps * 1e3
        extra["instrumented_pass"]["launch_streams"] = 2 if two_streams else 1
    per_rank = None
    if dist is not None:
        # every rank's own view, so a curve measured by the driver explains itself: the rank's wall time for the median
        # region and the HIP-event time of the gradient all-reduce (instrumented pass)
        try:
            ncpu = float(len(os.sched_getaffinity(0)))
        except Exception:
            ncpu = float("nan")
        mine = torch.tensor([own_regions[mid] / args.steps * 1e3, host_dt / args.steps * 1e3,
                             extra.get("all_reduce_us", float("nan")), ncpu], device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": i, "ms_per_step": t[0].item(), "host_enqueue_ms_per_step": t[1].item(),
                     "all_reduce_us": t[2].item(), "cpus_in_affinity": t[3].item()} for i, t in enumerate(allr)]
    return {"value": world * args.batch * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
            "timing": f"median of {len(regions)} regions of {args.steps} steps each",
            "region_ms": [round(r * 1e3, 4) for r in regions], "host_enqueue_ms_per_step": host_dt / args.steps * 1e3,
            "final_loss": loss_val, "launch": (graph_note or "eager launches"), "launch_calibration": calibration,
            "sustained": sustained, "extra": extra, "parity": parity,
            "per_rank": per_rank, "init": init, "cols": cols, "cols_cpu": cols, "norm": norm}


def _sig(x, n=6):
    """floats to n significant digits (the line is a digest; the report file keeps every digit)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{n}g}")


def _pick(o, *ks):
    return {k: _sig(o[k]) for k in ks if isinstance(o, dict) and k in o and not isinstance(o[k], (dict, list))}


def _parity_digest(pp):
    """ok / meets_north_star / the largest Q (or quantile, or logit) difference from the CPU oracle / gather bit-exact;
    `dw_flip_frac` = share of post-step weights further than 2e-5 from the oracle's (an Adam step of a weight whose
    gradient's sign differs moves it by 2*lr)"""
    if not isinstance(pp, dict):
        return None
    d = _pick(pp, "ok", "meets_north_star", "sane", "batch", "max_abs_dq", "max_abs_dquantile", "max_abs_dlogits",
              "gather_fields_bit_exact")
    if "frac_dw_beyond_2e-5" in pp:
        d["dw_flip_frac"] = _sig(pp["frac_dw_beyond_2e-5"], 3)
    if "error" in pp:
        d["error"] = str(pp["error"])[:120]
    return d


def _mode_digest(o, with_parity=True):
    """one measured region (a configuration in one precision) in six-odd fields"""
    if not isinstance(o, dict):
        return None
    if "error" in o:
        return {"dtype": o.get("dtype"), "error": str(o["error"])[:160]}
    d = _pick(o, "dtype", "value", "ms_per_step")
    fc = o.get("fc_roofline") if isinstance(o.get("fc_roofline"), dict) else {}
    if "frac" in fc:
        d["whole_fc_frac"] = _sig(fc["frac"], 4)
    if "executed_frac" in fc:
        d["executed_frac"] = _sig(fc["executed_frac"], 4)
    if isinstance(o.get("sustained"), dict):
        d["sustained_ms_per_step"] = _sig(o["sustained"].get("ms_per_step"))
    if with_parity:
        d["parity"] = _parity_digest(o.get("parity"))
    return d


def compact_line(res):
    """The ONE line rank 0 prints: the contract's keys, `roofline`, `cpu_baseline`, and six-field digests of the 1e-4-compliant
    mode (`compliant`, the co-headline: the mode north_star's tolerance binds) and of the other single-GPU configurations
    (`c3`, `c4`).  Everything else bench.py measured is in the report file the line names.  Never longer than LINE_LIMIT bytes."""
    line = {k: _sig(res[k], 10) for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step",
                                      "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in res}
    cfg = res.get("config", {})
    line["config"] = {"workload": str(cfg.get("workload", ""))[:200], **_pick(cfg, "name", "global_batch", "parallelism"),
                      "launch": str(cfg.get("launch", "")).split(":")[0].split(";")[0][:40]}
    if isinstance(res.get("sustained"), dict):
        line["sustained_ms_per_step"] = _sig(res["sustained"].get("ms_per_step"))
        sc = res["sustained"].get("sclk_mhz")
        if isinstance(sc, dict) and sc.get("mean") is not None:
            line["sclk_mhz"] = _sig(sc["mean"], 4)
    roof = res.get("roofline")
    if isinstance(roof, dict):
        r = _pick(roof, "bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_us", "launches_per_step", "traffic",
                  "algorithmic_bytes", "algorithmic_gflop_per_launch", "executed_frac")
        r.setdefault("traffic", None)
        fc = res.get("fc_roofline") if isinstance(res.get("fc_roofline"), dict) else {}
        if "frac" in fc:
            r["whole_fc_frac"] = _sig(fc["frac"], 4)
            r["whole_fc_ms_per_step"] = _sig(fc.get("fc_ms_per_step"), 4)
        g = res.get("gather") if isinstance(res.get("gather"), dict) else {}
        if "frac" in g:
            r["gather_hbm_frac"] = _sig(g["frac"], 4)
            r["gather_us"] = _sig(g.get("avg_launch_us"), 4)
        line["roofline"] = r
    line["parity"] = _parity_digest(res.get("parity"))
    if isinstance(res.get("accurate"), dict):
        line["compliant"] = _mode_digest(res["accurate"])
    for cfg_name, o in (res.get("also_measured") or {}).items():
        d = _mode_digest(o)
        if isinstance(o, dict) and isinstance(o.get("accurate"), dict):
            d["compliant"] = _mode_digest(o["accurate"])
        if cfg_name == "c3" and isinstance(d, dict) and "error" not in d:
            d["full_size_check"] = "grouped vs dense fp32 (HIP vs HIP); oracle parity at B<=8192"
        line[cfg_name] = d
    if res.get("per_rank"):
        line["per_rank_ms"] = [_sig(p.get("ms_per_step"), 5) for p in res["per_rank"]]
        ar = [p.get("all_reduce_us") for p in res["per_rank"] if p.get("all_reduce_us") == p.get("all_reduce_us")]
        if ar:
            line["all_reduce_us_max"] = _sig(max(ar), 4)
    pf = res.get("device_preflight")
    if isinstance(pf, dict):
        line["device_preflight"] = _pick(pf, "ok", "adopted")
        if pf.get("adopted"):  # a runtime workaround was needed on this node: not comparable with a healthy node's number
            line["value_valid"] = False
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, "value", "unit", "cores", "kind", "ms_per_step", "host_cpus")
        if "sample" in cb:
            c["sample"] = str(cb["sample"])[:150]
        if "error" in cb:
            c["error"] = str(cb["error"])[:160]
        line["cpu_baseline"] = c
    line["report"] = res.get("report_file", REPORT_FILE)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    # the line must stay parseable whatever a sub-measurement produced: shed the optional digests, largest first
    keep = {"metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "report", "shed", "value_valid"}
    while len(text.encode()) > LINE_LIMIT:
        optional = sorted((k for k in line if k not in keep), key=lambda k: -len(json.dumps(line[k])))
        if not optional:
            break
        line.pop(optional[0])
        line["shed"] = line.get("shed", 0) + 1
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(text.encode()) <= LINE_LIMIT, len(text)
    return text


def write_report(res, path=None):
    """the full record (every object of the run) beside bench.py, and under gpurun_out/ when that exists so a gpurun call brings
    it back; returns the path written (None if the directory is read-only — the line must still be printed)"""
    def clean(o):
        if isinstance(o, float) and (o != o or o in (float("inf"), float("-inf"))):
            return None
        if isinstance(o, dict):
            return {str(k): clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        return o

    text = json.dumps(clean(res), indent=1)
    written = None
    targets = [path or os.path.join(ROOT, REPORT_FILE)]
    if path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        targets.append(os.path.join(ROOT, "gpurun_out", REPORT_FILE))
    for t in targets:
        try:
            with open(t, "w") as fh:
                fh.write(text + "\n")
            written = written or t
        except OSError:
            pass
    return written


def main(device=None, backend="nccl"):
    """`device` / `backend` are None / "nccl" for every real run.  tests/bench_on_emu.py passes (cpu, "gloo") AFTER patching the
    C ABI to the host-compiled kernel sources, to run this function's whole control flow — launcher, ranks, secondary regions,
    report file, the one printed line — under torchrun on a box without GPUs; bench.py itself has no CPU path."""
    args = parse()
    under_torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not under_torchrun:
        relaunch_under_torchrun(args)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:  # before this process's first device touch: dmabuf IPC, the only form this pool's host driver supports for RCCL
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    preflight = None
    if device is None and world == 1 and os.path.exists("/dev/kfd") and not args.rendezvous_only and not os.environ.get("RG_SKIP_PREFLIGHT"):
        # One GPU: the torch-only first device touch in a SUBPROCESS, before this process starts its HSA runtime (so before
        # torch.cuda.is_available()).  A lease whose first touch faults (round 4's driver GPU record died that way) is tried
        # under the runtime alternatives of reagent_amd.device_preflight and named if none works: the record then blames the
        # node, not the benchmark.  (N > 1: every rank would spawn its own child on a shared host; the launcher's ranks go
        # straight to their devices.)
        from reagent_amd.device_preflight import NODE_FAULT, settle

        ok, log, adopted = settle()
        preflight = {"ok": bool(ok), "adopted": adopted, "attempts": len(log.splitlines())}
        if not ok and not log.endswith("no GPU visible to torch"):
            print(f"{NODE_FAULT}\n{log}", file=sys.stderr)
            raise SystemExit(97)
    have_gpu = torch.cuda.is_available()
    if args.rendezvous_only:  # launcher self-test (CPU boxes use gloo)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl" if have_gpu else "gloo", rank=rank, world_size=world)
        t = torch.ones(1, device="cuda" if have_gpu else "cpu")
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "ranks": int(t.item()), "backend": dist.get_backend(), "n_gpus": args.gpus}))
        dist.destroy_process_group()
        return
    if not have_gpu and device is None:
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False and there is no CPU fallback "
                         f"(rank {rank} of {world})")
    if device is None:
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {rank} has no GPU (device_count {torch.cuda.device_count()}, --gpus {args.gpus})")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    affinity = bind_rank_to_gpu_numa(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world))) if world > 1 else None
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if device.type == "cuda":
            dist.init_process_group(backend, device_id=device)  # "nccl" IS RCCL on ROCm
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus
    m = measure(args, device, rank, world, dist)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        res = {
            "metric": "transitions/sec at batch=65536 state_dim=128; 1/2/4/8 MI355X scaling",
            "value": m["value"],
            "unit": "transitions/s",
            "n_gpus": world, "rccl_ranks": world if dist is not None else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m["ms_per_step"],
            "timing": m["timing"],
            "region_ms": m["region_ms"],
            "host_enqueue_ms_per_step": m["host_enqueue_ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{CONFIGS[args.config]['name']}; batch={args.batch}/GPU, {args.layers}x{args.hidden} hidden",
                       "name": args.config, "global_batch": world * args.batch, "replay_capacity_per_gpu": args.capacity,
                       "parallelism": f"dp{world}", "final_loss": m["final_loss"], "launch": m["launch"]},
        }
        if world > 1:
            res["per_rank"] = m["per_rank"]
            res["cpu_affinity_rank0"] = affinity
            assert res["rccl_ranks"] == args.gpus == world, "the line must describe the group that ran"
        res["launch_calibration"] = m["launch_calibration"]
        if preflight is not None:
            res["device_preflight"] = preflight
        if m["sustained"] is not None:
            res["sustained"] = m["sustained"]
        res.update(m["extra"])
        if m["parity"] is not None:
            res["parity"] = m["parity"]
    def sub_object(mm, a, note=None):
        """what a secondary region contributes to the line"""
        o = {"dtype": a.precision, "value": mm["value"], "unit": "transitions/s", "ms_per_step": mm["ms_per_step"],
             "steps": a.steps, "timing": mm["timing"], "region_ms": mm["region_ms"], "launch": mm["launch"],
             "launch_calibration": mm["launch_calibration"]}
        if note:
            o["note"] = note
        if mm["sustained"] is not None:
            o["sustained"] = mm["sustained"]
        o.update({k: mm["extra"][k] for k in ("roofline", "fc_roofline", "instrumented_pass") if k in mm["extra"]})
        o["parity"] = mm["parity"]
        return o

    X3_NOTE = ("same workload, shard, initial weights and K-step region as its bf16 sibling, every FC operand split hi + lo "
               "(three bf16 MFMAs per product, fp32 accumulate): the mode held to north_star's 1e-4")
    # Secondary regions run on ONE rank only: with world > 1 a rank-local failure inside one of them (say an OOM) would
    # leave the other ranks waiting in its collectives; the scaling curve needs the headline value alone.
    secondary = world == 1
    # ---- the same K-step region in the mode that meets north_star's floating-point tolerance (Q within 1e-4 of the fp32
    # reference, dqn_trainer.py:204-238): split-bf16 operands on the bf16 MFMA pipe.  Same shard, same initial weights.
    if args.precision == "bf16" and not args.no_accurate and secondary:
        a2 = argparse.Namespace(**vars(args))
        a2.precision = "bf16x3"
        a2.repeats = max(1, min(args.repeats, 3))
        a2.sustained_steps = args.sustained_steps // 2
        try:
            ma = measure(a2, device, rank, world, dist, cols=m["cols"], profile_steps=min(args.steps, 6))
            res["accurate"] = sub_object(ma, a2, X3_NOTE)
            del ma
        except Exception as e:  # never fatal to the headline measurement
            res["accurate"] = {"dtype": "bf16x3", "error": repr(e)}
        torch.cuda.empty_cache()
    # ---- the other single-GPU BASELINE configurations, a few regions each in BOTH modes, so that every configuration has a
    # driver-timed throughput AND a driver-timed 1e-4-compliant throughput (`accurate`)
    if args.config == "c2" and secondary and not args.no_also:
        also = {}
        for cfg in ("c3", "c4"):
            c = CONFIGS[cfg]
            a3 = argparse.Namespace(**vars(args))
            a3.config, a3.repeats, a3.sustained_steps = cfg, 3, 0
            a3.state_dim, a3.actions, a3.algo, a3.atoms = c["state_dim"], c["actions"], c["algo"], c["atoms"]
            a3.no_parity = False
            shared_cols = m["cols"] if cfg == "c3" else None
            try:
                mo = measure(a3, device, rank, world, dist, cols=shared_cols, profile_steps=min(args.steps, 4))
                also[cfg] = {"workload": c["name"], **sub_object(mo, a3)}
                shared_cols = mo["cols"]
                del mo
            except Exception as e:  # never fatal to the headline measurement
                also[cfg] = {"error": repr(e)}
            torch.cuda.empty_cache()
            if args.precision == "bf16" and not args.no_accurate and "error" not in also[cfg]:
                a4 = argparse.Namespace(**vars(a3))
                a4.precision = "bf16x3"
                try:
                    mo = measure(a4, device, rank, world, dist, cols=shared_cols, profile_steps=min(args.steps, 4))
                    also[cfg]["accurate"] = sub_object(mo, a4, X3_NOTE)
                    del mo
                except Exception as e:
                    also[cfg]["accurate"] = {"dtype": "bf16x3", "error": repr(e)}
                torch.cuda.empty_cache()
            del shared_cols
        res["also_measured"] = also
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(args, m["init"], m["cols_cpu"], m["norm"])
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"error": repr(e)}
        where = write_report(res, args.report)
        res["report_file"] = os.path.relpath(where, ROOT) if where else None
        sys.stderr.flush()
        print(compact_line(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
