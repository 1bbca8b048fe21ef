#!/usr/bin/env python3
"""bench.py — transitions/sec of the DQN hot path (BASELINE.json configs[1]: Discrete DQN,
state_dim=128, |A|=16, 3x512 MLP, batch=65536, bf16) on N MI355X of one node.

One "step" = one pass of the whole hot path over one minibatch, everything already resident in HBM:
replay index sampling -> gather (rg_replay_*) -> input maker -> dense normalization x2 ->
3 FC forwards + TD/Huber head + FC backward -> [N>1: RCCL all-reduce of the flat fp32 gradient
slab] -> fused Adam + soft target update.

N>1 is launched by torch.distributed.run (one rank per GPU); every rank owns a disjoint shard of
the offline dataset (its own replay buffer), per-rank batch stays 65536 (weak scaling) and the
only collective is the gradient all-reduce.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     — dominant FC kernel: algorithmic FLOP of its launches / HIP-event time of those
                 launches (events on the launch stream, measured in a second, instrumented pass of
                 the same K steps so the timed region itself stays un-instrumented)
  fc_roofline  — all FC kernels together against BASELINE.md's 5.849 MFLOP/transition
  cpu_baseline — the CPU oracle (torch-CPU restatement of the reference step, oracle/restated.py)
                 timed on this box's host cores, rank 0, N == 1 only
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK = {"bf16": 2.5e15, "f32": 157.3e12}  # /opt/skills/guides/MI355X_MICROARCH.md:40-42
HBM_PEAK = 8.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--state-dim", type=int, default=128)
    ap.add_argument("--actions", type=int, default=16)
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--capacity", type=int, default=1 << 20)
    ap.add_argument("--precision", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--prefetch", action="store_true", help="gather the next batch on a second stream (slower, see runtime.py)")
    ap.add_argument("--cpu-steps", type=int, default=4)
    return ap.parse_args()


def fc_flops(layer_dims, batch):
    """Algorithmic FLOP of one DQN step: 3 forwards + wgrad(all) + dgrad(all but first)."""
    fwd = sum(a * b for a, b in zip(layer_dims, layer_dims[1:]))
    dgrad = sum(a * b for a, b in zip(layer_dims[1:], layer_dims[2:]))
    return 2 * batch * (3 * fwd + fwd + dgrad)


def build(args, device, rank):
    import reagent_amd._lib as L
    from reagent_amd import synthetic
    from reagent_amd.core.parameters import EvaluationParameters, NormalizationParameters, RLParameters
    from reagent_amd.models import FullyConnectedDQN, set_default_precision
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.preprocessing import Preprocessor
    from reagent_amd.replay_memory import ReplayBuffer
    from reagent_amd.runtime import OfflineDqnLoop
    from reagent_amd.training import DQNTrainer

    S, A, H = args.state_dim, args.actions, args.hidden
    set_default_precision(L.PREC_BF16 if args.precision == "bf16" else L.PREC_F32)
    torch.manual_seed(0)  # identical initial weights on every rank
    q = FullyConnectedDQN(S, A, [H] * args.layers, ["relu"] * args.layers)
    init = [p.detach().clone() for p in q.parameters()]
    q = q.to(device)
    trainer = DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(A)],
                         rl=RLParameters(gamma=0.99, target_update_rate=0.001, maxq_learning=True,
                                         q_network_loss="huber"),
                         double_q_learning=True, optimizer=Optimizer__Union.default(lr=1e-3),
                         evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(device)
    # this rank's shard of the offline dataset (different seed per rank), resident in HBM
    cols = synthetic.replay_contents(args.capacity, S, A, seed=100 + rank)
    rb = ReplayBuffer(replay_capacity=args.capacity, batch_size=args.batch, device=device)
    rb.load_columns({k: v.to(device) for k, v in cols.items()}, mark_all_valid=True)
    g = torch.Generator().manual_seed(7)
    mean, std = torch.randn(S, generator=g), torch.rand(S, generator=g) * 1.5 + 0.5
    norm = {i: NormalizationParameters(feature_type="CONTINUOUS", mean=mean[i].item(), stddev=std[i].item())
            for i in range(S)}
    pre = Preprocessor(norm, device=device)
    loop = OfflineDqnLoop(rb, trainer, args.batch, pre,
                          state_dtype=torch.bfloat16 if args.precision == "bf16" else torch.float32,
                          prefetch=args.prefetch)
    return loop, trainer, init, cols, (mean, std)


def cpu_baseline(args, init, cols, norm):
    """The reference step restated on torch-CPU (oracle/restated.py), on this box's host cores:
    numpy/torch gather of the same columns, (x-mean)/std normalization, DQN step.  Bounded sample."""
    from oracle import restated as R

    B, S, A = args.batch, args.state_dim, args.actions
    mean, std = norm
    acts = ["relu"] * args.layers + ["linear"]
    o = R.DQNOracle(init, init, acts, gamma=0.99, tau=0.001, loss="huber", lr=1e-3)
    g = torch.Generator().manual_seed(3)
    C = args.capacity

    def one():
        idx = torch.randint(C, (B,), generator=g)
        nxt = (idx + 1) % C
        state = torch.clamp((cols["observation"][idx] - mean) / std, -11.513, 11.513)
        next_state = torch.clamp((cols["observation"][nxt] - mean) / std, -11.513, 11.513)
        term = cols["terminal"][idx]
        b = dict(state=state, next_state=next_state,
                 action=torch.nn.functional.one_hot(cols["action"][idx], A).float(),
                 next_action=torch.nn.functional.one_hot(cols["action"][nxt], A).float(),
                 reward=cols["reward"][idx].unsqueeze(1), not_terminal=1.0 - term.float().unsqueeze(1),
                 possible_next_actions_mask=cols["possible_actions_mask"][nxt],
                 possible_actions_mask=cols["possible_actions_mask"][idx])
        o.step(b)

    one()  # warm-up
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        one()
    dt = time.perf_counter() - t0
    return {"value": args.cpu_steps * B / dt, "unit": "transitions/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{args.cpu_steps} steps of the same workload (B={B}, gather+normalize+DQN step, fp32, "
                      f"torch-CPU restatement of the reference trainer; {dt / args.cpu_steps * 1e3:.0f} ms/step)"}


def kernel_profile(args, loop, steps, layer_dims):
    """Instrumented pass: HIP events around every C-ABI launch (on the launch stream)."""
    from reagent_amd import ops

    B = args.batch
    with ops.profile() as prof:
        for _ in range(steps):
            loop.step()
    rows = prof.summary()
    fc_names = ("rg_fc_forward", "rg_fc_dgrad", "rg_fc_wgrad", "rg_fc_wgrad_frag", "rg_mlp_forward_fused",
                "rg_mlp_backward_fused", "rg_mlp_wgrad_fused")
    fc = [r for r in rows if r["name"] in fc_names]
    for r in fc:
        m = r["meta"]
        if r["name"] == "rg_mlp_forward_fused":
            d = m["dims"]
            r["flop_per_launch"] = 2.0 * m["B"] * sum(a * b for a, b in zip(d, d[1:]))
            r["label"] = f"rg_mlp_forward_fused B={m['B']} dims={list(d)} save={m['save']}"
        elif r["name"] == "rg_mlp_wgrad_fused":
            d = m["dims"]
            r["flop_per_launch"] = 2.0 * m["B"] * sum(a * b for a, b in zip(d, d[1:]))
            r["label"] = f"rg_mlp_wgrad_fused B={m['B']} dims={list(d)}"
        elif r["name"] == "rg_mlp_backward_fused":
            d = m["dims"]
            r["flop_per_launch"] = 2.0 * m["B"] * sum(a * b for a, b in zip(d[1:], d[2:]))
            r["label"] = f"rg_mlp_backward_fused B={m['B']} dims={list(d)}"
        else:
            r["flop_per_launch"] = 2.0 * m["M"] * m["N"] * m["K"]
            r["label"] = f"{r['name']} M={m['M']} N={m['N']} K={m['K']}"
    peak = MFMA_PEAK[args.precision]
    out = {}
    if fc:
        dom = fc[0]
        sec = dom["ms"] * 1e-3 / dom["calls"]
        ach = dom["flop_per_launch"] / sec
        out["roofline"] = {"bound": "mfma", "kernel": dom["label"],
                           "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak,
                           "avg_launch_us": sec * 1e6, "launches_per_step": dom["calls"] / steps, "traffic": None}
        fc_ms = sum(r["ms"] for r in fc) / steps
        alg = fc_flops(layer_dims, B)
        out["fc_roofline"] = {"algorithmic_gflop_per_step": alg / 1e9, "fc_ms_per_step": fc_ms,
                              "achieved": alg / (fc_ms * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                              "frac": alg / (fc_ms * 1e-3) / peak}
    g = [r for r in rows if r["name"] in ("rg_replay_dqn_batch", "rg_replay_gather")]
    if g:
        sec = g[0]["ms"] * 1e-3 / g[0]["calls"]
        bytes_ = g[0]["meta"]["bytes_per_row"] * B
        out["gather"] = {"bound": "hbm", "achieved": bytes_ / sec / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": bytes_ / sec / HBM_PEAK, "avg_launch_us": sec * 1e6,
                         "algorithmic_bytes_per_transition": g[0]["meta"]["bytes_per_row"], "kernel": g[0]["name"]}
    out["per_call_ms_per_step"] = {f"{r['name']}{tuple(r['meta'].values())}": round(r["ms"] / steps, 4) for r in rows[:18]}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)  # "nccl" IS RCCL on ROCm
    loop, trainer, init, cols, norm = build(args, device, rank)
    if world > 1:
        trainer.enable_data_parallel()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loop.step()
    loop.flush()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = loop.step()
    loop.flush()  # data parallel: the last step's update joins its all-reduce inside the timed region
    host_dt = time.perf_counter() - t0  # time the host needed to ENQUEUE the steps (diagnostic)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_val = float(loss.item())

    layer_dims = [args.state_dim] + [args.hidden] * args.layers + [args.actions]
    extra = {}
    if not args.no_kernel_profile:
        # every rank runs the instrumented steps (they contain the gradient all-reduce: a pass on rank 0
        # alone would never return); rank 0 reports
        extra = kernel_profile(args, loop, min(args.steps, 10), layer_dims)
        loop.flush()
    if dist is not None:
        dist.barrier()
    if rank == 0:
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        # the committed PMC figure belongs to the fused forward (save=0) launch only
        if "roofline" in extra and os.path.exists(traffic_file) and "forward_fused" in extra["roofline"].get("kernel", ""):
            try:
                extra["roofline"]["traffic"] = json.load(open(traffic_file)).get("hbm_bytes_per_launch")
            except Exception:
                pass
        res = {
            "metric": "transitions/sec at batch=65536 state_dim=128; 1/2/4/8 MI355X scaling",
            "value": world * args.batch * args.steps / dt,
            "unit": "transitions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "host_enqueue_ms_per_step": host_dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"Discrete DQN state_dim={args.state_dim} |A|={args.actions} "
                                   f"{args.layers}x{args.hidden} MLP batch={args.batch}/GPU "
                                   f"(BASELINE.json configs[1]; replay gather + normalize + 3 fwd + TD/Huber + bwd + Adam + soft update)",
                       "global_batch": world * args.batch, "replay_capacity_per_gpu": args.capacity,
                       "parallelism": f"dp{world}", "final_loss": loss_val},
        }
        res.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(args, init, cols, norm)
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
