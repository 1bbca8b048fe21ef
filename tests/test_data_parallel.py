"""Data-parallel path: one process per shard, gradient slab all-reduced (sum) after
backward, 1/world folded into the fused Adam.  world_size = 2 over gloo on the CPU, kernels through
the host-compiled sources (tests/emu) — checks the HOST logic: two ranks on disjoint half-batches end
with identical parameters, equal to one process stepping on the concatenated batch (every loss on
this path is a batch mean)."""
import os
import sys

import pytest

from conftest import free_port
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(kind):
    import reagent_amd._lib as L
    from reagent_amd.core.parameters import EvaluationParameters, RLParameters
    from reagent_amd.models import FullyConnectedCritic, FullyConnectedDQN, GaussianFullyConnectedActor
    from reagent_amd.optimizer import Optimizer__Union
    from reagent_amd.training import DQNTrainer, SACTrainer

    torch.manual_seed(0)  # identical initial weights on every rank
    if kind == "qr_fused":  # QR-DQN on the grouped engine (qr_engine.py): the one-launch update with the wide layer as a grouped layer
        import reagent_amd._lib as L
        from reagent_amd.models import set_default_precision
        from reagent_amd.training import QRDQNTrainer

        set_default_precision(L.PREC_BF16)
        try:
            q = FullyConnectedDQN(12, 4, [256, 256], ["relu", "relu"], num_atoms=16)
        finally:
            set_default_precision(L.PREC_F32)
        return QRDQNTrainer(q, q.get_target_network(), actions=["a", "b", "c", "d"], num_atoms=16,
                            rl=RLParameters(gamma=0.9, target_update_rate=0.1, maxq_learning=True), double_q_learning=True,
                            optimizer=Optimizer__Union.default(lr=0.001), evaluation=EvaluationParameters(calc_cpe_in_training=False))
    if kind in ("dqn_fused", "sac_fused", "dqn_x3"):  # fused kernels (bf16 / split-bf16): the one-launch updates under 1/world scaling
        import reagent_amd._lib as L
        from reagent_amd.models import set_default_precision

        set_default_precision(L.PREC_BF16X3 if kind == "dqn_x3" else L.PREC_BF16)
        try:
            if kind in ("dqn_fused", "dqn_x3"):
                q = FullyConnectedDQN(12, 4, [256, 256], ["relu", "relu"])
                return DQNTrainer(q, q.get_target_network(), None, actions=["a", "b", "c", "d"],
                                  rl=RLParameters(gamma=0.9, target_update_rate=0.1, q_network_loss="huber"),
                                  optimizer=Optimizer__Union.default(lr=0.001),
                                  evaluation=EvaluationParameters(calc_cpe_in_training=False))
            nets = [GaussianFullyConnectedActor(32, 2, [256, 256], ["relu", "relu"]), FullyConnectedCritic(32, 2, [256, 256], ["relu", "relu"]),
                    FullyConnectedCritic(32, 2, [256, 256], ["relu", "relu"])]
        finally:
            set_default_precision(L.PREC_F32)
        adam = lambda: Optimizer__Union.default(lr=0.001)  # noqa: E731
        return SACTrainer(nets[0], nets[1], nets[2], rl=RLParameters(gamma=0.9, target_update_rate=0.1), q_network_optimizer=adam(),
                          actor_network_optimizer=adam(), alpha_optimizer=adam())
    if kind.startswith("dqn"):
        q = FullyConnectedDQN(12, 4, [32, 16], ["relu", "relu"])
        return DQNTrainer(q, q.get_target_network(), None, actions=["a", "b", "c", "d"],
                          rl=RLParameters(gamma=0.9, target_update_rate=0.1, q_network_loss="huber"),
                          optimizer=Optimizer__Union.default(lr=0.01),
                          evaluation=EvaluationParameters(calc_cpe_in_training=False))
    if kind == "crr":  # twin critics, actor, CPE nets: five Adam-stepped networks, four targets
        from reagent_amd.models import FullyConnectedActor
        from reagent_amd.training import DiscreteCRRTrainer

        mk = lambda out: FullyConnectedDQN(12, out, [32, 16], ["relu", "relu"])  # noqa: E731
        actor, c1, c2, rn, qc = FullyConnectedActor(12, 4, [32, 16], ["relu", "relu"]), mk(4), mk(4), mk(4), mk(4)
        adam = lambda: Optimizer__Union.default(lr=0.01)  # noqa: E731
        return DiscreteCRRTrainer(
            actor_network=actor, actor_network_target=actor.get_target_network(), q1_network=c1,
            q1_network_target=c1.get_target_network(), reward_network=rn, q2_network=c2,
            q2_network_target=c2.get_target_network(), q_network_cpe=qc, q_network_cpe_target=qc.get_target_network(),
            metrics_to_score=[], evaluation=EvaluationParameters(calc_cpe_in_training=True),
            rl=RLParameters(gamma=0.9, target_update_rate=0.1), q_network_optimizer=adam(),
            actor_network_optimizer=adam(), actions=["a", "b", "c", "d"], entropy_coeff=0.05)
    if kind == "td3":
        from reagent_amd.models import FullyConnectedActor
        from reagent_amd.training import TD3Trainer

        return TD3Trainer(FullyConnectedActor(6, 2, [16, 16], ["relu", "relu"]),
                          FullyConnectedCritic(6, 2, [16, 16], ["relu", "relu"]),
                          FullyConnectedCritic(6, 2, [16, 16], ["relu", "relu"]),
                          rl=RLParameters(gamma=0.9, target_update_rate=0.1),
                          q_network_optimizer=Optimizer__Union.default(lr=0.01),
                          actor_network_optimizer=Optimizer__Union.default(lr=0.01), delayed_policy_update=1)
    actor = GaussianFullyConnectedActor(6, 2, [16, 16], ["relu", "relu"])
    q1 = FullyConnectedCritic(6, 2, [16, 16], ["relu", "relu"])
    q2 = FullyConnectedCritic(6, 2, [16, 16], ["relu", "relu"])
    return SACTrainer(actor, q1, q2, rl=RLParameters(gamma=0.9, target_update_rate=0.1),
                      q_network_optimizer=Optimizer__Union.default(lr=0.01),
                      actor_network_optimizer=Optimizer__Union.default(lr=0.01),
                      alpha_optimizer=Optimizer__Union.default(lr=0.01))


def _batches(kind, B):
    from reagent_amd import synthetic

    if kind == "sac_fused":
        return synthetic.policy_batch(B, 32, 2, seed=5)
    if kind.startswith("dqn") or kind == "qr_fused":
        return synthetic.dqn_batch(B, 12, 4, seed=5, p_impossible=0.2)
    if kind == "crr":
        return synthetic.dqn_batch(B, 12, 4, seed=5, p_impossible=0.2, with_propensity=True)
    return synthetic.policy_batch(B, 6, 2, seed=5)


def _step(kind, tr, d, noise=None):
    from reagent_amd import synthetic

    if kind == "dqn_generator":  # the Lightning protocol: the CALLER steps the optimizers (no grad_scale set by us)
        from test_dqn_trainer import lightning_like_step

        if not hasattr(tr, "_test_opts"):
            tr._test_opts = [o["optimizer"] for o in tr.configure_optimizers()]
        lightning_like_step(tr, tr._test_opts, synthetic.to_dqn_input(d))
    elif kind == "dqn_deferred":  # async all-reduce, Adam joined at the start of the next step
        tr.train_step_native(synthetic.to_dqn_input(d), defer_update=True)
    elif kind in ("dqn", "crr", "dqn_fused", "dqn_x3", "qr_fused"):
        tr.train_step_native(synthetic.to_dqn_input(d))
    elif kind == "td3":
        tr.train_step_native(synthetic.to_policy_input(d), noise[0])
    else:
        tr.train_step_native(synthetic.to_policy_input(d), noise[0], noise[1])


def _worker(rank, world, port, kind, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_backend

    emu_backend.install()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = 64
    full = _batches(kind, B)
    per = B // world  # equal shards: the mean over the batch is the mean of the ranks' means
    half = {k: v[rank * per : (rank + 1) * per].contiguous() for k, v in full.items()}
    g = torch.Generator().manual_seed(9)
    noise = (torch.randn(B, 2, generator=g), torch.randn(B, 2, generator=g))
    my_noise = tuple(n[rank * per : (rank + 1) * per].contiguous() for n in noise)
    tr = _build(kind).enable_data_parallel()
    fused = kind in ("dqn_fused", "sac_fused", "dqn_x3", "qr_fused")
    calls = []
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls.append((t.numel(), bool(k.get("async_op", False))))
        return real_all_reduce(t, *a, **k)

    dist.all_reduce = counting_all_reduce
    try:
        for _ in range(3 if fused else 2):  # (the one-launch update starts at the second step)
            _step(kind, tr, half, my_noise)
    finally:
        dist.all_reduce = real_all_reduce
    if kind in ("sac", "sac_fused"):
        # the twin critics' gradient slabs are ONE buffer and one collective, the temperature's gradient rides
        # as one float behind the actor's slab: TWO collectives per step (rounds 4-5: three; the reference's DDP would take
        # one per parameter bucket)
        n_steps = 3 if fused else 2
        s1, s2 = tr._e["q1"]["slab"], tr._e["q2"]["slab"]
        assert tr._dp_bucket_q.numel() == s1.total + s2.total and s1.grad.data_ptr() == tr._dp_bucket_q.data_ptr()
        assert s2.grad.data_ptr() == tr._dp_bucket_q.data_ptr() + 4 * s1.total
        per_step = calls[:len(calls) // n_steps]
        sa = tr._e["actor"]["slab"]
        assert len(calls) == 2 * n_steps and per_step[0] == (s1.total + s2.total, False), calls
        assert per_step[1] == (sa.total + 1, False) and sa.grad.data_ptr() == tr._dp_bucket_actor.data_ptr(), calls
    if kind == "qr_fused":
        from reagent_amd.qr_engine import GroupedQR

        assert isinstance(tr._qs, GroupedQR) and isinstance(tr._fused_plan, dict) and tr._fused_plan["desc"].group_rows[2] == 16
    elif fused:
        from reagent_amd.engine import FusedMLP

        st = tr._qs if kind.startswith("dqn") else tr._e["q1"]["stack"]
        assert isinstance(st, FusedMLP) and tr._fused_plan not in (None, False) and st.x3 == (kind == "dqn_x3")
    if kind == "dqn_deferred":
        assert tr._update_pending  # the last update is still waiting for its all-reduce
        tr.apply_pending_update()
    torch.save([p.detach().clone() for p in tr.parameters()], os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def _loop_worker(rank, world, port, out_dir, kind="dqn"):
    """bench.py's data-parallel flow on two ranks: own replay shard per rank, OfflineDqnLoop.step (one-launch
    sampler, deferred update under the asynchronous all-reduce), flush, barrier, max-reduce of a timing"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_backend

    emu_backend.install()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from reagent_amd import synthetic
    from reagent_amd.core.parameters import NormalizationParameters as NP
    from reagent_amd.preprocessing import Preprocessor
    from reagent_amd.replay_memory import ReplayBuffer
    from reagent_amd.runtime import OfflineDqnLoop

    S, A, C, B = 12, 4, 512, 64
    tr = _build(kind).enable_data_parallel()
    cols = synthetic.replay_contents(C, S, A, seed=100 + rank)  # this rank's shard
    rb = ReplayBuffer(replay_capacity=C, batch_size=B, device="cpu")
    rb.load_columns(cols, mark_all_valid=True)
    pre = Preprocessor({i: NP(feature_type="CONTINUOUS", mean=0.1 * i, stddev=1.0 + 0.1 * i) for i in range(S)}, device="cpu")
    loop = OfflineDqnLoop(rb, tr, B, pre)
    torch.manual_seed(1000 + rank)  # different index draws per rank
    for _ in range(3):
        loss = loop.step()
    loop.flush()
    dist.barrier()
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world - 1 and torch.isfinite(loss).all() and loop.fused_sampling
    torch.save([p.detach().clone() for p in tr.parameters()], os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["dqn", "dqn_fused", "dqn_x3"])
def test_offline_loop_two_ranks_stay_in_lockstep(tmp_path, emu_lib, kind):
    """the loop bench.py --gpus N runs (eager, asynchronous all-reduce, deferred one-launch update) on two ranks with
    different shards and index draws, per-layer fp32 engine and both fused engines: replicas stay bit-identical"""
    port = free_port()
    mp.spawn(_loop_worker, args=(2, port, str(tmp_path), kind), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    for a, b in zip(r0, r1):
        assert torch.equal(a, b) and torch.isfinite(a).all()


@pytest.mark.parametrize("kind", ["dqn", "dqn_deferred", "dqn_generator", "sac", "td3", "crr"])
def test_two_ranks_equal_single_process_on_concatenated_batch(tmp_path, emu_lib, kind):
    port = free_port()
    mp.spawn(_worker, args=(2, port, kind, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)  # replicas stay bit-identical
    # single process, whole batch
    B = 64
    full = _batches(kind, B)
    g = torch.Generator().manual_seed(9)
    noise = (torch.randn(B, 2, generator=g), torch.randn(B, 2, generator=g))
    tr = _build(kind)
    for _ in range(2):
        _step(kind, tr, full, noise)
    tr.apply_pending_update() if kind == "dqn_deferred" else None
    for a, p in zip(r0, tr.parameters()):
        assert (a.double() - p.detach().double()).abs().max() <= 2e-6, kind


@pytest.mark.parametrize("world,kind", [(4, "dqn"), (4, "dqn_deferred"), (4, "sac"), (8, "dqn"), (8, "dqn_deferred")])
def test_four_and_eight_ranks_equal_single_process_on_concatenated_batch(tmp_path, emu_lib, world, kind):
    """BASELINE C5's equivalence at reduced size: `world` ranks on disjoint equal shards of one batch, gradient
    slab all-reduced (sum) with 1/world folded into Adam — every replica bit-identical to every other, and equal to ONE
    process stepping on the concatenated batch.  world = 8 is the node the scaling curve is measured on."""
    port = free_port()
    mp.spawn(_worker, args=(world, port, kind, str(tmp_path)), nprocs=world, join=True)
    reps = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    for r in range(1, world):
        for a, b in zip(reps[0], reps[r]):
            assert torch.equal(a, b), (r, kind)
    B = 64
    full = _batches(kind, B)
    g = torch.Generator().manual_seed(9)
    noise = (torch.randn(B, 2, generator=g), torch.randn(B, 2, generator=g))
    tr = _build(kind)
    for _ in range(2):
        _step(kind, tr, full, noise)
    tr.apply_pending_update() if kind == "dqn_deferred" else None
    for a, p in zip(reps[0], tr.parameters()):
        assert (a.double() - p.detach().double()).abs().max() <= 2e-6, (world, kind)


@pytest.mark.gpu
def test_rccl_async_reduce_single_rank_group():
    """The RCCL code path on the real GPU (a 1-rank "nccl" group is all a 1-GPU box allows): the
    deferred update with an asynchronous all-reduce gives the same parameters as the plain step."""
    import reagent_amd._lib as L
    from reagent_amd import synthetic

    L.lib()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        d = {k: v.to(dev) for k, v in synthetic.dqn_batch(256, 12, 4, seed=5, p_impossible=0.2).items()}
        plain = _build("dqn").to(dev)
        dp = _build("dqn").to(dev).enable_data_parallel()
        for _ in range(3):
            plain.train_step_native(synthetic.to_dqn_input(d))
            dp.train_step_native(synthetic.to_dqn_input(d), defer_update=True)
        assert dp._update_pending
        dp.apply_pending_update()
        torch.cuda.synchronize()
        for a, b in zip(plain.parameters(), dp.parameters()):
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["dqn_fused", "sac_fused", "dqn_x3", "qr_fused"])
def test_two_ranks_on_the_fused_engine_stay_bit_identical(tmp_path, emu_lib, kind):
    """bf16 fused stacks under data parallelism: the one-launch Adam + soft update + re-staging (rg_mlp_update_fused,
    engine.FusedUpdate) with the 1/world factor folded in; replicas bit-identical, and close to the single-process
    run on the concatenated batch (bf16 gradients summed in another order: a few weights move by up to lr per step)"""
    port = free_port()
    mp.spawn(_worker, args=(2, port, kind, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    for a, b in zip(r0, r1):
        assert torch.equal(a, b) and torch.isfinite(a).all()
    B = 64
    full = _batches(kind, B)
    g = torch.Generator().manual_seed(9)
    noise = (torch.randn(B, 2, generator=g), torch.randn(B, 2, generator=g))
    tr = _build(kind)
    for _ in range(3):
        _step(kind, tr, full, noise)
    for a, p in zip(r0, tr.parameters()):
        assert (a.double() - p.detach().double()).abs().max() <= 3 * 2 * 0.001 + 1e-6, kind
