"""Graph-safe stepping (runtime._GraphedLoop, optimizer.AdamSchedule).

CPU (SIMT interpreter): the `_sched` entry points — lr and Adam's bias corrections read from a device-resident
table, the step counted on the device — give the same bits as the scalar-argument entry points, for FusedAdam,
the fused DQN update and the fp64 temperature optimizer; host step counters catch up on materialize / state_dict.
GPU: a step captured ONCE in a HIP graph and replayed equals the eager loop bit for bit (DQN loop at the fused
bf16 shape with static indices and with the device index draw; SAC loop; the three-graph data-parallel form on a
1-rank RCCL group).
"""
import os

import pytest
import torch

import reagent_amd._lib as L
from reagent_amd import synthetic
from reagent_amd.core.parameters import EvaluationParameters, NormalizationParameters, RLParameters
from reagent_amd.models import (FullyConnectedCritic, FullyConnectedDQN, GaussianFullyConnectedActor,
                                set_default_precision)
from reagent_amd.optimizer import AdamSchedule, FusedAdam, Optimizer__Union


def test_schedule_table_is_the_scalar_path_arithmetic():
    import math

    s = AdamSchedule(1e-3, (0.9, 0.999), 7, "cpu")
    assert s.buf[0].item() == 7.0 and s.buf[1].item() == 1e-3 and int(s.buf[2].item()) == s.n
    for t in (1, 2, 10, 349, 5000, s.n):
        assert s.buf[4 + 2 * (t - 1)].item() == 1.0 - 0.9**t
        assert s.buf[5 + 2 * (t - 1)].item() == math.sqrt(1.0 - 0.999**t)
    assert s.buf[-2].item() == 1.0 and s.buf[-1].item() == 1.0  # converged: later steps reuse the last entry
    with pytest.raises(NotImplementedError):
        AdamSchedule(1e-3, (0.9, 1.0), 0, "cpu")


def test_fused_adam_device_schedule_equals_scalar_steps(emu_lib):
    torch.manual_seed(0)

    def make():
        torch.manual_seed(1)
        ps = [torch.nn.Parameter(torch.randn(33, 17)), torch.nn.Parameter(torch.randn(33))]
        return ps, FusedAdam(ps, lr=3e-3, betas=(0.8, 0.99), weight_decay=0.01)

    (pa, a), (pb, b) = make(), make()
    g = torch.Generator().manual_seed(2)
    for k in range(12):
        if k == 3:
            b.enable_device_schedule()  # switch mid-run: the schedule starts from the host step
        if k == 8:
            for o in (a, b):
                o.param_groups[0]["lr"] = 1e-3  # a scheduler changed lr: the device copy follows
        grads = [torch.randn(p.shape, generator=g) for p in pa]
        for ps, o in ((pa, a), (pb, b)):
            for p, gr in zip(ps, grads):
                p.grad = gr.clone()
            o.step()
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)
    sd_a, sd_b = a.state_dict(), b.state_dict()
    for k in sd_a["state"]:
        assert float(sd_a["state"][k]["step"]) == float(sd_b["state"][k]["step"]) == 12.0
        assert torch.equal(sd_a["state"][k]["exp_avg_sq"], sd_b["state"][k]["exp_avg_sq"])
    # resume from the saved state keeps the schedule mode and the step
    (pc, c) = make()
    c.enable_device_schedule()
    with torch.no_grad():
        for x, y in zip(pc, pb):
            x.copy_(y)
    c.load_state_dict(sd_b)
    assert c.schedule_for(0) is not None and c.schedule_for(0).buf[0].item() == 12.0


def _dqn(dev, S=12, A=4, hidden=(32, 16)):
    from reagent_amd.training import DQNTrainer

    torch.manual_seed(0)
    q = FullyConnectedDQN(S, A, list(hidden), ["relu"] * len(hidden)).to(dev)
    return DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(A)],
                      rl=RLParameters(gamma=0.9, target_update_rate=0.1, q_network_loss="huber"),
                      optimizer=Optimizer__Union.default(lr=0.01),
                      evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)


def test_dqn_and_sac_steps_in_graph_mode_equal_plain_steps(backend):
    """enable_graph_mode changes where the coefficients come from, not the result (eager steps on both sides)"""
    from reagent_amd.training import SACTrainer
    from reagent_amd.training.dqn_trainer import enable_graph_mode, note_graph_replays

    dev = backend.device
    a, b = _dqn(dev), _dqn(dev)
    enable_graph_mode(b)
    for s in range(4):
        batch = synthetic.dqn_batch(64, 12, 4, seed=20 + s, p_impossible=0.2)
        la = a.train_step_native(synthetic.to_dqn_input(batch, dev))
        lb = b.train_step_native(synthetic.to_dqn_input(batch, dev))
        assert torch.equal(la, lb)
    for x, y in zip(a.parameters(), b.parameters()):
        assert torch.equal(x, y)
    ob = b.native_optimizers()[0]
    ob.materialize_steps()
    assert {float(st["step"]) for st in ob.state.values()} == {4.0}
    note_graph_replays(b, 3)
    assert b.all_batches_processed == a.all_batches_processed + 3
    assert {float(st["step"]) for st in ob.state_dict()["state"].values()} == {7.0}

    def sac():
        torch.manual_seed(3)
        adam = lambda: Optimizer__Union.default(lr=0.01)  # noqa: E731
        return SACTrainer(GaussianFullyConnectedActor(6, 2, [16, 16], ["relu", "relu"]).to(dev),
                          FullyConnectedCritic(6, 2, [16, 16], ["relu", "relu"]).to(dev),
                          FullyConnectedCritic(6, 2, [16, 16], ["relu", "relu"]).to(dev),
                          rl=RLParameters(gamma=0.9, target_update_rate=0.1), q_network_optimizer=adam(),
                          actor_network_optimizer=adam(), alpha_optimizer=adam()).to(dev)

    a, b = sac(), sac()
    enable_graph_mode(b)
    g = torch.Generator().manual_seed(9)
    for s in range(3):
        pb = synthetic.to_policy_input(synthetic.policy_batch(32, 6, 2, seed=40 + s), dev)
        n1, n2 = torch.randn(32, 2, generator=g).to(dev), torch.randn(32, 2, generator=g).to(dev)
        oa, ob_ = a.train_step_native(pb, n1, n2), b.train_step_native(pb, n1, n2)
        for k in oa:
            assert torch.equal(oa[k], ob_[k]), k
    for x, y in zip(a.parameters(), b.parameters()):
        assert torch.equal(x, y)
    assert torch.equal(a.log_alpha, b.log_alpha)


# ---- GPU: replay == eager -------------------------------------------------------------------------
def _c2_loop(dev, capacity=8192, batch=1024, precision=L.PREC_BF16):
    from reagent_amd.preprocessing import Preprocessor
    from reagent_amd.replay_memory import ReplayBuffer
    from reagent_amd.runtime import OfflineDqnLoop
    from reagent_amd.training import DQNTrainer

    S, A, H = 128, 16, [512, 512, 512]
    set_default_precision(precision)
    try:
        q = FullyConnectedDQN(S, A, H, ["relu"] * 3)
    finally:
        set_default_precision(L.PREC_F32)
    with torch.no_grad():
        for p, w in zip(q.parameters(), synthetic.fc_init([S] + H + [A], ["relu"] * 3 + ["linear"], seed=40)):
            p.copy_(w)
    q = q.to(dev)
    tr = DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(A)],
                    rl=RLParameters(gamma=0.99, target_update_rate=0.01, q_network_loss="huber"),
                    optimizer=Optimizer__Union.default(lr=1e-3),
                    evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)
    cols = synthetic.replay_contents(capacity, S, A, seed=100)
    rb = ReplayBuffer(replay_capacity=capacity, batch_size=batch, device=dev)
    rb.load_columns({k: v.to(dev) for k, v in cols.items()}, mark_all_valid=True)
    mean, std = synthetic.normalization_table(S, 7)
    pre = Preprocessor({i: NormalizationParameters(feature_type="CONTINUOUS", mean=mean[i].item(), stddev=std[i].item())
                        for i in range(S)}, device=dev)
    return OfflineDqnLoop(rb, tr, batch, pre, state_dtype=torch.bfloat16 if precision == L.PREC_BF16 else torch.float32), tr


@pytest.mark.gpu
@pytest.mark.parametrize("draw", ["static_indices", "device_rng", "device_pool", "device_pool_2_steps_per_graph"])
def test_dqn_loop_graph_replay_equals_eager(draw):
    dev = torch.device("cuda")
    L.lib()
    N, W = 6, 2
    idx = [torch.randint(8192, (1024,), generator=torch.Generator().manual_seed(50 + k)).to(dev) for k in range(N)]
    losses = {}
    params = {}
    for mode in ("eager", "graph"):
        loop, tr = _c2_loop(dev)
        per = 2 if draw == "device_pool_2_steps_per_graph" else 1  # consecutive steps recorded per graph (round 4)
        if not draw.startswith("device_pool"):
            loop.index_pool_steps = 1  # a draw per step, inside the captured graph too
        # device_pool (the default): eager steps and replays both take their indices from the loop's pool (32 steps per
        # torch.randint); a replay copies its row into the buffer the captured sampler reads
        torch.cuda.manual_seed(77)
        out = []
        if mode == "eager":
            for _ in range(W):
                loop.step()
            for k in range(N):
                out.append(loop.step(idx[k] if draw == "static_indices" else None).clone())
        else:
            step = loop.capture(warmup=W, static_indices=draw == "static_indices", steps_per_replay=per)
            assert loop.replay_steps == per
            for k in range(N // per):
                out.append(step(idx[k] if draw == "static_indices" else None).clone())
        loop.flush()
        torch.cuda.synchronize()
        losses[mode] = torch.stack(out).cpu()
        params[mode] = [p.detach().cpu().clone() for p in list(tr.q_network.parameters()) + list(tr.q_network_target.parameters())]
        adam = tr.native_optimizers()[0]
        assert {float(st["step"]) for st in adam.state_dict()["state"].values()} == {float(N + W)}
        assert tr.all_batches_processed == N + W
    if per > 1:  # a replay returns its LAST step's loss
        losses["eager"] = losses["eager"][per - 1::per]
    assert torch.equal(losses["eager"], losses["graph"]), (losses["eager"], losses["graph"])
    for a, b in zip(params["eager"], params["graph"]):
        assert torch.equal(a, b)
    # and an eager step after the replays continues from the replayed state (weights re-staged, step counted)
    l2 = loop.step(idx[0])
    loop.flush()
    assert torch.isfinite(l2).all() and tr.all_batches_processed == N + W + 1


def _qr_loop(dev, capacity=8192, batch=1024, precision=L.PREC_BF16, atoms=200):
    """C3's shapes on the grouped engine (two streams, the per-action mean layer after the update)"""
    from reagent_amd.preprocessing import Preprocessor
    from reagent_amd.replay_memory import ReplayBuffer
    from reagent_amd.runtime import OfflineDqnLoop
    from reagent_amd.training import QRDQNTrainer

    S, A, H = 128, 16, [512, 512, 512]
    set_default_precision(precision)
    try:
        q = FullyConnectedDQN(S, A, H, ["relu"] * 3, num_atoms=atoms)
    finally:
        set_default_precision(L.PREC_F32)
    with torch.no_grad():
        for p, w in zip(q.parameters(), synthetic.fc_init([S] + H + [A * atoms], ["relu"] * 3 + ["linear"], seed=41)):
            p.copy_(w)
    q = q.to(dev)
    tr = QRDQNTrainer(q, q.get_target_network(), num_atoms=atoms, actions=[str(i) for i in range(A)],
                      rl=RLParameters(gamma=0.99, target_update_rate=0.01), double_q_learning=True,
                      optimizer=Optimizer__Union.default(lr=1e-3),
                      evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)
    cols = synthetic.replay_contents(capacity, S, A, seed=100)
    rb = ReplayBuffer(replay_capacity=capacity, batch_size=batch, device=dev)
    rb.load_columns({k: v.to(dev) for k, v in cols.items()}, mark_all_valid=True)
    mean, std = synthetic.normalization_table(S, 7)
    pre = Preprocessor({i: NormalizationParameters(feature_type="CONTINUOUS", mean=mean[i].item(), stddev=std[i].item())
                        for i in range(S)}, device=dev)
    return OfflineDqnLoop(rb, tr, batch, pre, state_dtype=torch.bfloat16 if precision == L.PREC_BF16 else torch.float32), tr


@pytest.mark.gpu
@pytest.mark.parametrize("draw", ["device_rng", "pools"])
def test_sac_loop_graph_replay_equals_eager(draw):
    import numpy as np

    from reagent_amd.core.parameters import CONTINUOUS_TRAINING_ACTION_RANGE as R
    from reagent_amd.preprocessing import PolicyNetworkInputMaker
    from reagent_amd.replay_memory import ReplayBuffer
    from reagent_amd.runtime import OfflinePolicyLoop
    from reagent_amd.training import SACTrainer

    dev = torch.device("cuda")
    S, A, B, C = 24, 6, 512, 4096

    def build():
        torch.manual_seed(5)
        adam = lambda: Optimizer__Union.default(lr=1e-3)  # noqa: E731
        tr = SACTrainer(GaussianFullyConnectedActor(S, A, [64, 64], ["relu", "relu"]).to(dev),
                        FullyConnectedCritic(S, A, [64, 64], ["relu", "relu"]).to(dev),
                        FullyConnectedCritic(S, A, [64, 64], ["relu", "relu"]).to(dev),
                        rl=RLParameters(gamma=0.99, target_update_rate=0.05), q_network_optimizer=adam(),
                        actor_network_optimizer=adam(), alpha_optimizer=adam()).to(dev)
        cols = synthetic.replay_contents(C, S, A, seed=3)
        cols["action"] = torch.rand(C, A, generator=torch.Generator().manual_seed(4)) * 1.8 - 0.9
        del cols["possible_actions_mask"]
        rb = ReplayBuffer(replay_capacity=C, batch_size=B, device=dev)
        rb.load_columns({k: v.to(dev) for k, v in cols.items()}, mark_all_valid=True)
        maker = PolicyNetworkInputMaker(np.full(A, R[0], dtype=np.float32), np.full(A, R[1], dtype=np.float32))
        return OfflinePolicyLoop(rb, tr, B, maker), tr

    res = {}
    for mode in ("eager", "graph"):
        loop, tr = build()
        if draw == "device_rng":  # a draw per step, inside the captured graph too
            loop.index_pool_steps = loop.noise_pool_steps = 1
        # pools (the default): indices and the actor's noise of several steps per draw; a replay copies its rows into the
        # buffers the captured step reads
        torch.cuda.manual_seed(11)
        out = []
        if mode == "eager":
            for _ in range(2):
                loop.step()
            for _ in range(4):
                out.append(torch.cat([v.float().reshape(1) for v in loop.step().values()]).clone())
        else:
            step = loop.capture(warmup=2)
            for _ in range(4):
                out.append(torch.cat([v.float().reshape(1) for v in step().values()]).clone())
        loop.flush()
        torch.cuda.synchronize()
        res[mode] = (torch.stack(out).cpu(), [p.detach().cpu().clone() for p in tr.parameters()])
    assert torch.equal(res["eager"][0], res["graph"][0]), (res["eager"][0], res["graph"][0])
    for a, b in zip(res["eager"][1], res["graph"][1]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_data_parallel_three_graph_replay_on_a_one_rank_rccl_group():
    import torch.distributed as dist

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29300 + os.getpid() % 200))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        N = 5
        idx = [torch.randint(8192, (1024,), generator=torch.Generator().manual_seed(60 + k)).to(dev) for k in range(N)]
        res = {}
        for mode in ("eager", "graph"):
            loop, tr = _c2_loop(dev)
            tr.enable_data_parallel()
            torch.cuda.manual_seed(78)
            out = []
            if mode == "eager":
                for _ in range(2):
                    loop.step()
                step = loop.step
            else:
                step = loop.capture(warmup=2, static_indices=True)
                assert loop._graph["dp"] and len(loop._graph["graphs"]) == 3
            for k in range(N):
                out.append(step(idx[k]).clone())
            loop.flush()
            torch.cuda.synchronize()
            res[mode] = (torch.stack(out).cpu(), [p.detach().cpu().clone() for p in tr.q_network.parameters()])
            assert tr.all_batches_processed == N + 2
        assert torch.equal(res["eager"][0], res["graph"][0])
        for a, b in zip(res["eager"][1], res["graph"][1]):
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()


def test_leaving_graph_mode_keeps_the_step_count(emu_lib):
    from reagent_amd.training.dqn_trainer import disable_graph_mode, enable_graph_mode

    a, b = _dqn("cpu"), _dqn("cpu")
    enable_graph_mode(b)
    for s in range(5):
        if s == 3:
            disable_graph_mode(b)
            assert b.native_optimizers()[0].schedule_for(0) is None
        batch = synthetic.dqn_batch(64, 12, 4, seed=20 + s, p_impossible=0.2)
        a.train_step_native(synthetic.to_dqn_input(batch, "cpu"))
        b.train_step_native(synthetic.to_dqn_input(batch, "cpu"))
    for x, y in zip(a.parameters(), b.parameters()):
        assert torch.equal(x, y)


def test_replayed_step_protocol_on_the_interpreter(emu_lib):
    """What a captured step with a device-side index cursor consists of, launch for launch, issued by hand on the SIMT
    interpreter (graphs need the GPU; the protocol does not): rg_replay_dqn_batch_pooled reads row cursor[0] of the index pool
    and counts the step in the Adam schedule, rg_mlp_update_fused_sched(sched_pre_ticked, post_tick) takes the coefficients of
    the counted step and advances the cursor — and leaves the bits of eager steps on the same pool rows (sampler on explicit
    indices, update with the count taken afterwards by rg_sched_tick): losses, weights, targets, Adam state, the cursor."""
    from reagent_amd import ops
    from reagent_amd.preprocessing import Preprocessor
    from reagent_amd.replay_memory import ReplayBuffer
    from reagent_amd.runtime import OfflineDqnLoop
    from reagent_amd.training import DQNTrainer
    from reagent_amd.training.dqn_trainer import enable_graph_mode

    S, A, C, B, P = 24, 4, 512, 64, 3

    def make():
        set_default_precision(L.PREC_BF16)
        try:
            torch.manual_seed(3)
            q = FullyConnectedDQN(S, A, [256, 256], ["relu", "relu"])
        finally:
            set_default_precision(L.PREC_F32)
        tr = DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(A)],
                        rl=RLParameters(gamma=0.9, target_update_rate=0.05, q_network_loss="huber"),
                        optimizer=Optimizer__Union.default(lr=0.003), evaluation=EvaluationParameters(calc_cpe_in_training=False))
        rb = ReplayBuffer(replay_capacity=C, batch_size=B, device="cpu")
        rb.load_columns(synthetic.replay_contents(C, S, A, seed=9), mark_all_valid=True)
        mean, std = synthetic.normalization_table(S, 7)
        pre = Preprocessor({i: NormalizationParameters(feature_type="CONTINUOUS", mean=mean[i].item(), stddev=std[i].item())
                            for i in range(S)}, device="cpu")
        loop = OfflineDqnLoop(rb, tr, B, pre, state_dtype=torch.bfloat16)
        loop.index_pool_steps = P
        enable_graph_mode(tr)
        return loop, tr

    eager, te = make()
    hand, th = make()
    torch.manual_seed(77)
    for _ in range(2):  # the warm-up of capture(): the one-launch update shows itself from the second step on
        eager.step()
    eager.flush()
    torch.manual_seed(77)
    for _ in range(2):
        hand.step()
    hand.flush()
    assert torch.equal(eager._pool, hand._pool) and eager._pool_pos == hand._pool_pos == 2
    rng = torch.get_rng_state()  # both continuations refill their pools from the same position of the random stream
    tick = hand._cursor_protocol(torch.device("cpu"))
    assert tick is not None and tick["mod"] == P
    cursor, sched = tick["cursor"], tick["sched"]
    losses_e, losses_h = [], []
    for k in range(5):  # crosses the pool boundary twice (P = 3)
        losses_e.append(eager.step().clone())
        eager.flush()
    torch.set_rng_state(rng)
    for k in range(5):
        # --- one "replay" by hand
        hand._ensure_pool(torch.device("cpu"))
        if k == 0:
            cursor.fill_(hand._pool_pos)
        assert int(cursor.item()) == hand._pool_pos
        before = float(sched[0])
        th._graph_tick = tick
        try:
            losses_h.append(hand._eager_step(ops.PooledIndices(hand._pool, cursor, sched)).clone())
        finally:
            th._graph_tick = None
        assert tick["used"] and float(sched[0]) == before + 1.0
        hand._pool_pos += 1
        assert int(cursor.item()) == hand._pool_pos % P
    # (outside a capture the host counts these steps itself; after real replays runtime.flush() does — note_graph_replays)
    assert all(torch.equal(a, b) for a, b in zip(losses_e, losses_h))
    for a, b in zip(list(te.q_network.parameters()) + list(te.q_network_target.parameters()),
                    list(th.q_network.parameters()) + list(th.q_network_target.parameters())):
        assert torch.equal(a, b)
    oe, oh = te.native_optimizers()[0], th.native_optimizers()[0]
    oe.materialize_steps()
    oh.materialize_steps()
    for pa, pb in zip(te.q_network.parameters(), th.q_network.parameters()):
        assert torch.equal(oe.state[pa]["exp_avg"], oh.state[pb]["exp_avg"]) and torch.equal(oe.state[pa]["exp_avg_sq"], oh.state[pb]["exp_avg_sq"])
        assert float(oe.state[pa]["step"]) == float(oh.state[pb]["step"]) == 7.0


# ---- the data-parallel three-graph replay ORDER on two real ranks (gloo), graphs replaced by re-executing stand-ins -------
class _ReplayedLaunches:
    """Stands for a HIP graph on a box without one: `replay()` re-issues the recorded launches (calls the recorded function
    again) and leaves the results in the buffers of the first run — what a replayed graph does with its fixed addresses."""

    def __init__(self, fn, trainer_ref, run_now):
        # A capture RECORDS: nothing executes.  Only the sampler region is run here (it has no state, and the batch buffers it
        # returns are what the third region reads); the update and forward/backward regions run at replay() only, their
        # result landing in a placeholder the caller already holds.
        self.fn, self.tr = fn, trainer_ref
        self.out = fn() if run_now else torch.empty(0)

    def _host_counters(self, restore=None):
        """a real replay runs no python: what the re-executed function counted on the HOST (steps taken, Adam's pending device
        steps) is put back — runtime.flush() counts the replays (note_graph_replays), as it does for real graphs"""
        tr = self.tr["tr"]
        scheds = [s for o in tr.native_optimizers() for s in getattr(o, "_scheds", {}).values()]
        if restore is None:
            return tr.all_batches_processed, [s.pending for s in scheds]
        tr.all_batches_processed = restore[0]
        for s, v in zip(scheds, restore[1]):
            s.pending = v

    @staticmethod
    def _copy_into(dst, src):
        import dataclasses

        if isinstance(dst, torch.Tensor):
            if dst.shape != src.shape:
                dst.resize_(src.shape)
            dst.copy_(src)
        elif dataclasses.is_dataclass(dst):
            for f in dataclasses.fields(dst):
                a, b = getattr(dst, f.name), getattr(src, f.name)
                if a is not None and b is not None:
                    _ReplayedLaunches._copy_into(a, b)
        elif isinstance(dst, (list, tuple)):
            for a, b in zip(dst, src):
                _ReplayedLaunches._copy_into(a, b)

    def replay(self):
        saved = self._host_counters()
        res = self.fn()
        if res is not None:
            self._copy_into(self.out, res)
        self._host_counters(saved)


def _dp_graph_worker(rank, world, port, out_dir, mode):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import emu_backend

    emu_backend.install()
    import torch.distributed as dist

    from reagent_amd.runtime import _GraphedLoop

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _GraphedLoop._graphs_available = staticmethod(lambda dev: True)
    _GraphedLoop._new_graph_pool = staticmethod(lambda: None)

    ref = {}

    def record(fn, pool=None):
        g = _ReplayedLaunches(fn, ref, run_now=not ref.setdefault("recorded", 0))
        ref["recorded"] += 1
        return g, g.out

    _GraphedLoop._record = staticmethod(record)
    loop, tr = _small_dp_loop(rank)
    ref["tr"] = tr
    tr.enable_data_parallel()
    N = 5
    idx = [torch.randint(512, (64,), generator=torch.Generator().manual_seed(60 + 10 * rank + k)) for k in range(N)]
    losses = []
    if mode == "eager":
        for _ in range(2):
            loop.step()
        step = loop.step
    else:
        step = loop.capture(warmup=2, static_indices=True)
        assert loop._graph["dp"] and len(loop._graph["graphs"]) == 3
    for k in range(N):
        losses.append(step(idx[k]).clone())
    loop.flush()
    assert tr.all_batches_processed == N + 2
    torch.save(dict(losses=torch.stack(losses), params=[p.detach().clone() for p in tr.parameters()]),
               os.path.join(out_dir, f"{mode}_rank{rank}.pt"))
    dist.destroy_process_group()


def _small_dp_loop(rank):
    from reagent_amd.preprocessing import Preprocessor
    from reagent_amd.replay_memory import ReplayBuffer
    from reagent_amd.runtime import OfflineDqnLoop
    from reagent_amd.training import DQNTrainer

    S, A, C, B = 24, 4, 512, 64
    set_default_precision(L.PREC_BF16)
    try:
        torch.manual_seed(3)  # identical initial weights on every rank
        q = FullyConnectedDQN(S, A, [256, 256], ["relu", "relu"])
    finally:
        set_default_precision(L.PREC_F32)
    tr = DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(A)],
                    rl=RLParameters(gamma=0.9, target_update_rate=0.05, q_network_loss="huber"),
                    optimizer=Optimizer__Union.default(lr=0.003), evaluation=EvaluationParameters(calc_cpe_in_training=False))
    rb = ReplayBuffer(replay_capacity=C, batch_size=B, device="cpu")
    rb.load_columns(synthetic.replay_contents(C, S, A, seed=9 + rank), mark_all_valid=True)  # this rank's shard
    mean, std = synthetic.normalization_table(S, 7)
    pre = Preprocessor({i: NormalizationParameters(feature_type="CONTINUOUS", mean=mean[i].item(), stddev=std[i].item())
                        for i in range(S)}, device="cpu")
    return OfflineDqnLoop(rb, tr, B, pre, state_dtype=torch.bfloat16), tr


def test_data_parallel_three_graph_replay_order_on_two_gloo_ranks(tmp_path, emu_lib):
    """the three-graph data-parallel replay (sample | update | forward + backward, the gradient all-reduce
    launched eagerly between them, the update of step k joined at step k + 1) had only ever run on a ONE-rank RCCL group.
    Here its host-side ORDER runs on two real ranks with a real collective (gloo), the graphs replaced by stand-ins that
    re-issue the recorded launches: both ranks finish (no mismatched collective), stay bit-identical to each other, and
    equal the eager data-parallel loop on the same indices step for step."""
    import torch.multiprocessing as mp

    from conftest import free_port

    for mode in ("eager", "graph"):
        mp.spawn(_dp_graph_worker, args=(2, free_port(), str(tmp_path), mode), nprocs=2, join=True)
    e0, e1 = torch.load(tmp_path / "eager_rank0.pt"), torch.load(tmp_path / "eager_rank1.pt")
    g0, g1 = torch.load(tmp_path / "graph_rank0.pt"), torch.load(tmp_path / "graph_rank1.pt")
    for a, b in zip(g0["params"], g1["params"]):
        assert torch.equal(a, b)  # replicas in lockstep
    for a, b in zip(e0["params"], g0["params"]):
        assert torch.equal(a, b)  # the replayed order == the eager loop
    assert torch.equal(e0["losses"], g0["losses"]) and torch.equal(e1["losses"], g1["losses"])
