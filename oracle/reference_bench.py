"""TEST INFRASTRUCTURE (oracle).  The UNMODIFIED reference's CPU path, timed — bench.py's `cpu_baseline` leg with
`kind: "reference"`.

What runs per step, all of it the reference's own code imported from /root/reference (build container) or from its byte
code in oracle/_ref (GPU box; `python -m oracle.build_ref`):
    ReplayBuffer.sample_transition_batch        reagent/replay_memory/circular_replay_buffer.py:614-706 (indices drawn
                                                by its own sample_index_batch: nonzero() over the capacity + randint)
    DiscreteDqnInputMaker / PolicyNetworkInputMaker   reagent/gym/preprocessors/trainer_preprocessor.py:100-227
    Preprocessor.forward on state and next_state      reagent/preprocessing/preprocessor.py:115-170
    {DQN,QRDQN,SAC}Trainer.train_step_gen             driven by the Lightning-1.6 loop emulation (reference_harness.PLLoop)
The buffer's columns are the ones the GPU run holds in HBM (the reference's per-transition `add` takes 143 us each —
150 s for 2^20 rows — so after one `add` has made the buffer infer its storage types the columns are written into its
storage arrays directly; the sampling path does not know the difference).  Network weights = the GPU run's initial weights.
"""
import time

import numpy as np
import torch

from . import reference_harness as rh
from . import stubs


def available() -> bool:
    return stubs.runtime_root() is not None


def where() -> str:
    r = stubs.runtime_root()
    return "oracle/_ref (byte code of the unmodified reference, oracle/build_ref.py)" if r == stubs.BUILT_ROOT else str(r)


def _load(net, init):
    params = list(net.parameters())
    assert len(params) == len(init) and all(p.shape == w.shape for p, w in zip(params, init))
    with torch.no_grad():
        for p, w in zip(params, init):
            p.copy_(w)


def build(algo, state_dim, actions, hidden, layers, atoms, capacity, batch, init, cols, norm):
    """(sample, step) closures over the reference objects"""
    stubs.install_gym()
    import reagent.core.types as rlt
    from reagent.core.parameters import NormalizationParameters as NP
    from reagent.gym.preprocessors.trainer_preprocessor import DiscreteDqnInputMaker, PolicyNetworkInputMaker
    from reagent.preprocessing.preprocessor import Preprocessor
    from reagent.replay_memory.circular_replay_buffer import ReplayBuffer

    S, A, H = state_dim, actions, [hidden] * layers
    acts = ["relu"] * layers
    if algo == "sac":
        tr = rh.build_sac(S, A, H, acts, dict(gamma=0.99, target_update_rate=0.001), 1e-3, seed=0)
        _load(tr.actor_network, init[0])
        for net, tgt, w in ((tr.q1_network, tr.q1_network_target, init[1]), (tr.q2_network, tr.q2_network_target, init[2])):
            _load(net, w)
            _load(tgt, w)
    else:
        rl = dict(gamma=0.99, target_update_rate=0.001, maxq_learning=True)
        if algo == "dqn":
            rl["q_network_loss"] = "huber"
        tr = rh.build_dqn(S, A, H, acts, rl, 1e-3, double_q=True, seed=0, num_atoms=atoms)
        _load(tr.q_network, init[0])
        _load(tr.q_network_target, init[0])
    rb = ReplayBuffer(replay_capacity=capacity, batch_size=batch)
    first = {}
    for k, v in cols.items():  # one example per key, typed as the gym flow's `add` calls type them
        x = v[0]
        first[k] = bool(x) if k == "terminal" else (x.numpy() if x.ndim else x.numpy()[()])
    rb.add(**first)
    for k, v in cols.items():  # storage arrays are torch tensors of the inferred dtype (DenseMetadata.create_storage)
        rb._store[k][:] = v.to(rb._store[k].dtype)
    rb.add_count = np.array(capacity)
    rb._is_index_valid[:] = True
    rb._num_valid_indices = capacity
    mean, std = norm
    pre = Preprocessor({i: NP(feature_type="CONTINUOUS", mean=mean[i].item(), stddev=std[i].item()) for i in range(S)},
                       device=torch.device("cpu"))
    pre.eval()
    if algo == "sac":
        from reagent.core.parameters import CONTINUOUS_TRAINING_ACTION_RANGE as R

        maker = PolicyNetworkInputMaker(np.full(A, R[0], dtype=np.float32), np.full(A, R[1], dtype=np.float32))
    else:
        maker = DiscreteDqnInputMaker(A)
    presence = torch.ones(batch, S, dtype=torch.uint8)
    loop = rh.PLLoop(tr)

    def sample():
        inp = maker(rb.sample_transition_batch(batch_size=batch))
        with torch.no_grad():
            inp.state = rlt.FeatureData(pre(inp.state.float_features, presence))
            inp.next_state = rlt.FeatureData(pre(inp.next_state.float_features, presence))
        return inp

    return sample, loop.step


def run(algo, state_dim, actions, hidden, layers, atoms, capacity, batch, init, cols, norm, steps=2, budget_s=25.0):
    """Times the loop; threads: torch's default on this box and, when that is more than 32, 32 and 16 as well (a
    65536 x 512 fp32 GEMM does not scale to 128 hyper-threads: oversubscribed intra-op pools were what made round 3's port
    slower on 128 threads than the survey's run on 8) — the fastest setting is the baseline, all are listed."""
    sample, step = build(algo, state_dim, actions, hidden, layers, atoms, capacity, batch, init, cols, norm)
    default = torch.get_num_threads()
    settings = [default] + [t for t in (32, 16) if t < default]
    tried, t_start = [], time.perf_counter()
    try:
        for k, th in enumerate(settings):
            torch.set_num_threads(th)
            step(sample())  # warm-up at this setting (allocator, thread pool)
            ts = tt = 0.0
            n = 0
            for _ in range(steps):
                t0 = time.perf_counter()
                b = sample()
                t1 = time.perf_counter()
                step(b)
                t2 = time.perf_counter()
                ts, tt, n = ts + (t1 - t0), tt + (t2 - t1), n + 1
                if time.perf_counter() - t_start > budget_s:
                    break
            tried.append(dict(threads=th, ms_per_step=(ts + tt) / n * 1e3, sample_ms=ts / n * 1e3, train_ms=tt / n * 1e3, steps=n))
            if time.perf_counter() - t_start > budget_s:
                break
    finally:
        torch.set_num_threads(default)
    best = min(tried, key=lambda r: r["ms_per_step"])
    return best, tried
