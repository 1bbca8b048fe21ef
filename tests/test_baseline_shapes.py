"""The BENCHMARKED code path against the unmodified reference at BASELINE.json's layer shapes
(tests/golden/baseline_c{2,3,4}.npz, made by `python -m oracle.make_golden baseline`):

  C2  replay gather -> input maker -> Preprocessor x2 -> DQN step, net 128-512-512-512-16, B=2048 —
      driven exactly as bench.py drives it (OfflineDqnLoop: rg_replay_dqn_batch -> train_step_native with
      the deferred rg_mlp_update_fused), in fp32 parity mode, bf16x3 accurate-MFMA mode and bf16 throughput mode
  C3  QR-DQN N=200 (3200-wide output layer), B=256
  C4  SAC, actor + twin critics, S=256, A=32, H=3x512, B=1024

Tolerances (BASELINE.json north_star): accurate modes (f32, bf16x3) — Q-values / policy logits within 1e-4,
loss within 1e-4 rel, post-step weights within 2e-5.  bf16 throughput mode: the measured errors are
PRINTED (run with -s) and bounded by stated, looser limits.
One exception, measured rather than assumed (profiles/microbench/diag_c4_actor.py on the MI355X): the C4 ACTOR
gradient is ill-conditioned — it is the difference of the alpha*log_prob term and dQ/da pushed through two
4-layer networks with |Q| ~ 5 — so that torch-CPU fp32 autograd (the reference's own arithmetic) is itself
5e-6 away from an fp64 evaluation of the same step, at a mean |g| of 2e-4 (the HIP fp32 path: 8e-6).  Adam's
first steps move a weight by lr * g / (|g| + 1e-8): for the ~1 % of actor weights with |g| below that noise
floor the SIGN of the update is not determined by the fp32 reference, and two correct implementations differ
by up to 2 * lr there.  The actor's weights are therefore held to: at most 2 % of the sampled elements beyond
2e-5, none beyond 2 * lr per step taken; critics (well-conditioned MSE gradients) stay at 2e-5 everywhere.
Big tensors are compared through the fixture's digest: full biases, every 61st weight, fp64 sums.
"""
import numpy as np
import pytest
import torch

import reagent_amd._lib as L
from golden_util import Golden
from reagent_amd import synthetic
from reagent_amd.core.parameters import EvaluationParameters, NormalizationParameters, RLParameters
from reagent_amd.models import (FullyConnectedCritic, FullyConnectedDQN, GaussianFullyConnectedActor,
                                set_default_precision)
from reagent_amd.optimizer import Optimizer__Union

W_STRIDE = 61  # oracle/make_golden.py


def digest(t):
    a = t.detach().cpu().numpy().reshape(-1)
    return torch.from_numpy(a if a.size <= 8192 else a[::W_STRIDE].copy())


def digest_err(t, g, key):
    return (digest(t) - g.t(key)).abs().max().item()


def worst(t, g, key, init_key=None):
    """(|error|, reference value, ours, reference initial value) at the worst element of the digest"""
    d = digest(t)
    r = g.t(key)
    i = int((d - r).abs().argmax())
    return dict(err=float((d - r).abs().max()), ref=float(r[i]), got=float(d[i]),
                init=float(g.t(init_key)[i]) if init_key else None)


def frac_beyond(t, g, key, tol=2e-5):
    return float(((digest(t) - g.t(key)).abs() > tol).double().mean())


def grad_err(grads, g, prefix):
    """max over the tensors of max|dg| / max|g_ref| on the digests — the backward pass against what the
    reference's autograd handed its optimizer (well conditioned, unlike post-Adam weights)"""
    worst_rel = 0.0
    for i, gr in enumerate(grads):
        ref = g.t(f"{prefix}{i}")
        worst_rel = max(worst_rel, ((digest(gr) - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item())
    return worst_rel


# Gradient bounds (relative to the tensor's largest entry), ~3x what the MI355X measures.  fp32: rounding only.
# bf16x3: its own rounding is ~1e-5, the bound is set by ReLU masks — a forward error of 1e-5 puts ~30 of the
# 3.1 M hidden pre-activations of a 2048-row batch on the other side of zero, and each flipped unit changes one
# row of dW by that sample's whole contribution (1e-3 of the row's largest entry at these batch sizes).
GRAD_TOL = {"f32": 3e-5, "bf16x3": 3e-3, "bf16": 3e-1}


def check_regenerated(t, g, key):
    """inputs / initial weights are regenerated from seeds on both sides: must be the same bits"""
    assert torch.equal(digest(t), g.t(key)), key
    a = t.detach().cpu().numpy().reshape(-1).astype(np.float64)
    assert np.array_equal(np.array([a.sum(), (a * a).sum()]), g.a(key + "_sums")), key


def load_init(net, dims, acts, seed, g=None, prefix=None):
    init = synthetic.fc_init(dims, acts, seed)
    with torch.no_grad():
        for p, w in zip(net.parameters(), init):
            assert p.shape == w.shape
            p.copy_(w)
    if g is not None:
        for i, p in enumerate(net.parameters()):
            check_regenerated(p, g, f"{prefix}{i}")


MODES = {"f32": L.PREC_F32, "bf16x3": L.PREC_BF16X3, "bf16": L.PREC_BF16}
ACCURATE = ("f32", "bf16x3")


def _with_precision(prec, fn):
    set_default_precision(prec)
    try:
        return fn()
    finally:
        set_default_precision(L.PREC_F32)


# ---- C2: the loop bench.py times -----------------------------------------------------------------
def build_c2(g, device, mode):
    from reagent_amd.preprocessing import Preprocessor
    from reagent_amd.replay_memory import ReplayBuffer
    from reagent_amd.runtime import OfflineDqnLoop
    from reagent_amd.training import DQNTrainer

    c = g.cfg
    S, A, B, C = c["state_dim"], c["num_actions"], c["batch"], c["capacity"]
    q = _with_precision(MODES[mode], lambda: FullyConnectedDQN(S, A, c["sizes"], c["activations"]))
    load_init(q, [S] + c["sizes"] + [A], c["activations"] + ["linear"], c["init_seed"], g, "init_param_")
    q = q.to(device)
    tr = DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(A)], rl=RLParameters(**c["rl"]),
                    double_q_learning=c["double_q"], optimizer=Optimizer__Union.default(lr=c["lr"]),
                    evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(device)
    cols = synthetic.replay_contents(C, S, A, seed=c["replay_seed"], p_terminal=c["p_terminal"])
    rb = ReplayBuffer(replay_capacity=C, batch_size=B, device=device)
    rb.load_columns({k: v.to(device) for k, v in cols.items()})
    assert np.array_equal(rb._is_index_valid.numpy(), g.a("valid_mask"))  # closed-form validity == 8192 adds
    mean, std = synthetic.normalization_table(S, c["norm_seed"])
    pre = Preprocessor({i: NormalizationParameters(feature_type="CONTINUOUS", mean=mean[i].item(), stddev=std[i].item())
                        for i in range(S)}, device=device)
    loop = OfflineDqnLoop(rb, tr, B, pre, state_dtype=torch.bfloat16 if mode == "bf16" else torch.float32)
    return loop, tr


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
def test_c2_loop_matches_reference(backend, mode):
    from reagent_amd.engine import FusedMLP

    if backend.name == "emu" and mode != "bf16":
        pytest.skip("the SIMT interpreter runs the 2048-row C2 step in the fused bf16 mode only (time)")
    g = Golden("baseline_c2")
    loop, tr = build_c2(g, backend.device, mode)
    steps = g.cfg["steps"] if backend.name == "hip" else 1
    for s in range(steps):
        idx = g.t(f"step{s}_indices").to(backend.device)
        batch = loop.make_batch(idx)
        # gather + input maker: exact; normalize-on-gather: the reference Preprocessor's fp32 arithmetic
        for k in ("action", "next_action", "reward", "not_terminal", "possible_next_actions_mask"):
            assert torch.equal(getattr(batch, k).float().cpu().reshape(g.t(f"step{s}_{k}").shape), g.t(f"step{s}_{k}")), k
        tol_state = 1e-5 if mode != "bf16" else 4e-2  # bf16 rows: 2^-8 relative of |x| <= 11.5
        for k in ("state", "next_state"):
            got = getattr(batch, k).float_features.float().cpu()
            assert (got[:16] - g.t(f"step{s}_{k}_rows")).abs().max() <= tol_state
            assert (digest(got) - g.t(f"step{s}_{k}")).abs().max() <= tol_state
        loss = loop.step(idx)
        loop.flush()
        if mode != "f32":
            assert isinstance(tr._qs, FusedMLP) and isinstance(tr._ts, FusedMLP), "the fused kernels must be the ones running"
        ref_loss = g.t(f"step{s}_loss").item()
        dq = (tr.all_action_scores.cpu() - g.t(f"step{s}_q")).abs().max().item()
        dl = abs(loss.item() - ref_loss) / abs(ref_loss)
        dw = max(digest_err(p, g, f"step{s}_param_{i}") for i, p in enumerate(tr.q_network.parameters()))
        dt = max(digest_err(p, g, f"step{s}_target_{i}") for i, p in enumerate(tr.q_network_target.parameters()))
        dg = grad_err(tr._slab.grad_views(), g, f"step{s}_grad_")
        frac = max(frac_beyond(p, g, f"step{s}_param_{i}") for i, p in enumerate(tr.q_network.parameters()))
        print(f"\n[baseline_c2 {mode} {backend.name} step {s}] max|dQ| {dq:.3e}  rel dloss {dl:.3e}  max|dg|/max|g| {dg:.3e}  "
              f"max|dW| {dw:.3e} ({100 * frac:.2f} % of the sampled weights beyond 2e-5)  max|dW_target| {dt:.3e}")
        if mode == "f32":
            assert dq <= 1e-4 and dl <= 1e-4 and dw <= 2e-5 and dt <= 2e-5 and dg <= GRAD_TOL[mode]
        elif mode == "bf16x3":
            # Q, loss and the GRADIENT are fp32-class.  Post-Adam weights: the first steps move every weight by
            # lr * g / (|g| + 1e-8), i.e. by +-lr whatever |g| is, so the ~1e-5 relative gradient error of the
            # split-bf16 products flips the direction of the few weights whose gradient is smaller than that
            # error; those differ by 2 * lr, every other one agrees to 2e-5
            if s == 0:
                assert dq <= 1e-4 and dl <= 1e-4 and dg <= GRAD_TOL[mode]
                assert dw <= 2.0 * g.cfg["lr"] * 1.05 and frac <= 0.01 and dt <= 1e-5
            else:  # starts from weights that differ by 2 * lr at the 0.2 % direction-flipped elements of step 0
                assert dq <= 1e-2 and dl <= 1e-2 and dw <= 2.0 * (s + 1) * g.cfg["lr"] * 1.05 and frac <= 0.1
        else:
            # bf16 inputs, fp32 accumulate: SURVEY.md §7.3 measured 1.9e-2 on this net.  A first Adam step
            # moves every weight by +-lr whatever the gradient's size, so an element whose tiny gradient
            # changes sign under bf16 rounding differs by 2*lr; the target moves by tau times that
            assert dq <= 6e-2 and dl <= 3e-2 and dw <= 2.0 * (s + 1) * g.cfg["lr"] * 1.05 and dt <= 1e-5
    if backend.name == "hip" and mode == "f32":  # (bf16x3's second step starts from the direction-flipped weights)
        adam = tr.native_optimizers()[0]
        for i, p in enumerate(tr.q_network.parameters()):
            ref_m, ref_v = g.t(f"final_exp_avg_{i}"), g.t(f"final_exp_avg_sq_{i}")
            assert (digest(adam.state[p]["exp_avg"]) - ref_m).abs().max() <= 1e-6 + 1e-3 * ref_m.abs().max()
            assert (digest(adam.state[p]["exp_avg_sq"]) - ref_v).abs().max() <= 1e-9 + 1e-3 * ref_v.abs().max()


# ---- C3: QR-DQN, N = 200 -------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
def test_c3_qrdqn_matches_reference(mode):
    from reagent_amd.training import QRDQNTrainer

    assert torch.cuda.is_available()
    dev = "cuda"
    g = Golden("baseline_c3")
    c = g.cfg
    S, A, N, B = c["state_dim"], c["num_actions"], c["num_atoms"], c["batch"]
    q = _with_precision(MODES[mode], lambda: FullyConnectedDQN(S, A, c["sizes"], c["activations"], num_atoms=N))
    load_init(q, [S] + c["sizes"] + [A * N], c["activations"] + ["linear"], c["init_seed"], g, "init_param_")
    q = q.to(dev)
    tr = QRDQNTrainer(q, q.get_target_network(), actions=[str(i) for i in range(A)], rl=RLParameters(**c["rl"]),
                      double_q_learning=c["double_q"], num_atoms=N, optimizer=Optimizer__Union.default(lr=c["lr"]),
                      evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)
    for s in range(c["steps"]):
        b = synthetic.dqn_batch(B, S, A, seed=700 + s, p_impossible=c["p_impossible"])
        check_regenerated(b["state"], g, f"step{s}_batch_state")
        batch = synthetic.to_dqn_input(b, dev)
        z = tr.q_network(batch.state)
        assert z.shape == (B, A, N)
        dz = (z[:8].cpu() - g.t(f"step{s}_quantile_rows")).abs().max().item()
        dq = (z.mean(dim=2).cpu() - g.t(f"step{s}_q_mean")).abs().max().item()
        loss = tr.train_step_native(batch)
        ref = g.t(f"step{s}_loss").item()
        dl = abs(loss.item() - ref) / abs(ref)
        dw = max(digest_err(p, g, f"step{s}_param_{i}") for i, p in enumerate(tr.q_network.parameters()))
        dt = max(digest_err(p, g, f"step{s}_target_{i}") for i, p in enumerate(tr.q_network_target.parameters()))
        dg = grad_err(tr._slab.grad_views(), g, f"step{s}_grad_")
        print(f"\n[baseline_c3 {mode} step {s}] max|dquantile| {dz:.3e} max|dQmean| {dq:.3e} rel dloss {dl:.3e} "
              f"max|dg|/max|g| {dg:.3e} max|dW| {dw:.3e} max|dW_target| {dt:.3e}")
        if mode == "f32":  # the dense [B, A * N] path on exact-fp32 GEMMs
            assert tr._gq_active is None
            assert dz <= 1e-4 and dq <= 1e-4 and dl <= 1e-4 and dw <= 2e-5 and dt <= 2e-5 and dg <= GRAD_TOL["f32"]
        elif mode == "bf16x3":
            # round 4: the GROUPED engine on split-bf16 operands (qr_engine.py; the dense forward above ran exact fp32 — it is
            # the model's own stack).  What the STEP computed: the logged action's quantiles in grouped space and the
            # per-action means a* was chosen from, against the reference's rows; loss; every gradient; weights by the
            # split-bf16 rule (Adam moves a weight by lr whatever |g|: the few whose gradient is below the arithmetic's
            # error change direction).
            gq = tr._gq_active
            assert gq is not None and gq.x3
            rowmap, key = gq.sp_cur.rowmap.cpu().long(), gq.key_cur.cpu().long()
            zq, ref_rows = gq.z.cpu(), g.t(f"step{s}_quantile_rows")
            dzg = max((zq[r, :N] - ref_rows[b_, key[b_]]).abs().max().item()
                      for r, b_ in enumerate(rowmap.tolist()) if 0 <= b_ < ref_rows.shape[0])
            fw = max(frac_beyond(p, g, f"step{s}_param_{i}") for i, p in enumerate(tr.q_network.parameters()))
            print(f"[baseline_c3 bf16x3 grouped step {s}] max|dquantile| (grouped rows) {dzg:.3e} weights beyond 2e-5 {fw:.4f}")
            if s == 0:
                assert dz <= 1e-4 and dq <= 1e-4 and dzg <= 1e-4 and dl <= 1e-4 and dg <= GRAD_TOL["bf16x3"]
                assert fw <= 0.02 and dw <= 2.0 * c["lr"] * 1.05 and dt <= 2.1e-6
            else:  # starts from weights that differ by 2 * lr at the direction-flipped elements of the steps before (C2's rule)
                assert dz <= 1e-2 and dzg <= 1e-2 and dl <= 1e-2 and dw <= 2.0 * (s + 1) * c["lr"] * 1.05 and fw <= 0.1
        else:
            assert dz <= 6e-2 and dl <= 3e-2 and dw <= 2.0 * (s + 1) * c["lr"] * 1.05 and dt <= 1e-5


# ---- C4: SAC --------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
def test_c4_sac_matches_reference(mode):
    from reagent_amd.training import SACTrainer

    assert torch.cuda.is_available()
    dev = "cuda"
    g = Golden("baseline_c4")
    c = g.cfg
    S, A, B = c["state_dim"], c["action_dim"], c["batch"]
    acts = c["activations"] + ["linear"]

    def nets():
        return (GaussianFullyConnectedActor(S, A, c["sizes"], c["activations"]),
                FullyConnectedCritic(S, A, c["sizes"], c["activations"]),
                FullyConnectedCritic(S, A, c["sizes"], c["activations"]))

    actor, q1, q2 = _with_precision(MODES[mode], nets)
    load_init(actor, [S] + c["sizes"] + [2 * A], acts, c["init_seed"], g, "init_actor_")
    load_init(q1, [S + A] + c["sizes"] + [1], acts, c["init_seed"] + 1, g, "init_q1_")
    load_init(q2, [S + A] + c["sizes"] + [1], acts, c["init_seed"] + 2, g, "init_q2_")
    adam = lambda: Optimizer__Union.default(lr=c["lr"])  # noqa: E731
    tr = SACTrainer(actor.to(dev), q1.to(dev), q2.to(dev), rl=RLParameters(**c["rl"]), q_network_optimizer=adam(),
                    actor_network_optimizer=adam(), alpha_optimizer=adam()).to(dev)
    for s in range(c["steps"]):
        b = synthetic.policy_batch(B, S, A, seed=800 + s)
        check_regenerated(b["state"], g, f"step{s}_batch_state")
        batch = synthetic.to_policy_input(b, dev)
        loc, scale_log = tr.actor_network._get_loc_and_scale_log(batch.state)
        d_loc = (loc.cpu() - g.t(f"step{s}_loc")).abs().max().item()
        d_sl = (scale_log.cpu() - g.t(f"step{s}_scale_log")).abs().max().item()
        d_q1 = (tr.q1_network(batch.state, batch.action).cpu() - g.t(f"step{s}_q1")).abs().max().item()
        torch.manual_seed(3000 + s)
        noise_next, noise_cur = torch.randn(B, A), torch.randn(B, A)
        check_regenerated(noise_next, g, f"step{s}_noise_next")
        check_regenerated(noise_cur, g, f"step{s}_noise_cur")
        out = tr.train_step_native(batch, noise_next, noise_cur)
        dl = {}
        for nm in ("q1_loss", "q2_loss", "actor_loss", "alpha_loss"):
            ref = float(g.t(f"step{s}_{nm}"))
            dl[nm] = abs(float(out[nm]) - ref) / max(abs(ref), 1e-3)
        dw = {}
        for n, net in dict(actor=tr.actor_network, q1=tr.q1_network, q2=tr.q2_network, q1_target=tr.q1_network_target,
                           q2_target=tr.q2_network_target).items():
            dw[n] = max(digest_err(p, g, f"step{s}_{n}_{i}") for i, p in enumerate(net.parameters()))
        d_alpha = abs(tr.log_alpha.item() - g.t(f"step{s}_log_alpha").item())
        if s == 0 and max(v for k, v in dw.items() if k != "actor") > 2e-5:
            for n_ in ("q1", "q2"):
                for i, p in enumerate(getattr(tr, n_ + "_network").parameters()):
                    print(" ", n_, "param", i, tuple(p.shape), worst(p, g, f"step0_{n_}_{i}", f"init_{n_}_{i}"))
        print(f"\n[baseline_c4 {mode} step {s}] max|dloc| {d_loc:.3e} max|dscale_log| {d_sl:.3e} max|dq1| {d_q1:.3e} "
              f"rel dloss { {k: float('%.2e' % v) for k, v in dl.items()} } max|dW| { {k: float('%.2e' % v) for k, v in dw.items()} } "
              f"|dlog_alpha| {d_alpha:.2e}")
        actor_frac = max(frac_beyond(p, g, f"step{s}_actor_{i}") for i, p in enumerate(tr.actor_network.parameters()))
        dg = {n: grad_err(tr._e[n]["slab"].grad_views(), g, f"step{s}_grad_{n}_") for n in ("q1", "q2", "actor")}
        print(f"  max|dg|/max|g| { {k: float('%.2e' % v) for k, v in dg.items()} }; actor weights beyond 2e-5: "
              f"{100 * actor_frac:.2f} % of the sampled elements (worst tensor)")
        if mode in ACCURATE and s == 0:
            # (later steps start from actor weights that differ at the sign-undetermined elements — see the header —
            # and are printed, not bounded: the reference itself is not reproducible there across thread counts)
            assert d_loc <= 1e-4 and d_sl <= 1e-4 and d_q1 <= 1e-4
            assert all(v <= 2e-4 for v in dl.values()), dl
            crit_tol = GRAD_TOL[mode] * (10 if mode == "bf16x3" else 1)  # B = 1024: a flipped mask weighs more
            assert dg["q1"] <= crit_tol and dg["q2"] <= crit_tol and dg["actor"] <= 6e-2, dg
            crit = {k: v for k, v in dw.items() if k != "actor"}
            if mode == "f32":
                assert all(v <= 2e-5 for v in crit.values()), dw
            else:  # bf16x3: a handful of critic weights with |g| below the split-product error (see test_c2)
                assert all(v <= 2.0 * c["lr"] * 1.05 for v in crit.values()), dw
            assert dw["actor"] <= 2.0 * c["lr"] * 1.05 and actor_frac <= 0.02, (dw, actor_frac)
            assert d_alpha <= 1e-6
        elif mode in ACCURATE:
            assert d_loc <= 0.1 and d_q1 <= 0.1 and all(v <= 1e-2 for v in dl.values())
        else:
            # step 0 is the clean bf16 figure; later steps start from weights +-2 lr apart at sign-flipped elements
            assert d_loc <= 6e-2 * (1 + 2 * s) and d_sl <= 6e-2 * (1 + 2 * s) and d_q1 <= 6e-2 * (1 + 2 * s)
            assert all(v <= 5e-2 for v in dl.values()), dl
            assert all(v <= 2.0 * (s + 1) * c["lr"] * 1.05 for v in dw.values()), dw


# ---- multi-step drift -------------------------------------------------------------------------------
@pytest.mark.gpu
def test_c2_multi_step_drift_of_the_compliant_mode_is_what_fp32_itself_shows():
    """Twenty consecutive steps at C2's layer shapes (B = 2048, a fresh batch per step) against the oracle stepping on the same
    batches.  Adam's first steps move a weight by ~lr whatever |g| is, so ANY two arithmetics — torch-CPU fp32 and this
    library's exact-fp32 mode included — separate: measured on the MI355X (profiles/microbench/drift.py) after 20 steps the
    fp32 mode is 4.2e-2 from the oracle in Q on a probe batch (rms weight difference 1.3e-4), split-bf16 6.6e-2 (2.5e-4), bf16
    1.3e-1 (5.9e-4).  What a multi-step bound can therefore state is RELATIVE: the 1e-4-compliant mode drifts like fp32
    itself does (within 3x of the fp32 mode's own distance from the oracle, in Q and in rms weight difference), and the
    single-step bounds above are where north_star's 1e-4 lives."""
    from oracle import restated as R
    from reagent_amd.training import DQNTrainer

    assert torch.cuda.is_available()
    dev = torch.device("cuda")
    S, A, H, B, steps = 128, 16, [512, 512, 512], 2048, 20
    acts = ["relu"] * 3 + ["linear"]
    init = synthetic.fc_init([S] + H + [A], acts, seed=40)
    probe = synthetic.dqn_batch(B, S, A, seed=999)
    drift = {}
    for mode in ("f32", "bf16x3", "bf16"):
        q = _with_precision(MODES[mode], lambda: FullyConnectedDQN(S, A, H, ["relu"] * 3))
        with torch.no_grad():
            for p, w in zip(q.parameters(), init):
                p.copy_(w)
        q = q.to(dev)
        tr = DQNTrainer(q, q.get_target_network(), None, actions=[str(i) for i in range(A)],
                        rl=RLParameters(gamma=0.99, target_update_rate=0.001, q_network_loss="huber"),
                        optimizer=Optimizer__Union.default(lr=1e-3), evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)
        o = R.DQNOracle(init, init, acts, gamma=0.99, tau=0.001, loss="huber", lr=1e-3)
        for s in range(steps):
            b = synthetic.dqn_batch(B, S, A, seed=100 + s, p_impossible=0.1)
            loss = tr.train_step_native(synthetic.to_dqn_input(b, dev))
            ref = o.step(b)
            if s == 0 and mode != "bf16":  # the single-step statement, once more, on this path
                assert (tr.all_action_scores.cpu() - ref["q"]).abs().max() <= 1e-4
        with torch.no_grad():
            qp = tr.q_network(synthetic.to_dqn_input(probe, dev).state).float().cpu()
            qr = R.fc_forward(o.params, acts, probe["state"])
        dws = [(p.detach().cpu() - r.detach()).double() for p, r in zip(tr.q_network.parameters(), o.params)]
        drift[mode] = dict(dq=(qp - qr).abs().max().item(),
                           rms=float((sum((d ** 2).sum() for d in dws) / sum(d.numel() for d in dws)).sqrt()),
                           dloss=abs(loss.item() - ref["loss"].item()) / abs(ref["loss"].item()))
        assert torch.isfinite(qp).all()
    print("\n[c2 drift after 20 steps] " + "  ".join(f"{m}: dQ {d['dq']:.2e} rms dW {d['rms']:.2e} rel dloss {d['dloss']:.2e}"
                                                      for m, d in drift.items()))
    f, x = drift["f32"], drift["bf16x3"]
    assert f["dq"] <= 0.2 and f["rms"] <= 1e-3  # fp32 against fp32: the chaos of Adam's first steps, bounded
    assert x["dq"] <= 3.0 * f["dq"] + 1e-3 and x["rms"] <= 3.0 * f["rms"] + 1e-6 and x["dloss"] <= 5e-2
    assert drift["bf16"]["dq"] <= 8.0 * f["dq"] + 1e-3 and drift["bf16"]["rms"] <= 8.0 * f["rms"] + 1e-6
