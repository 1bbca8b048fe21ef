"""The restated CPU oracle (oracle/restated.py) against the golden vectors produced by the real
reference trainers / replay buffer / preprocessor (tests/golden, oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from golden_util import Golden
from oracle import restated as R

TIGHT = dict(rtol=1e-6, atol=1e-7)


def _acts(cfg):
    return list(cfg["activations"]) + ["linear"]


def _boosts(cfg):
    rb = cfg["rl"].get("reward_boost")
    if not rb:
        return None
    t = torch.zeros(1, cfg["num_actions"])
    for k, v in rb.items():
        t[0, int(k)] = v
    return t


@pytest.mark.parametrize("name", ["dqn_c1", "dqn_huber_masks", "dqn_sarsa_multistep", "dqn_timediff", "dqn_dueling"])
def test_dqn_oracle_matches_reference(name):
    g = Golden(name)
    c = g.cfg
    init = g.seq("init_param_")
    o = R.DQNOracle(init, init, _acts(c), gamma=c["rl"]["gamma"], tau=c["rl"]["target_update_rate"],
                    loss=c["rl"]["q_network_loss"], double_q=c["double_q"], maxq=c["rl"]["maxq_learning"],
                    lr=c["lr"], reward_boosts=_boosts(c),
                    use_seq_num_diff_as_time_diff=c["rl"].get("use_seq_num_diff_as_time_diff", False),
                    multi_steps=c["rl"].get("multi_steps"), dueling=c.get("dueling", False))
    for s in range(c["steps"]):
        out = o.step(g.batch(s))
        torch.testing.assert_close(out["loss"], g.t(f"step{s}_loss"), **TIGHT)
        torch.testing.assert_close(out["q"], g.t(f"step{s}_q"), **TIGHT)
        for i, p in enumerate(o.params):
            torch.testing.assert_close(p.detach(), g.t(f"step{s}_param_{i}"), **TIGHT)
        for i, p in enumerate(o.target):
            torch.testing.assert_close(p, g.t(f"step{s}_target_{i}"), **TIGHT)
    for i, p in enumerate(o.params):
        torch.testing.assert_close(o.opt.state[p]["exp_avg"], g.t(f"final_exp_avg_{i}"), **TIGHT)
        torch.testing.assert_close(o.opt.state[p]["exp_avg_sq"], g.t(f"final_exp_avg_sq_{i}"), **TIGHT)


@pytest.mark.parametrize("name", ["qrdqn_double", "qrdqn_single_sarsa", "qrdqn_dueling"])
def test_qrdqn_oracle_matches_reference(name):
    g = Golden(name)
    c = g.cfg
    init = g.seq("init_param_")
    o = R.QRDQNOracle(init, init, _acts(c), num_actions=c["num_actions"], num_atoms=c["num_atoms"],
                      gamma=c["rl"]["gamma"], tau=c["rl"]["target_update_rate"], double_q=c["double_q"],
                      maxq=c["rl"]["maxq_learning"], lr=c["lr"], dueling=c.get("dueling", False))
    for s in range(c["steps"]):
        out = o.step(g.batch(s))
        torch.testing.assert_close(out["loss"], g.t(f"step{s}_loss"), **TIGHT)
        for i, p in enumerate(o.params):
            torch.testing.assert_close(p.detach(), g.t(f"step{s}_param_{i}"), **TIGHT)
        for i, p in enumerate(o.target):
            torch.testing.assert_close(p, g.t(f"step{s}_target_{i}"), **TIGHT)


def test_sac_oracle_matches_reference():
    g = Golden("sac_twin")
    c = g.cfg
    acts = _acts(c)
    o = R.SACOracle(g.seq("init_actor_"), g.seq("init_q1_"), g.seq("init_q2_"), acts, acts, c["action_dim"],
                    gamma=c["rl"]["gamma"], tau=c["rl"]["target_update_rate"], lr=c["lr"])
    for s in range(c["steps"]):
        out = o.step(g.batch(s), g.t(f"step{s}_noise_next"), g.t(f"step{s}_noise_cur"))
        for nm in ["q1_loss", "q2_loss", "actor_loss", "alpha_loss"]:
            torch.testing.assert_close(out[nm].to(torch.float64), g.t(f"step{s}_{nm}").to(torch.float64),
                                       rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(o.log_alpha.detach(), g.t(f"step{s}_log_alpha"), rtol=1e-6, atol=1e-7)
        for n, ps in dict(actor=o.actor, q1=o.q1, q2=o.q2, q1_target=o.q1_t, q2_target=o.q2_t).items():
            for i, p in enumerate(ps):
                torch.testing.assert_close(p.detach(), g.t(f"step{s}_{n}_{i}"), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["replay_basic", "replay_nstep", "replay_stack"])
def test_replay_oracle_matches_reference_bit_exact(name):
    g = Golden(name)
    c = g.cfg
    rb = R.ReplayOracle(stack_size=c["stack_size"], replay_capacity=c["replay_capacity"],
                        update_horizon=c["update_horizon"], gamma=c["gamma"])
    keys = ["observation", "action", "reward", "terminal", "possible_actions_mask", "log_prob", "mdp_id"]
    n = c["n_add"]
    for i in range(n):
        rb.add(**{k: g.a(f"add_{k}")[i] for k in keys})
    assert rb.add_count == int(g.a("add_count"))
    np.testing.assert_array_equal(rb.valid, g.a("valid_mask"))
    assert rb.valid.sum() == int(g.a("size"))
    out = rb.sample(g.a("indices"), extra_keys=["possible_actions_mask", "log_prob", "mdp_id"])
    checked = 0
    for k, v in out.items():
        ref = g.a(f"out_{k}")
        assert v.shape == ref.shape, (k, v.shape, ref.shape)
        assert v.dtype == ref.dtype, (k, v.dtype, ref.dtype)
        np.testing.assert_array_equal(v, ref, err_msg=k)  # bit exact, incl. the n-step fp32 reward
        checked += 1
    assert checked == len([f for f in g.z.files if f.startswith("out_")])


def test_preprocessor_oracle_matches_reference():
    from types import SimpleNamespace

    g = Golden("preprocessor_all_types")
    norm = {int(k): SimpleNamespace(**v) for k, v in g.cfg["norm"].items()}
    assert R.sort_features(norm) == g.cfg["sorted_features"]
    out = R.preprocess(norm, g.t("x"), g.t("presence"))
    torch.testing.assert_close(out, g.t("out"), rtol=1e-6, atol=1e-6)


def test_fc_options_oracle_matches_reference():
    """restated batch-norm / layer-norm / residual forward (training and eval mode, moved running statistics) against the
    reference module's outputs in tests/golden/fc_options.npz"""
    g = Golden("fc_options")
    c = g.cfg
    names = [str(n) for n in g.a("names")]
    state = {n: g.t(f"init_{i}") for i, n in enumerate(names)}
    kw = dict(use_batch_norm=True, use_layer_norm=True, use_skip_connections=True)
    out, moved = R.fc_forward_options(state, c["layers"], c["activations"], g.t("x"), training=True, **kw)
    assert (out - g.t("train_out")).abs().max() <= 2e-5
    after = {n: g.t(f"after_{i}") for i, n in enumerate(names)}
    assert moved and all((moved[k] - after[k]).abs().max() <= 1e-6 for k in moved)
    out_e, _ = R.fc_forward_options(after, c["layers"], c["activations"], g.t("x"), training=False, **kw)
    assert (out_e - g.t("eval_out")).abs().max() <= 2e-5


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/reagent"), reason="needs the reference checkout")
def test_committed_goldens_are_what_the_reference_produces():
    """`python -m oracle.make_golden --check`: every fixture under tests/golden regenerated from the
    unmodified reference and compared array by array with the committed file (build container only)"""
    import subprocess
    import sys

    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "oracle.make_golden", "--check"], cwd=root, capture_output=True, text=True,
                         timeout=1500)
    assert out.returncode == 0 and "all identical" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
