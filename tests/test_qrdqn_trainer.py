"""QRDQNTrainer (reagent_amd.training) against golden vectors of the reference QRDQNTrainer
(tests/golden/qrdqn_*.npz) — loss within 1e-4 rel, post-step weights within 2e-5 abs (fp32 mode)."""
import pytest
import torch

import reagent_amd._lib as L
from golden_util import Golden
from reagent_amd import synthetic
from reagent_amd.core.parameters import EvaluationParameters, RLParameters
from reagent_amd.models import FullyConnectedDQN, set_default_precision
from reagent_amd.optimizer import Optimizer__Union
from reagent_amd.training import QRDQNTrainer
from test_dqn_trainer import lightning_like_step


def build(g, device, precision):
    c = g.cfg
    set_default_precision(precision)
    try:
        q = FullyConnectedDQN(c["state_dim"], c["num_actions"], c["sizes"], c["activations"], num_atoms=c["num_atoms"])
    finally:
        set_default_precision(L.PREC_F32)
    with torch.no_grad():
        for p, init in zip(q.parameters(), g.seq("init_param_")):
            p.copy_(init)
    q = q.to(device)
    return QRDQNTrainer(q, q.get_target_network(), actions=[str(i) for i in range(c["num_actions"])],
                        rl=RLParameters(**c["rl"]), double_q_learning=c["double_q"], num_atoms=c["num_atoms"],
                        optimizer=Optimizer__Union.default(lr=c["lr"]),
                        evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(device)


@pytest.mark.parametrize("name", ["qrdqn_double", "qrdqn_single_sarsa"])
def test_qrdqn_matches_reference_fp32_mode(backend, name):
    g = Golden(name)
    tr = build(g, backend.device, L.PREC_F32)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    assert len(opts) == 2
    for s in range(g.cfg["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        losses = lightning_like_step(tr, opts, batch)
        ref = g.t(f"step{s}_loss").item()
        assert abs(losses[0].item() - ref) <= 1e-4 * abs(ref) + 1e-6, (losses[0].item(), ref)
        for i, p in enumerate(tr.q_network.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_param_{i}")).abs().max() <= 2e-5, (s, i)
        for i, p in enumerate(tr.q_network_target.parameters()):
            assert (p.detach().cpu() - g.t(f"step{s}_target_{i}")).abs().max() <= 2e-5, (s, i)
    assert "quantiles" in tr.state_dict() and tr.state_dict()["quantiles"].shape == (1, g.cfg["num_atoms"])
    out = tr.q_network(batch.state)
    assert out.shape == (g.cfg["batch"], g.cfg["num_actions"], g.cfg["num_atoms"])  # (B, A, N) view


def test_qrdqn_native_step_and_bf16(backend):
    g = Golden("qrdqn_double")
    tr_a, tr_b = build(g, backend.device, L.PREC_F32), build(g, backend.device, L.PREC_F32)
    opts = [o["optimizer"] for o in tr_a.configure_optimizers()]
    batch = synthetic.to_dqn_input(g.batch(0), backend.device)
    la = lightning_like_step(tr_a, opts, batch)[0]
    lb = tr_b.train_step_native(batch)
    assert torch.equal(la.cpu().reshape(()), lb.cpu().reshape(()))
    for pa, pb in zip(tr_a.q_network.parameters(), tr_b.q_network.parameters()):
        assert torch.equal(pa.detach().cpu(), pb.detach().cpu())
    tr_c = build(g, backend.device, L.PREC_BF16)
    lc = tr_c.train_step_native(batch)
    ref = g.t("step0_loss").item()
    assert abs(lc.item() - ref) <= 3e-2 * abs(ref) + 1e-3


def test_qrdqn_cpe_matches_reference(backend):
    """calc_cpe_in_training for QR-DQN: the CPE targets use q_network(next_state).mean(atoms) after the
    q-network step (qrdqn_trainer.py:161-177); two extra metrics"""
    from test_dqn_trainer import CPE_NETS

    g = Golden("qrdqn_cpe")
    c = g.cfg
    A, n_out = c["num_actions"], (len(c["cpe_metrics"]) + 1) * c["num_actions"]

    def net(out, prefix, num_atoms=None):
        m = FullyConnectedDQN(c["state_dim"], out, c["sizes"], c["activations"], num_atoms=num_atoms)
        with torch.no_grad():
            for p, init in zip(m.parameters(), g.seq(prefix)):
                p.copy_(init)
        return m.to(backend.device)

    q = net(A, "init_param_", c["num_atoms"])
    reward_net, q_cpe = net(n_out, "init_reward_network_"), net(n_out, "init_q_network_cpe_")
    tr = QRDQNTrainer(q, q.get_target_network(), metrics_to_score=list(c["cpe_metrics"]), reward_network=reward_net,
                      q_network_cpe=q_cpe, q_network_cpe_target=q_cpe.get_target_network(),
                      actions=[str(i) for i in range(A)], rl=RLParameters(**c["rl"]),
                      double_q_learning=c["double_q"], num_atoms=c["num_atoms"],
                      optimizer=Optimizer__Union.default(lr=c["lr"]), cpe_optimizer=Optimizer__Union.default(lr=c["lr"]),
                      evaluation=EvaluationParameters(calc_cpe_in_training=True)).to(backend.device)
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    assert len(opts) == 4
    for s in range(c["steps"]):
        batch = synthetic.to_dqn_input(g.batch(s), backend.device)
        losses = lightning_like_step(tr, opts, batch)
        for got, key in zip(losses[:3], ["loss", "reward_loss", "cpe_loss"]):
            want = g.t(f"step{s}_{key}").item()
            assert abs(got.item() - want) <= 1e-4 * abs(want) + 1e-6, (key, got.item(), want)
        for netname in CPE_NETS:
            for i, p in enumerate(getattr(tr, netname).parameters()):
                assert (p.detach().cpu() - g.t(f"step{s}_{netname}_{i}")).abs().max() <= 2e-5, (s, netname, i)


# ---- grouped wide layer (qr_engine.py / csrc/qr_grouped.hip) against the dense [B, A * N] path -------------------
def _qr_pair(device, S, A, N, hidden, rl, double_q, seed=3, precision=L.PREC_BF16):
    def one(grouped):
        torch.manual_seed(seed)
        set_default_precision(precision)
        try:
            q = FullyConnectedDQN(S, A, hidden, ["relu"] * len(hidden), num_atoms=N)
        finally:
            set_default_precision(L.PREC_F32)
        q = q.to(device)
        tr = QRDQNTrainer(q, q.get_target_network(), actions=[str(i) for i in range(A)], rl=RLParameters(**rl),
                          double_q_learning=double_q, num_atoms=N, optimizer=Optimizer__Union.default(lr=1e-3),
                          evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(device)
        tr.use_grouped_head = grouped
        return tr

    return one(True), one(False)


@pytest.mark.parametrize("rl,double_q", [
    (dict(gamma=0.9, target_update_rate=0.1, maxq_learning=True), True),
    (dict(gamma=0.9, target_update_rate=0.1, maxq_learning=True), False),
    (dict(gamma=0.9, target_update_rate=0.1, maxq_learning=False, reward_boost={"1": 0.5}), True),
])
@pytest.mark.parametrize("N", [10, 72])  # 72 quantiles: a wide output leaves through the LDS staging area as whole rows
def test_grouped_head_equals_dense_path(backend, rl, double_q, N):
    """Same bf16 trunk kernels on both sides, so the comparison isolates the grouped machinery: the per-action
    mean layer for a*, the device-built grouped spaces, forward / loss / input gradient / weight gradient of the
    wide layer on [rows, N] instead of [B, A * N]."""
    from reagent_amd.qr_engine import GroupedQR

    dev = backend.device
    S, A, B = 24, 4, 300
    tg, td = _qr_pair(dev, S, A, N, [256, 256], rl, double_q)
    assert GroupedQR.eligible(tg)

    class Reporter:  # with a reporter attached the step itself evaluates all_q_values (model_values of qrdqn_trainer.py:189),
        def log(self, **kw):  # with the step's weights; without one the property evaluates the network on first read
            pass

    tg.set_reporter(Reporter())
    b = synthetic.dqn_batch(B, S, A, seed=21, p_impossible=0.3)
    # (1) exactly one possible next action per row: a* is forced, so both paths regress the same targets and the
    #     comparison is tight (the trunks are the same kernels; the wide layer is summed in the same K order)
    g = torch.Generator().manual_seed(5)
    forced = torch.nn.functional.one_hot(torch.randint(A, (B,), generator=g), A).float()
    b1 = dict(b, possible_next_actions_mask=forced, next_action=forced * b["not_terminal"])
    batch = synthetic.to_dqn_input(b1, dev)
    lg, ld = tg.train_step_native(batch), td.train_step_native(batch)
    assert tg._gq_active is not None and getattr(td, "_gq_active", None) is None
    assert abs(lg.item() - ld.item()) <= 1e-5 * abs(ld.item()), (lg.item(), ld.item())
    for i, (x, y) in enumerate(zip(tg._slab.grad_views(), td._slab.grad_views())):
        rel = ((x - y).norm() / (y.norm() + 1e-12)).item()
        assert rel <= 3e-3, (i, rel)  # bf16 rounding of dZ at different points of the two backward paths
    assert (tg.all_q_values - td.all_q_values).abs().max() <= 3e-2  # mean layer vs mean of bf16-operand logits
    # (2) free masks: a* comes from the per-action mean layer; it may differ from the dense path's on near ties
    #     (bf16 noise, ~1e-3 on these values), everywhere else the rows' losses agree
    batch = synthetic.to_dqn_input(b, dev)
    lg, ld = tg.train_step_native(batch), td.train_step_native(batch)
    if rl["maxq_learning"]:
        sel = td._qn_online if double_q else td._qn_target
        qn = sel.view(B, A, N).mean(2)
        a_dense = (qn + -1e9 * (1 - b["possible_next_actions_mask"].to(dev))).argmax(1)
        flips = (a_dense != tg._gq.key_next.long()).float().mean().item()
        assert flips <= 0.02, flips
    assert abs(lg.item() - ld.item()) <= 3e-3 * abs(ld.item()), (lg.item(), ld.item())
    # the generator / Lightning path goes through the same engine
    opts = [o["optimizer"] for o in tg.configure_optimizers()]
    losses = lightning_like_step(tg, opts, synthetic.to_dqn_input(b, dev))
    assert torch.isfinite(losses[0]).all()


@pytest.mark.parametrize("rl,double_q", [
    (dict(gamma=0.9, target_update_rate=0.1, maxq_learning=True), True),
    (dict(gamma=0.9, target_update_rate=0.1, maxq_learning=False, reward_boost={"1": 0.5}), True),
])
@pytest.mark.parametrize("N", [72])  # the wide output leaves through the LDS staging area behind both planes (N = 10, the
def test_grouped_head_split_bf16_meets_the_fp32_bound(backend, rl, double_q, N):  # per-element update path: bf16 tests below)
    """The grouped engine on split-bf16 operands (PREC_BF16X3: what BASELINE config 3 runs in its 1e-4-compliant mode)
    against the DENSE path of the same trainer, which for a [B, A * N] head runs exact-fp32 GEMMs: quantiles of the
    logged action and the per-action means within 1e-4, loss within 1e-5 rel, every gradient within the split-bf16 bound
    (ReLU masks that flip under a 1e-5 forward error), and the update in one launch bit-identical to separate launches."""
    from reagent_amd.qr_engine import GroupedQR

    dev = backend.device
    S, A, B = 24, 4, 300
    tg, td = _qr_pair(dev, S, A, N, [256, 256], rl, double_q, precision=L.PREC_BF16X3)
    assert GroupedQR.eligible(tg)

    class Reporter:  # with a reporter attached all_q_values is evaluated inside the step, with the step's weights
        def log(self, **kw):
            pass

    tg.set_reporter(Reporter())
    b = synthetic.dqn_batch(B, S, A, seed=21, p_impossible=0.3)
    g = torch.Generator().manual_seed(5)
    forced = torch.nn.functional.one_hot(torch.randint(A, (B,), generator=g), A).float()
    b1 = dict(b, possible_next_actions_mask=forced, next_action=forced * b["not_terminal"])  # a* forced: same targets
    batch = synthetic.to_dqn_input(b1, dev)
    with torch.no_grad():
        z_ref = td.q_network(batch.state)  # [B, A, N], exact fp32
    lg, ld = tg.train_step_native(batch), td.train_step_native(batch)
    gq = tg._gq_active
    assert gq is not None and gq.x3 and gq.online.st.x3 and getattr(td, "_gq_active", None) is None
    assert abs(lg.item() - ld.item()) <= 1e-5 * abs(ld.item()), (lg.item(), ld.item())
    # quantiles of the logged action, row by row of the grouped space
    rowmap, key = gq.sp_cur.rowmap.cpu().long(), gq.key_cur.cpu().long()
    live = rowmap >= 0
    rows = rowmap[live]
    z_got = gq.z.cpu()[live][:, :N]
    z_want = z_ref.cpu()[rows, key[rows]]
    assert live.sum().item() == B and (z_got - z_want).abs().max() <= 1e-4, (z_got - z_want).abs().max()
    assert (tg.all_q_values.cpu() - z_ref.cpu().mean(dim=2)).abs().max() <= 1e-4
    for i, (x, y) in enumerate(zip(tg._slab.grad_views(), td._slab.grad_views())):
        rel = ((x - y).abs().max() / (y.abs().max() + 1e-30)).item()
        assert rel <= 3e-3, (i, rel)
    assert isinstance(tg._fused_plan, dict) and tg._fused_plan["desc"].x3 == 1 and tg._fused_plan["desc"].group_rows[2] == N
    if not rl["maxq_learning"]:
        return
    # one-launch update == separate launches, bit for bit, in this mode too (both planes of every fragment set)
    sep, _ = _qr_pair(dev, S, A, N, [256, 256], rl, double_q, precision=L.PREC_BF16X3)
    sep._fused_plan = False
    fused, _ = _qr_pair(dev, S, A, N, [256, 256], rl, double_q, precision=L.PREC_BF16X3)
    for s_ in range(2):
        bb = synthetic.to_dqn_input(synthetic.dqn_batch(B, S, A, seed=60 + s_, p_impossible=0.3), dev)
        assert torch.equal(fused.train_step_native(bb), sep.train_step_native(bb)), s_
    for a_, b_ in zip(list(fused.q_network.parameters()) + list(fused.q_network_target.parameters()),
                      list(sep.q_network.parameters()) + list(sep.q_network_target.parameters())):
        assert torch.equal(a_, b_)
    gf, gs = fused._gq, sep._gq
    for net in (gs.online, gs.target):
        net.stage()
    for net in (gf.online, gf.target):
        net.ensure_mean()
    for nf, ns in ((gf.online, gs.online), (gf.target, gs.target)):
        assert torch.equal(nf.gh.wf, ns.gh.wf)
        for a_, b_ in zip(nf.st._wf, ns.st._wf):
            assert torch.equal(a_, b_)
    assert torch.equal(gf.online.gh.wb, gs.online.gh.wb)


@pytest.mark.parametrize("hidden", [[256, 256], [512, 512]])  # LDS row pitch 264 / 520: the masked dZ copy behind / inside the tile
@pytest.mark.parametrize("precision", [L.PREC_BF16, L.PREC_BF16X3])
def test_dense_grouped_space_equals_the_padded_one(backend, monkeypatch, hidden, precision):
    """Round 4: the grouped spaces are DENSE (no padding between the groups: B / 128 tiles; a tile with rows of several groups
    runs the grouped layer once per group) — against the rounds 2-3 layout (every group padded to whole tiles, one group per
    tile), same kernels.  Per-row results (quantiles, targets, dz) are identical bit for bit — a row's arithmetic does not
    depend on its tile; sums over rows (loss, weight and bias gradients) are taken in another order: fp32 rounding."""
    dev = backend.device
    S, A, B, N = 24, 4, 300, 72
    rl = dict(gamma=0.9, target_update_rate=0.1, maxq_learning=True, reward_boost={"1": 0.5})
    batch = synthetic.to_dqn_input(synthetic.dqn_batch(B, S, A, seed=33, p_impossible=0.3), dev)
    out = {}
    for dense in (True, False):
        monkeypatch.setenv("RG_QR_DENSE", "1" if dense else "0")
        tr, _ = _qr_pair(dev, S, A, N, hidden, rl, True, precision=precision)
        loss = tr.train_step_native(batch)
        gq = tr._gq_active
        assert gq is not None and gq.dense == dense and gq.sp_cur.dense == dense
        assert gq.sp_cur.n_tiles == (B + 127) // 128 + (0 if dense else A)
        rm = gq.sp_cur.rowmap.cpu().long()
        live = rm >= 0
        order = torch.argsort(rm[live])  # grouped rows in batch order
        out[dense] = dict(loss=loss.item(), z=gq.z.cpu()[live][order], dz=gq.dz.cpu()[live][order], zt=gq.zt.cpu().clone(),
                          key=gq.key_next.cpu().clone(), grads=[g.clone() for g in tr._slab.grad_views()],
                          params=[p.detach().clone() for p in tr.q_network.parameters()])
        assert live.sum().item() == B
        if dense:  # three tiles, four groups of ~75 rows: every tile holds rows of two or three groups
            rb = gq.sp_cur.row_begin.cpu()
            assert all(int(rb[g]) % 128 != 0 for g in range(1, A))
    d, p = out[True], out[False]
    # A row's sums over K are taken in an order that depends on its workgroup (k_rotation), so the two layouts differ by fp32
    # summation order — which, where a hidden activation sits on a bf16 rounding boundary, becomes one bf16 ulp of that
    # activation: ~1e-3 on a quantile in bf16, ~1e-6 in split-bf16 (whose lo plane carries the next 8 bits).
    x3 = precision == L.PREC_BF16X3
    tol = 2e-5 if x3 else 2e-2
    flips = (d["key"] != p["key"]).float().mean().item()  # a* on a near tie
    assert flips <= (0.0 if x3 else 0.01), flips
    same = d["key"] == p["key"]
    assert (d["zt"] - p["zt"])[same].abs().max() <= tol and (d["z"] - p["z"]).abs().max() <= tol
    assert (d["dz"] - p["dz"])[same].abs().max() <= (1e-3 if x3 else 0.3) * p["dz"].abs().max()
    assert abs(d["loss"] - p["loss"]) <= (1e-5 if x3 else 3e-3) * abs(p["loss"])
    for i, (x, y) in enumerate(zip(d["grads"], p["grads"])):
        rel = ((x - y).norm() / (y.norm() + 1e-30)).item()
        assert rel <= (1e-4 if x3 else 1e-2), (i, rel)


@pytest.mark.parametrize("rl,double_q,N", [
    (dict(gamma=0.9, target_update_rate=0.1, maxq_learning=True), True, 16),
    (dict(gamma=0.9, target_update_rate=0.1, maxq_learning=True), False, 16),   # a* from the TARGET network's mean layer
    (dict(gamma=0.9, target_update_rate=0.1, maxq_learning=False), True, 16),   # SARSA: no mean layer in the step at all
    (dict(gamma=0.9, target_update_rate=0.1, maxq_learning=True), True, 10),    # 10 rows per action: records of 8 would
])                                                                              # straddle groups -> per-element path
def test_grouped_fused_update_equals_separate_launches(backend, rl, double_q, N):
    """The QR-DQN step's update in ONE launch (rg_mlp_update_fused with the wide layer as a grouped last layer: Adam +
    soft update + the trunk fragments and the per-action fragment blocks of both networks) + the mean layer by
    rg_wide_head_mean_staged, against the separate launches (rg_adam_step, rg_soft_update, rg_mlp_stage_weights_fused x 2,
    rg_group_weights_stage x 2, rg_wide_head_mean x 2): same bits in the parameters, the targets, the Adam moments, every
    fragment buffer, the mean layer, and the following steps' losses."""
    from reagent_amd.qr_engine import GroupedQR

    dev = backend.device
    S, A, B = 24, 4, 300

    def make():
        tr, _ = _qr_pair(dev, S, A, N, [256, 256], rl, double_q)
        return tr

    fused, separate = make(), make()
    separate._fused_plan = False
    for s in range(3):
        batch = synthetic.to_dqn_input(synthetic.dqn_batch(B, S, A, seed=60 + s, p_impossible=0.3), dev)
        la, lb = fused.train_step_native(batch), separate.train_step_native(batch)
        assert torch.equal(la, lb), s
    assert isinstance(fused._qs, GroupedQR) and isinstance(fused._fused_plan, dict) and fused._fused_plan["desc"].group_rows[2] == N
    for a, b in zip(list(fused.q_network.parameters()) + list(fused.q_network_target.parameters()),
                    list(separate.q_network.parameters()) + list(separate.q_network_target.parameters())):
        assert torch.equal(a, b)
    oa, ob = fused.native_optimizers()[0], separate.native_optimizers()[0]
    for pa, pb in zip(fused.q_network.parameters(), separate.q_network.parameters()):
        assert torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"])
        assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"])
    # the staged state itself: bring both engines fully up to date (the fused path computes a mean layer only when a
    # forward needs it, the separate path stages lazily before its next forward)
    gf, gs = fused._gq, separate._gq
    for net in (gs.online, gs.target):
        net.stage()
    for net in (gf.online, gf.target):
        net.ensure_mean()
    for nf, ns in ((gf.online, gs.online), (gf.target, gs.target)):
        assert torch.equal(nf.wbar, ns.wbar) and torch.equal(nf.bbar, ns.bbar)
        assert torch.equal(nf.gh.wf, ns.gh.wf)
        for a, b in zip(nf.st._wf, ns.st._wf):
            assert torch.equal(a, b)
    assert torch.equal(gf.online.gh.wb, gs.online.gh.wb)
    for a, b in zip(gf.online.st._wb[:-1], gs.online.st._wb[:-1]):  # (the mean layer has no backward)
        assert torch.equal(a, b)
    assert torch.equal(fused.all_q_values, separate.all_q_values)


@pytest.mark.parametrize("dense", [True, False])
@pytest.mark.parametrize("B,G", [(1000, 5), (37, 3), (4096, 16), (300, 40), (1500, 129)])  # 129: the most groups the kernels take
def test_grouped_space_layout(backend, B, G, dense):
    """dense: the groups follow each other without padding (ceil(B / 128) tiles); otherwise every group starts on a tile.
    (300, 40): groups of ~7 rows — a tile holds many groups, and some groups are empty"""
    from reagent_amd.qr_engine import TILE, GroupedSpace

    key = torch.randint(0, G + 1, (B,), generator=torch.Generator().manual_seed(0)).to(torch.int32)  # G = "no group"
    if G == 40:
        key[key == 7] = 8  # an empty group in the middle
    sp = GroupedSpace(B, G, backend.device, dense=dense).build(key.to(backend.device))
    rm, tk, rb = sp.rowmap.cpu(), sp.tile_key.cpu(), sp.row_begin.cpu()
    assert sp.n_tiles == (B + TILE - 1) // TILE + (0 if dense else G)
    assert rm.shape == (sp.n_tiles * TILE,) and tk.shape == (sp.n_tiles,) and rb.shape == (G + 1,)
    seen = rm[rm >= 0]
    assert sorted(seen.tolist()) == sorted(torch.nonzero(key < G).reshape(-1).tolist())  # every grouped row exactly once
    assert rb[0] == 0 and (rb[1:] >= rb[:-1]).all()
    for g in range(G):
        n = int((key == g).sum())
        lo, hi = int(rb[g]), int(rb[g + 1])
        assert hi - lo == (n if dense else (n + TILE - 1) // TILE * TILE)
        rows = rm[lo:lo + n]
        assert (key[rows.long()] == g).all() and (rm[lo + n:hi] == -1).all()  # the group's rows first, then its padding
        if n > 1:  # batch order inside a group: deterministic
            assert (rows[1:] > rows[:-1]).all()
    assert (rm[int(rb[G]):] == -1).all()
    for t in range(sp.n_tiles):
        first = [g for g in range(G) if rb[g + 1] > rb[g] and rb[g + 1] > t * TILE and rb[g] < (t + 1) * TILE]
        assert tk[t] == (first[0] if first else -1)


def test_compact_head_against_the_pair_loop(backend):
    """rg_qr_compact_head (sorted targets + prefix sums, O(N log N) per row) == the N x N pair loop of
    qrdqn_trainer.py:143-160 evaluated in float64 — values chosen so that T_i - C_j hits -1, 0 and 1 exactly and
    repeats, where the ranges of the closed form meet."""
    from reagent_amd import ops
    from reagent_amd.qr_engine import TILE, GroupedSpace

    dev = backend.device
    B, G, N = 150, 3, 37
    g = torch.Generator().manual_seed(4)
    key = torch.randint(0, G, (B,), generator=g).to(torch.int32)
    sp = GroupedSpace(B, G, dev).build(key.to(dev))
    R = sp.rows
    z = torch.randn(R, 40, generator=g) * 1.5
    zt = torch.randn(B, 40, generator=g) * 1.5
    z[:, :5] = torch.round(z[:, :5])      # integers: differences of exactly 0 and +-1 against ...
    zt[:, :7] = torch.round(zt[:, :7])    # ... integer targets (reward 0, discount 1 below for half of the rows)
    reward = torch.where(torch.arange(B) % 2 == 0, torch.zeros(B), torch.randn(B, generator=g))
    nt = torch.where(torch.arange(B) % 5 == 0, torch.zeros(B), torch.ones(B))
    boosts = torch.tensor([0.0, 0.25, -0.5])
    gamma = 1.0
    tau = ((0.5 + torch.arange(N).float()) / N)
    dz = torch.full((R, 40), 7.0)
    lp, tl = torch.zeros(R), torch.zeros(sp.n_tiles)
    D = lambda t: t.to(dev)  # noqa: E731
    dzd, lpd, tld = D(dz), D(lp), D(tl)
    ops.qr_compact_head(D(z), D(zt), sp.rowmap, D(key), D(reward), D(boosts), D(nt), gamma, None, D(tau), B, N, dzd,
                        lpd, tld)
    rm = sp.rowmap.cpu()
    ref_dz = torch.zeros(R, 40, dtype=torch.float64)
    ref_l = torch.zeros(R, dtype=torch.float64)
    for r in range(R):
        b = int(rm[r])
        if b < 0:
            continue
        T = (reward[b] + boosts[int(key[b])]).double() + gamma * nt[b].double() * zt[b, :N].double()
        C = z[r, :N].double()
        td = T[:, None] - C[None, :]                       # [i, j]
        ad = td.abs()
        hub = torch.where(ad < 1, 0.5 * td * td, ad - 0.5)
        dh = torch.where(ad < 1, td, torch.sign(td))
        w = (tau.double()[None, :] - (td < 0).double()).abs()
        inv = 1.0 / (N * B * N)
        ref_l[r] = (hub * w).sum() * inv
        ref_dz[r, :N] = -(dh * w).sum(0) * inv
    assert (dzd.cpu().double() - ref_dz).abs().max() <= 1e-6 * ref_dz.abs().max() + 1e-9
    assert (lpd.cpu().double() - ref_l).abs().max() <= 2e-6 * ref_l.abs().max()
    assert abs(tld.cpu().double().sum() - ref_l.sum()) <= 1e-6 * ref_l.sum()


@pytest.mark.parametrize("name", ["qrdqn_double", "qrdqn_single_sarsa"])
def test_reporter_fields_match_the_reference(backend, name):
    """qrdqn_trainer.py:178-192: what the step hands its reporter — td_loss, logged action indices, propensities, boosted
    rewards, the mean-over-atoms Q-values and their masked arg-max — against what the reference's reporter received on the
    golden batches"""
    g = Golden(name)
    tr = build(g, backend.device, L.PREC_F32)
    seen = {}

    class Reporter:
        def log(self, **kw):
            seen.update({k: v for k, v in kw.items() if v is not None})

    tr.set_reporter(Reporter())
    opts = [o["optimizer"] for o in tr.configure_optimizers()]
    for s in range(g.cfg["steps"]):
        seen.clear()
        lightning_like_step(tr, opts, synthetic.to_dqn_input(g.batch(s), backend.device))
        want = {k[len(f"step{s}_report_"):]: g.t(k) for k in g.z.files if k.startswith(f"step{s}_report_")}
        assert set(seen) == set(want) and len(want) == 6
        for k, ref in want.items():
            v = seen[k].detach().cpu()
            assert v.reshape(ref.shape).dtype == ref.dtype or ref.dtype.is_floating_point, k
            if ref.dtype.is_floating_point:
                assert (v.reshape(ref.shape) - ref).abs().max() <= 1e-4 * max(1.0, ref.abs().max().item()), (s, k)
            else:
                assert torch.equal(v.reshape(ref.shape), ref), (s, k)
