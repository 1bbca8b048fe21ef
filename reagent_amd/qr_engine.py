"""QR-DQN step with the wide output layer as a GROUPED layer (csrc/qr_grouped.hip): what
reagent/training/qrdqn_trainer.py:108-160 computes, without ever writing the [B, A * N] logits.

    q_next     = sel_net(next_state).mean(dim=2)         one fused forward whose output layer is the per-action
                                                         MEAN of the wide layer's rows (rg_wide_head_mean)
    a*         = arg max with the possible-actions mask   rg_qr_select_action                    (:125-135, :210-214)
    grouped space of a*: rows sorted by a* (rg_qr_select_group_rows: on the device)
    zt[b, :]   = target_net(next_state)[b, a*, :]         ONE fused forward in grouped space whose output layer is the
                                                         tile's action slice of the wide layer        (:137-141)
    grouped space of the logged action
    z[r, :]    = q_net(state)[b, logged a, :]              the same, saving for the backward          (:143-146)
    loss, dz   = quantile Huber on [rows, N]               rg_qr_compact_head                      (:148-160, :217-218)
    backward   = ONE rg_mlp_backward_fused (the wide layer's input gradient is its first step, per-tile W_g^T),
                 rg_mlp_wgrad_fused for the trunk, rg_group_head_wgrad for the wide layer
Arithmetic is that of the fused stack in the network's precision — bf16 operands, or split-bf16 ("bf16x3": hi + lo planes,
three MFMAs per product, quantiles within 1e-4 of the fp32 reference) — with fp32 accumulation; the only algebraic rewrite
is mean_n(h . W[a, n] + b[a, n]) = h . mean_n W[a, n] + mean_n b[a, n].  A transition whose logged action row is all
zero contributes nothing (the reference would regress C = 0 for it; one-hot actions are the trainer's contract,
dqn_trainer_base.py `_check_input`).
"""
from typing import Optional

import os

import torch

from . import _lib as L
from . import ops
from .engine import FusedMLP, GroupedHead, fused_backward_grouped, fused_forward_grouped

TILE = 128


class GroupedSpace:
    """rows of a batch sorted by an int32 key in [0, G] (G = "no group": dropped) — rg_group_rows: a stable counting sort on
    the device, static shapes, no host synchronisation.  dense (round 4, the default; RG_QR_DENSE=0 for the rounds 2-3
    layout): the groups follow each other without padding, ceil(B / TILE) tiles — C3: 512, two full rounds of the 256 CUs
    where the padded layout's ~520 made every grouped launch a third, nearly empty one; otherwise each key's rows are padded
    to whole tiles.  rowmap [TILE * n_tiles], tile_key [n_tiles] (first group with rows in the tile), row_begin [G + 1], all int32"""

    def __init__(self, B: int, G: int, device, dense: bool = True):
        self.B, self.G, self.dense = B, G, bool(dense)
        self.n_tiles = (B + TILE - 1) // TILE + (0 if dense else G)
        self.rows = self.n_tiles * TILE
        i32 = dict(dtype=torch.int32, device=device)
        self.rowmap = torch.empty(self.rows, **i32)
        self.tile_key = torch.empty(self.n_tiles, **i32)
        self.row_begin = torch.empty(G + 1, **i32)
        self._ws = torch.empty(max(1, L.lib().rg_group_rows_workspace_bytes(B, G) // 4), **i32)

    def build(self, key: torch.Tensor):
        ops.group_rows(key, self.G, self.n_tiles, self.rowmap, self.tile_key, self.row_begin, self._ws, dense=self.dense)
        return self

    def build_selected(self, q, mask, maxq: bool, key: torch.Tensor):
        """key = rg_qr_select_action(q, mask, maxq), then build(key) — in the two launches of the latter"""
        ops.qr_select_group_rows(q, mask, maxq, key, self.n_tiles, self.rowmap, self.tile_key, self.row_begin, self._ws,
                                 dense=self.dense)
        return self


class _Net:
    """one Q-network seen as trunk + grouped wide layer"""

    def __init__(self, net, A: int, N: int, need_bwd: bool, x3: bool = False):
        lin = net.fc.linears()
        self.x3 = bool(x3)
        self.lin, self.head = lin, lin[-1]
        self.A, self.N, self.need_bwd = A, N, need_bwd
        H = self.head.weight.shape[1]
        dev = self.head.weight.device
        self.H = H
        self.wbar = torch.zeros(A, H, device=dev)
        self.bbar = torch.zeros(A, device=dev)
        acts = [L.ACT[a] for a in net.fc.activation_names]
        self.st = FusedMLP([l.weight for l in lin[:-1]] + [self.wbar], [l.bias for l in lin[:-1]] + [self.bbar],
                           acts[:-1] + [L.ACT["linear"]], x3=self.x3)
        self.gh = GroupedHead(self.head.weight, self.head.bias, A, N, need_bwd, x3=self.x3)
        self._staged = None
        self._mean_stale = False

    def _versions(self):
        return tuple((p._version, getattr(p, "_rg_version", 0), p.data_ptr()) for l in self.lin for p in (l.weight, l.bias))

    def stage(self):
        ver = self._versions()
        if ver == self._staged:
            return
        ops.wide_head_mean(self.head.weight.detach(), self.head.bias.detach(), self.A, self.N, self.wbar, self.bbar)
        self.st.stage_weights(need_transposed=self.need_bwd, force=True)
        self.gh.stage()
        self._staged = ver
        self._mean_stale = False

    def fragments_updated(self, mean_now: bool):
        """after the one-launch update (rg_mlp_update_fused with this network's fragment buffers as destinations): trunk
        and grouped-head fragments are those of the new parameters; the per-action mean layer (wbar, bbar and its forward
        fragments) follows in one more launch — now, or when somebody needs it (`ensure_mean`)"""
        self._staged = self._versions()
        self._mean_stale = True
        if mean_now:
            self.ensure_mean()

    def ensure_mean(self):
        if self._mean_stale:
            ops.wide_head_mean(self.head.weight.detach(), self.head.bias.detach(), self.A, self.N, self.wbar, self.bbar,
                               wfrag_fwd=self.st._wf[self.st.L - 1], x3=self.x3)
            self._mean_stale = False


class GroupedQR:
    def __init__(self, trainer):
        self.tr = trainer
        self.A, self.N = trainer.num_actions, trainer.num_atoms
        # PREC_BF16X3: the same engine on split-bf16 operands (three MFMAs per product, every fragment buffer two planes) —
        # the mode held to north_star's 1e-4 on the quantiles; the loss head runs in fp32 / fp64 either way
        self.x3 = trainer.q_network.fc.precision == L.PREC_BF16X3
        self.online = _Net(trainer.q_network, self.A, self.N, need_bwd=True, x3=self.x3)
        self.target = _Net(trainer.q_network_target, self.A, self.N, need_bwd=False, x3=self.x3)
        self._B = -1
        self._side = None
        self.two_streams = os.environ.get("RG_QR_STREAMS", "1") != "0"  # the forward's two halves on two streams
        # the native step's loss sum rides in the reduce launch of the trunk's weight gradient (round 6; 0 = on the weight gradients'
        # side stream under the backward launch, rounds 4-5: one more fork of the main stream — a ~7 us marker gap — and a launch
        # that waited 60 us for CUs in front of the side stream's weight gradient: same box C3 0.8095 / 0.8002 / 0.8094 ->
        # 0.7752 / 0.7747 / 0.7756 ms, split-bf16 1.4777 / 1.4643 -> 1.4483 / 1.4481)
        self.loss_in_reduce = os.environ.get("RG_QR_LOSS_IN_REDUCE", "1") != "0"
        self.wgrad_streams = os.environ.get("RG_QR_WGRAD_STREAMS", "1") != "0"  # the backward's two weight-gradient launches
        self.dense = os.environ.get("RG_QR_DENSE", "1") != "0"  # grouped spaces without per-group padding (GroupedSpace)

    def after_fused_update(self):
        """DQNTrainer._fused_update ran Adam + soft update + re-staging of both networks' trunk and grouped-head
        fragments in one launch; the mean layer of the network that selects a* follows (the other one's on demand)"""
        tr = self.tr
        sel_online = (not tr.maxq_learning) or tr.double_q_learning
        self.online.fragments_updated(mean_now=True)  # a* under double-Q, and all_q_values()
        self.target.fragments_updated(mean_now=not sel_online)

    @staticmethod
    def eligible(trainer) -> bool:
        try:
            fc = trainer.q_network.fc
            if not hasattr(fc, "dnn"):  # a composite (dueling) network: three stacks, not one trunk + head
                return False
            if not fc.is_plain():  # LayerNorm / batch-norm / dropout / residual layers: per-layer path
                return False
            lin = fc.linears()
            names = fc.activation_names
        except AttributeError:
            return False
        if fc.precision not in (L.PREC_BF16, L.PREC_BF16X3) or len(lin) < 3 or getattr(trainer, "_cpe", None) is not None:
            return False
        if getattr(trainer.q_network_target.fc, "precision", fc.precision) != fc.precision:
            return False
        H = lin[0].weight.shape[0]
        A, N = trainer.num_actions, trainer.num_atoms
        pitch = 264 if H == 256 else 520
        return (H in (256, 512) and all(l.weight.shape[0] == H for l in lin[:-1]) and lin[0].weight.shape[1] <= 512
                and lin[-1].weight.shape[0] == A * N and all(n in ("relu", "leaky_relu") for n in names[:-1])
                and names[-1] == "linear" and (N + 31) // 32 * 32 + 8 <= pitch and N <= 256 and A <= 128
                and len(lin) <= L.MLP_MAX_LAYERS)

    def _ws(self, B, dev):
        if self._B == B:
            return
        A, N, H = self.A, self.N, self.online.H
        f32 = dict(dtype=torch.float32, device=dev)
        self.sp_next, self.sp_cur = GroupedSpace(B, A, dev, self.dense), GroupedSpace(B, A, dev, self.dense)
        R = self.sp_cur.rows
        ldz = (N + 7) // 8 * 8
        self.key_next = torch.empty(B, dtype=torch.int32, device=dev)
        self.key_cur = torch.empty(B, dtype=torch.int32, device=dev)
        self.qbar_next = torch.empty(B, A, **f32)
        self.qbar_cur = torch.empty(B, A, **f32)
        self.zt = torch.zeros(B, ldz, **f32)
        self.z = torch.empty(R, ldz, **f32)
        self.dz = torch.empty(R, ldz, **f32)
        self.loss_partials = torch.empty(R, **f32)
        # batch splits of a group's head weight gradient: A groups x 2 k-groups x splits workgroups, run beside the trunk's
        # weight gradient on the other stream.  Round 4, same box, C3 step in ms (bf16 / split-bf16): 16 splits 0.936 / 1.566,
        # 8 (rounds 2-4) 0.916 / 1.553, **4: 0.896 / 1.542**, 2: 0.902 / 1.564 — half the partial slabs (26 MB) and half of
        # their reduce launch against workgroups twice as long (`profiles/scripts/gpu_batch21.sh`)
        self.splits = int(os.environ.get("RG_QR_HEAD_SPLITS", "4"))
        nb = L.lib().rg_group_head_wgrad_workspace_bytes(A, N, H, self.splits)
        self.wg_ws = torch.empty(nb // 4, **f32)
        self._B = B

    def forward(self, b) -> torch.Tensor:
        tr = self.tr
        state, next_state = tr._net_in(b.state.float_features), tr._net_in(b.next_state.float_features)
        L.require_cuda(state, "training_batch.state")
        B, dev = state.shape[0], state.device
        self._ws(B, dev)
        on, tg = self.online, self.target
        on.stage()
        tg.stage()
        # The two halves of the forward are independent until the loss — (1) a* and the target quantiles of next_state,
        # (2) the current quantiles of the logged action — and each ends in a launch of B / 128 workgroups (rounds 2-3:
        # B / 128 + ~A / 2, two full rounds of the 256 CUs plus a sliver).  On two streams their tails fill each other.
        main = torch.cuda.current_stream() if state.is_cuda and self.two_streams else None
        if main is not None:
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                self._forward_current(b, state)
        # a*: the next action whose target quantiles form the Bellman target
        if tr.maxq_learning:
            sel = on if tr.double_q_learning else tg
            sel.ensure_mean()
            sel.st.forward(next_state, self.qbar_next, save=False)
            sp2 = self.sp_next.build_selected(self.qbar_next, tr._f32c(b.possible_next_actions_mask), True, self.key_next)
        else:  # SARSA: the logged next action (qrdqn_trainer.py:139-141); terminal rows carry none
            sp2 = self.sp_next.build_selected(None, tr._f32c(b.next_action), False, self.key_next)
        if not tr.maxq_learning:  # rows without a next action (SARSA: terminal) keep zero quantiles; the
            self.zt.zero_()       # masked arg max gives every row one, and the scatter then writes every row
        fused_forward_grouped(tg.st, tg.gh, next_state, sp2, self.zt, scatter=True, save=False)
        if main is not None:
            main.wait_stream(self._side)
        else:
            self._forward_current(b, state)
        sp1 = self.sp_cur
        self._state = state
        gamma_exp = None
        if tr.use_seq_num_diff_as_time_diff:
            gamma_exp = tr._f32c(b.time_diff).reshape(-1)
        if tr.multi_steps is not None:
            gamma_exp = tr._f32c(b.step).reshape(-1)
        boosts = tr.reward_boosts.reshape(-1).to(dev) if tr._has_reward_boost else None
        if tr.quantiles.device != dev:
            tr.quantiles = tr.quantiles.to(dev)
        ops.qr_compact_head(self.z, self.zt, sp1.rowmap, self.key_cur, tr._f32c(b.reward).reshape(-1), boosts,
                            tr._f32c(b.not_terminal).reshape(-1), tr.gamma, gamma_exp, tr.quantiles.reshape(-1), B, self.N,
                            self.dz, self.loss_partials)
        # the rows' loss terms (padding rows: 0) summed in fixed order.  Nothing on the device waits for the loss, so in the native
        # step the sum leaves the critical path: in the reduce launch of the trunk's weight gradient (loss_in_reduce, the default)
        # — or on the weight gradients' side stream, under the backward launch (joined where fused_backward_grouped joins it).
        # (Rounds 2-3: per-tile sums + their sum, two launch-bound launches = 13 us between the loss head and the backward.)
        if getattr(tr, "_loss_tail_wanted", False) and state.is_cuda and self.wgrad_streams and not self.loss_in_reduce:
            from .engine import side_stream

            side = side_stream(dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ops.reduce_sum(self.loss_partials, self.loss_partials.numel(), 1.0, tr._loss)
                # whoever hands tr._loss out waits for THIS (train_step_native's finally): the join inside
                # fused_backward_grouped only happens when that backward runs, with two_streams on
                tr._loss_side_event = side.record_event()
        elif getattr(tr, "_loss_tail_wanted", False):
            tr._loss_tail = (self.loss_partials, 1.0, tr._loss)
        else:
            ops.reduce_sum(self.loss_partials, self.loss_partials.numel(), 1.0, tr._loss)
        tr._dq = self.dz
        self._all_q = None
        return tr._loss

    def _forward_current(self, b, state):
        """current quantiles of the logged action (grouped space of the logged action; saved for the backward)"""
        tr, on = self.tr, self.online
        sp1 = self.sp_cur.build_selected(None, tr._f32c(b.action), False, self.key_cur)
        fused_forward_grouped(on.st, on.gh, state, sp1, self.z, scatter=False, save=True)

    def all_q_values(self) -> torch.Tensor:
        """q_network(state).mean(dim=2) [B, A] (the trainer's logged `all_q_values`): one more forward with the
        per-action mean layer, run only when somebody asks (reporters, CPE)"""
        if self._all_q is None:
            self.online.ensure_mean()
            self.online.st.forward(self._state, self.qbar_cur, save=False)
            self._all_q = self.qbar_cur
        return self._all_q

    # the trainer's `_qs.backward(dq, xt, dw, db)` contract
    def backward(self, dq, xt, dw, db, tail_sum=None, **_):
        fused_backward_grouped(self.online.st, self.online.gh, self.sp_cur, dq, dw, db, self.wg_ws, self.splits,
                               two_streams=self.wgrad_streams, tail_sum=tail_sum)
