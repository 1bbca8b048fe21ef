"""QR-DQN step with the wide output layer as a GROUPED layer (csrc/qr_grouped.hip): what
reagent/training/qrdqn_trainer.py:108-160 computes, without ever writing the [B, A * N] logits.

    q_next     = sel_net(next_state).mean(dim=2)         one fused forward whose output layer is the per-action
                                                         MEAN of the wide layer's rows (rg_wide_head_mean)
    a*         = arg max with the possible-actions mask   rg_qr_select_action                    (:125-135, :210-214)
    grouped space of a*: rows sorted by a*, each action padded to whole 128-row tiles (index bookkeeping, torch)
    zt[b, :]   = target_net(next_state)[b, a*, :]         fused trunk in grouped space + rg_group_head_forward (:137-141)
    grouped space of the logged action
    z[r, :]    = q_net(state)[b, logged a, :]              fused trunk (saving) + rg_group_head_forward  (:143-146)
    loss, dz   = quantile Huber on [rows, N]               rg_qr_compact_head                      (:148-160, :217-218)
    backward   = rg_group_head_dgrad -> trunk backward (rg_mlp_backward_fused on the trunk) + rg_mlp_wgrad_fused,
                 rg_group_head_wgrad for the wide layer
Arithmetic is that of the bf16 fused stack (bf16 operands, fp32 accumulation); the only algebraic rewrite is
mean_n(h . W[a, n] + b[a, n]) = h . mean_n W[a, n] + mean_n b[a, n].  A transition whose logged action row is all
zero contributes nothing (the reference would regress C = 0 for it; one-hot actions are the trainer's contract,
dqn_trainer_base.py `_check_input`).
"""
from typing import Optional

import torch

from . import _lib as L
from . import ops
from .engine import FusedMLP, fused_backward_trunk

TILE = 128


class GroupedSpace:
    """rows of a batch sorted by an int32 key in [0, G] (G = "no group": dropped) and padded per key to whole tiles;
    static shapes throughout (no host synchronisation): rowmap [TILE * n_tiles], tile_key [n_tiles], tile_begin [G + 1]"""

    def __init__(self, B: int, G: int, device):
        self.B, self.G = B, G
        self.n_tiles = (B + TILE - 1) // TILE + G
        self.rows = self.n_tiles * TILE
        self._ar_b = torch.arange(B, device=device)
        self._ar_t = torch.arange(self.n_tiles, device=device)
        self._zero = torch.zeros(1, dtype=torch.int64, device=device)
        self._ones = torch.ones(B, dtype=torch.int64, device=device)
        self.rowmap = self.tile_key = self.tile_begin = None

    def build(self, key: torch.Tensor):
        B, G = self.B, self.G
        k = key.long()
        counts = torch.zeros(G + 1, dtype=torch.int64, device=key.device).scatter_add_(0, k, self._ones)  # (bincount syncs)
        tiles = (counts[:G] + (TILE - 1)) // TILE
        tile_begin = torch.cat((self._zero, torch.cumsum(tiles, 0)))          # [G + 1]
        ks, order = torch.sort(k, stable=True)                                 # deterministic: batch order inside a group
        group_start = torch.cumsum(counts, 0) - counts                         # first sorted position of each key
        rank = self._ar_b - group_start[ks]
        grouped = ks < G
        tb_ext = torch.cat((tile_begin[:G], self._zero))
        dest = torch.where(grouped, tb_ext[ks] * TILE + rank, torch.full_like(rank, self.rows))
        rowmap = torch.full((self.rows + 1,), -1, dtype=torch.int32, device=key.device)
        rowmap[dest] = torch.where(grouped, order, torch.full_like(order, -1)).to(torch.int32)
        self.rowmap = rowmap[: self.rows]
        g_of_tile = torch.searchsorted(tile_begin[1:].contiguous(), self._ar_t, right=True)
        self.tile_key = torch.where(self._ar_t < tile_begin[G], g_of_tile, torch.full_like(g_of_tile, -1)).to(torch.int32)
        self.tile_begin = tile_begin.to(torch.int32)
        return self


class _Net:
    """one Q-network seen as trunk + grouped wide layer"""

    def __init__(self, net, A: int, N: int, need_bwd: bool):
        lin = net.fc.linears()
        self.lin, self.head = lin, lin[-1]
        self.A, self.N, self.need_bwd = A, N, need_bwd
        H = self.head.weight.shape[1]
        dev = self.head.weight.device
        self.H = H
        self.wbar = torch.zeros(A, H, device=dev)
        self.bbar = torch.zeros(A, device=dev)
        acts = [L.ACT[a] for a in net.fc.activation_names]
        self.st = FusedMLP([l.weight for l in lin[:-1]] + [self.wbar], [l.bias for l in lin[:-1]] + [self.bbar],
                           acts[:-1] + [L.ACT["linear"]])
        self.wf = torch.empty(A * ops.group_wfrag_elems(N, H, False), dtype=torch.bfloat16, device=dev)
        self.wb = torch.empty(A * ops.group_wfrag_elems(N, H, True), dtype=torch.bfloat16, device=dev) if need_bwd else None
        self._staged = None

    def stage(self):
        ver = tuple((p._version, getattr(p, "_rg_version", 0), p.data_ptr()) for l in self.lin for p in (l.weight, l.bias))
        if ver == self._staged:
            return
        ops.wide_head_mean(self.head.weight.detach(), self.head.bias.detach(), self.A, self.N, self.wbar, self.bbar)
        self.st.stage_weights(need_transposed=self.need_bwd, force=True)
        ops.group_weights_stage(self.head.weight.detach(), self.A, self.N, self.wf, self.wb)
        self._staged = ver

    def h_frag(self):
        return self.st._ws["act_frag"][self.st.L - 1]


class GroupedQR:
    def __init__(self, trainer):
        self.tr = trainer
        self.A, self.N = trainer.num_actions, trainer.num_atoms
        self.online = _Net(trainer.q_network, self.A, self.N, need_bwd=True)
        self.target = _Net(trainer.q_network_target, self.A, self.N, need_bwd=False)
        self._B = -1

    @staticmethod
    def eligible(trainer) -> bool:
        try:
            fc = trainer.q_network.fc
            lin = fc.linears()
            names = fc.activation_names
        except AttributeError:
            return False
        if fc.precision != L.PREC_BF16 or len(lin) < 3 or getattr(trainer, "_cpe", None) is not None:
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
        self.sp_next, self.sp_cur = GroupedSpace(B, A, dev), GroupedSpace(B, A, dev)
        R = self.sp_cur.rows
        ldz = (N + 7) // 8 * 8
        self.key_next = torch.empty(B, dtype=torch.int32, device=dev)
        self.key_cur = torch.empty(B, dtype=torch.int32, device=dev)
        self.qbar_next = torch.empty(B, A, **f32)
        self.qbar_t = torch.empty(R, A, **f32)
        self.qbar_cur = torch.empty(R, A, **f32)
        self.zt = torch.zeros(B, ldz, **f32)
        self.z = torch.empty(R, ldz, **f32)
        self.dz = torch.empty(R, ldz, **f32)
        self.dz3 = torch.empty(R, H, **f32)
        NgP = (N + 31) // 32 * 32
        self.dzw_frag = torch.empty(R * NgP, dtype=torch.bfloat16, device=dev)
        self.db_part = torch.empty(self.sp_cur.n_tiles * NgP, **f32)
        self.loss_partials = torch.empty(R, **f32)
        self.tile_losses = torch.empty(self.sp_cur.n_tiles, **f32)
        self.splits = 8
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
        # a*: the next action whose target quantiles form the Bellman target
        if tr.maxq_learning:
            sel = on if tr.double_q_learning else tg
            sel.st.forward(next_state, self.qbar_next, save=False)
            ops.qr_select_action(self.qbar_next, tr._f32c(b.possible_next_actions_mask), True, self.key_next)
        else:  # SARSA: the logged next action (qrdqn_trainer.py:139-141); terminal rows carry none
            ops.qr_select_action(None, tr._f32c(b.next_action), False, self.key_next)
        sp2 = self.sp_next.build(self.key_next)
        tg.st.forward(next_state, self.qbar_t, save=True, rowmap=sp2.rowmap)
        self.zt.zero_()
        ops.group_head_forward(tg.h_frag(), sp2.rowmap, sp2.tile_key, tg.wf, tg.head.bias.detach(), self.N, tg.H, True, self.zt)
        # current quantiles of the logged action
        ops.qr_select_action(None, tr._f32c(b.action), False, self.key_cur)
        sp1 = self.sp_cur.build(self.key_cur)
        on.st.forward(state, self.qbar_cur, save=True, rowmap=sp1.rowmap)
        ops.group_head_forward(on.h_frag(), sp1.rowmap, sp1.tile_key, on.wf, on.head.bias.detach(), self.N, on.H, False, self.z)
        gamma_exp = None
        if tr.use_seq_num_diff_as_time_diff:
            gamma_exp = tr._f32c(b.time_diff).reshape(-1)
        if tr.multi_steps is not None:
            gamma_exp = tr._f32c(b.step).reshape(-1)
        boosts = tr.reward_boosts.reshape(-1).to(dev) if tr._has_reward_boost else None
        if tr.quantiles.device != dev:
            tr.quantiles = tr.quantiles.to(dev)
        ops.qr_compact_head(self.z, self.zt, sp1.rowmap, sp1.tile_key, tr._f32c(b.reward).reshape(-1), boosts,
                            tr._f32c(b.not_terminal).reshape(-1), tr.gamma, gamma_exp, tr.quantiles.reshape(-1), B, self.N,
                            self.dz, self.loss_partials, self.tile_losses)
        ops.reduce_sum(self.tile_losses, self.tile_losses.numel(), 1.0, tr._loss)
        tr._dq = self.dz
        self._all_q = None
        return tr._loss

    def all_q_values(self) -> torch.Tensor:
        """mean over quantiles of q_network(state), batch order (the trainer's logged `all_q_values`)"""
        if self._all_q is None:
            rm = self.sp_cur.rowmap.long()
            out = torch.zeros(self._B + 1, self.A, device=rm.device)
            out[torch.where(rm >= 0, rm, torch.full_like(rm, self._B))] = self.qbar_cur
            self._all_q = out[: self._B]
        return self._all_q

    # the trainer's `_qs.backward(dq, xt, dw, db)` contract
    def backward(self, dq, xt, dw, db, **_):
        on, sp = self.online, self.sp_cur
        nl = len(dw) - 1
        leaky = on.st.acts[nl - 1] == L.ACT["leaky_relu"]
        ops.group_head_dgrad(dq, sp.tile_key, sp.tile_begin, self.A, on.wb, on.h_frag(), self.N, on.H, leaky, self.dz3,
                             self.dzw_frag, self.db_part, db[nl])
        ops.group_head_wgrad(self.dzw_frag, on.h_frag(), sp.tile_begin, self.A, self.N, on.H, self.splits, dw[nl], self.wg_ws)
        fused_backward_trunk(on.st, self.dz3, dw[:nl], db[:nl])
