"""Thin tensor-level wrappers over the C ABI (one function per entry point of reagent_hip.h).

All tensors must be on the GPU; leading dimensions are taken from ``stride(0)``.  Nothing here
falls back to torch math.  ``profile()`` optionally brackets every C-ABI call with HIP events on
the stream the kernels are enqueued on (bench.py's per-kernel roofline numbers come from that).
"""
import contextlib
import ctypes
import os
from typing import List, Optional, Tuple

import torch

from . import _lib as L

BF16 = torch.bfloat16
F32 = torch.float32


def compute_dtype(precision: int) -> torch.dtype:
    return F32 if precision == L.PREC_F32 else BF16


def dt_code(dtype: torch.dtype) -> int:
    if dtype == F32:
        return L.DT_F32
    if dtype == BF16:
        return L.DT_BF16
    raise L.ReagentHipError(f"unsupported dtype {dtype}")


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D tensor"
    return t.stride(0)


def _chk_dev(*ts):
    for t in ts:
        if t is not None:
            L.require_cuda(t)


# ---- optional per-call timing -----------------------------------------------------------------
class CallProfile:
    """(entry point, meta) -> HIP-event durations in ms, on torch's current stream."""

    def __init__(self):
        self.records: List[Tuple[str, dict, torch.cuda.Event, torch.cuda.Event]] = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, meta, s, e in self.records:
            key = (name, tuple(sorted(meta.items())))
            d = out.setdefault(key, {"name": name, "meta": meta, "calls": 0, "ms": 0.0})
            d["calls"] += 1
            d["ms"] += s.elapsed_time(e)
        return sorted(out.values(), key=lambda d: -d["ms"])


_prof: Optional[CallProfile] = None


@contextlib.contextmanager
def profile():
    global _prof
    prev, _prof = _prof, CallProfile()
    try:
        yield _prof
    finally:
        _prof = prev


def _run(name: str, meta: dict, call):
    if _prof is None:
        L.check(call(), name)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = call()
    e.record()
    L.check(rc, name)
    _prof.records.append((name, meta, s, e))


def profiling() -> bool:
    return _prof is not None


@contextlib.contextmanager
def profile_span(name: str, meta: dict):
    """events around a region that is not one C-ABI call (the RCCL all-reduce of the gradient slab); no-op unless
    a profile() pass is active"""
    if _prof is None:
        yield
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    yield
    e.record()
    _prof.records.append((name, meta, s, e))


# ---- FullyConnected ---------------------------------------------------------------------------
def fc_forward(x, w, bias, act: int, precision: int, y=None, y32=None, yt=None):
    """y = act(x @ w.T + bias); any of y (compute type), y32 (fp32), yt (transposed) may be given."""
    _chk_dev(x, w, bias, y, y32, yt)
    batch, in_f = x.shape
    out_f = w.shape[0]
    ldy = _ld(y) if y is not None else (_ld(y32) if y32 is not None else out_f)
    if y is not None and y32 is not None:
        assert _ld(y) == _ld(y32)
    _run("rg_fc_forward", dict(M=batch, N=out_f, K=in_f, prec=precision),
         lambda: L.lib().rg_fc_forward(L.ptr(x), _ld(x), L.ptr(w), _ld(w), L.ptr(bias), L.ptr(y), L.ptr(y32),
                                       ldy, L.ptr(yt), _ld(yt) if yt is not None else 0, batch, out_f, in_f,
                                       act, precision, L.stream_ptr()))


def fc_dgrad(dz, wt, ht, act_below: int, precision: int, dx=None, dx32=None, dxt=None):
    _chk_dev(dz, wt, ht, dx, dx32, dxt)
    batch, out_f = dz.shape
    in_f = wt.shape[0]
    lddx = _ld(dx) if dx is not None else (_ld(dx32) if dx32 is not None else in_f)
    _run("rg_fc_dgrad", dict(M=batch, N=in_f, K=out_f, prec=precision),
         lambda: L.lib().rg_fc_dgrad(L.ptr(dz), _ld(dz), L.ptr(wt), _ld(wt), L.ptr(ht),
                                     _ld(ht) if ht is not None else 0, act_below, L.ptr(dx), L.ptr(dx32), lddx,
                                     L.ptr(dxt), _ld(dxt) if dxt is not None else 0, batch, in_f, out_f,
                                     precision, L.stream_ptr()))


def fc_wgrad_workspace_bytes(out_f, in_f, batch, precision) -> int:
    return int(L.lib().rg_fc_wgrad_workspace_bytes(out_f, in_f, batch, precision))


def fc_wgrad(dzt, xt, dw, db, workspace, precision: int):
    """dw[out,in] (contiguous fp32) = dz^T x, db[out] = sum_b dz; inputs are transposed copies."""
    _chk_dev(dzt, xt, dw, db, workspace)
    out_f, batch = dzt.shape
    in_f = xt.shape[0]
    assert dw.is_contiguous() and dw.dtype == F32
    _run("rg_fc_wgrad", dict(M=out_f, N=in_f, K=batch, prec=precision),
         lambda: L.lib().rg_fc_wgrad(L.ptr(dzt), _ld(dzt), L.ptr(xt), _ld(xt), L.ptr(dw), L.ptr(db),
                                     L.ptr(workspace), workspace.numel() * workspace.element_size(), out_f,
                                     in_f, batch, precision, L.stream_ptr()))


def act_backward(dy, y, act: int, dz):
    """dz = dy * act'(z) through the activation output y (all fp32 [rows, cols], unit column stride)"""
    _chk_dev(dy, y, dz)
    rows, cols = dy.shape
    assert dy.dtype == y.dtype == dz.dtype == F32 and dy.stride(1) == y.stride(1) == dz.stride(1) == 1
    _run("rg_act_backward", dict(rows=rows, cols=cols),
         lambda: L.lib().rg_act_backward(L.ptr(dy), _ld(dy), L.ptr(y), _ld(y), act, L.ptr(dz), _ld(dz), rows, cols,
                                         L.stream_ptr()))


def td3_target_action(next_actor, noise, noise_variance, noise_clip_range, lo, hi, out):
    """out (a [B, A] view, e.g. the action columns of the critic input) = smoothed target action"""
    _chk_dev(next_actor, noise, out)
    B, A = next_actor.shape
    assert noise.is_contiguous() and noise.shape == (B, A) and next_actor.stride(1) == 1 and out.stride(1) == 1
    _run("rg_td3_target_action", dict(B=B, A=A),
         lambda: L.lib().rg_td3_target_action(L.ptr(next_actor), _ld(next_actor), L.ptr(noise), float(noise_variance),
                                              float(noise_clip_range[0]), float(noise_clip_range[1]), float(lo), float(hi), L.ptr(out), _ld(out), B, A,
                                              L.stream_ptr()))


def transpose_cast(src, dst=None, dst_t=None):
    _chk_dev(src, dst, dst_t)
    rows, cols = src.shape
    out = dst if dst is not None else dst_t
    _run("rg_transpose_cast", dict(rows=rows, cols=cols),
         lambda: L.lib().rg_transpose_cast(L.ptr(src), dt_code(src.dtype), _ld(src), rows, cols, L.ptr(dst),
                                           _ld(dst) if dst is not None else 0, L.ptr(dst_t),
                                           _ld(dst_t) if dst_t is not None else 0, dt_code(out.dtype),
                                           L.stream_ptr()))


# ---- layer norm (use_layer_norm of FullyConnectedNetwork) ------------------------------------------------------
def layer_norm_forward(z32, gamma, beta, eps, act: int, y=None, y32=None, mean=None, rstd=None):
    _chk_dev(z32, gamma, beta, y, y32, mean, rstd)
    B, n = z32.shape
    _run("rg_layer_norm_forward", dict(B=B, n=n),
         lambda: L.lib().rg_layer_norm_forward(L.ptr(z32), _ld(z32), L.ptr(gamma), L.ptr(beta), float(eps), act, B, n,
                                               L.ptr(y), dt_code(y.dtype) if y is not None else 0,
                                               _ld(y) if y is not None else 0, L.ptr(y32),
                                               _ld(y32) if y32 is not None else 0, L.ptr(mean), L.ptr(rstd), L.stream_ptr()))


def layer_norm_backward(g32, z32, mean, rstd, gamma, dgamma, dbeta, workspace, dz=None, dz32=None):
    _chk_dev(g32, z32, mean, rstd, gamma, dgamma, dbeta, workspace, dz, dz32)
    B, n = z32.shape
    assert dgamma.is_contiguous() and dbeta.is_contiguous()
    _run("rg_layer_norm_backward", dict(B=B, n=n),
         lambda: L.lib().rg_layer_norm_backward(L.ptr(g32), _ld(g32), L.ptr(z32), _ld(z32), L.ptr(mean), L.ptr(rstd),
                                                L.ptr(gamma), B, n, L.ptr(dz), dt_code(dz.dtype) if dz is not None else 0,
                                                _ld(dz) if dz is not None else 0, L.ptr(dz32),
                                                _ld(dz32) if dz32 is not None else 0, L.ptr(dgamma), L.ptr(dbeta),
                                                L.ptr(workspace), workspace.numel() * workspace.element_size(),
                                                L.stream_ptr()))



# ---- batch norm / dropout (use_batch_norm, dropout_ratio of FullyConnectedNetwork; fcopts.hip) ------------------
def batch_norm_workspace(batch: int, n: int, device) -> torch.Tensor:
    nbytes = L.lib().rg_batch_norm_workspace_bytes(batch, n)
    return torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=device)


def batch_norm_forward(x32, gamma, beta, running_mean, running_var, training: bool, momentum, eps, y32, save_mean=None,
                       save_rstd=None, workspace=None, stat_updates: int = 1):
    _chk_dev(x32, gamma, beta, running_mean, running_var, y32, save_mean, save_rstd, workspace)
    B, n = x32.shape
    assert x32.dtype == y32.dtype == F32 and x32.stride(1) == y32.stride(1) == 1
    _run("rg_batch_norm_forward", dict(B=B, n=n),
         lambda: L.lib().rg_batch_norm_forward(L.ptr(x32), _ld(x32), L.ptr(gamma), L.ptr(beta), L.ptr(running_mean),
                                               L.ptr(running_var), int(bool(training)), int(stat_updates), float(momentum),
                                               float(eps), B, n,
                                               L.ptr(y32), _ld(y32), L.ptr(save_mean), L.ptr(save_rstd), L.ptr(workspace),
                                               workspace.numel() * workspace.element_size() if workspace is not None else 0,
                                               L.stream_ptr()))


def batch_norm_backward(g32, x32, gamma, mean, rstd, running_var, training: bool, eps, workspace, dx32=None, dgamma=None,
                        dbeta=None):
    _chk_dev(g32, x32, gamma, mean, rstd, running_var, workspace, dx32, dgamma, dbeta)
    B, n = x32.shape
    assert g32.dtype == x32.dtype == F32 and g32.stride(1) == x32.stride(1) == 1
    _run("rg_batch_norm_backward", dict(B=B, n=n),
         lambda: L.lib().rg_batch_norm_backward(L.ptr(g32), _ld(g32), L.ptr(x32), _ld(x32), L.ptr(gamma), L.ptr(mean),
                                                L.ptr(rstd), L.ptr(running_var), int(bool(training)), float(eps), B, n,
                                                L.ptr(dx32), _ld(dx32) if dx32 is not None else 0, L.ptr(dgamma),
                                                L.ptr(dbeta), L.ptr(workspace),
                                                workspace.numel() * workspace.element_size(), L.stream_ptr()))


def dropout(x32, p: float, keep, y32, seed: int = 0, offset: int = 0, generate: bool = True):
    """y = x * keep / (1 - p); generate: draw `keep` (uint8 [B * n]) from (seed, offset), else read it (backward)"""
    _chk_dev(x32, keep, y32)
    B, n = x32.shape
    assert x32.dtype == y32.dtype == F32 and keep.dtype == torch.uint8 and keep.numel() >= B * n and keep.is_contiguous()
    _run("rg_dropout", dict(B=B, n=n),
         lambda: L.lib().rg_dropout(L.ptr(x32), _ld(x32), B, n, float(p), int(bool(generate)), int(seed) & (2**64 - 1),
                                    int(offset) & (2**64 - 1), L.ptr(keep), L.ptr(y32), _ld(y32), L.stream_ptr()))

# ---- replay -----------------------------------------------------------------------------------
def replay_nstep(indices, terminal_u8, reward, decays, capacity, horizon, steps, next_indices,
                 out_terminal, out_reward):
    _chk_dev(indices, terminal_u8, reward, decays, steps, next_indices, out_terminal, out_reward)
    _run("rg_replay_nstep", dict(B=indices.numel()),
         lambda: L.lib().rg_replay_nstep(L.ptr(indices), L.ptr(terminal_u8), L.ptr(reward), L.ptr(decays),
                                         capacity, horizon, indices.numel(), L.ptr(steps),
                                         L.ptr(next_indices), L.ptr(out_terminal), L.ptr(out_reward),
                                         L.stream_ptr()))


def replay_gather(cols, capacity: int, stack: int, batch: int):
    """cols: list of (src [C, ...], dst, indices[int64 B][, norm]) tensors; one launch per <=16 columns.
    norm = (col_table_dev, quantiles_dev): normalize-on-gather epilogue (dst dtype fp32 or bf16)."""
    for i in range(0, len(cols), L.MAX_GATHER_COLS):
        chunk = cols[i : i + L.MAX_GATHER_COLS]
        arr = (L.GatherCol * len(chunk))()
        nbytes = 0
        for j, item in enumerate(chunk):
            src, dst, idx = item[:3]
            norm = item[3] if len(item) > 3 else None
            _chk_dev(src, dst, idx)
            assert src.is_contiguous() and dst.is_contiguous() and idx.dtype == torch.int64
            row_elems = 1
            for s in src.shape[1:]:
                row_elems *= s
            arr[j].src = src.data_ptr()
            arr[j].dst = dst.data_ptr()
            arr[j].indices = idx.data_ptr()
            arr[j].row_elems = row_elems
            arr[j].elem_bytes = src.element_size()
            if norm is not None:
                assert src.dtype == F32 and stack == 1
                arr[j].norm = norm[0].data_ptr()
                arr[j].norm_quantiles = norm[1].data_ptr()
                arr[j].out_dtype = dt_code(dst.dtype)
            nbytes += row_elems * (src.element_size() + dst.element_size()) * stack + 8
        _run("rg_replay_gather", dict(B=batch, bytes_per_row=nbytes),
             lambda: L.lib().rg_replay_gather(arr, len(chunk), capacity, stack, batch, L.stream_ptr()))


def sumtree_set(tree, depth: int, capacity: int, indices, values, claim=None):
    """SumTree.set for every (index, value) pair, in order (sum_tree.py:159-189)."""
    _chk_dev(tree, indices, values, claim)
    assert tree.dtype == torch.float64 and values.dtype == torch.float64 and indices.dtype == torch.int64
    _run("rg_sumtree_set", dict(n=indices.numel()),
         lambda: L.lib().rg_sumtree_set(L.ptr(tree), depth, capacity, L.ptr(indices), L.ptr(values), indices.numel(),
                                        L.ptr(claim), L.stream_ptr()))


def sumtree_sample(tree, depth: int, query01, out_indices):
    """SumTree.sample for a batch of query values in [0, 1] (sum_tree.py:97-131)."""
    _chk_dev(tree, query01, out_indices)
    assert tree.dtype == torch.float64 and query01.dtype == torch.float64 and out_indices.dtype == torch.int64
    _run("rg_sumtree_sample", dict(n=query01.numel()),
         lambda: L.lib().rg_sumtree_sample(L.ptr(tree), depth, L.ptr(query01), query01.numel(), L.ptr(out_indices),
                                           L.stream_ptr()))


def sumtree_get(tree, depth: int, capacity: int, indices, out32=None, out64=None):
    _chk_dev(tree, indices, out32, out64)
    assert tree.dtype == torch.float64 and indices.dtype == torch.int64
    _run("rg_sumtree_get", dict(n=indices.numel()),
         lambda: L.lib().rg_sumtree_get(L.ptr(tree), depth, capacity, L.ptr(indices), indices.numel(), L.ptr(out32),
                                        L.ptr(out64), L.stream_ptr()))


def make_dqn_input(action, next_action, terminal, log_prob, num_actions, action_1h, next_action_1h,
                   not_terminal, action_probability=None):
    _chk_dev(action, next_action, terminal, log_prob, action_1h, next_action_1h, not_terminal,
             action_probability)
    _run("rg_make_dqn_input", dict(B=action.numel()),
         lambda: L.lib().rg_make_dqn_input(L.ptr(action), L.ptr(next_action), L.ptr(terminal), L.ptr(log_prob),
                                           action.numel(), num_actions, L.ptr(action_1h), L.ptr(next_action_1h),
                                           L.ptr(not_terminal), L.ptr(action_probability), L.stream_ptr()))


def ragged_gather(ids, scores, lens, indices):
    """(offsets int32 [B], ids int64 [total], scores float32 [total] or None): the sampled rows of a padded-slot ragged
    column back to back (IDListMetadata / IDScoreListMetadata.sample_to_output).  One host read: the total length."""
    _chk_dev(ids, scores, lens, indices)
    B, dev = indices.numel(), ids.device
    offsets = torch.empty(B, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    _run("rg_ragged_offsets", dict(B=B),
         lambda: L.lib().rg_ragged_offsets(L.ptr(lens), L.ptr(indices), B, L.ptr(offsets), L.ptr(total), L.stream_ptr()))
    n = int(total.item())  # the output's length is data dependent, as in the reference
    ids_out = torch.empty(n, dtype=torch.int64, device=dev)
    scores_out = torch.empty(n, dtype=torch.float32, device=dev) if scores is not None else None
    if n:
        _run("rg_ragged_copy", dict(B=B),
             lambda: L.lib().rg_ragged_copy(L.ptr(ids), L.ptr(scores), ids.shape[1], L.ptr(lens), L.ptr(indices), L.ptr(offsets),
                                            B, L.ptr(ids_out), L.ptr(scores_out), L.stream_ptr()))
    return offsets, ids_out, scores_out


def make_policy_input(action, next_action, terminal, log_prob, ranges, action_out, next_action_out, not_terminal,
                      action_probability=None):
    """PolicyNetworkInputMaker's arithmetic in one launch; ranges [4, A] = prev_min, prev_max, new_min, new_max"""
    _chk_dev(action, next_action, terminal, log_prob, ranges, action_out, next_action_out, not_terminal, action_probability)
    B, A = action.shape
    assert action.stride(1) == 1 and next_action.stride(1) == 1 and action_out.is_contiguous() and next_action_out.is_contiguous()
    assert ranges.is_contiguous() and ranges.numel() == 4 * A and terminal.element_size() == 1
    _run("rg_make_policy_input", dict(B=B, A=A),
         lambda: L.lib().rg_make_policy_input(L.ptr(action), action.stride(0), L.ptr(next_action), next_action.stride(0),
                                              L.ptr(terminal), L.ptr(log_prob), L.ptr(ranges), B, A, L.ptr(action_out),
                                              L.ptr(next_action_out), L.ptr(not_terminal), L.ptr(action_probability),
                                              L.stream_ptr()))


def normalize_dense(x, presence_u8, cols_dev, n_out, quantiles, out):
    _chk_dev(x, presence_u8, cols_dev, quantiles, out)
    _run("rg_normalize_dense", dict(B=x.shape[0], n_out=n_out),
         lambda: L.lib().rg_normalize_dense(L.ptr(x), _ld(x), L.ptr(presence_u8),
                                            _ld(presence_u8) if presence_u8 is not None else 0,
                                            L.ptr(cols_dev), n_out, L.ptr(quantiles), L.ptr(out), _ld(out),
                                            x.shape[0], L.stream_ptr()))


def table_dqn_batch(table, indices, cols_dev, n_out, quantiles, out: dict):
    """out: name -> device tensor for every field of rg_dqn_batch_out (a missing name = NULL)"""
    _chk_dev(indices, cols_dev, quantiles, *out.values())
    assert indices.dtype == torch.int64 and indices.is_contiguous()
    o = L.DqnBatchOut()
    for name in L.BATCH_OUT_FIELDS:
        t = out.get(name)
        assert t is None or t.is_contiguous()
        setattr(o, name, t.data_ptr() if t is not None else None)
    o.state_dtype = dt_code(out["state"].dtype)
    assert out["next_state"].dtype == out["state"].dtype
    B = indices.numel()
    _run("rg_table_dqn_batch", dict(B=B, n_out=n_out, F=table.num_features, A=table.num_actions),
         lambda: L.lib().rg_table_dqn_batch(ctypes.byref(table.desc()), L.ptr(indices), B, L.ptr(cols_dev), n_out,
                                            L.ptr(quantiles), ctypes.byref(o), L.stream_ptr()))


class PooledIndices:
    """The indices of a REPLAYED step (runtime._GraphedLoop): row `cursor[0]` (device int64) of `pool` [rows, batch]; the
    sampler launch also counts the step in the device-resident Adam schedule `sched` (nullable), the step's one-launch update
    advances the cursor (rg_replay_dqn_batch_pooled / rg_mlp_update_desc.sched_pre_ticked, post_tick)."""

    def __init__(self, pool, cursor, sched=None):
        assert pool.dtype == torch.int64 and pool.dim() == 2 and pool.is_contiguous() and cursor.dtype == torch.int64
        self.pool, self.cursor, self.sched = pool, cursor, sched

    def numel(self):
        return self.pool.shape[1]


def replay_dqn_batch(view: "L.ReplayView", indices, cols_dev, quantiles, out: dict) -> bool:
    """one-launch sample + input maker (+ normalize); False = shape not supported by the fused kernel"""
    pooled = indices if isinstance(indices, PooledIndices) else None
    if pooled is not None:
        _chk_dev(pooled.pool, pooled.cursor, pooled.sched, cols_dev, quantiles, *out.values())
    else:
        _chk_dev(indices, cols_dev, quantiles, *out.values())
        assert indices.dtype == torch.int64 and indices.is_contiguous()
    o = L.DqnBatchOut()
    for name in L.BATCH_OUT_FIELDS:
        t = out.get(name)
        assert t is None or t.is_contiguous()
        setattr(o, name, t.data_ptr() if t is not None else None)
    o.state_dtype = dt_code(out["state"].dtype)
    B = indices.numel()
    rc = [0]

    def call():
        if pooled is not None:
            rc[0] = L.lib().rg_replay_dqn_batch_pooled(ctypes.byref(view), pooled.pool.data_ptr(), pooled.cursor.data_ptr(),
                                                       L.ptr(pooled.sched), B, L.ptr(cols_dev), L.ptr(quantiles),
                                                       ctypes.byref(o), L.stream_ptr())
        else:
            rc[0] = L.lib().rg_replay_dqn_batch(ctypes.byref(view), L.ptr(indices), B, L.ptr(cols_dev), L.ptr(quantiles),
                                                ctypes.byref(o), L.stream_ptr())
        return 0 if rc[0] == L.EUNSUPPORTED else rc[0]

    F_, A_, H_ = view.n_features, view.n_actions, view.update_horizon
    es = out["state"].element_size()
    # algorithmic bytes per transition: rows in (2 F fp32) and out (2 F), n-step window, action / log_prob /
    # index reads, masks in (if stored) and out, one-hots and scalars out
    nbytes = (2 * F_ * 4 + 2 * F_ * es + 5 * H_ + 2 * 8 + 4 + 8 + (2 * A_ * 4 if view.possible_actions_mask else 0)
              + 4 * A_ * 4 + 3 * 4)
    _run("rg_replay_dqn_batch", dict(B=B, bytes_per_row=nbytes), call)
    return rc[0] == 0


def replay_policy_batch(view: "L.PolicyReplayView", indices, cols_dev, quantiles, out: dict) -> bool:
    """one-launch sample + PolicyNetworkInputMaker (+ normalize) — rg_replay_policy_batch; False = shape not supported by the
    fused kernel (the caller takes rg_replay_nstep + rg_replay_gather + rg_make_policy_input)"""
    _chk_dev(indices, cols_dev, quantiles, *out.values())
    assert indices.dtype == torch.int64 and indices.is_contiguous()
    o = L.PolicyBatchOut()
    for name in L.POLICY_OUT_FIELDS:
        t = out.get(name)
        assert t is None or t.is_contiguous()
        setattr(o, name, t.data_ptr() if t is not None else None)
    o.state_dtype = dt_code(out["state"].dtype)
    B = indices.numel()
    rc = [0]

    def call():
        rc[0] = L.lib().rg_replay_policy_batch(ctypes.byref(view), L.ptr(indices), B, L.ptr(cols_dev), L.ptr(quantiles),
                                               ctypes.byref(o), L.stream_ptr())
        return 0 if rc[0] == L.EUNSUPPORTED else rc[0]

    F_, A_, H_ = view.n_features, view.action_dim, view.update_horizon
    es = out["state"].element_size()
    # algorithmic bytes per transition: state rows in (2 F fp32) and out (2 F), action rows in and out (4 A fp32), the n-step
    # window (terminal byte + reward per slot), index, log_prob, three scalars out
    nbytes = 2 * F_ * 4 + 2 * F_ * es + 4 * A_ * 4 + 5 * H_ + 8 + 4 + 3 * 4
    _run("rg_replay_policy_batch", dict(B=B, bytes_per_row=nbytes), call)
    return rc[0] == 0


def table_check_actions(table, indices) -> int:
    _chk_dev(indices)
    flag = torch.zeros(1, dtype=torch.int32, device=indices.device)
    _run("rg_table_check_actions", dict(B=indices.numel()),
         lambda: L.lib().rg_table_check_actions(ctypes.byref(table.desc()), L.ptr(indices), indices.numel(),
                                                L.ptr(flag), L.stream_ptr()))
    return int(flag.item())


# ---- heads ------------------------------------------------------------------------------------
def bcq_filter(imitator_logits, drop_threshold: float, mask):
    """mask *= (softmax(logits) / rowmax >= drop_threshold), in place"""
    _chk_dev(imitator_logits, mask)
    B, A = mask.shape
    assert imitator_logits.shape == (B, A) and imitator_logits.dtype == F32 and mask.dtype == F32
    assert imitator_logits.is_contiguous() and mask.is_contiguous()
    _run("rg_bcq_filter", dict(B=B, A=A),
         lambda: L.lib().rg_bcq_filter(L.ptr(imitator_logits), B, A, float(drop_threshold), L.ptr(mask), L.stream_ptr()))


def dqn_head_partials(batch: int) -> int:
    return int(L.lib().rg_dqn_head_partials(batch))


def dqn_head(q, qn_online, qn_target, action, next_mask, reward, reward_boosts, not_terminal, gamma,
             gamma_exponent, double_q, loss_type, dq, loss_partials, next_q=None, next_idx=None,
             q_sel=None):
    _chk_dev(q, qn_online, qn_target, action, next_mask, reward, reward_boosts, not_terminal,
             gamma_exponent, dq, loss_partials, next_q, next_idx, q_sel)
    batch, A = q.shape
    for t in (q, qn_online, qn_target, action, next_mask, dq):
        assert t.is_contiguous() and t.dtype == F32 and t.shape == (batch, A)
    for t in (reward, not_terminal, gamma_exponent):
        assert t is None or (t.is_contiguous() and t.dtype == F32 and t.numel() == batch)
    _run("rg_dqn_head", dict(B=batch, A=A),
         lambda: L.lib().rg_dqn_head(L.ptr(q), L.ptr(qn_online), L.ptr(qn_target), L.ptr(action),
                                     L.ptr(next_mask), L.ptr(reward), L.ptr(reward_boosts),
                                     L.ptr(not_terminal), float(gamma), L.ptr(gamma_exponent), batch, A,
                                     int(double_q), loss_type, L.ptr(dq), L.ptr(loss_partials), L.ptr(next_q),
                                     L.ptr(next_idx), L.ptr(q_sel), L.stream_ptr()))


def cpe_head(reward_est, q_cpe, q_cpe_tgt_next, next_scores, next_mask, action, reward, extra_metrics,
             not_terminal, gamma, gamma_exponent, temperature, num_metrics, loss_type, d_reward_est, d_q_cpe,
             reward_partials, cpe_partials, propensities_out=None):
    _chk_dev(reward_est, q_cpe, q_cpe_tgt_next, next_scores, next_mask, action, reward, extra_metrics,
             not_terminal, gamma_exponent, d_reward_est, d_q_cpe, reward_partials, cpe_partials, propensities_out)
    batch, A = action.shape
    M = num_metrics
    for t in (reward_est, q_cpe, q_cpe_tgt_next, d_reward_est, d_q_cpe):
        assert t.is_contiguous() and t.dtype == F32 and t.shape == (batch, M * A)
    for t in (next_scores, next_mask, action):
        assert t.is_contiguous() and t.dtype == F32 and t.shape == (batch, A)
    assert extra_metrics is None or (extra_metrics.is_contiguous() and extra_metrics.shape == (batch, M - 1))
    _run("rg_cpe_head", dict(B=batch, A=A, M=M),
         lambda: L.lib().rg_cpe_head(L.ptr(reward_est), L.ptr(q_cpe), L.ptr(q_cpe_tgt_next), L.ptr(next_scores),
                                     L.ptr(next_mask), L.ptr(action), L.ptr(reward), L.ptr(extra_metrics),
                                     L.ptr(not_terminal), float(gamma), L.ptr(gamma_exponent), float(temperature),
                                     batch, A, M, loss_type, L.ptr(d_reward_est), L.ptr(d_q_cpe),
                                     L.ptr(reward_partials), L.ptr(cpe_partials), L.ptr(propensities_out),
                                     L.stream_ptr()))


def c51_head(q, qn_online, qn_target, action, next_mask, reward, reward_boosts, not_terminal, gamma,
             gamma_exponent, support, qmin, qmax, num_atoms, maxq, dq, loss_partials, all_q=None):
    _chk_dev(q, qn_online, qn_target, action, next_mask, reward, reward_boosts, not_terminal, gamma_exponent,
             support, dq, loss_partials, all_q)
    batch, A = action.shape
    for t in (q, qn_target, dq):
        assert t.is_contiguous() and t.dtype == F32 and t.shape == (batch, A * num_atoms)
    _run("rg_c51_head", dict(B=batch, A=A, N=num_atoms),
         lambda: L.lib().rg_c51_head(L.ptr(q), L.ptr(qn_online), L.ptr(qn_target), L.ptr(action), L.ptr(next_mask),
                                     L.ptr(reward), L.ptr(reward_boosts), L.ptr(not_terminal), float(gamma),
                                     L.ptr(gamma_exponent), L.ptr(support), float(qmin), float(qmax), batch, A,
                                     num_atoms, int(maxq), L.ptr(dq), L.ptr(loss_partials), L.ptr(all_q),
                                     L.stream_ptr()))


def crr_partials(batch: int) -> int:
    return int(L.lib().rg_crr_partials(batch))


def crr_critic_head(q1, q2, q1_next_t, q2_next_t, next_logits, action, reward, reward_boosts, not_terminal, gamma,
                    target, dq1, dq2, partials1, partials2):
    _chk_dev(q1, q2, q1_next_t, q2_next_t, next_logits, action, reward, reward_boosts, not_terminal, target, dq1, dq2,
             partials1, partials2)
    batch, A = action.shape
    for t in (q1, q2, q1_next_t, q2_next_t, next_logits, action, dq1, dq2):
        assert t is None or (t.is_contiguous() and t.dtype == F32 and t.shape == (batch, A))
    _run("rg_crr_critic_head", dict(B=batch, A=A),
         lambda: L.lib().rg_crr_critic_head(L.ptr(q1), L.ptr(q2), L.ptr(q1_next_t), L.ptr(q2_next_t),
                                            L.ptr(next_logits), L.ptr(action), L.ptr(reward), L.ptr(reward_boosts),
                                            L.ptr(not_terminal), float(gamma), batch, A, L.ptr(target), L.ptr(dq1),
                                            L.ptr(dq2), L.ptr(partials1), L.ptr(partials2), L.stream_ptr()))


def crr_actor_head(q, logits, action, logged_prob, beta, max_weight, entropy_coeff, clip_limit, dlogits,
                   plain_partials, entropy_partials):
    _chk_dev(q, logits, action, logged_prob, dlogits, plain_partials, entropy_partials)
    batch, A = action.shape
    for t in (q, logits, action, dlogits):
        assert t.is_contiguous() and t.dtype == F32 and t.shape == (batch, A)
    _run("rg_crr_actor_head", dict(B=batch, A=A),
         lambda: L.lib().rg_crr_actor_head(L.ptr(q), L.ptr(logits), L.ptr(action), L.ptr(logged_prob), float(beta),
                                           float(max_weight), float(entropy_coeff), float(clip_limit), batch, A,
                                           L.ptr(dlogits), L.ptr(plain_partials), L.ptr(entropy_partials),
                                           L.stream_ptr()))


def qr_head(q, qn_online, qn_target, action, next_mask, reward, reward_boosts, not_terminal, gamma,
            gamma_exponent, quantiles, num_atoms, maxq, dq, loss_partials, all_q=None):
    _chk_dev(q, qn_online, qn_target, action, next_mask, reward, reward_boosts, not_terminal, gamma_exponent,
             quantiles, dq, loss_partials, all_q)
    batch, A = action.shape
    for t in (q, qn_online, qn_target, dq):
        assert t is None or (t.is_contiguous() and t.dtype == F32 and t.shape == (batch, A * num_atoms))
    _run("rg_qr_head", dict(B=batch, A=A, N=num_atoms),
         lambda: L.lib().rg_qr_head(L.ptr(q), L.ptr(qn_online), L.ptr(qn_target), L.ptr(action), L.ptr(next_mask),
                                    L.ptr(reward), L.ptr(reward_boosts), L.ptr(not_terminal), float(gamma),
                                    L.ptr(gamma_exponent), L.ptr(quantiles), batch, A, num_atoms, int(maxq),
                                    L.ptr(dq), L.ptr(loss_partials), L.ptr(all_q), L.stream_ptr()))


def reduce_sum(inp, n: int, scale: float, out):
    _chk_dev(inp, out)
    _run("rg_reduce_sum", dict(n=n), lambda: L.lib().rg_reduce_sum(L.ptr(inp), n, scale, L.ptr(out), L.stream_ptr()))


# ---- optimizer --------------------------------------------------------------------------------
def adam_step(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt,
              grad_scale=1.0, offset=0):
    """Flat fp32 slabs; [offset, offset+n) is updated."""
    _chk_dev(param, grad, exp_avg, exp_avg_sq)
    o = offset * 4
    _run("rg_adam_step", dict(n=n),
         lambda: L.lib().rg_adam_step(param.data_ptr() + o, grad.data_ptr() + o, exp_avg.data_ptr() + o,
                                      exp_avg_sq.data_ptr() + o, n, lr, beta1, beta2, eps, weight_decay, bc1,
                                      bc2_sqrt, grad_scale, L.stream_ptr()))


def adam_step_sched(param, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, weight_decay, sched, grad_scale=1.0,
                    offset=0):
    """rg_adam_step with lr and the bias corrections read from the device schedule `sched` (graph-safe)."""
    _chk_dev(param, grad, exp_avg, exp_avg_sq, sched)
    tick_fence(sched)
    o = offset * 4
    _run("rg_adam_step", dict(n=n),
         lambda: L.lib().rg_adam_step_sched(param.data_ptr() + o, grad.data_ptr() + o, exp_avg.data_ptr() + o,
                                            exp_avg_sq.data_ptr() + o, n, beta1, beta2, eps, weight_decay,
                                            grad_scale, sched.data_ptr(), L.stream_ptr()))


# ---- schedule ticks of a native step in ONE launch ---------------------------------------------------------
# A device-scheduled update is followed by the tick of ITS schedule (sched[0] += 1).  Nothing reads a schedule again before that
# optimizer's next step, so inside `deferred_ticks()` (the trainers' native steps, dqn_trainer.native_step) the ticks are collected
# and leave as one rg_sched_tick_many launch when the scope closes: SAC's four launch-bound ticks (15 us per step) become one.
# `tick_fence(sched)` — called by every launch that READS a schedule — flushes first if that schedule has a tick waiting.
MAX_TICKS = 8
_tick_scopes = []
_DEFER_TICKS = os.environ.get("RG_DEFER_TICKS", "1") != "0"  # 0: a tick launch per schedule, where it was (same-box A/B)


def _flush_ticks(pending):
    if not pending:
        return
    if len(pending) == 1:
        s = pending[0]
        _run("rg_sched_tick", {}, lambda: L.lib().rg_sched_tick(s.data_ptr(), L.stream_ptr()))
    else:
        arr = (ctypes.c_void_p * len(pending))(*[s.data_ptr() for s in pending])
        _run("rg_sched_tick", dict(n=len(pending)), lambda: L.lib().rg_sched_tick_many(arr, len(pending), L.stream_ptr()))
    del pending[:]


class deferred_ticks:
    def __enter__(self):
        _tick_scopes.append([])
        return self

    def __exit__(self, *exc):
        _flush_ticks(_tick_scopes.pop())
        return False


def tick_fence(sched):
    """before a launch that reads `sched`: its waiting tick (if any) must have been enqueued"""
    for pending in _tick_scopes:
        if any(s.data_ptr() == sched.data_ptr() for s in pending):
            _flush_ticks(pending)


def sched_tick(sched):
    _chk_dev(sched)
    if _tick_scopes and _DEFER_TICKS:
        pending = _tick_scopes[-1]
        if any(s.data_ptr() == sched.data_ptr() for s in pending) or len(pending) == MAX_TICKS:
            _flush_ticks(pending)
        pending.append(sched)
        return
    _run("rg_sched_tick", {}, lambda: L.lib().rg_sched_tick(sched.data_ptr(), L.stream_ptr()))


def soft_update(target, source, n, tau, t_off=0, s_off=0):
    _chk_dev(target, source)
    _run("rg_soft_update", dict(n=n),
         lambda: L.lib().rg_soft_update(target.data_ptr() + 4 * t_off, source.data_ptr() + 4 * s_off, n, tau,
                                        L.stream_ptr()))


# ---- SAC --------------------------------------------------------------------------------------
def gaussian_head_forward(loc_scale, noise, action, log_prob=None, squashed_mean=None):
    _chk_dev(loc_scale, noise, action, log_prob, squashed_mean)
    B, A = noise.shape
    assert noise.is_contiguous() and loc_scale.shape == (B, 2 * A)
    _run("rg_gaussian_head_forward", dict(B=B, A=A),
         lambda: L.lib().rg_gaussian_head_forward(L.ptr(loc_scale), _ld(loc_scale), L.ptr(noise), B, A,
                                                  L.ptr(action), _ld(action), L.ptr(log_prob),
                                                  L.ptr(squashed_mean), L.stream_ptr()))


def gaussian_log_prob(loc_scale, action, log_prob):
    _chk_dev(loc_scale, action, log_prob)
    B, A = action.shape
    _run("rg_gaussian_log_prob", dict(B=B, A=A),
         lambda: L.lib().rg_gaussian_log_prob(L.ptr(loc_scale), _ld(loc_scale), L.ptr(action), _ld(action), B, A,
                                              L.ptr(log_prob), L.stream_ptr()))


def gaussian_head_backward(loc_scale, noise, g_action, g_log_prob, d_loc_scale, kld_coef=None, kld_on_mean=False):
    _chk_dev(loc_scale, noise, g_action, g_log_prob, d_loc_scale, kld_coef)
    B, A = noise.shape
    _run("rg_gaussian_head_backward", dict(B=B, A=A),
         lambda: L.lib().rg_gaussian_head_backward_kld(L.ptr(loc_scale), _ld(loc_scale), L.ptr(noise),
                                                       L.ptr(g_action), _ld(g_action) if g_action is not None else 0,
                                                       L.ptr(g_log_prob), B, A, L.ptr(d_loc_scale),
                                                       _ld(d_loc_scale), L.ptr(kld_coef), int(kld_on_mean),
                                                       L.stream_ptr()))


def sac_kld(x, squash, emb_mean, emb_var, weight, coef, kld_terms, kld=None, loss_inout=None):
    """action-embedding KLD term (sac_trainer.py:282-306) over the columns of x [B, A]: gradient coefficients into
    coef [2A], the value into kld [1], weight * kld added to loss_inout [1]"""
    _chk_dev(x, emb_mean, emb_var, coef, kld_terms, kld, loss_inout)
    B, A = x.shape[0], emb_mean.numel()
    assert coef.numel() == 2 * A and kld_terms.numel() >= A and x.stride(1) == 1
    _run("rg_sac_kld", dict(B=B, A=A),
         lambda: L.lib().rg_sac_kld(L.ptr(x), x.stride(0), int(squash), B, A, L.ptr(emb_mean), L.ptr(emb_var), float(weight),
                                    L.ptr(coef), L.ptr(kld_terms), L.ptr(kld), L.ptr(loss_inout), L.stream_ptr()))


def sac_partials(batch: int) -> int:
    return int(L.lib().rg_sac_partials(batch))


def sac_critic_head(q1, q2, q1t, q2t, lp_next, reward, not_terminal, gamma, alpha, target, dq1, dq2, l1, l2):
    _chk_dev(q1, q2, q1t, q2t, lp_next, reward, not_terminal, alpha, target, dq1, dq2, l1, l2)
    B = q1.numel()
    _run("rg_sac_critic_head", dict(B=B),
         lambda: L.lib().rg_sac_critic_head(L.ptr(q1), L.ptr(q2), L.ptr(q1t), L.ptr(q2t), L.ptr(lp_next),
                                            L.ptr(reward), L.ptr(not_terminal), float(gamma), L.ptr(alpha), B,
                                            L.ptr(target), L.ptr(dq1), L.ptr(dq2), L.ptr(l1), L.ptr(l2),
                                            L.stream_ptr()))


def sac_actor_head(lp, q1a, q2a, alpha, target_entropy, g_lp, dq1a, dq2a, loss_part, ent_part, v_cur=None, crr_mode=0,
                   crr_p0=0.0, crr_clamp=0.0, backprop_log_prob=True):
    """crr_mode 0: alpha * clamp(log_prob) - min q; 1 / 2: -(clamp(log_prob) * CRR weight of (min q - v_cur)), indicator
    (threshold crr_p0) / exponent (beta crr_p0, clamp crr_clamp)"""
    _chk_dev(lp, q1a, q2a, alpha, g_lp, dq1a, dq2a, loss_part, ent_part, v_cur)
    B = lp.numel()
    _run("rg_sac_actor_head", dict(B=B),
         lambda: L.lib().rg_sac_actor_head(L.ptr(lp), L.ptr(q1a), L.ptr(q2a), L.ptr(alpha), float(target_entropy),
                                           B, L.ptr(v_cur), int(crr_mode), float(crr_p0), float(crr_clamp),
                                           int(bool(backprop_log_prob)), L.ptr(g_lp), L.ptr(dq1a), L.ptr(dq2a),
                                           L.ptr(loss_part), L.ptr(ent_part), L.stream_ptr()))


def sac_alpha_grad(ent_part, batch, log_alpha, grad, alpha_loss=None):
    _chk_dev(ent_part, log_alpha, grad, alpha_loss)
    _run("rg_sac_alpha_grad", dict(B=batch),
         lambda: L.lib().rg_sac_alpha_grad(L.ptr(ent_part), batch, L.ptr(log_alpha), L.ptr(grad),
                                           L.ptr(alpha_loss), L.stream_ptr()))


def adam_step_f64(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, bc1, bc2_sqrt, exp_out=None):
    _chk_dev(param, grad, exp_avg, exp_avg_sq, exp_out)
    _run("rg_adam_step_f64", dict(n=param.numel()),
         lambda: L.lib().rg_adam_step_f64(L.ptr(param), L.ptr(grad), L.ptr(exp_avg), L.ptr(exp_avg_sq),
                                          param.numel(), lr, beta1, beta2, eps, bc1, bc2_sqrt, L.ptr(exp_out),
                                          L.stream_ptr()))


def adam_step_f64_sched(param, grad, exp_avg, exp_avg_sq, beta1, beta2, eps, sched, exp_out=None):
    _chk_dev(param, grad, exp_avg, exp_avg_sq, exp_out, sched)
    tick_fence(sched)
    _run("rg_adam_step_f64", dict(n=param.numel()),
         lambda: L.lib().rg_adam_step_f64_sched(L.ptr(param), L.ptr(grad), L.ptr(exp_avg), L.ptr(exp_avg_sq),
                                                param.numel(), beta1, beta2, eps, sched.data_ptr(), L.ptr(exp_out),
                                                L.stream_ptr()))


def add_cols(a, b, out):
    _chk_dev(a, b, out)
    B, C = out.shape
    _run("rg_add_cols", dict(B=B, C=C),
         lambda: L.lib().rg_add_cols(L.ptr(a), _ld(a), L.ptr(b), _ld(b) if b is not None else 0, B, C, L.ptr(out),
                                     _ld(out), L.stream_ptr()))


# ---- QR-DQN with a grouped output layer (qr_grouped.hip) ------------------------------------------------------
def group_rows(key, n_groups, n_tiles, rowmap, tile_key, row_begin, workspace, dense: bool = True):
    _chk_dev(key, rowmap, tile_key, row_begin, workspace)
    assert key.dtype == torch.int32 and rowmap.numel() == n_tiles * 128 and tile_key.numel() == n_tiles
    _run("rg_group_rows", dict(B=key.numel(), G=n_groups),
         lambda: L.lib().rg_group_rows(key.data_ptr(), key.numel(), n_groups, n_tiles, int(dense), rowmap.data_ptr(),
                                       tile_key.data_ptr(), row_begin.data_ptr(), workspace.data_ptr(), workspace.numel() * 4,
                                       L.stream_ptr()))


def group_wfrag_elems(group_rows: int, in_features: int, transposed: bool) -> int:
    return int(L.lib().rg_group_wfrag_elems(group_rows, in_features, int(transposed)))


def group_weights_stage(w, n_groups, group_rows, wf, wb, x3: bool = False):
    """x3: split-bf16 — every group's fragment set is [hi plane | lo plane] (wf / wb hold 2 x the elements)"""
    _chk_dev(w, wf, wb)
    assert w.is_contiguous() and w.dtype == F32 and w.shape[0] == n_groups * group_rows
    _run("rg_group_weights_stage", dict(G=n_groups, Ng=group_rows, K=w.shape[1]),
         lambda: L.lib().rg_group_weights_stage(w.data_ptr(), n_groups, group_rows, w.shape[1], int(x3), L.ptr(wf), L.ptr(wb),
                                                L.stream_ptr()))


def wide_head_mean(w, b, n_groups, group_rows, wbar, bbar, wfrag_fwd=None, x3: bool = False):
    """wfrag_fwd: the (already staged once) forward fragments of the [n_groups, in] mean layer, rewritten in the same launch
    (x3: both planes)"""
    _chk_dev(w, b, wbar, bbar)
    assert w.is_contiguous() and wbar.is_contiguous() and wbar.shape == (n_groups, w.shape[1])
    if wfrag_fwd is None:
        _run("rg_wide_head_mean", dict(G=n_groups, Ng=group_rows, K=w.shape[1]),
             lambda: L.lib().rg_wide_head_mean(w.data_ptr(), L.ptr(b), n_groups, group_rows, w.shape[1], wbar.data_ptr(),
                                               bbar.data_ptr(), L.stream_ptr()))
    else:
        _chk_dev(wfrag_fwd)
        _run("rg_wide_head_mean", dict(G=n_groups, Ng=group_rows, K=w.shape[1]),
             lambda: L.lib().rg_wide_head_mean_staged(w.data_ptr(), L.ptr(b), n_groups, group_rows, w.shape[1],
                                                      wbar.data_ptr(), bbar.data_ptr(), wfrag_fwd.data_ptr(), int(x3),
                                                      L.stream_ptr()))


def qr_select_action(q, mask, maxq: bool, key):
    _chk_dev(q, mask, key)
    B, A = mask.shape
    assert mask.is_contiguous() and mask.dtype == F32 and key.dtype == torch.int32 and key.numel() == B
    _run("rg_qr_select_action", dict(B=B, A=A),
         lambda: L.lib().rg_qr_select_action(L.ptr(q), _ld(q) if q is not None else 0, mask.data_ptr(), B, A, int(maxq),
                                             key.data_ptr(), L.stream_ptr()))


def qr_select_group_rows(q, mask, maxq: bool, key, n_tiles, rowmap, tile_key, row_begin, workspace, dense: bool = True):
    """qr_select_action + group_rows (n_groups = number of actions) in two launches"""
    _chk_dev(q, mask, key, rowmap, tile_key, row_begin, workspace)
    B, A = mask.shape
    assert mask.is_contiguous() and mask.dtype == F32 and key.dtype == torch.int32 and key.numel() == B
    assert rowmap.numel() == n_tiles * 128 and tile_key.numel() == n_tiles
    _run("rg_group_rows", dict(B=B, G=A),
         lambda: L.lib().rg_qr_select_group_rows(L.ptr(q), _ld(q) if q is not None else 0, mask.data_ptr(), B, A, int(maxq),
                                                 key.data_ptr(), n_tiles, int(dense), rowmap.data_ptr(), tile_key.data_ptr(),
                                                 row_begin.data_ptr(), workspace.data_ptr(), workspace.numel() * 4,
                                                 L.stream_ptr()))


def qr_compact_head(z, zt, rowmap, row_key, reward, reward_boosts, not_terminal, gamma, gamma_exponent, quantiles,
                    batch, num_atoms, dz, loss_partials, tile_losses=None):
    """row_key [batch] int32: the group (logged action) of every batch row — read for the reward boost only"""
    _chk_dev(z, zt, rowmap, row_key, reward, reward_boosts, not_terminal, gamma_exponent, quantiles, dz, loss_partials,
             tile_losses)
    rows = rowmap.numel()
    assert z.shape[0] == rows and dz.shape[0] == rows and loss_partials.numel() == rows
    assert reward_boosts is None or (row_key is not None and row_key.numel() == batch)
    _run("rg_qr_compact_head", dict(rows=rows, N=num_atoms),
         lambda: L.lib().rg_qr_compact_head(z.data_ptr(), _ld(z), zt.data_ptr(), _ld(zt), rowmap.data_ptr(),
                                            L.ptr(row_key), rows, reward.data_ptr(), L.ptr(reward_boosts),
                                            not_terminal.data_ptr(), float(gamma), L.ptr(gamma_exponent),
                                            quantiles.data_ptr(), batch, num_atoms, dz.data_ptr(), _ld(dz),
                                            loss_partials.data_ptr(), L.ptr(tile_losses), L.stream_ptr()))


def grouped_dz_rows(rows: int, n_groups: int) -> int:
    """rows of the fragment matrix that holds a grouped layer's dZ (include/reagent_hip.h, rg_mlp_desc: group g's blocks g late)"""
    return rows + 32 * n_groups


def group_head_wgrad(dzw_frag, h_frag, row_begin, n_groups, group_rows, in_features, splits, dw, workspace, x3: bool = False,
                     rows: int = 0):
    """x3: split-bf16 operands ([hi plane | lo plane] over the `rows` rows of the grouped space)"""
    _chk_dev(dzw_frag, h_frag, row_begin, dw, workspace)
    assert dw.is_contiguous() and dw.shape == (n_groups * group_rows, in_features)
    _run("rg_group_head_wgrad", dict(G=n_groups, Ng=group_rows, K=in_features),
         lambda: L.lib().rg_group_head_wgrad(dzw_frag.data_ptr(), h_frag.data_ptr(), row_begin.data_ptr(), n_groups,
                                             group_rows, in_features, splits, int(x3), int(rows), dw.data_ptr(),
                                             workspace.data_ptr(), workspace.numel() * 4, L.stream_ptr()))


# ---- dueling aggregation -------------------------------------------------------------------------------------
def dueling_combine(value, adv, num_actions, num_atoms, q):
    _chk_dev(value, adv, q)
    B = q.shape[0]
    _run("rg_dueling_combine", dict(B=B, A=num_actions, N=num_atoms),
         lambda: L.lib().rg_dueling_combine(value.data_ptr(), _ld(value), adv.data_ptr(), _ld(adv), B, num_actions, num_atoms,
                                            q.data_ptr(), _ld(q), L.stream_ptr()))


def dueling_split(dq, num_actions, num_atoms, dadv, dvalue):
    _chk_dev(dq, dadv, dvalue)
    B = dq.shape[0]
    _run("rg_dueling_split", dict(B=B, A=num_actions, N=num_atoms),
         lambda: L.lib().rg_dueling_split(dq.data_ptr(), _ld(dq), B, num_actions, num_atoms, dadv.data_ptr(), _ld(dadv),
                                          dvalue.data_ptr(), _ld(dvalue), L.stream_ptr()))
