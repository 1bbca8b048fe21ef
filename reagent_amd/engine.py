"""Host-side execution engine for FullyConnected stacks on the HIP kernels.

``ParamSlab``  : one flat fp32 arena per network; the nn.Parameters become views into it so that
                 the fused Adam / soft-update kernels and the RCCL all-reduce see ONE buffer while
                 ``state_dict()`` / ``parameters()`` keep working.
``FCStack``    : forward / backward of a FullyConnectedNetwork
                 (reference: reagent/models/fully_connected_network.py:101-163 + autograd) as a
                 sequence of rg_fc_forward / rg_fc_dgrad / rg_fc_wgrad launches, with the
                 compute-type weight copies and the activation workspace it needs.
"""
import contextlib
from typing import List, Optional, Sequence

import torch

from . import _lib as L
from . import ops


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _mat(rows: int, cols: int, dtype, device) -> torch.Tensor:
    """[rows, cols] view with a row pitch padded to 16 bytes worth of elements (vector loads)."""
    ld = _round_up(max(cols, 1), 8)
    return torch.empty(rows, ld, dtype=dtype, device=device)[:, :cols]


class ParamSlab:
    """Flat fp32 storage for a list of parameters (+ an equally shaped gradient slab)."""

    ALIGN = 4  # elements (16 bytes) so every tensor can be an MFMA-GEMM operand directly

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params = list(params)
        assert len(self.params) > 0
        self.offsets = []
        off = 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise L.ReagentHipError("parameters must be float32 (fp32 master weights)")
            self.offsets.append(off)
            off += _round_up(p.numel(), self.ALIGN)
        self.total = off
        self.data = None
        self.grad = None
        self.rebind()

    def rebind(self):
        """(Re)allocate on the parameters' current device and point every parameter into the slab."""
        dev = self.params[0].device
        data = torch.zeros(self.total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                data[off : off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = data[off : off + p.numel()].view(p.shape)
        self.data = data
        self.grad = torch.zeros_like(data)

    def is_bound(self) -> bool:
        base = self.data.data_ptr()
        return all(p.data_ptr() == base + 4 * off and p.device == self.data.device
                   for p, off in zip(self.params, self.offsets))

    def ensure_bound(self):
        if not self.is_bound():
            self.rebind()

    def view(self, slab: torch.Tensor, i: int) -> torch.Tensor:
        p, off = self.params[i], self.offsets[i]
        return slab[off : off + p.numel()].view(p.shape)

    def grad_views(self) -> List[torch.Tensor]:
        return [self.view(self.grad, i) for i in range(len(self.params))]

    def attach_grads(self):
        """Make p.grad alias the gradient slab (what the HIP backward writes)."""
        for i, p in enumerate(self.params):
            p.grad = self.view(self.grad, i)


def ensure_slab(params) -> ParamSlab:
    """The ParamSlab that holds exactly ``params`` (created, and the parameters re-homed, if needed)."""
    params = list(params)
    slab = getattr(params[0], "_rg_slab", None)
    if slab is None or len(slab.params) != len(params) or any(a is not b for a, b in zip(slab.params, params)):
        slab = ParamSlab(params)
        for p in params:
            p._rg_slab = slab
    else:
        slab.ensure_bound()
    return slab


def _through_output_activation(act: int, dout32: torch.Tensor, out32: Optional[torch.Tensor]) -> torch.Tensor:
    """d loss / d (pre-activation of the last layer) from d loss / d output"""
    if act == L.ACT["linear"]:
        return dout32
    if out32 is None:
        raise ValueError("backward through a non-linear output layer needs the forward output (out32=)")
    dz = torch.empty_like(dout32)
    ops.act_backward(dout32, out32, act, dz)
    return dz


class FCStack:
    """Runs one FullyConnectedNetwork on the GPU kernels.

    weights[i]: [out_i, in_i] fp32 Parameter (nn.Linear layout), biases[i]: [out_i], acts[i]: rg
    activation code of layer i (the last one is the output activation).
    """

    def __init__(self, weights, biases, acts: List[int], precision: int, layer_norms=None):
        """layer_norms (optional): per layer None or an nn.LayerNorm-like holder (.weight, .bias, .eps) applied between
        the Linear and its activation (fully_connected_network.py:128-130)"""
        assert len(weights) == len(biases) == len(acts)
        self.weights, self.biases, self.acts = list(weights), list(biases), list(acts)
        self.lns = list(layer_norms) if layer_norms is not None else [None] * len(weights)
        assert len(self.lns) == len(weights)
        self._ln_grads = [None] * len(weights)  # (dgamma, dbeta) destinations, bound by the trainer (bind_ln_grads)
        self.precision = precision
        self.cdtype = ops.compute_dtype(precision)
        self.dims = [self.weights[0].shape[1]] + [w.shape[0] for w in self.weights]
        self.L = len(self.weights)
        self._wc = [None] * self.L   # compute-type weights [out, in]
        self._wtc = [None] * self.L  # compute-type transposed weights [in, out]
        self._batch = -1
        self._ws = {}
        self._staged_versions = None

    # ---- weights -------------------------------------------------------------------------
    def stage_weights(self, need_transposed: bool = True, force: bool = False):
        """fp32 master weights -> compute-type copies (and W^T for dgrad).  Skipped when the
        parameters have not been modified since the last staging."""
        versions = tuple((w._version, getattr(w, "_rg_version", 0)) for w in self.weights) + (need_transposed,)
        if not force and versions == self._staged_versions and all(
            self._wsrc_ptrs[i] == self.weights[i].data_ptr() for i in range(self.L)
        ):
            return
        dev = self.weights[0].device
        for i, w in enumerate(self.weights):
            out_f, in_f = w.shape
            wd = w.detach()
            if self.precision == L.PREC_F32:
                self._wc[i] = wd  # fp32 master weights are the operand
                dst = None
            else:
                if self._wc[i] is None or self._wc[i].device != dev:
                    self._wc[i] = _mat(out_f, in_f, self.cdtype, dev)
                dst = self._wc[i]
            dst_t = None
            if need_transposed and (i > 0 or self._need_dx):
                if self._wtc[i] is None or self._wtc[i].device != dev:
                    self._wtc[i] = _mat(in_f, out_f, self.cdtype, dev)
                dst_t = self._wtc[i]
            if dst is not None or dst_t is not None:
                ops.transpose_cast(wd, dst, dst_t)
        self._staged_versions = versions
        self._wsrc_ptrs = [w.data_ptr() for w in self.weights]

    _need_dx = False
    _wsrc_ptrs = ()

    def set_need_input_grad(self, flag: bool):
        if flag != self._need_dx:
            self._need_dx = flag
            self._staged_versions = None

    # ---- workspace -----------------------------------------------------------------------
    def _ensure_ws(self, batch: int, device):
        if self._batch == batch and self._ws.get("device") == device:
            return
        ws = {"device": device}
        hidden = self.dims[1:-1]
        wmax = max(hidden) if hidden else 1
        cd = self.cdtype
        ws["h"] = [_mat(batch, wmax, cd, device) for _ in range(2)] if hidden else []
        ws["ht"] = [_mat(n, batch, cd, device) for n in hidden]           # saved (transposed) activations
        dmax = max(self.dims[1:])
        ws["dz"] = [_mat(batch, dmax, cd, device) for _ in range(2)]
        ws["dzt"] = [_mat(dmax, batch, cd, device) for _ in range(2)]
        nbytes = max(
            ops.fc_wgrad_workspace_bytes(self.dims[i + 1], self.dims[i], batch, self.precision)
            for i in range(self.L)
        )
        ws["wgrad"] = torch.empty(_round_up(nbytes, 16) // 4, dtype=torch.float32, device=device)
        if any(ln is not None for ln in self.lns):
            f32 = dict(dtype=torch.float32, device=device)
            ws["ln_z"] = [_mat(batch, self.dims[i + 1], torch.float32, device) if ln is not None else None
                          for i, ln in enumerate(self.lns)]  # pre-norm outputs of the Linear, kept for the backward
            ws["ln_mean"] = [torch.empty(batch, **f32) if ln is not None else None for ln in self.lns]
            ws["ln_rstd"] = [torch.empty(batch, **f32) if ln is not None else None for ln in self.lns]
            ws["ln_g"] = _mat(batch, dmax, torch.float32, device)  # d loss / d (LayerNorm output) of the layer at hand
            nb = max(L.lib().rg_layer_norm_backward_workspace_bytes(batch, self.dims[i + 1])
                     for i, ln in enumerate(self.lns) if ln is not None)
            ws["ln_ws"] = torch.empty(_round_up(nb, 16) // 4, **f32)
        self._ws = ws
        self._batch = batch

    def bind_ln_grads(self, dst):
        """dst[i] = (dgamma, dbeta) fp32 destinations (gradient-slab views) of layer i's LayerNorm, or None"""
        assert len(dst) == self.L and all((d is None) == (ln is None) for d, ln in zip(dst, self.lns))
        self._ln_grads = list(dst)

    def stage_input(self, x32: torch.Tensor, need_transposed: bool):
        """fp32 [B, in] network input -> (row-major compute-type operand, transposed copy or None)."""
        B, in_f = x32.shape
        dev = x32.device
        xt = _mat(in_f, B, self.cdtype, dev) if need_transposed else None
        if self.precision == L.PREC_F32 and x32.stride(1) == 1 and x32.dtype == torch.float32:
            xc = x32
            if xt is not None:
                ops.transpose_cast(x32, None, xt)
        else:
            xc = _mat(B, in_f, self.cdtype, dev)
            ops.transpose_cast(x32, xc, xt)
        return xc, xt

    # ---- forward -------------------------------------------------------------------------
    def forward(self, xc: torch.Tensor, out32: torch.Tensor, save: bool = False):
        """xc: staged input (compute type) [B, in]; out32: fp32 [B, out_last] (written)."""
        B = xc.shape[0]
        self._ensure_ws(B, xc.device)
        ws = self._ws
        cur = xc
        for i in range(self.L):
            last = i == self.L - 1
            out_f = self.dims[i + 1]
            ln = self.lns[i]
            if ln is not None:  # Linear (fp32 out) -> LayerNorm -> activation
                z = ws["ln_z"][i]
                ops.fc_forward(cur, self._wc[i], self.biases[i].detach(), L.ACT["linear"], self.precision, y=None, y32=z, yt=None)
                y = None if last else ws["h"][i % 2][:, :out_f]
                ops.layer_norm_forward(z, ln.weight.detach(), ln.bias.detach(), ln.eps, self.acts[i], y=y,
                                       y32=out32 if last else None, mean=ws["ln_mean"][i], rstd=ws["ln_rstd"][i])
                if not last:
                    if save:
                        ops.transpose_cast(y, None, ws["ht"][i])
                    cur = y
            elif last:
                ops.fc_forward(cur, self._wc[i], self.biases[i].detach(), self.acts[i], self.precision,
                               y=None, y32=out32, yt=None)
            else:
                y = ws["h"][i % 2][:, :out_f]
                yt = ws["ht"][i] if save else None
                ops.fc_forward(cur, self._wc[i], self.biases[i].detach(), self.acts[i], self.precision,
                               y=y, y32=None, yt=yt)
                cur = y
        return out32

    # ---- backward ------------------------------------------------------------------------
    def backward(self, dout32: torch.Tensor, xt: torch.Tensor, dw: List[torch.Tensor],
                 db: List[torch.Tensor], dx32: Optional[torch.Tensor] = None, skip_wgrad: bool = False,
                 out32: Optional[torch.Tensor] = None):
        """Gradients of a scalar loss given d loss / d output (fp32 [B, out_last]).
        out32: the forward output, needed only when the LAST layer is non-linear (its derivative is
        applied first with rg_act_backward; the kernels below assume a linear output layer).
        skip_wgrad: only propagate to the input (frozen network, e.g. SAC's critics in the actor step).

        Requires a preceding ``forward(..., save=True)`` on the same batch.  xt: transposed staged
        input [in, B].  dw[i] / db[i]: contiguous fp32 destinations (gradient-slab views).
        dx32 (optional): fp32 [B, in] destination for the gradient w.r.t. the network input.
        """
        dout32 = _through_output_activation(self.acts[-1], dout32, out32)
        B = dout32.shape[0]
        ws = self._ws
        n_last = self.dims[-1]
        dz = ws["dz"][0][:, :n_last]
        dzt = ws["dzt"][0][:n_last]
        if self.lns[-1] is not None:  # the gradient first passes the output layer's LayerNorm
            self._ln_backward(self.L - 1, dout32, dz, dzt, skip_wgrad)
        else:
            ops.transpose_cast(dout32, dz, dzt)
        for i in range(self.L - 1, -1, -1):
            in_f, out_f = self.dims[i], self.dims[i + 1]
            x_t = xt if i == 0 else ws["ht"][i - 1]
            if not skip_wgrad:
                ops.fc_wgrad(dzt, x_t, dw[i], db[i], ws["wgrad"], self.precision)
            if i > 0:
                nxt = (self.L - i) % 2
                dz_n = ws["dz"][nxt][:, :in_f]
                dzt_n = ws["dzt"][nxt][:in_f]
                if self.lns[i - 1] is not None:  # dgrad x act' gives d loss / d (LayerNorm output) in fp32 ...
                    g = ws["ln_g"][:, :in_f]
                    ops.fc_dgrad(dz, self._wtc[i], ws["ht"][i - 1], self.acts[i - 1], self.precision, dx=None, dx32=g,
                                 dxt=None)
                    self._ln_backward(i - 1, g, dz_n, dzt_n, skip_wgrad)  # ... which its backward turns into dZ
                else:
                    ops.fc_dgrad(dz, self._wtc[i], ws["ht"][i - 1], self.acts[i - 1], self.precision,
                                 dx=dz_n, dx32=None, dxt=dzt_n)
                dz, dzt = dz_n, dzt_n
            elif dx32 is not None:
                ops.fc_dgrad(dz, self._wtc[0], None, L.ACT["linear"], self.precision, dx=None,
                             dx32=dx32, dxt=None)


def _ln_backward(self, i: int, g32: torch.Tensor, dz: torch.Tensor, dzt: torch.Tensor, skip_wgrad: bool):
    """LayerNorm backward of layer i: g32 = d loss / d (LN output) -> dz (compute type, + transposed copy); the
    parameter gradients go to the bound destinations (scratch when the stack is frozen)"""
    ws, ln = self._ws, self.lns[i]
    dst = self._ln_grads[i]
    if dst is None or skip_wgrad:
        n = self.dims[i + 1]
        scratch = ws.setdefault("ln_scratch", torch.empty(2 * max(self.dims[1:]), dtype=torch.float32, device=g32.device))
        if dst is None and not skip_wgrad:
            raise RuntimeError("LayerNorm gradients have no destination: the trainer did not call bind_ln_grads")
        dst = (scratch[:n], scratch[n : 2 * n])
    ops.layer_norm_backward(g32, ws["ln_z"][i], ws["ln_mean"][i], ws["ln_rstd"][i], ln.weight.detach(), dst[0], dst[1],
                            ws["ln_ws"], dz=dz)
    ops.transpose_cast(dz, None, dzt)


FCStack._ln_backward = _ln_backward

SAVE_FOR_DX = 2  # FusedMLP.forward(save=SAVE_FOR_DX): keep only what a dx-only backward needs (rg_mlp_forward_fused save = 2)


class FusedMLP:
    """bf16-MFMA engine for stacks the fused kernels support (hidden width 256/512 shared by
    all hidden layers, input <= 512, output <= 128): rg_mlp_forward_fused / rg_mlp_backward_fused /
    rg_mlp_wgrad_fused.  Same interface as FCStack.

    x3=False: bf16 operands (throughput mode, ~2e-2 from fp32).
    x3=True : split-bf16 operands ("bf16x3", PREC_BF16X3): every operand as hi + lo bf16 planes, three MFMAs
              per product, fp32-class results (Q within 1e-4 of the reference); every fragment buffer holds
              two planes."""

    def __init__(self, weights, biases, acts: List[int], x3: bool = False):
        self.weights, self.biases, self.acts = list(weights), list(biases), list(acts)
        self.x3 = bool(x3)
        self.planes = 2 if self.x3 else 1
        self.precision = L.PREC_BF16X3 if self.x3 else L.PREC_BF16
        self.cdtype = torch.bfloat16
        self.dims = [self.weights[0].shape[1]] + [w.shape[0] for w in self.weights]
        self.L = len(self.weights)
        self._wf = [None] * self.L
        self._wb = [None] * self.L
        self._need_dx = False
        self._saved = 0
        self._staged_versions = None
        self._wsrc_ptrs = ()
        self._batch = -1
        self._ws = {}
        self._desc = L.MlpDesc()

    fold_tails = True  # bias-gradient column reduce inside the weight gradient's reduce launch (False: its own launch)

    @staticmethod
    def supported(weights, acts) -> bool:
        d = L.MlpDesc()
        n = len(weights)
        if n < 2 or n > L.MLP_MAX_LAYERS:
            return False
        d.n_layers = n
        dims = [weights[0].shape[1]] + [w.shape[0] for w in weights]
        for i, v in enumerate(dims):
            d.dims[i] = v
        return bool(L.lib().rg_mlp_fused_supported(d))

    def set_need_input_grad(self, flag: bool):
        if flag != self._need_dx:
            self._need_dx = flag
            self._staged_versions = None

    def stage_weights(self, need_transposed: bool = True, force: bool = False):
        versions = tuple((w._version, getattr(w, "_rg_version", 0)) for w in self.weights) + (need_transposed,)
        if not force and versions == self._staged_versions and all(
            self._wsrc_ptrs[i] == self.weights[i].data_ptr() for i in range(self.L)
        ):
            return
        dev = self.weights[0].device
        lib = L.lib()
        d = self._desc
        d.n_layers = self.L
        d.x3 = int(self.x3)
        for i, v in enumerate(self.dims):
            d.dims[i] = v
        need_bwd = bool(need_transposed)
        P = self.planes
        for i, w in enumerate(self.weights):
            out_f, in_f = w.shape
            if self._wf[i] is None or self._wf[i].device != dev:
                self._wf[i] = torch.empty(P * lib.rg_wfrag_elems(out_f, in_f), dtype=torch.bfloat16, device=dev)
            if need_bwd and (self._wb[i] is None or self._wb[i].device != dev):
                self._wb[i] = torch.empty(P * lib.rg_wfrag_elems(in_f, out_f), dtype=torch.bfloat16, device=dev)
            wd = w.detach()
            assert wd.is_contiguous()
            L.require_cuda(wd)
            d.w[i] = wd.data_ptr()
            d.wfrag_fwd[i] = self._wf[i].data_ptr()
            d.wfrag_bwd[i] = self._wb[i].data_ptr() if self._wb[i] is not None else None
        ops._run("rg_mlp_stage_weights_fused", dict(L=self.L),
                 lambda: lib.rg_mlp_stage_weights_fused(d, int(need_bwd), L.stream_ptr()))
        self._staged_versions = versions
        self._wsrc_ptrs = [w.data_ptr() for w in self.weights]

    def _ensure_ws(self, batch: int, device, training: bool):
        key = (batch, device, training)
        if self._ws.get("key") == key:
            return
        if not training and self._ws.get("key", (0, device))[1] == device:
            # a non-saving forward needs no workspace of its own: a training workspace of another batch size (QR-DQN:
            # 65 536 rows here, B + 128 A rows in the grouped space) stays as it is instead of being rebuilt every step
            self._ws.setdefault("key", key)
            return
        lib = L.lib()
        ws = {"key": key}
        if training:
            bf = dict(dtype=torch.bfloat16, device=device)
            P = self.planes
            ws["act_frag"] = [torch.empty(P * lib.rg_frag_elems(batch, self.dims[l]), **bf) for l in range(self.L)]
            ws["dz_frag"] = [torch.empty(P * lib.rg_frag_elems(batch, self.dims[l + 1]), **bf) for l in range(self.L)]
            ws["act_sign"] = [torch.empty(lib.rg_sign_bytes(batch, self.dims[l]), dtype=torch.uint8, device=device)
                              if l >= 1 else None for l in range(self.L)]
            dd = L.MlpDesc()
            dd.n_layers = self.L
            dd.x3 = int(self.x3)
            for i, v in enumerate(self.dims):
                dd.dims[i] = v
            nbytes = lib.rg_mlp_wgrad_fused_workspace_bytes(dd, batch)
            ws["wgrad"] = torch.empty(_round_up(nbytes, 16) // 4, dtype=torch.float32, device=device)
            self._ws = ws
            nb = lib.rg_mlp_backward_fused_workspace_bytes(self._fill_desc(), batch)
            ws["bwd"] = torch.empty(_round_up(nb, 16) // 4, dtype=torch.float32, device=device)
        self._ws = ws
        self._batch = batch

    def _fill_desc(self):
        d = self._desc
        d.n_layers = self.L
        d.x3 = int(self.x3)
        d.dx_col0 = 0
        d.dx_only = 0
        d.defer_db, d.sum_n = 0, 0
        d.db_partials, d.sum_in, d.sum_out = None, None, None
        for i, v in enumerate(self.dims):
            d.dims[i] = v
        ws = self._ws
        for l in range(self.L):
            d.acts[l] = self.acts[l]
            d.wfrag_fwd[l] = self._wf[l].data_ptr() if self._wf[l] is not None else None
            d.wfrag_bwd[l] = self._wb[l].data_ptr() if self._wb[l] is not None else None
            d.bias[l] = self.biases[l].data_ptr()
            d.act_frag[l] = ws["act_frag"][l].data_ptr() if "act_frag" in ws else None
            d.dz_frag[l] = ws["dz_frag"][l].data_ptr() if "dz_frag" in ws else None
            d.act_sign[l] = ws["act_sign"][l].data_ptr() if "act_sign" in ws and ws["act_sign"][l] is not None else None
        return d

    def stage_input(self, x32: torch.Tensor, need_transposed: bool):
        return x32, None  # the kernel reads fp32 (or bf16) rows directly and casts in flight

    def forward(self, xc: torch.Tensor, out32: torch.Tensor, save: bool = False, x2: Optional[torch.Tensor] = None,
                rowmap: Optional[torch.Tensor] = None):
        """x2 (optional): second input panel — the network input is cat(xc, x2) (FullyConnectedCritic's
        cat(state, action), critic.py:79-92) read in place by the kernel; xc.shape[1] must be a multiple of 32.
        rowmap (optional, int32, length a multiple of 128): the stack runs on len(rowmap) rows, row r reading input
        row rowmap[r] of xc (-1: zeros) — "grouped space" of qr_engine.py; out32 has len(rowmap) rows."""
        L.require_cuda(xc)
        B = xc.shape[0] if rowmap is None else rowmap.shape[0]
        save = int(save)  # 0 / 1 (everything backward + wgrad read) / SAVE_FOR_DX (what a dx-only backward reads)
        self._ensure_ws(B, xc.device, training=bool(save))
        d = self._fill_desc()
        assert xc.stride(1) == 1 and out32.stride(1) == 1
        if x2 is not None:
            L.require_cuda(x2)
            assert x2.stride(1) == 1 and xc.shape[1] % 32 == 0  # (the panels may differ in element type)
            assert xc.shape[1] + x2.shape[1] == self.dims[0] and x2.shape[0] == B
            d.x2, d.ldx2, d.x_split, d.x2_dtype = x2.data_ptr(), x2.stride(0), xc.shape[1], ops.dt_code(x2.dtype)
        else:
            assert xc.shape[1] == self.dims[0]
            d.x2, d.ldx2, d.x_split = None, 0, 0
        if rowmap is not None:
            assert rowmap.dtype == torch.int32 and rowmap.is_contiguous() and B % 128 == 0 and out32.shape[0] == B
            L.require_cuda(rowmap)
        d.rowmap = rowmap.data_ptr() if rowmap is not None else None
        ops._run("rg_mlp_forward_fused", dict(B=B, save=int(save), dims=tuple(self.dims)),
                 lambda: L.lib().rg_mlp_forward_fused(d, xc.data_ptr(), ops.dt_code(xc.dtype), xc.stride(0), B,
                                                      out32.data_ptr(), out32.stride(0), save,
                                                      L.stream_ptr()))
        self._saved = save
        return out32

    def backward(self, dout32: torch.Tensor, xt, dw: List[torch.Tensor], db: List[torch.Tensor],
                 dx32: Optional[torch.Tensor] = None, skip_wgrad: bool = False,
                 out32: Optional[torch.Tensor] = None, dx_col0: int = 0, tail_sum=None):
        """dx_col0: dx32 receives the gradient of input columns [dx_col0, in_features) only (a multiple of 32).
        tail_sum = (partials fp32 [n], scale, out fp32 [1]): out = scale * sum(partials) is evaluated in the weight
        gradient's reduce launch (the mean loss of the step from the loss head's partials: one launch fewer)."""
        dout32 = _through_output_activation(self.acts[-1], dout32, out32)
        B = dout32.shape[0]
        assert self._ws.get("key") == (B, dout32.device, True), "backward needs a saving forward first"
        d = self._fill_desc()
        d.x2, d.ldx2, d.x_split = None, 0, 0
        d.dx_col0 = dx_col0
        # a frozen network (only dx wanted): the dZ fragments have no reader and are not written
        d.dx_only = int(skip_wgrad and dx32 is not None and (db is None or all(b is None for b in db)))
        assert self._saved == 1 or d.dx_only, "a forward saved with SAVE_FOR_DX serves a dx-only backward (skip_wgrad) alone"
        if dx32 is not None:
            assert dx_col0 % 32 == 0 and dx32.shape[1] == self.dims[0] - dx_col0
        lib = L.lib()
        ws = self._ws
        want_db = False
        for l in range(self.L):
            d.db[l] = db[l].data_ptr() if (db is not None and not skip_wgrad and db[l] is not None) else None
            want_db = want_db or bool(d.db[l])
        # the bias gradients' column reduce (and the step's loss mean) ride in the weight gradient's reduce launch
        defer = bool(want_db and not skip_wgrad and self.fold_tails)
        d.defer_db = int(defer)
        d.db_partials, d.sum_in, d.sum_out, d.sum_n, d.sum_scale = None, None, None, 0, 0.0
        ops._run("rg_mlp_backward_fused", dict(B=B, dims=tuple(self.dims)),
                 lambda: lib.rg_mlp_backward_fused(d, dout32.data_ptr(), dout32.stride(0), B,
                                                   dx32.data_ptr() if dx32 is not None else None,
                                                   dx32.stride(0) if dx32 is not None else 0,
                                                   ws["bwd"].data_ptr(), ws["bwd"].numel() * 4, L.stream_ptr()))
        d.dx_only = 0
        d.defer_db = 0
        if not skip_wgrad:
            for l in range(self.L):
                d.dw[l] = dw[l].data_ptr()
            if defer:
                d.db_partials = ws["bwd"].data_ptr()
            if tail_sum is not None:
                part, scale, out = tail_sum
                L.require_cuda(part)
                L.require_cuda(out)
                assert part.dtype == torch.float32 and part.is_contiguous() and out.dtype == torch.float32
                d.sum_in, d.sum_n, d.sum_scale, d.sum_out = part.data_ptr(), part.numel(), float(scale), out.data_ptr()
            wsb = ws["wgrad"].numel() * 4
            ops._run("rg_mlp_wgrad_fused", dict(B=B, dims=tuple(self.dims)),
                     lambda: lib.rg_mlp_wgrad_fused(d, B, ws["wgrad"].data_ptr(), wsb, L.stream_ptr()))
            d.db_partials, d.sum_in, d.sum_out, d.sum_n = None, None, None, 0
        else:
            assert tail_sum is None, "tail_sum rides in the weight gradient's launch"


class GroupedHead:
    """The output layer of a fused stack as n_groups layers [group_rows, H], one per group of rows of a grouped row
    space (qr_engine.py): per-group MFMA fragments of W_g / W_g^T (rg_group_weights_stage), the bias, and the
    buffers the backward of that layer needs."""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor, n_groups: int, group_rows: int, need_bwd: bool, x3: bool = False):
        self.weight, self.bias, self.G, self.Ng = weight, bias, n_groups, group_rows
        H = weight.shape[1]
        dev = weight.device
        self.H = H
        # split-bf16 (x3): a group's fragment set is [hi plane | lo plane]; per_f / per_b are the strides between groups
        self.x3, self.planes = bool(x3), 2 if x3 else 1
        self.per_f = self.planes * ops.group_wfrag_elems(group_rows, H, False)
        self.per_b = self.planes * ops.group_wfrag_elems(group_rows, H, True)
        self.wf = torch.empty(n_groups * self.per_f, dtype=torch.bfloat16, device=dev)
        self.wb = torch.empty(n_groups * self.per_b, dtype=torch.bfloat16, device=dev) if need_bwd else None
        self._rows = -1

    def stage(self):
        ops.group_weights_stage(self.weight.detach(), self.G, self.Ng, self.wf, self.wb, x3=self.x3)

    def workspace(self, rows: int, st: "FusedMLP"):
        if self._rows != rows:
            dev = self.weight.device
            lib = L.lib()
            # (group g's 32-row blocks are written g blocks late — a block two groups share once per group: ops.grouped_dz_rows)
            self.dz_frag = torch.zeros(self.planes * lib.rg_frag_elems(ops.grouped_dz_rows(rows, self.G), self.Ng), dtype=torch.bfloat16,
                                       device=dev)
            d = self.desc(st, None, None, None)
            self.bwd_ws = torch.empty(lib.rg_mlp_backward_fused_workspace_bytes(d, rows) // 4 + 4, dtype=torch.float32, device=dev)
            self._rows = rows

    def desc(self, st: "FusedMLP", space, scatter, save_dz) -> "L.MlpDesc":
        """the stack's descriptor with its last layer replaced by this grouped layer"""
        src = st._fill_desc()
        d = L.MlpDesc()
        ctypes_copy(d, src)
        n = st.L
        d.dims[n] = self.Ng
        d.wfrag_fwd[n - 1] = self.wf.data_ptr()
        d.wfrag_bwd[n - 1] = self.wb.data_ptr() if self.wb is not None else None
        d.bias[n - 1] = self.bias.data_ptr()
        d.group_stride_fwd, d.group_stride_bwd, d.n_groups = self.per_f, self.per_b, self.G
        if space is not None:
            d.rowmap, d.tile_key, d.row_begin = space.rowmap.data_ptr(), space.tile_key.data_ptr(), space.row_begin.data_ptr()
            d.out_scatter = int(bool(scatter))
        if save_dz:
            d.dz_frag[n - 1] = self.dz_frag.data_ptr()
        return d


def ctypes_copy(dst, src):
    import ctypes

    ctypes.memmove(ctypes.byref(dst), ctypes.byref(src), ctypes.sizeof(type(src)))


def fused_forward_grouped(st: "FusedMLP", head: GroupedHead, x: torch.Tensor, space, out32: torch.Tensor, scatter: bool,
                          save: bool):
    """The stack in the grouped row space `space` with `head` as its output layer:
    out32[dst(r), 0:group_rows] = head_g(trunk(x[rowmap[r]])), dst(r) = rowmap[r] (scatter) or r."""
    L.require_cuda(x)
    R = space.rows
    st._ensure_ws(R, x.device, training=save)
    head.workspace(R, st)
    d = head.desc(st, space, scatter, save)
    d.x2, d.ldx2, d.x_split = None, 0, 0
    assert x.stride(1) == 1 and out32.stride(1) == 1 and x.shape[1] == st.dims[0]
    ops._run("rg_mlp_forward_fused", dict(B=R, save=int(save), dims=tuple(st.dims[:-1]) + (head.Ng,)),
             lambda: L.lib().rg_mlp_forward_fused(d, x.data_ptr(), ops.dt_code(x.dtype), x.stride(0), R, out32.data_ptr(),
                                                  out32.stride(0), int(save), L.stream_ptr()))


_side_streams = {}


def side_stream(device):
    """the engine's second HIP stream of `device` (the grouped head's weight gradient, the QR step's loss sum): one per device,
    joined by whoever forks it (fused_backward_grouped)"""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())  # "cuda" == "cuda:<current>"
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def fused_backward_grouped(st: "FusedMLP", head: GroupedHead, space, dz32: torch.Tensor, dw: List[torch.Tensor],
                           db: List[torch.Tensor], wgrad_ws: torch.Tensor, splits: int, two_streams: bool = False,
                           tail_sum=None):
    """Backward of the stack + grouped head from dz32 = d loss / d (head output) [grouped rows, group_rows]:
    one rg_mlp_backward_fused launch (the head's input gradient is its first layer step, per-tile W_g^T), the
    trunk's weight gradients by rg_mlp_wgrad_fused, the head's by rg_group_head_wgrad.  dw / db: all L layers."""
    R = space.rows
    assert st._ws.get("key") == (R, dz32.device, True), "needs a saving grouped forward first"
    lib = L.lib()
    n = st.L
    d = head.desc(st, space, False, True)
    for l in range(n):
        d.db[l] = db[l].data_ptr()
    # the trunk's bias-gradient column reduce and the step's loss mean (tail_sum, FusedMLP.backward) ride in the reduce launch
    # of the trunk's weight gradient: two launch-bound tails fewer on the step's critical path
    fold = bool(st.fold_tails)
    d.defer_db = int(fold)
    ops._run("rg_mlp_backward_fused", dict(B=R, dims=tuple(st.dims[:-1]) + (head.Ng,)),
             lambda: lib.rg_mlp_backward_fused(d, dz32.data_ptr(), dz32.stride(0), R, None, 0, head.bwd_ws.data_ptr(),
                                               head.bwd_ws.numel() * 4, L.stream_ptr()))
    t = L.MlpDesc()  # the trunk: layers 0 .. L-2
    t.n_layers = n - 1
    t.x3 = int(st.x3)
    for i in range(n):
        t.dims[i] = st.dims[i]
    for l in range(n - 1):
        t.acts[l] = st.acts[l]
        t.act_frag[l], t.dz_frag[l], t.dw[l] = d.act_frag[l], d.dz_frag[l], dw[l].data_ptr()
        if fold:
            t.db[l] = db[l].data_ptr()
    if fold:
        t.db_partials = head.bwd_ws.data_ptr()
    if tail_sum is not None:
        part, scale, out = tail_sum
        if fold:
            t.sum_in, t.sum_n, t.sum_scale, t.sum_out = part.data_ptr(), part.numel(), float(scale), out.data_ptr()
        else:
            ops.reduce_sum(part, part.numel(), scale, out)
    ws = st._ws
    # (ABI 10) beside the head's weight gradient on the second stream the trunk's launch does not own the dispatch order its
    # uneven split plan leans on: measured 113.6 -> 122.6 us with it, so it is told to keep its splits even
    t.wgrad_flags = 1 if (dz32.is_cuda and two_streams) else 0
    # the trunk is its own launch plan (the splits of a launch are shared out over ITS layers): its own workspace size
    need = lib.rg_mlp_wgrad_fused_workspace_bytes(t, R)
    if ws["wgrad"].numel() * 4 < need:
        ws["wgrad"] = torch.empty(_round_up(need, 16) // 4, dtype=torch.float32, device=ws["wgrad"].device)
    # The two weight-gradient launches read what the backward launch wrote and nothing of each other: on two streams the
    # head's (n_groups x splits x 2 workgroups) fills the tail of the trunk's and the other way round (RG_QR_WGRAD_STREAMS=0:
    # one after the other, as in rounds 2-3).
    side = None
    if dz32.is_cuda and two_streams:
        main = torch.cuda.current_stream()
        side = side_stream(dz32.device)
        side.wait_stream(main)
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        ops.group_head_wgrad(head.dz_frag, ws["act_frag"][n - 1], space.row_begin, head.G, head.Ng, head.H, splits, dw[n - 1],
                             wgrad_ws, x3=head.x3, rows=R)
    ops._run("rg_mlp_wgrad_fused", dict(B=R, dims=tuple(st.dims[:n])),
             lambda: lib.rg_mlp_wgrad_fused(t, R, ws["wgrad"].data_ptr(), ws["wgrad"].numel() * 4, L.stream_ptr()))
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)


class FusedUpdate:
    """Adam [+ the soft update of a target copy] + bf16 re-staging of the network's (and the target's) MFMA fragments
    in ONE launch (rg_mlp_update_fused) for a network on the fused bf16 kernels whose optimizer is one FusedAdam group
    over exactly its parameters.  Arithmetic per element is that of the separate launches (rg_adam_step,
    rg_soft_update, rg_mlp_stage_weights_fused): bit-identical results.  `make` returns None when the shape of the
    step is not the plain one."""

    @classmethod
    def make(cls, adam, params, lin, stack, target_params=None, target_stack=None, tau=None):
        ok = (isinstance(stack, FusedMLP) and len(adam.param_groups) == 1
              and len(adam.param_groups[0]["params"]) == len(params)
              and all(a is b for a, b in zip(adam.param_groups[0]["params"], params)))
        if ok and target_params is not None:
            ok = (isinstance(target_stack, FusedMLP) and target_stack.x3 == stack.x3 and len(target_params) == len(params)
                  and all(t is not s for t, s in zip(target_params, params)))
        if not ok:
            return None
        slab = ensure_slab(params)
        tslab = ensure_slab(target_params) if target_params is not None else None
        if tslab is not None and (tslab.offsets != slab.offsets or tslab.total != slab.total):
            return None
        self = cls()
        self.adam, self.params, self.slab, self.stack = adam, list(params), slab, stack
        self.tparams, self.tslab, self.tstack, self.tau = target_params, tslab, target_stack, tau
        index = {id(p): i for i, p in enumerate(params)}
        d = L.MlpUpdateDesc()
        d.n_layers = len(lin)
        d.x3 = int(stack.x3)  # split-bf16: both planes of every fragment set are re-staged
        for i, v in enumerate(stack.dims):
            d.dims[i] = v
        for l, layer in enumerate(lin):
            d.w_off[l] = slab.offsets[index[id(layer.weight)]]
            d.b_off[l] = slab.offsets[index[id(layer.bias)]]
        self.desc = d
        return self

    def staged(self) -> bool:
        """fragments staged at least once (their padding is written by the first staging), slabs in place"""
        st, ts = self.stack, self.tstack
        if any(w is None for w in st._wf) or any(w is None for w in st._wb):
            return False
        if ts is not None and (any(w is None for w in ts._wf) or not self.tslab.is_bound()):
            return False
        return self.adam.moments_for(0)[0] is self.slab

    def step(self, grad_scale: float = 1.0):
        import math

        from .optimizer import _bump, capturing

        adam, d, st, ts = self.adam, self.desc, self.stack, self.tstack
        assert all(p.grad is not None for p in self.params), "the fused update reads dense gradients from the slab"
        slab, exp_avg, exp_avg_sq = adam.moments_for(0)
        group = adam.param_groups[0]
        beta1, beta2 = group["betas"]
        sched = adam.schedule_for(0)
        if sched is None:
            steps = {adam.advance(0, i) for i in range(len(slab.params))}
            if len(steps) != 1:
                raise RuntimeError("fused update needs every parameter at the same Adam step")
            step = steps.pop()
        d.param, d.grad = slab.data.data_ptr(), slab.grad.data_ptr()
        d.exp_avg, d.exp_avg_sq = exp_avg.data_ptr(), exp_avg_sq.data_ptr()
        d.target = self.tslab.data.data_ptr() if ts is not None else None
        for l in range(d.n_layers):
            d.wfrag_fwd[l], d.wfrag_bwd[l] = st._wf[l].data_ptr(), st._wb[l].data_ptr()
            d.target_wfrag_fwd[l] = ts._wf[l].data_ptr() if ts is not None else None
        tau = self.tau if ts is not None else 0.0
        if sched is not None:  # graph-safe: lr and the bias corrections come from HBM, the step is counted there
            if not capturing():
                sched.set_lr(group["lr"])
                sched.pending += 1
            ops.tick_fence(sched.buf)
            ops._run("rg_mlp_update_fused", dict(P=slab.total),
                     lambda: L.lib().rg_mlp_update_fused_sched(d, beta1, beta2, group["eps"], group["weight_decay"],
                                                               grad_scale, tau, sched.buf.data_ptr(), L.stream_ptr()))
            ops.sched_tick(sched.buf)
        else:
            ops._run("rg_mlp_update_fused", dict(P=slab.total),
                     lambda: L.lib().rg_mlp_update_fused(d, group["lr"], beta1, beta2, group["eps"], group["weight_decay"],
                                                         1.0 - beta1**step, math.sqrt(1.0 - beta2**step), grad_scale,
                                                         tau, L.stream_ptr()))
        _bump(slab.params)
        if ts is not None:
            _bump(self.tparams)
        # the fragments are current for the bumped versions: no separate staging launch
        for st_, need_t in ((st, True), (ts, False)):
            if st_ is not None:
                st_._staged_versions = tuple((w._version, getattr(w, "_rg_version", 0)) for w in st_.weights) + (need_t,)
                st_._wsrc_ptrs = [w.data_ptr() for w in st_.weights]


def dx_save(stack):
    """`save` argument of a forward whose backward will be input-gradient only (skip_wgrad): the fused kernels then
    keep the sign planes instead of the activations"""
    return SAVE_FOR_DX if isinstance(stack, FusedMLP) else True


def make_stack(weights, biases, acts: List[int], precision: int, layer_norms=None, batch_norms=None, dropouts=None,
               residuals=None, training=None):
    """Engine selection: the fused bf16-MFMA kernels when the shape allows (plain bf16 operands for
    PREC_BF16, split-bf16 for PREC_BF16X3), else the per-layer GEMMs.  A PREC_BF16X3 stack whose shape the
    fused kernels do not serve runs on the exact-fp32 MFMA GEMMs: the accuracy class is what was asked for.
    Batch-norm / dropout / residual layers (off in every BASELINE configuration) go to engine_general."""
    has_ln = layer_norms is not None and any(ln is not None for ln in layer_norms)
    if ((batch_norms is not None and any(b is not None for b in batch_norms)) or (dropouts is not None and any(dropouts))
            or (residuals is not None and any(residuals))):
        from .engine_general import GeneralFCStack

        return GeneralFCStack(weights, biases, acts, L.PREC_F32 if precision == L.PREC_BF16X3 else precision,
                              layer_norms=layer_norms, batch_norms=batch_norms, dropouts=dropouts, residuals=residuals,
                              training=training)
    if not has_ln and precision in (L.PREC_BF16, L.PREC_BF16X3) and FusedMLP.supported(weights, acts):
        return FusedMLP(weights, biases, acts, x3=precision == L.PREC_BF16X3)
    # (a stack with LayerNorm between its layers is not of the fused kernels' shape: per-layer GEMMs + rg_layer_norm_*)
    return FCStack(weights, biases, acts, L.PREC_F32 if precision == L.PREC_BF16X3 else precision, layer_norms=layer_norms)


def grad_views(fc, slab, params):
    """(dw, db): views of the flat gradient slab for every linear layer of `fc`, in layer order; a stack with
    LayerNorm also gets the destinations of its gamma / beta gradients bound (FCStack.bind_ln_grads)"""
    index = {id(p): i for i, p in enumerate(params)}
    view = lambda p: slab.view(slab.grad, index[id(p)])  # noqa: E731
    lin = fc.linears()
    lns = fc.layer_norms() if hasattr(fc, "layer_norms") else []
    if any(ln is not None for ln in lns):
        fc.stack().bind_ln_grads([(view(ln.weight), view(ln.bias)) if ln is not None else None for ln in lns])
    bns = fc.batch_norms() if hasattr(fc, "batch_norms") else []
    if any(bn is not None for bn in bns):
        fc.stack().bind_bn_grads([(view(bn.weight), view(bn.bias)) if bn is not None else None for bn in bns])
    return [view(l.weight) for l in lin], [view(l.bias) for l in lin]
