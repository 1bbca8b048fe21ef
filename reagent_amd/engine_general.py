"""Per-layer engine for FullyConnectedNetwork's optional layer components — batch-norm on a layer's input, layer-norm,
dropout after the activation, the residual wrapper (reagent/models/fully_connected_network.py:101-153,
reagent/models/residual_wrapper.py:21-22).  They are off in every BASELINE configuration, so this path favours
plainness over launches: the GEMMs run in the stack's precision (rg_fc_*), everything between them is an fp32 row
sweep (rg_batch_norm_*, rg_layer_norm_*, rg_dropout, rg_act_backward, rg_add_cols).

    x_i (fp32) -> [BatchNorm1d] -> Linear -> [LayerNorm] -> activation -> [Dropout] -> [+ x_i] = x_{i+1}
"""
from typing import List, Optional

import torch

from . import _lib as L
from . import ops
from .engine import FCStack, _mat

_LINEAR = L.ACT["linear"]


class GeneralFCStack(FCStack):
    def __init__(self, weights, biases, acts: List[int], precision: int, layer_norms=None, batch_norms=None,
                 dropouts=None, residuals=None, training=None):
        """batch_norms[i]: None or an nn.BatchNorm1d-like holder (weight, bias, running_mean, running_var,
        num_batches_tracked, eps, momentum) applied to layer i's input; dropouts[i]: drop probability after layer i's
        activation; residuals[i]: layer i's output is x_i + layer(x_i); training: callable -> bool (the owning module's
        mode: batch statistics and dropout masks in training mode, running statistics and no dropout in eval mode)"""
        super().__init__(weights, biases, acts, precision, layer_norms=layer_norms)
        n = self.L
        self.bns = list(batch_norms) if batch_norms is not None else [None] * n
        self.drops = [float(p) for p in dropouts] if dropouts is not None else [0.0] * n
        self.res = [bool(r) for r in residuals] if residuals is not None else [False] * n
        assert len(self.bns) == len(self.drops) == len(self.res) == n
        for i, r in enumerate(self.res):
            assert not r or self.dims[i] == self.dims[i + 1], "a residual layer keeps its width"
        self._training = training if training is not None else (lambda: True)
        self._bn_grads = [None] * n
        self._bufs = {}
        self._calls = 0            # dropout: one Philox offset per forward
        self._fwd_training = True  # mode of the last saving forward (its backward follows it)
        self._need_dx = self.bns[0] is not None  # BatchNorm's parameter gradients need d loss / d (its output)
        GeneralFCStack._instances += 1
        self._rank = GeneralFCStack._instances
        self._outs = [None] * n    # x_{i+1} of the last saving forward
        self.stat_updates = 1      # running-statistics updates per forward (2: a module the reference evaluates twice)

    # ---- plumbing ------------------------------------------------------------------------
    def set_need_input_grad(self, flag: bool):
        super().set_need_input_grad(bool(flag) or self.bns[0] is not None)

    def bind_bn_grads(self, dst):
        """dst[i] = (dgamma, dbeta) fp32 destinations (gradient-slab views) of layer i's BatchNorm, or None"""
        assert len(dst) == self.L and all((d is None) == (bn is None) for d, bn in zip(dst, self.bns))
        self._bn_grads = list(dst)

    def stage_input(self, x32: torch.Tensor, need_transposed: bool):
        """the network input stays fp32 (batch-norm and the residual add read it); its compute-type and transposed
        copies are made inside forward()"""
        x = x32 if x32.dtype == torch.float32 else x32.float()
        return (x if x.stride(-1) == 1 else x.contiguous()), None

    def _ensure_ws(self, batch: int, device):
        if self._batch != batch or self._ws.get("device") != device:
            self._bufs = {}
        super()._ensure_ws(batch, device)

    def _buf(self, role: str, i: int, rows: int, cols: int, dtype=torch.float32):
        key = (role, i)
        t = self._bufs.get(key)
        if t is None:
            dev = self._ws["device"]
            t = torch.empty(rows, dtype=dtype, device=dev) if cols == 0 else _mat(rows, cols, dtype, dev)
            self._bufs[key] = t
        return t

    _instances = 0

    def _seed(self) -> int:
        """dropout key: torch's seed and this stack's construction rank (reproducible under torch.manual_seed)"""
        return (int(torch.initial_seed()) * 1000003 + self._rank) & (2 ** 63 - 1)

    # ---- forward -------------------------------------------------------------------------
    def forward(self, xc: torch.Tensor, out32: torch.Tensor, save: bool = False):
        """xc: fp32 [B, in] (stage_input); out32: fp32 [B, out_last] (written).  save: keep what backward() reads."""
        B, dev = xc.shape[0], xc.device
        self._ensure_ws(B, dev)
        training = bool(self._training())
        if save:
            self._fwd_training = training
        self._calls += 1
        # a saving forward and the non-saving ones around it (target / next-state evaluations) keep separate buffers:
        # backward() reads the saving one's
        tag = "/s" if save else "/n"
        buf = lambda role, *a: self._buf(role + tag, *a)  # noqa: E731
        x_in = xc
        if save:
            self._x0 = xc
        for i in range(self.L):
            last = i == self.L - 1
            in_f, out_f = self.dims[i], self.dims[i + 1]
            bn, ln = self.bns[i], self.lns[i]
            drop = self.drops[i] if training else 0.0
            x = x_in
            if bn is not None:
                x = buf("xbn", i, B, in_f)
                if training:
                    ws = buf("bn_ws", i, L.lib().rg_batch_norm_workspace_bytes(B, in_f) // 8 + 1, 0, torch.float64)
                    ops.batch_norm_forward(x_in, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                           True, 0.1 if bn.momentum is None else bn.momentum, bn.eps, x,
                                           buf("bn_mean", i, in_f, 0), buf("bn_rstd", i, in_f, 0), ws,
                                           stat_updates=self.stat_updates)
                    bn.num_batches_tracked.add_(self.stat_updates)
                else:
                    ops.batch_norm_forward(x_in, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                           False, 0.0, bn.eps, x)
            # compute-type operand of the GEMM (+ its transposed copy for the weight gradient)
            xt = buf("xt", i, in_f, B, self.cdtype) if save else None
            if self.precision == L.PREC_F32:
                xop = x
                if xt is not None:
                    ops.transpose_cast(x, None, xt)
            else:
                xop = buf("xc", i, B, in_f, self.cdtype)
                ops.transpose_cast(x, xop, xt)
            tail = drop > 0.0 or self.res[i]  # something still follows the activation
            a = out32 if (last and not tail) else buf("act", i, B, out_f)
            if ln is not None:
                z = buf("z", i, B, out_f)
                ops.fc_forward(xop, self._wc[i], self.biases[i].detach(), _LINEAR, self.precision, y=None, y32=z, yt=None)
                ops.layer_norm_forward(z, ln.weight.detach(), ln.bias.detach(), ln.eps, self.acts[i], y=None, y32=a,
                                       mean=buf("ln_mean", i, B, 0), rstd=buf("ln_rstd", i, B, 0))
            else:
                ops.fc_forward(xop, self._wc[i], self.biases[i].detach(), self.acts[i], self.precision, y=None, y32=a, yt=None)
            if save:
                self._bufs[("act_out/s", i)] = a
            cur = a
            if drop > 0.0:
                dst = out32 if (last and not self.res[i]) else buf("drop", i, B, out_f)
                ops.dropout(cur, drop, buf("keep", i, B * out_f, 0, torch.uint8), dst, seed=self._seed(),
                            offset=self._calls * self.L + i)
                cur = dst
            if self.res[i]:
                dst = out32 if last else buf("sum", i, B, out_f)
                ops.add_cols(cur, x_in, dst)
                cur = dst
            x_in = cur
            if save:
                self._outs[i] = cur
        return out32

    # ---- backward ------------------------------------------------------------------------
    def backward(self, dout32: torch.Tensor, xt, dw: List[torch.Tensor], db: List[torch.Tensor],
                 dx32: Optional[torch.Tensor] = None, skip_wgrad: bool = False, out32: Optional[torch.Tensor] = None):
        """Gradients of a scalar loss given d loss / d output (fp32 [B, out_last]); follows a forward(..., save=True) on
        the same batch (xt is not used: the transposed operands were kept by the forward).  dw / db: fp32 slab views;
        dx32 (optional): fp32 [B, in] destination of the input gradient; skip_wgrad: input gradient only."""
        B = dout32.shape[0]
        training = self._fwd_training
        g = dout32
        for i in range(self.L - 1, -1, -1):
            in_f, out_f = self.dims[i], self.dims[i + 1]
            bn, ln = self.bns[i], self.lns[i]
            drop = self.drops[i] if training else 0.0
            g_skip = g if self.res[i] else None
            if drop > 0.0:
                t = self._buf("g_drop", i, B, out_f)
                ops.dropout(g, drop, self._buf("keep/s", i, B * out_f, 0, torch.uint8), t, generate=False)
                g = t
            if self.acts[i] != _LINEAR:
                t = self._buf("g_act", i, B, out_f)
                ops.act_backward(g, self._bufs[("act_out/s", i)], self.acts[i], t)
                g = t
            dz = self._buf("dz", i, B, out_f, self.cdtype)
            dzt = self._buf("dzt", i, out_f, B, self.cdtype)
            if ln is not None:
                dst = self._ln_grads[i]
                if dst is None or skip_wgrad:
                    if dst is None and not skip_wgrad:
                        raise RuntimeError("LayerNorm gradients have no destination: the trainer did not call bind_ln_grads")
                    s = self._buf("ln_scratch", i, 2 * out_f, 0)
                    dst = (s[:out_f], s[out_f:])
                ws = self._buf("ln_ws", i, L.lib().rg_layer_norm_backward_workspace_bytes(B, out_f) // 4 + 4, 0)
                ops.layer_norm_backward(g, self._buf("z/s", i, B, out_f), self._buf("ln_mean/s", i, B, 0),
                                        self._buf("ln_rstd/s", i, B, 0), ln.weight.detach(), dst[0], dst[1], ws, dz=dz)
                ops.transpose_cast(dz, None, dzt)
            else:
                ops.transpose_cast(g, dz, dzt)
            if not skip_wgrad:
                ops.fc_wgrad(dzt, self._buf("xt/s", i, in_f, B, self.cdtype), dw[i], db[i], self._ws["wgrad"], self.precision)
            if i == 0 and dx32 is None and bn is None:
                break
            gx = self._buf("gx", i, B, in_f)
            direct = i == 0 and bn is None and not self.res[i]  # the GEMM can write the caller's buffer
            ops.fc_dgrad(dz, self._wtc[i], None, _LINEAR, self.precision, dx=None, dx32=dx32 if direct else gx, dxt=None)
            if direct:
                break
            if bn is not None:
                dst = self._bn_grads[i]
                if dst is None or skip_wgrad:
                    if dst is None and not skip_wgrad:
                        raise RuntimeError("BatchNorm gradients have no destination: the trainer did not call bind_bn_grads")
                    s = self._buf("bn_scratch", i, 2 * in_f, 0)
                    dst = (s[:in_f], s[in_f:])
                need_dx = i > 0 or dx32 is not None
                x_in = self._x0 if i == 0 else self._outs[i - 1]
                ws = self._buf("bn_ws", i, L.lib().rg_batch_norm_workspace_bytes(B, in_f) // 8 + 1, 0, torch.float64)
                t = (dx32 if (i == 0 and not self.res[i]) else self._buf("gbn", i, B, in_f)) if need_dx else None
                if training:
                    ops.batch_norm_backward(gx, x_in, bn.weight.detach(), self._buf("bn_mean/s", i, in_f, 0),
                                            self._buf("bn_rstd/s", i, in_f, 0), None, True, bn.eps, ws, t, dst[0], dst[1])
                else:
                    ops.batch_norm_backward(gx, x_in, bn.weight.detach(), bn.running_mean, None, bn.running_var, False,
                                            bn.eps, ws, t, dst[0], dst[1])
                if not need_dx:
                    break
                gx = t
            if self.res[i]:
                t = dx32 if i == 0 else self._buf("gsum", i, B, in_f)
                ops.add_cols(gx, g_skip, t)
                gx = t
            g = gx
