"""DuelingQNetwork (reagent/models/dueling_q_network.py:16-120): the reference's default Q-network for discrete DQN
and (with `num_atoms`) QR-DQN — a shared trunk, an advantage stream and a value stream, each a FullyConnectedDQN:

    shared_state = shared_network(state)                                  [B, E]   (linear output layer)
    value        = value_network(shared_state)                            [B, 1] or [B, 1, N]
    raw_adv      = advantage_network(shared_state)                        [B, A] or [B, A, N]
    q            = value + raw_adv - raw_adv.mean(all dims but the batch)

On the HIP engine the three streams are three FC stacks (fused or per-layer, `engine.make_stack`) composed by
`DuelingStack` with the FCStack interface, so DQNTrainer / QRDQNTrainer / the CRR critics drive a dueling network
through the same native step: forward = 3 stack forwards + rg_dueling_combine; backward = rg_dueling_split, the two
stream backwards (each returning its input gradient), their sum, the trunk's backward.
"""
from typing import List, Optional

import torch

from .. import _lib as L
from .. import ops
from ..core import types as rlt
from .base import ModelBase
from .dqn import INVALID_ACTION_CONSTANT, FullyConnectedDQN


class DuelingStack:
    """FCStack-shaped composite over the (shared, advantage, value) stacks"""

    def __init__(self, shared, advantage, value, num_actions: int, num_atoms: int):
        self.s, self.a, self.v = shared, advantage, value
        self.A, self.N = num_actions, num_atoms
        self.precision = shared.precision
        self.dims = [shared.dims[0], num_actions * num_atoms]  # what callers size their buffers with
        self.ns, self.na = shared.L, advantage.L
        self._B = -1
        self.a.set_need_input_grad(True)
        self.v.set_need_input_grad(True)

    def set_need_input_grad(self, flag: bool):
        self.s.set_need_input_grad(flag)

    def _bind(self, which: str, dst):
        """per-layer (dgamma, dbeta) destinations of the three streams' normalisation layers, in layer order"""
        ns, na = self.ns, self.na
        for st, part in ((self.s, dst[:ns]), (self.a, dst[ns:ns + na]), (self.v, dst[ns + na:])):
            if any(d is not None for d in part):
                getattr(st, which)(part)

    def bind_ln_grads(self, dst):
        self._bind("bind_ln_grads", dst)

    def bind_bn_grads(self, dst):
        self._bind("bind_bn_grads", dst)

    def stage_weights(self, need_transposed: bool = True, force: bool = False):
        for st in (self.s, self.a, self.v):
            st.stage_weights(need_transposed=need_transposed, force=force)

    def stage_input(self, x32: torch.Tensor, need_transposed: bool):
        return self.s.stage_input(x32, need_transposed)

    def _ws(self, B, dev):
        if self._B != B or self._e.device != dev:
            f = dict(dtype=torch.float32, device=dev)
            E = self.s.dims[-1]
            self._e, self._de_a, self._de_v, self._de = (torch.empty(B, E, **f) for _ in range(4))
            self._adv, self._dadv = torch.empty(B, self.A * self.N, **f), torch.empty(B, self.A * self.N, **f)
            self._val, self._dval = torch.empty(B, self.N, **f), torch.empty(B, self.N, **f)
            self._B = B

    def forward(self, xc: torch.Tensor, out32: torch.Tensor, save: bool = False):
        B = xc.shape[0]
        self._ws(B, xc.device)
        self.s.forward(xc, self._e, save=save)
        ea, ea_t = self.a.stage_input(self._e, need_transposed=save)
        self.a.forward(ea, self._adv, save=save)
        ev, ev_t = self.v.stage_input(self._e, need_transposed=save)
        self.v.forward(ev, self._val, save=save)
        if save:  # what the streams' weight gradients read; later non-saving forwards leave it alone
            self._ea_t, self._ev_t = ea_t, ev_t
        ops.dueling_combine(self._val, self._adv, self.A, self.N, out32)
        return out32

    def backward(self, dout32: torch.Tensor, xt, dw: List[torch.Tensor], db: List[torch.Tensor],
                 dx32: Optional[torch.Tensor] = None, skip_wgrad: bool = False, out32: Optional[torch.Tensor] = None):
        ns, na = self.ns, self.na
        ops.dueling_split(dout32, self.A, self.N, self._dadv, self._dval)
        sl = (lambda lst, lo, hi: lst[lo:hi] if lst is not None else None)
        self.a.backward(self._dadv, self._ea_t, sl(dw, ns, ns + na), sl(db, ns, ns + na), dx32=self._de_a, skip_wgrad=skip_wgrad)
        self.v.backward(self._dval, self._ev_t, sl(dw, ns + na, None), sl(db, ns + na, None), dx32=self._de_v,
                        skip_wgrad=skip_wgrad)
        ops.add_cols(self._de_a, self._de_v, self._de)
        self.s.backward(self._de, xt, sl(dw, 0, ns), sl(db, 0, ns), dx32=dx32, skip_wgrad=skip_wgrad)


class _DuelingFC:
    """what the trainers read through `net.fc`: the ordered linear layers (= parameter order) and the engine"""

    def __init__(self, net: "DuelingQNetwork"):
        self._net = net
        self._stack = None

    @property
    def precision(self):
        return self._net.shared_network.fc.precision

    @property
    def activation_names(self):
        n = self._net
        return (list(n.shared_network.fc.activation_names) + list(n.advantage_network.fc.activation_names)
                + list(n.value_network.fc.activation_names))

    def linears(self):
        n = self._net
        return n.shared_network.fc.linears() + n.advantage_network.fc.linears() + n.value_network.fc.linears()

    def layer_norms(self):
        n = self._net
        return n.shared_network.fc.layer_norms() + n.advantage_network.fc.layer_norms() + n.value_network.fc.layer_norms()

    def batch_norms(self):
        n = self._net
        return n.shared_network.fc.batch_norms() + n.advantage_network.fc.batch_norms() + n.value_network.fc.batch_norms()

    def stack(self) -> DuelingStack:
        n = self._net
        subs = (n.shared_network.fc.stack(), n.advantage_network.fc.stack(), n.value_network.fc.stack())
        if self._stack is None or any(a is not b for a, b in zip(subs, (self._stack.s, self._stack.a, self._stack.v))):
            self._stack = DuelingStack(*subs, num_actions=n.advantage_network.output_dim,
                                       num_atoms=n.advantage_network.num_atoms or 1)
        return self._stack


class DuelingQNetwork(ModelBase):
    def __init__(self, *, shared_network: ModelBase, advantage_network: ModelBase, value_network: ModelBase) -> None:
        super().__init__()
        self.shared_network = shared_network
        assert isinstance(shared_network.input_prototype(), rlt.FeatureData), "shared_network should expect FeatureData as input"
        self.advantage_network = advantage_network
        self.value_network = value_network
        _check_connection(self)
        self._name = "unnamed"

    @classmethod
    def make_fully_connected(cls, state_dim: int, action_dim: int, layers: List[int], activations: List[str],
                             num_atoms: Optional[int] = None, use_batch_norm: bool = False):
        """dueling_q_network.py:44-86"""
        assert len(layers) > 0, "Must have at least one layer"
        state_embedding_dim = layers[-1]
        assert state_embedding_dim % 2 == 0, "The last size must be divisible by 2"
        shared_network = FullyConnectedDQN(state_dim, state_embedding_dim, sizes=layers[:-1], activations=activations[:-1],
                                           normalized_output=True, use_batch_norm=use_batch_norm)
        advantage_network = FullyConnectedDQN(state_embedding_dim, action_dim, sizes=[state_embedding_dim // 2],
                                              activations=activations[-1:], num_atoms=num_atoms)
        value_network = FullyConnectedDQN(state_embedding_dim, 1, sizes=[state_embedding_dim // 2],
                                          activations=activations[-1:], num_atoms=num_atoms)
        return cls(shared_network=shared_network, advantage_network=advantage_network, value_network=value_network)

    @property
    def fc(self) -> _DuelingFC:  # not a submodule: the engine view of the three streams
        view = self.__dict__.get("_fc_view")
        if view is None:
            view = self.__dict__["_fc_view"] = _DuelingFC(self)
        return view

    def __deepcopy__(self, memo):
        from copy import deepcopy

        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_fc_view":  # the engine view (and its workspaces) belongs to the original
                new.__dict__[k] = deepcopy(v, memo)
        return new

    def input_prototype(self):
        return self.shared_network.input_prototype()

    def forward(self, state: rlt.FeatureData, possible_actions_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = state.float_features
        L.require_cuda(x, "state.float_features")
        st = self.fc.stack()
        st.stage_weights(need_transposed=False)
        xc, _ = st.stage_input(x if x.dtype in (torch.float32, torch.bfloat16) else x.float(), need_transposed=False)
        A, N = st.A, st.N
        q = torch.empty(x.shape[0], A * N, dtype=torch.float32, device=x.device)
        st.forward(xc, q, save=False)
        if self.advantage_network.num_atoms is not None:
            q = q.view(-1, A, N)
        if possible_actions_mask is not None:
            q = q + (1 - possible_actions_mask.float()) * INVALID_ACTION_CONSTANT
        return q


def _check_connection(model):
    """dueling_q_network.py:_check_connection: the streams must take what the trunk emits"""
    emb = model.shared_network.output_dim
    for name in ("advantage_network", "value_network"):
        net = getattr(model, name)
        assert net.state_dim == emb, f"{name} expects {net.state_dim} features, the shared network emits {emb}"
    assert model.value_network.output_dim == 1
