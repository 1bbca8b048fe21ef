"""TEST INFRASTRUCTURE (oracle).  Drives the UNMODIFIED reference trainers from /root/reference.

Used only in the build container (where /root/reference exists) to (a) generate the golden
fixtures committed under tests/golden/ (oracle/make_golden.py) and (b) validate the restated
oracle (oracle/restated.py).  It emulates the optimizer loop of pytorch-lightning 1.6.0 — a pinned,
un-vendored dependency of the reference (setup.cfg:32) whose call site is
reagent/workflow/utils.py:155-165 — as described in SURVEY.md §3.3 / §8(c):
    for each optimizer i (in configure_optimizers() order):
        toggle_optimizer(i)  (requires_grad=False on parameters owned only by other optimizers)
        loss = training_step(batch, batch_idx, i); opt.zero_grad(); loss.backward(); opt.step()
        untoggle
No reference test pins these stepping semantics ("parity unpinned" at the Lightning boundary).
"""
from typing import List

import torch

from . import stubs


def _install():
    if stubs.runtime_root() is None:
        raise RuntimeError("reference not available: neither /root/reference (build container) nor oracle/_ref (its byte code, "
                           "python -m oracle.build_ref) is here")
    stubs.install()


class PLLoop:
    """pytorch-lightning 1.6 automatic-optimization loop for ONE module, CPU."""

    def __init__(self, trainer):
        self.trainer = trainer
        self.optimizers = [o["optimizer"] for o in trainer.configure_optimizers()]
        self.batch_idx = 0
        self.last_grads = {}

        class _Logger:  # SACTrainer calls self.logger.log_metrics unconditionally (:343); the last step's values are kept
            def __init__(self):
                self.metrics = {}

            def log_metrics(self, metrics, step=None):
                for k, v in metrics.items():  # DQN's per-action dicts are flattened to "<metric>/<action name>"
                    if isinstance(v, dict):
                        self.log_metrics({f"{k}/{a}": x for a, x in v.items()})
                    elif isinstance(v, torch.Tensor):
                        self.metrics[k] = v.detach().clone()
                    elif isinstance(v, (int, float)):
                        self.metrics[k] = torch.tensor(float(v))

        trainer.logger = _Logger()
        if getattr(trainer, "trainer", None) is None:  # TD3 reads self.trainer.log_every_n_steps (:158)
            import types

            trainer.trainer = types.SimpleNamespace(log_every_n_steps=50)

    def _toggle(self, idx):
        saved = {}
        for opt in self.optimizers:
            for g in opt.param_groups:
                for p in g["params"]:
                    if p not in saved:
                        saved[p] = p.requires_grad
                        p.requires_grad = False
        for g in self.optimizers[idx].param_groups:
            for p in g["params"]:
                p.requires_grad = saved[p]
        return saved

    def step(self, batch) -> List[torch.Tensor]:
        losses = []
        for i, opt in enumerate(self.optimizers):
            saved = self._toggle(i)
            loss = self.trainer.training_step(batch, self.batch_idx, i)
            if loss is not None:  # PL skips the optimizer step when training_step returns None
                opt.zero_grad()
                loss.backward()
                # what autograd handed the optimizer (golden fixtures pin the backward pass with it)
                self.last_grads[i] = [p.grad.detach().clone() if p.grad is not None else None
                                      for g in opt.param_groups for p in g["params"]]
                opt.step()
            for p, rg in saved.items():
                p.requires_grad = rg
            losses.append(loss.detach().clone() if loss is not None else None)
        self.batch_idx += 1
        return losses


def make_rl_parameters(**kw):
    _install()
    from reagent.core.parameters import RLParameters

    return RLParameters(**kw)


def make_adam(lr=1e-3, **kw):
    _install()
    from reagent.optimizer.union import Optimizer__Union

    return Optimizer__Union.default(lr=lr, **kw)


def build_dqn(state_dim, num_actions, sizes, activations, rl_kwargs, lr, double_q=True, seed=0,
              num_atoms=None, cpe_metrics=None, bcq_threshold=None, dueling=False, layer_norm=False, batch_norm=False):
    """Reference FullyConnectedDQN (+ target) and DQNTrainer / QRDQNTrainer.  cpe_metrics: None = CPE
    off; a list of extra metric names (may be empty) = calc_cpe_in_training with reward_network,
    q_network_cpe and its target of output width (len(cpe_metrics) + 1) * num_actions
    (model_managers/discrete/discrete_dqn.py:84-104).  dueling=True: DuelingQNetwork.make_fully_connected over the
    same sizes (what net_builder/discrete_dqn/dueling.py:38-46 builds — the reference's default DQN net builder)."""
    _install()
    from reagent.core.parameters import EvaluationParameters
    from reagent.models.dqn import FullyConnectedDQN

    torch.manual_seed(seed)
    if dueling:
        from reagent.models.dueling_q_network import DuelingQNetwork

        q = DuelingQNetwork.make_fully_connected(state_dim, num_actions, sizes, activations, num_atoms=num_atoms,
                                                 use_batch_norm=batch_norm)
    else:  # layer_norm: Linear -> LayerNorm -> activation on the hidden layers (fully_connected_network.py:128-130)
        # batch_norm: BatchNorm1d on every layer's input (:107-108); the trainer leaves all networks in training mode
        q = FullyConnectedDQN(state_dim, num_actions, sizes, activations, num_atoms=num_atoms, use_layer_norm=layer_norm,
                              use_batch_norm=batch_norm)
    qt = q.get_target_network()
    actions = [str(i) for i in range(num_actions)]
    cpe = cpe_metrics is not None
    common = dict(actions=actions, rl=make_rl_parameters(**rl_kwargs), double_q_learning=double_q,
                  optimizer=make_adam(lr), evaluation=EvaluationParameters(calc_cpe_in_training=cpe))
    if num_atoms is None:
        from reagent.training.dqn_trainer import DQNTrainer

        if cpe:
            n_out = (len(cpe_metrics) + 1) * num_actions
            reward_net = FullyConnectedDQN(state_dim, n_out, sizes, activations)
            q_cpe = FullyConnectedDQN(state_dim, n_out, sizes, activations)
            trainer = DQNTrainer(q, qt, reward_net, q_network_cpe=q_cpe, q_network_cpe_target=q_cpe.get_target_network(),
                                 metrics_to_score=list(cpe_metrics), **common)
        elif bcq_threshold is not None:
            # batch-constrained q-learning: `imitator` is any module mapping state features to action logits
            # (imitator_training.py:12-25); a two-layer torch net, its weights saved with the fixture
            from reagent.training.dqn_trainer import BCQConfig

            imitator = torch.nn.Sequential(torch.nn.Linear(state_dim, 16), torch.nn.ReLU(), torch.nn.Linear(16, num_actions))
            trainer = DQNTrainer(q, qt, None, imitator=imitator, bcq=BCQConfig(drop_threshold=bcq_threshold), **common)
        else:
            trainer = DQNTrainer(q, qt, None, **common)
    else:
        from reagent.training.qrdqn_trainer import QRDQNTrainer

        if cpe:  # the CPE networks are plain (non-distributional) FC nets, discrete_qrdqn.py:84-103
            n_out = (len(cpe_metrics) + 1) * num_actions
            reward_net = FullyConnectedDQN(state_dim, n_out, sizes, activations)
            q_cpe = FullyConnectedDQN(state_dim, n_out, sizes, activations)
            trainer = QRDQNTrainer(q, qt, metrics_to_score=list(cpe_metrics), reward_network=reward_net,
                                   q_network_cpe=q_cpe, q_network_cpe_target=q_cpe.get_target_network(),
                                   num_atoms=num_atoms, cpe_optimizer=make_adam(lr), **common)
        else:
            trainer = QRDQNTrainer(q, qt, num_atoms=num_atoms, **common)
    return trainer


def build_sac(state_dim, action_dim, sizes, activations, rl_kwargs, lr, seed=0, value=False, crr=None,
              critic_layer_norm=False, actor_layer_norm=False, critic_batch_norm=False, actor_batch_norm=False, **trainer_kw):
    """value=True adds a value network (FloatFeatureFullyConnected state -> 1, what the reference's value net builder
    makes); crr = CRRWeightFn arguments."""
    _install()
    from reagent.models.actor import GaussianFullyConnectedActor
    from reagent.models.critic import FullyConnectedCritic
    from reagent.training.sac_trainer import CRRWeightFn, SACTrainer

    torch.manual_seed(seed)
    actor = GaussianFullyConnectedActor(state_dim, action_dim, sizes, activations, use_layer_norm=actor_layer_norm,
                                        use_batch_norm=actor_batch_norm)
    q1 = FullyConnectedCritic(state_dim, action_dim, sizes, activations, use_layer_norm=critic_layer_norm,
                              use_batch_norm=critic_batch_norm)
    q2 = FullyConnectedCritic(state_dim, action_dim, sizes, activations, use_layer_norm=critic_layer_norm,
                              use_batch_norm=critic_batch_norm)
    if value:
        from reagent.models.fully_connected_network import FloatFeatureFullyConnected

        trainer_kw["value_network"] = FloatFeatureFullyConnected(state_dim, 1, sizes, activations)
        trainer_kw["value_network_optimizer"] = make_adam(lr)
    if crr is not None:
        trainer_kw["crr_config"] = CRRWeightFn(**crr)
    return SACTrainer(actor, q1, q2, rl=make_rl_parameters(**rl_kwargs), q_network_optimizer=make_adam(lr),
                      actor_network_optimizer=make_adam(lr), alpha_optimizer=make_adam(lr), **trainer_kw)


def build_td3(state_dim, action_dim, sizes, activations, rl_kwargs, lr, seed=0, actor_batch_norm=False, **trainer_kw):
    """actor_batch_norm: BatchNorm1d on the inputs of the actor's layers.  (Batch-normed CRITICS make TD3's actor loss
    -mean_b q1(s, actor(s)) constant in the action — the batch mean of a batch-normed layer's output is its beta — so that
    configuration has an exactly-zero actor gradient and pins nothing but rounding noise.)"""
    _install()
    from reagent.models.actor import FullyConnectedActor
    from reagent.models.critic import FullyConnectedCritic
    from reagent.training.td3_trainer import TD3Trainer

    torch.manual_seed(seed)
    actor = FullyConnectedActor(state_dim, action_dim, sizes, activations, use_batch_norm=actor_batch_norm)
    q1 = FullyConnectedCritic(state_dim, action_dim, sizes, activations)
    q2 = FullyConnectedCritic(state_dim, action_dim, sizes, activations)
    return TD3Trainer(actor, q1, q2, rl=make_rl_parameters(**rl_kwargs), q_network_optimizer=make_adam(lr),
                      actor_network_optimizer=make_adam(lr), **trainer_kw)


def build_c51(state_dim, num_actions, sizes, activations, rl_kwargs, lr, num_atoms, qmin, qmax, double_q=True, seed=0):
    """CategoricalDQN over FullyConnectedDQN(num_atoms) as net_builder/categorical_dqn/categorical.py:36-56"""
    _install()
    from reagent.models.categorical_dqn import CategoricalDQN
    from reagent.models.dqn import FullyConnectedDQN
    from reagent.training.c51_trainer import C51Trainer

    torch.manual_seed(seed)
    dist_net = FullyConnectedDQN(state_dim, num_actions, sizes, activations, num_atoms=num_atoms)
    q = CategoricalDQN(dist_net, qmin=qmin, qmax=qmax, num_atoms=num_atoms)
    return C51Trainer(q, q.get_target_network(), actions=[str(i) for i in range(num_actions)],
                      rl=make_rl_parameters(**rl_kwargs), double_q_learning=double_q, num_atoms=num_atoms, qmin=qmin,
                      qmax=qmax, optimizer=make_adam(lr))


def build_crr(state_dim, num_actions, sizes, activations, rl_kwargs, lr, twin=True, cpe_metrics=None, seed=0,
              actor_activation="tanh", **trainer_kw):
    """DiscreteCRRTrainer over FullyConnectedActor (net_builder/discrete_actor/fully_connected.py:36-52)
    and FullyConnectedDQN critics / CPE nets, wired as model_managers/discrete/discrete_crr.py:105-178."""
    _install()
    from reagent.core.parameters import EvaluationParameters
    from reagent.models.actor import FullyConnectedActor
    from reagent.models.dqn import FullyConnectedDQN
    from reagent.training.discrete_crr_trainer import DiscreteCRRTrainer

    torch.manual_seed(seed)
    actor = FullyConnectedActor(state_dim, num_actions, sizes, activations, action_activation=actor_activation)
    q1 = FullyConnectedDQN(state_dim, num_actions, sizes, activations)
    q2 = FullyConnectedDQN(state_dim, num_actions, sizes, activations) if twin else None
    cpe = cpe_metrics is not None
    reward_net = q_cpe = q_cpe_t = None
    if cpe:
        n_out = (len(cpe_metrics) + 1) * num_actions
        reward_net = FullyConnectedDQN(state_dim, n_out, sizes, activations)
        q_cpe = FullyConnectedDQN(state_dim, n_out, sizes, activations)
        q_cpe_t = q_cpe.get_target_network()
    return DiscreteCRRTrainer(
        actor_network=actor, actor_network_target=actor.get_target_network(), q1_network=q1,
        q1_network_target=q1.get_target_network(), reward_network=reward_net, q2_network=q2,
        q2_network_target=q2.get_target_network() if twin else None, q_network_cpe=q_cpe,
        q_network_cpe_target=q_cpe_t, metrics_to_score=list(cpe_metrics) if cpe else None,
        evaluation=EvaluationParameters(calc_cpe_in_training=cpe), rl=make_rl_parameters(**rl_kwargs),
        q_network_optimizer=make_adam(lr), actor_network_optimizer=make_adam(lr),
        actions=[str(i) for i in range(num_actions)], **trainer_kw)


def dqn_batch_to_reference(b: dict):
    """dict of tensors (see oracle/synthetic.py) -> reference rlt.DiscreteDqnInput."""
    _install()
    import reagent.core.types as rlt

    return rlt.DiscreteDqnInput(
        state=rlt.FeatureData(b["state"]), next_state=rlt.FeatureData(b["next_state"]),
        reward=b["reward"], time_diff=b["time_diff"], step=b["step"], not_terminal=b["not_terminal"],
        action=b["action"], next_action=b["next_action"], possible_actions_mask=b["possible_actions_mask"],
        possible_next_actions_mask=b["possible_next_actions_mask"],
        extras=rlt.ExtraData(action_probability=b.get("action_probability", torch.ones_like(b["reward"])),
                             metrics=b.get("metrics")),
    )


def policy_batch_to_reference(b: dict):
    _install()
    import reagent.core.types as rlt

    return rlt.PolicyNetworkInput(
        state=rlt.FeatureData(b["state"]), next_state=rlt.FeatureData(b["next_state"]),
        reward=b["reward"], time_diff=b["time_diff"], step=b["step"], not_terminal=b["not_terminal"],
        action=rlt.FeatureData(b["action"]), next_action=rlt.FeatureData(b["next_action"]), extras=None,
    )
