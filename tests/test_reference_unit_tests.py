"""The reference's own unit tests for this path, restated on the HIP-backed classes with the same sizes
and the same assertions at the result level (grad_fn-type assertions cannot hold for fused losses):
reagent/test/training/test_dqn.py (init, train_step_gen counts, configure_optimizers order,
get_detached_model_outputs, compute_discount_tensor known answers, compute_td_loss),
reagent/test/training/test_qrdqn.py (quantile midpoints, the same structure), and the model tests
reagent/test/models/test_dqn.py / test_critic.py / test_actor.py (shapes, state_dict round trip)."""
import pytest
import torch

from reagent_amd.core import types as rlt
from reagent_amd.core.parameters import EvaluationParameters, RLParameters
from reagent_amd.models import (FullyConnectedActor, FullyConnectedCritic, FullyConnectedDQN,
                                GaussianFullyConnectedActor)
from reagent_amd.training import DQNTrainer, QRDQNTrainer

B, S, A = 3, 10, 2


def nets(dev, out=A, atoms=None):
    mk = lambda o, n=None: FullyConnectedDQN(S, o, [20, 20], ["relu", "relu"], num_atoms=n).to(dev)  # noqa: E731
    return mk(out, atoms), mk(A), mk(A)  # q, reward, cpe (reward metric only: one block of |A| outputs)


def batch(dev, time_diff=2, steps=2):
    g = torch.Generator().manual_seed(0)
    t = lambda x: x.to(dev)  # noqa: E731
    return rlt.DiscreteDqnInput(
        state=rlt.FeatureData(t(torch.rand(B, S, generator=g))), next_state=rlt.FeatureData(t(torch.rand(B, S, generator=g))),
        reward=t(torch.ones(B, 1)), time_diff=t(torch.ones(B, 1) * time_diff), step=t(torch.ones(B, 1) * steps),
        not_terminal=t(torch.ones(B, 1)), action=t(torch.tensor([[0, 1], [1, 0], [0, 1]])),
        next_action=t(torch.tensor([[1, 0], [0, 1], [1, 0]])), possible_actions_mask=t(torch.ones(B, A)),
        possible_next_actions_mask=t(torch.ones(B, A)), extras=rlt.ExtraData())


def dqn(dev, no_cpe=False, **rl):
    q, r, c = nets(dev)
    return DQNTrainer(q, q.get_target_network(), None if no_cpe else r, q_network_cpe=None if no_cpe else c,
                      q_network_cpe_target=None if no_cpe else c.get_target_network(), metrics_to_score=[],
                      actions=["1", "2"], rl=RLParameters(**rl),
                      evaluation=EvaluationParameters(calc_cpe_in_training=not no_cpe)).to(dev)


def qrdqn(dev, no_cpe=False, num_atoms=11, **rl):
    q, r, c = nets(dev, atoms=num_atoms)
    return QRDQNTrainer(q, q.get_target_network(), metrics_to_score=[], reward_network=None if no_cpe else r,
                        q_network_cpe=None if no_cpe else c,
                        q_network_cpe_target=None if no_cpe else c.get_target_network(), actions=["1", "2"],
                        num_atoms=num_atoms, rl=RLParameters(**rl),
                        evaluation=EvaluationParameters(calc_cpe_in_training=not no_cpe)).to(dev)


@pytest.mark.parametrize("make", [dqn, qrdqn])
def test_init(backend, make):  # test_dqn.py:84-102, test_qrdqn.py:83-100
    tr = make(backend.device)
    assert isinstance(tr.reward_boosts, torch.Tensor) and torch.isclose(tr.reward_boosts.cpu(), torch.zeros(2)).all()
    boosted = make(backend.device, reward_boost={"1": 1, "2": 2})
    assert torch.isclose(boosted.reward_boosts.cpu(), torch.tensor([1.0, 2.0])).all()
    if make is qrdqn:
        assert torch.isclose(tr.quantiles.cpu(), (0.5 + torch.arange(11).float()) / 11.0).all()


@pytest.mark.parametrize("make", [dqn, qrdqn])
def test_train_step_gen(backend, make):  # test_dqn.py:104-189, test_qrdqn.py:102-187
    dev = backend.device
    inp = batch(dev)
    losses = list(make(dev).train_step_gen(inp, batch_idx=1))
    assert len(losses) == 4 and all(torch.isfinite(l).all() and l.requires_grad for l in losses)
    assert len(list(make(dev, no_cpe=True).train_step_gen(inp, batch_idx=1))) == 2
    for rl in (dict(maxq_learning=False), dict(use_seq_num_diff_as_time_diff=True), dict(multi_steps=2),
               dict(q_network_loss="huber")):
        assert len(list(make(dev, **rl).train_step_gen(inp, batch_idx=1))) == 4, rl


@pytest.mark.parametrize("make", [dqn, qrdqn])
def test_configure_optimizers(backend, make):  # test_dqn.py:191-216, test_qrdqn.py:189-206
    tr = make(backend.device)
    optimizers = tr.configure_optimizers()
    assert len(optimizers) == 4
    for opt, net in zip(optimizers, [tr.q_network, tr.reward_network, tr.q_network_cpe, tr.q_network]):
        opt_param = opt["optimizer"].param_groups[0]["params"][0]
        assert torch.isclose(opt_param, list(net.parameters())[0]).all()
    assert len(make(backend.device, no_cpe=True).configure_optimizers()) == 2


def test_get_detached_model_outputs(backend):  # test_dqn.py:218-223, test_qrdqn.py:208-212
    x = rlt.FeatureData(torch.rand(B, S).to(backend.device))
    for tr in (dqn(backend.device), qrdqn(backend.device)):
        q_out, q_target = tr.get_detached_model_outputs(x)
        assert q_target is not None and q_out.shape[0] == q_target.shape[0] == B
        assert q_out.shape[1] == q_target.shape[1] == A


def test_compute_discount_tensor(backend):  # test_dqn.py:225-289 (known answers)
    dev = backend.device
    time_diff, steps = 4, 3
    inp = batch(dev, time_diff=time_diff, steps=steps)
    for rl, expo in ((dict(), 1), (dict(use_seq_num_diff_as_time_diff=True), time_diff), (dict(multi_steps=steps), steps)):
        tr = dqn(dev, **rl)
        d = tr.compute_discount_tensor(batch=inp, boosted_rewards=inp.reward)
        assert d.shape == (B, 1)
        assert torch.isclose(d.cpu(), torch.tensor(tr.gamma ** expo)).all(), rl


@pytest.mark.parametrize("loss", ["mse", "huber"])
def test_compute_td_loss(backend, loss):  # test_dqn.py:291-344, value instead of grad_fn type
    dev = backend.device
    inp = batch(dev)
    tr = dqn(dev, q_network_loss=loss)
    d = tr.compute_discount_tensor(batch=inp, boosted_rewards=inp.reward)
    got = tr.compute_td_loss(batch=inp, boosted_rewards=inp.reward, discount_tensor=d)
    assert got.requires_grad
    with torch.no_grad():  # dqn_trainer.py:179-239 on the same networks
        q_next, q_next_t = tr.q_network(inp.next_state), tr.q_network_target(inp.next_state)
        next_q, _ = tr.get_max_q_values_with_target(q_next, q_next_t, inp.possible_next_actions_mask.float())
        target = inp.reward + d * (next_q * inp.not_terminal.float())
        q = (tr.q_network(inp.state) * inp.action.float()).sum(1, keepdim=True)
        f = torch.nn.functional.mse_loss if loss == "mse" else torch.nn.functional.smooth_l1_loss
        ref = f(q, target)
    assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item()) + 1e-7


MODELS = [  # test_dqn.py / test_critic.py / test_actor.py: sizes [8, 4] (critic / actors [7, 6]), relu
    (lambda: FullyConnectedDQN(8, 4, sizes=[8, 4], activations=["relu", "relu"]), (1, 4)),
    (lambda: FullyConnectedCritic(8, 4, sizes=[7, 6], activations=["relu", "relu"]), (1, 1)),
    (lambda: FullyConnectedActor(8, 4, sizes=[7, 6], activations=["relu", "relu"]), (1, 4)),
    (lambda: GaussianFullyConnectedActor(8, 4, sizes=[7, 6], activations=["relu", "relu"]), (1, 4)),
    # test_save_load_batch_norm of the same files (and test_dueling_q_network.py:110-122): batch-normed, frozen by eval()
    (lambda: FullyConnectedDQN(8, 4, sizes=[8, 4], activations=["relu", "relu"], use_batch_norm=True).eval(), (1, 4)),
    (lambda: FullyConnectedCritic(8, 4, sizes=[7, 6], activations=["relu", "relu"], use_batch_norm=True).eval(), (1, 1)),
    (lambda: FullyConnectedActor(8, 4, sizes=[7, 6], activations=["relu", "relu"], use_batch_norm=True).eval(), (1, 4)),
    (lambda: GaussianFullyConnectedActor(8, 4, sizes=[7, 6], activations=["relu", "relu"], use_batch_norm=True).eval(), (1, 4)),
    (lambda: _dueling_bn().eval(), (1, 4)),
]


def _dueling_bn():
    from reagent_amd.models import DuelingQNetwork

    return DuelingQNetwork.make_fully_connected(8, 4, [8, 4], ["relu", "relu"], use_batch_norm=True)


@pytest.mark.parametrize("make,out_shape", MODELS)
def test_model_basic_and_save_load(backend, make, out_shape):
    dev = backend.device
    model = make().to(dev)
    proto = model.input_prototype()
    inputs = proto if isinstance(proto, tuple) else (proto,)
    inputs = tuple(rlt.FeatureData(x.float_features.to(dev)) for x in inputs)
    assert inputs[0].float_features.shape == (1, 8)
    # check_save_load (models/test_utils.py): a fresh model loading the state dict gives the same output
    if any(isinstance(m, torch.nn.BatchNorm1d) for m in model.modules()):  # running statistics that are not the defaults
        with torch.no_grad():
            for m in [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm1d)]:
                m.running_mean.normal_(0, 0.5)
                m.running_var.uniform_(0.5, 2.0)
    clone = make().to(dev)
    clone.load_state_dict(model.state_dict())
    assert [k for k in clone.state_dict()] == [k for k in model.state_dict()]
    if isinstance(model, GaussianFullyConnectedActor):  # same reparameterisation draw on both
        noise = torch.randn(1, 4)
        model.noise_override, clone.noise_override = noise, noise.clone()
    out = model(*inputs)
    out = out.action if hasattr(out, "action") else out
    assert tuple(out.shape) == out_shape
    out2 = clone(*inputs)
    out2 = out2.action if hasattr(out2, "action") else out2
    assert torch.equal(out, out2)
