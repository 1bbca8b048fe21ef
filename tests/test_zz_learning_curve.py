"""The reference's integration level ("train to a reward bar") on this package's path: online Discrete DQN on
cart-pole through the replay-buffer flow of reagent/gym/tests/test_gym.py:186-268 (`run_test_replay_buffer`) with the
model of reagent/gym/tests/configs/cartpole/discrete_dqn_cartpole_online.yaml — FullyConnectedDQN [128, 64] leaky_relu,
Adam lr 0.01, gamma 0.99, target_update_rate 0.2, double-Q, max-Q, MSE loss, minibatch 512, one training step per
environment step, greedy training and serving policies, passing_score_bar 100 over 20 evaluation episodes:

    random policy fills the ReplayBuffer (one `add` per transition through the inserter)
    ReplayBufferDataset: act greedily on q_network -> env.step -> inserter -> one-launch sampler + input maker
    trainer.train_step_native(batch)                  (three forwards, TD head, backward, Adam, soft update: HIP kernels)
    evaluation: DiscreteDqnPredictorWrapper(DiscreteDqnWithPreprocessor(q_network, Preprocessor(gym normalizer)))

Everything the step loop of a user of the reference touches, end to end, and the only test here whose pass criterion is
that the agent LEARNS.  Sized down from the YAML where that only costs time (5 000 random transitions instead of 30 000,
60 training episodes instead of 120: a torch restatement of the same loop reaches 110-200 on eight seeds out of eight);
the environment is tests/cartpole_env.py (gym is not installed).  The run is deterministic — seeded numpy environment and
random policy, seeded device index draws, kernels with fixed summation orders — so its outcome does not depend on the box.
It is the LAST test file on purpose: a `-x` run has judged every parity test before this one starts.
On the SIMT interpreter the DQN and SAC flows run for one or two short episodes (no bar: seconds, not learning).
"""
import numpy as np
import pytest
import torch

from cartpole_env import CartPoleEnv
from pendulum_env import PendulumEnv
from reagent_amd.core import types as rlt
from reagent_amd.core.parameters import EvaluationParameters, NormalizationParameters, RLParameters
from reagent_amd.gym.datasets import ReplayBufferDataset
from reagent_amd.gym.preprocessors import make_replay_buffer_inserter
from reagent_amd.gym.types import Transition
from reagent_amd.core.parameters import CONTINUOUS_TRAINING_ACTION_RANGE
from reagent_amd.models import FullyConnectedCritic, FullyConnectedDQN, GaussianFullyConnectedActor
from reagent_amd.models.fully_connected_network import FloatFeatureFullyConnected
from reagent_amd.optimizer import Optimizer__Union
from reagent_amd.prediction.predictor_wrapper import (DiscreteDqnPredictorWrapper, DiscreteDqnWithPreprocessor,
                                                      ServingFeatureData)
from reagent_amd.preprocessing import Preprocessor
from reagent_amd.preprocessing.trainer_preprocessor import DiscreteDqnInputMaker, PolicyNetworkInputMaker, rescale_actions
from reagent_amd.replay_memory import ReplayBuffer
from reagent_amd.training import DQNTrainer, SACTrainer

import os

SEED = 0
PASSING_SCORE_BAR = 100.0  # discrete_dqn_cartpole_online.yaml:35
# The driver's GPU run spends its time on parity: the DQN and SAC flows (the path's two trainers with a published bar) always
# run; the QR-DQN, C51 and TD3 flows only with RG_ALL_LEARNING_CURVES=1 (they passed on the MI355X in round 5,
# profiles/r05_run11/learning_curves.txt).
all_flows = pytest.mark.skipif(not os.environ.get("RG_ALL_LEARNING_CURVES"), reason="set RG_ALL_LEARNING_CURVES=1 for the QR-DQN / C51 / TD3 flows")


class RandomAgent:
    """make_random_policy_for_env (reagent/gym/policies/random_policies.py): uniform over the discrete actions"""

    post_step = None

    def __init__(self, num_actions, seed):
        self.n, self.rng = num_actions, np.random.RandomState(seed)

    def act(self, obs, possible_actions_mask=None):
        return int(self.rng.randint(self.n)), float(-np.log(self.n))


class GreedyQAgent:
    """Policy(scorer=discrete_dqn_scorer(q_network), sampler=GreedyActionSampler()) — the training policy of
    DiscreteDQNBase.create_policy(serving=False), reagent/model_managers/discrete_dqn_base.py:97-102"""

    post_step = None

    def __init__(self, q_network, device):
        self.q, self.device = q_network, device

    @torch.no_grad()
    def act(self, obs, possible_actions_mask=None):
        x = torch.from_numpy(np.asarray(obs, dtype=np.float32))[None].to(self.device)
        mask = torch.from_numpy(possible_actions_mask)[None].to(self.device) if possible_actions_mask is not None else None
        scores = self.q(rlt.FeatureData(float_features=x), mask)
        return int(scores.argmax(dim=1).item()), 0.0


def fill_replay_buffer(env, rb, desired_size, agent, inserter, max_steps):
    """reagent/gym/utils.py fill_replay_buffer: whole episodes of `agent` until the buffer holds desired_size transitions"""
    mdp_id = 0
    while rb.size < desired_size:
        obs, t, terminal = env.reset(), 0, False
        while not terminal and rb.size < desired_size:
            mask = env.possible_actions_mask
            action, log_prob = agent.act(obs, mask)
            next_obs, reward, terminal, _ = env.step(action)
            terminal = terminal or t >= max_steps
            inserter(rb, Transition(mdp_id=mdp_id, sequence_number=t, observation=obs, action=action, reward=float(reward),
                                    terminal=bool(terminal), log_prob=log_prob, possible_actions_mask=mask))
            obs, t = next_obs, t + 1
        mdp_id += 1


def evaluate(env, predictor, episodes, device):
    """evaluate_for_n_episodes with the serving policy: DiscreteDQNPredictorPolicy, greedy (rl.softmax_policy is False)"""
    presence = torch.ones(1, 4, dtype=torch.uint8, device=device)
    rewards = []
    for _ in range(episodes):
        obs, total, terminal, t = env.reset(), 0.0, False, 0
        while not terminal and t < env.max_steps:
            x = torch.from_numpy(obs)[None].to(device)
            _, q = predictor(ServingFeatureData(float_features_with_presence=(x, presence)))
            obs, r, terminal, _ = env.step(int(q.argmax(dim=1).item()))
            total, t = total + r, t + 1
        rewards.append(total)
    return np.array(rewards)


def test_online_dqn_reaches_the_reference_bar_on_cartpole(backend):
    full = backend.name == "hip"
    prefill, episodes, batch, eval_episodes = (5000, 60, 512, 20) if full else (48, 2, 16, 1)
    dev = torch.device(backend.device)
    torch.manual_seed(SEED)
    if dev.type == "cuda":
        torch.cuda.manual_seed(SEED)
    env = CartPoleEnv(seed=SEED)
    q = FullyConnectedDQN(4, 2, [128, 64], ["leaky_relu", "leaky_relu"]).to(dev)
    trainer = DQNTrainer(q, q.get_target_network(), None, actions=["0", "1"],
                         rl=RLParameters(gamma=0.99, target_update_rate=0.2, maxq_learning=True, temperature=1.0),
                         double_q_learning=True, minibatches_per_step=1, optimizer=Optimizer__Union.default(lr=0.01),
                         evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)
    rb = ReplayBuffer(replay_capacity=100000 if full else 4096, batch_size=batch, device=dev)
    inserter = make_replay_buffer_inserter(env)
    fill_replay_buffer(env, rb, max(prefill, batch), RandomAgent(2, SEED + 1), inserter, env.max_steps)
    assert rb.size >= prefill

    train_rewards = []
    ds = ReplayBufferDataset.create_for_trainer(
        trainer, env, GreedyQAgent(trainer.q_network, dev), rb, batch_size=batch, training_frequency=1,
        num_episodes=episodes, max_steps=env.max_steps,
        post_episode_callback=lambda traj, info: train_rewards.append(traj.calculate_cumulative_reward()))
    assert type(ds._trainer_preprocessor.maker) is DiscreteDqnInputMaker
    steps = 0
    for b in ds:  # pl.Trainer.fit(trainer, DataLoader(dataset)) of test_gym.py:253-261: one training step per batch
        assert isinstance(b, rlt.DiscreteDqnInput) and b.state.float_features.shape == (batch, 4)
        loss = trainer.train_step_native(b)
        steps += 1
    assert len(train_rewards) == episodes and steps == int(sum(train_rewards)) and trainer.all_batches_processed == steps
    assert torch.isfinite(loss).all()

    norm = {i: NormalizationParameters(feature_type="CONTINUOUS", mean=0.0, stddev=1.0) for i in range(4)}  # gym/normalizers.py:16-52
    predictor = DiscreteDqnPredictorWrapper(DiscreteDqnWithPreprocessor(trainer.q_network, Preprocessor(norm, device=dev)), ["0", "1"])
    eval_rewards = evaluate(env, predictor, eval_episodes, dev)
    print(f"\ncart-pole: {steps} training steps over {episodes} episodes (last ten: {np.mean(train_rewards[-10:]):.1f} per episode); "
          f"evaluation over {eval_episodes} episodes: mean {eval_rewards.mean():.1f}, min {eval_rewards.min():.0f}, max {eval_rewards.max():.0f}")
    if full:
        assert eval_rewards.mean() >= PASSING_SCORE_BAR, f"Eval reward is {eval_rewards.mean()}, less than < {PASSING_SCORE_BAR}."
        assert np.mean(train_rewards[:5]) < np.mean(train_rewards[-10:])  # and it got there by training



@all_flows
def test_online_qrdqn_reaches_the_reference_bar_on_cartpole(backend):
    """reagent/gym/tests/configs/cartpole/discrete_qr_cartpole_online.yaml through the same flow: the DuelingQuantile network
    ([64, 64] leaky_relu: shared trunk, advantage and value streams, 11 quantiles per action), QRDQNTrainer with gamma 0.9,
    target_update_rate 0.05, double-Q, AdamW lr 1e-3 with amsgrad (one of Optimizer__Union's torch-built members: the native
    step's separate-launch update), minibatch 512, 40 training episodes; passing_score_bar 100 over 20 evaluation episodes.
    Sized down where that only costs time: 5 000 random transitions before training instead of 20 000.  The policies act
    greedily on the mean over the quantiles."""
    from reagent_amd.models.dueling_q_network import DuelingQNetwork
    from reagent_amd.optimizer import AdamW
    from reagent_amd.training import QRDQNTrainer

    if backend.name == "emu":
        pytest.skip("the interpreter runs this flow for DQN and SAC (time); the trainer itself has its own interpreter tests")
    full = backend.name == "hip"
    prefill, episodes, batch, eval_episodes = (5000, 40, 512, 20) if full else (96, 2, 32, 1)
    dev = torch.device(backend.device)
    torch.manual_seed(SEED)
    if dev.type == "cuda":
        torch.cuda.manual_seed(SEED)
    env = CartPoleEnv(seed=SEED)
    q = DuelingQNetwork.make_fully_connected(4, 2, [64, 64], ["leaky_relu", "leaky_relu"], num_atoms=11).to(dev)
    trainer = QRDQNTrainer(q, q.get_target_network(), num_atoms=11, actions=["0", "1"],
                           rl=RLParameters(gamma=0.9, target_update_rate=0.05, maxq_learning=True, temperature=1.0),
                           double_q_learning=True, minibatches_per_step=1,
                           optimizer=Optimizer__Union(AdamW=AdamW(lr=0.001, amsgrad=True)),
                           evaluation=EvaluationParameters(calc_cpe_in_training=False)).to(dev)

    class MeanQ(torch.nn.Module):  # what the policies score: the expectation over the quantiles, [B, A, N] -> [B, A]
        def forward(self, state, possible_actions_mask=None):
            return q(state).mean(dim=2)

    rb = ReplayBuffer(replay_capacity=100000 if full else 4096, batch_size=batch, device=dev)
    inserter = make_replay_buffer_inserter(env)
    fill_replay_buffer(env, rb, max(prefill, batch), RandomAgent(2, SEED + 1), inserter, env.max_steps)
    agent = GreedyQAgent(MeanQ(), dev)
    train_rewards = []
    ds = ReplayBufferDataset.create_for_trainer(
        trainer, env, agent, rb, batch_size=batch, training_frequency=1, num_episodes=episodes, max_steps=env.max_steps,
        post_episode_callback=lambda traj, info: train_rewards.append(traj.calculate_cumulative_reward()))
    steps = 0
    for b in ds:
        loss = trainer.train_step_native(b)
        steps += 1
    assert len(train_rewards) == episodes and steps == int(sum(train_rewards)) and trainer.all_batches_processed == steps
    assert torch.isfinite(loss).all()
    eval_rewards = []
    for _ in range(eval_episodes):
        obs, total, terminal, t = env.reset(), 0.0, False, 0
        while not terminal and t < env.max_steps:
            obs, r, terminal, _ = env.step(agent.act(obs)[0])
            total, t = total + r, t + 1
        eval_rewards.append(total)
    eval_rewards = np.array(eval_rewards)
    print(f"\ncart-pole, QR-DQN: {steps} training steps over {episodes} episodes (last ten: {np.mean(train_rewards[-10:]):.1f} per "
          f"episode); evaluation over {eval_episodes} episodes: mean {eval_rewards.mean():.1f}, min {eval_rewards.min():.0f}, "
          f"max {eval_rewards.max():.0f}")
    if full:
        assert eval_rewards.mean() >= PASSING_SCORE_BAR, f"Eval reward is {eval_rewards.mean()}, less than < {PASSING_SCORE_BAR}."


# ---- SAC on the pendulum: the ONLY check of SAC the reference has -------------------------------------
class UniformBoxAgent:
    """make_random_policy_for_env for a box action space (ContinuousRandomPolicy): uniform in [low, high]"""

    post_step = None

    def __init__(self, space, seed):
        self.low, self.high, self.rng = space.low, space.high, np.random.RandomState(seed)

    def act(self, obs, possible_actions_mask=None):
        a = self.rng.uniform(self.low, self.high).astype(np.float32)
        return a, float(-np.log(self.high - self.low).sum())


class ActorAgent:
    """ActorPolicyWrapper(actor_network) (reagent/model_managers/actor_critic_base.py:51-65: the actor's forward IS the
    act — a tanh-squashed Gaussian sample) + EnvWrapper.action_extractor (reagent/gym/envs/env_wrapper.py:78-87: the sample
    leaves the model's range for the environment's; that is what the replay buffer stores and PolicyNetworkInputMaker
    scales back).  greedy=True acts on the squashed mean (evaluation)."""

    post_step = None

    def __init__(self, actor, space, device, greedy=False):
        self.actor, self.device, self.greedy = actor, device, greedy
        self.low, self.high = torch.from_numpy(space.low), torch.from_numpy(space.high)
        self.m_low, self.m_high = (torch.tensor(v) for v in CONTINUOUS_TRAINING_ACTION_RANGE)

    @torch.no_grad()
    def act(self, obs, possible_actions_mask=None):
        out = self.actor(rlt.FeatureData(float_features=torch.from_numpy(obs)[None].to(self.device)))
        a = (out.squashed_mean if self.greedy else out.action).cpu().reshape(-1)
        env_a = rescale_actions(a, new_min=self.low, new_max=self.high, prev_min=self.m_low, prev_max=self.m_high)
        return env_a.numpy().astype(np.float32), float(out.log_prob.cpu().reshape(-1)[0])


def test_online_sac_reaches_the_reference_bar_on_the_pendulum(backend):
    """reagent/gym/tests/configs/pendulum/sac_pendulum_online.yaml through run_test_replay_buffer: Gaussian actor, twin critics
    and a value network (all [64, 64] leaky_relu), Adam lr 1e-3 for each and for the temperature (starting at 0.3),
    gamma 0.99, target_update_rate 0.005, minibatch 256, one training step per environment step, 40 training episodes of
    200 steps; passing_score_bar -500 over 20 evaluation episodes (a random policy: about -1240).
    Sizing: 10 000 random transitions before training (YAML: 20 000) and 60 training episodes (YAML: 40).  The UNMODIFIED
    reference trainer driven through the same loop on this environment (torch-CPU, oracle/learning_probe.py) scores -129 /
    -133 on two seeds at this size (worst episode -302 / -357) but -262 / -141 with episodes below -1100 at 5 000 / 40 —
    the YAML's own comment says its bar is low "to let tests finish in time"; this package at 5 000 / 40 on the GPU: -447
    (squashed mean -202), the same class.  The extra 20 episodes buy the margin for 10 s."""
    full = backend.name == "hip"
    prefill, episodes, batch, eval_episodes = (10000, 60, 256, 20) if full else (32, 1, 16, 1)
    dev = torch.device(backend.device)
    torch.manual_seed(SEED)
    if dev.type == "cuda":
        torch.cuda.manual_seed(SEED)
    env = PendulumEnv(seed=SEED)
    if not full:
        env.max_steps = 8
    S, A, H, acts = 3, 1, [64, 64], ["leaky_relu", "leaky_relu"]
    adam = lambda: Optimizer__Union.default(lr=1e-3)  # noqa: E731
    trainer = SACTrainer(GaussianFullyConnectedActor(S, A, H, acts).to(dev), FullyConnectedCritic(S, A, H, acts).to(dev),
                         FullyConnectedCritic(S, A, H, acts).to(dev), value_network=FloatFeatureFullyConnected(S, 1, H, acts).to(dev),
                         rl=RLParameters(gamma=0.99, target_update_rate=0.005, softmax_policy=True), entropy_temperature=0.3,
                         q_network_optimizer=adam(), value_network_optimizer=adam(), actor_network_optimizer=adam(),
                         alpha_optimizer=adam()).to(dev)
    rb = ReplayBuffer(replay_capacity=100000 if full else 4096, batch_size=batch, device=dev)
    inserter = make_replay_buffer_inserter(env)
    fill_replay_buffer(env, rb, max(prefill, batch), UniformBoxAgent(env.action_space, SEED + 1), inserter, env.max_steps)

    train_rewards = []
    ds = ReplayBufferDataset.create_for_trainer(
        trainer, env, ActorAgent(trainer.actor_network, env.action_space, dev), rb, batch_size=batch, training_frequency=1,
        num_episodes=episodes, max_steps=env.max_steps,
        post_episode_callback=lambda traj, info: train_rewards.append(traj.calculate_cumulative_reward()))
    assert type(ds._trainer_preprocessor.maker) is PolicyNetworkInputMaker
    steps = 0
    for b in ds:
        assert isinstance(b, rlt.PolicyNetworkInput) and b.action.float_features.shape == (batch, A)
        losses = trainer.train_step_native(b)
        steps += 1
    assert len(train_rewards) == episodes and trainer.all_batches_processed == steps == episodes * (env.max_steps + 1)
    assert all(torch.isfinite(v).all() for v in losses.values())

    # evaluation with the serving policy: ActorWithPreprocessor returns the SAMPLED action unless serve_mean_policy is set
    # (reagent/prediction/predictor_wrapper.py:315,332-338; the SAC manager's default is False); the squashed mean's score
    # is printed beside it
    def run_eval(agent):
        out = []
        for _ in range(eval_episodes):
            obs, total = env.reset(), 0.0
            for _ in range(env.max_steps):
                obs, r, _, _ = env.step(agent.act(obs)[0])
                total += r
            out.append(total)
        return np.array(out)

    eval_rewards = run_eval(ActorAgent(trainer.actor_network, env.action_space, dev))
    mean_policy = run_eval(ActorAgent(trainer.actor_network, env.action_space, dev, greedy=True))
    print(f"\npendulum: {steps} training steps over {episodes} episodes (first five {np.mean(train_rewards[:5]):.0f}, last ten "
          f"{np.mean(train_rewards[-10:]):.0f} per episode); evaluation over {eval_episodes} episodes: mean {eval_rewards.mean():.0f}, "
          f"min {eval_rewards.min():.0f}, max {eval_rewards.max():.0f} (acting on the squashed mean: {mean_policy.mean():.0f}); "
          f"temperature {float(trainer.log_alpha.detach().exp()):.3f}")
    if full:
        assert eval_rewards.mean() >= -500.0, f"Eval reward is {eval_rewards.mean()}, less than < -500."  # sac_pendulum_online.yaml:56
        assert np.mean(train_rewards[-10:]) > np.mean(train_rewards[:5]) + 300


# ---- the widened rows: C51 on cart-pole, TD3 on the pendulum --------------------------------------------
@all_flows
def test_online_c51_reaches_the_reference_bar_on_cartpole(backend):
    """reagent/gym/tests/configs/cartpole/discrete_c51_cartpole_online.yaml: Categorical network [64, 64] leaky_relu with 21
    atoms on [0, 40], C51Trainer with gamma 0.9, target_update_rate 0.05, double-Q, AdamW lr 1e-3 amsgrad, minibatch 512,
    40 training episodes; passing_score_bar 100 over 20 evaluation episodes.  5 000 random transitions before training
    (YAML: 20 000); the policies act greedily on the expected values."""
    from reagent_amd.models.categorical_dqn import CategoricalDQN
    from reagent_amd.optimizer import AdamW
    from reagent_amd.training import C51Trainer

    if backend.name == "emu":
        pytest.skip("the interpreter runs this flow for DQN and SAC (time); the trainer itself has its own interpreter tests")
    full = backend.name == "hip"
    prefill, episodes, batch, eval_episodes = (5000, 40, 512, 20) if full else (96, 2, 32, 1)
    dev = torch.device(backend.device)
    torch.manual_seed(SEED)
    if dev.type == "cuda":
        torch.cuda.manual_seed(SEED)
    env = CartPoleEnv(seed=SEED)
    q = CategoricalDQN(FullyConnectedDQN(4, 2, [64, 64], ["leaky_relu", "leaky_relu"], num_atoms=21), qmin=0, qmax=40,
                       num_atoms=21).to(dev)
    trainer = C51Trainer(q, q.get_target_network(), actions=["0", "1"],
                         rl=RLParameters(gamma=0.9, target_update_rate=0.05, maxq_learning=True, temperature=1.0),
                         double_q_learning=True, minibatches_per_step=1, num_atoms=21, qmin=0, qmax=40,
                         optimizer=Optimizer__Union(AdamW=AdamW(lr=0.001, amsgrad=True))).to(dev)

    class ExpectedQ(torch.nn.Module):
        def forward(self, state, possible_actions_mask=None):
            return q(state)

    rb = ReplayBuffer(replay_capacity=100000 if full else 4096, batch_size=batch, device=dev)
    inserter = make_replay_buffer_inserter(env)
    fill_replay_buffer(env, rb, max(prefill, batch), RandomAgent(2, SEED + 1), inserter, env.max_steps)
    agent = GreedyQAgent(ExpectedQ(), dev)
    train_rewards = []
    ds = ReplayBufferDataset.create_for_trainer(
        trainer, env, agent, rb, batch_size=batch, training_frequency=1, num_episodes=episodes, max_steps=env.max_steps,
        post_episode_callback=lambda traj, info: train_rewards.append(traj.calculate_cumulative_reward()))
    steps = 0
    for b in ds:
        loss = trainer.train_step_native(b)
        steps += 1
    assert len(train_rewards) == episodes and steps == int(sum(train_rewards)) and trainer.all_batches_processed == steps
    assert torch.isfinite(loss).all()
    eval_rewards = []
    for _ in range(eval_episodes):
        obs, total, terminal, t = env.reset(), 0.0, False, 0
        while not terminal and t < env.max_steps:
            obs, r, terminal, _ = env.step(agent.act(obs)[0])
            total, t = total + r, t + 1
        eval_rewards.append(total)
    eval_rewards = np.array(eval_rewards)
    print(f"\ncart-pole, C51: {steps} training steps over {episodes} episodes (last ten: {np.mean(train_rewards[-10:]):.1f} per "
          f"episode); evaluation over {eval_episodes} episodes: mean {eval_rewards.mean():.1f}, min {eval_rewards.min():.0f}, "
          f"max {eval_rewards.max():.0f}")
    if full:
        assert eval_rewards.mean() >= PASSING_SCORE_BAR, f"Eval reward is {eval_rewards.mean()}, less than < {PASSING_SCORE_BAR}."


@all_flows
def test_online_td3_reaches_the_reference_bar_on_the_pendulum(backend):
    """reagent/gym/tests/configs/pendulum/td3_pendulum_online.yaml: deterministic actor [64, 64] leaky_relu with exploration
    variance 0.01, twin critics, Adam lr 0.005 (actor) / 0.01 (critics), gamma 0.99, target_update_rate 0.005, target-policy
    noise 0.2 clipped at 0.5, the actor updated every second step, minibatch 256, 5 000 random transitions, 40 training
    episodes (the YAML's own sizes); passing_score_bar -750 (the YAML evaluates one episode; twenty here)."""
    from reagent_amd.models.actor import FullyConnectedActor
    from reagent_amd.training import TD3Trainer

    if backend.name == "emu":
        pytest.skip("the interpreter runs this flow for DQN and SAC (time); the trainer itself has its own interpreter tests")
    full = backend.name == "hip"
    prefill, episodes, batch, eval_episodes = (5000, 40, 256, 20) if full else (64, 1, 32, 1)
    dev = torch.device(backend.device)
    torch.manual_seed(SEED)
    if dev.type == "cuda":
        torch.cuda.manual_seed(SEED)
    env = PendulumEnv(seed=SEED)
    if not full:
        env.max_steps = 8
    S, A, H, acts = 3, 1, [64, 64], ["leaky_relu", "leaky_relu"]
    trainer = TD3Trainer(FullyConnectedActor(S, A, H, acts, exploration_variance=0.01).to(dev),
                         FullyConnectedCritic(S, A, H, acts).to(dev), FullyConnectedCritic(S, A, H, acts).to(dev),
                         rl=RLParameters(gamma=0.99, target_update_rate=0.005),
                         q_network_optimizer=Optimizer__Union.default(lr=0.01),
                         actor_network_optimizer=Optimizer__Union.default(lr=0.005), noise_variance=0.2, noise_clip=0.5,
                         delayed_policy_update=2).to(dev)
    rb = ReplayBuffer(replay_capacity=100000 if full else 4096, batch_size=batch, device=dev)
    inserter = make_replay_buffer_inserter(env)
    fill_replay_buffer(env, rb, max(prefill, batch), UniformBoxAgent(env.action_space, SEED + 1), inserter, env.max_steps)
    agent = ActorAgent(trainer.actor_network, env.action_space, dev)  # the actor's forward, exploration noise included
    train_rewards = []
    ds = ReplayBufferDataset.create_for_trainer(
        trainer, env, agent, rb, batch_size=batch, training_frequency=1, num_episodes=episodes, max_steps=env.max_steps,
        post_episode_callback=lambda traj, info: train_rewards.append(traj.calculate_cumulative_reward()))
    steps = 0
    for b in ds:
        trainer.train_step_native(b)
        steps += 1
    assert len(train_rewards) == episodes and steps == episodes * (env.max_steps + 1)
    eval_rewards = []
    for _ in range(eval_episodes):
        obs, total = env.reset(), 0.0
        for _ in range(env.max_steps):
            obs, r, _, _ = env.step(agent.act(obs)[0])
            total += r
        eval_rewards.append(total)
    eval_rewards = np.array(eval_rewards)
    print(f"\npendulum, TD3: {steps} training steps over {episodes} episodes (first five {np.mean(train_rewards[:5]):.0f}, last ten "
          f"{np.mean(train_rewards[-10:]):.0f} per episode); evaluation over {eval_episodes} episodes: mean {eval_rewards.mean():.0f}, "
          f"min {eval_rewards.min():.0f}, max {eval_rewards.max():.0f}")
    if full:
        assert eval_rewards.mean() >= -750.0, f"Eval reward is {eval_rewards.mean()}, less than < -750."  # td3_pendulum_online.yaml
