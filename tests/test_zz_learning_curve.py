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
On the SIMT interpreter the same flow runs for three short episodes (no bar: seconds, not learning).
"""
import numpy as np
import pytest
import torch

from cartpole_env import CartPoleEnv
from reagent_amd.core import types as rlt
from reagent_amd.core.parameters import EvaluationParameters, NormalizationParameters, RLParameters
from reagent_amd.gym.datasets import ReplayBufferDataset
from reagent_amd.gym.preprocessors import make_replay_buffer_inserter
from reagent_amd.gym.types import Transition
from reagent_amd.models import FullyConnectedDQN
from reagent_amd.optimizer import Optimizer__Union
from reagent_amd.prediction.predictor_wrapper import (DiscreteDqnPredictorWrapper, DiscreteDqnWithPreprocessor,
                                                      ServingFeatureData)
from reagent_amd.preprocessing import Preprocessor
from reagent_amd.preprocessing.trainer_preprocessor import DiscreteDqnInputMaker
from reagent_amd.replay_memory import ReplayBuffer
from reagent_amd.training import DQNTrainer

SEED = 0
PASSING_SCORE_BAR = 100.0  # discrete_dqn_cartpole_online.yaml:35


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
    prefill, episodes, batch, eval_episodes = (5000, 60, 512, 20) if full else (96, 3, 32, 2)
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
