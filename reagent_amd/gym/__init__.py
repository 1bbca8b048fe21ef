"""The replay-buffer training flow on either side of the hot path: what the reference's gym
integration runs per environment step — Transition -> ReplayBuffer.add, and per training step —
sample_transition_batch -> input maker -> trainer.  Environments, agents and policies stay the caller's
(duck-typed: `reset / step / possible_actions_mask / action_space`, `act / post_step`); nothing here imports gym.
"""
