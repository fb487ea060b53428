"""An SB3-surface model built on the CPU oracle -- lets the host-side mirror (OnPolicyAgent, MultiAgentEnv, trainer
graph) run without a GPU.  Test infrastructure only."""
from __future__ import annotations

from collections import deque

import numpy as np
import torch as th

from oracle import sb3_oracle as orc
from pantheonrl_amd.logger import Logger


def _spec_of(space) -> orc.SpaceSpec:
    k = type(space).__name__
    if k == "Box":
        return orc.SpaceSpec("box", dim=int(np.prod(space.shape)))
    if k == "Discrete":
        return orc.SpaceSpec("discrete", nvec=(space.n,))
    return orc.SpaceSpec("multidiscrete", nvec=tuple(int(v) for v in space.nvec))


class _Policy:
    def __init__(self, observation_space, action_space, seed):
        self.observation_space, self.action_space, self.device = observation_space, action_space, "cpu"
        th.manual_seed(seed)
        self.net = orc.MlpPolicyOracle(_spec_of(observation_space), _spec_of(action_space))

    def forward(self, obs, deterministic=False, action_mask=None, uniforms=None):
        obs = th.as_tensor(np.asarray(obs, np.float32)).reshape(-1, self.net.obs_space.stored_len)
        with th.no_grad():
            a, v, lp = self.net.forward(obs, deterministic=deterministic)
        return a.reshape((-1,) + tuple(self.action_space.shape)), v, lp

    def reset_noise(self, n=1):
        return None


class OraclePPO:
    """`stable_baselines3.PPO`-shaped object on the oracle: exactly the attributes OnPolicyAgent touches."""

    def __init__(self, env, n_steps=2048, batch_size=64, n_epochs=10, seed=0, verbose=0):
        self.observation_space, self.action_space = env.observation_space, env.action_space
        self.env, self.n_steps, self.verbose = env, n_steps, verbose
        self.policy = _Policy(self.observation_space, self.action_space, seed)
        obs_spec, act_spec = _spec_of(self.observation_space), _spec_of(self.action_space)
        self.rollout_buffer = orc.RolloutBufferOracle(n_steps, 1, obs_spec.stored_len, act_spec.stored_len)
        self.rollout_buffer.obs_shape = (obs_spec.stored_len,)
        self.hp = orc.PPOHyper(batch_size=batch_size, n_epochs=n_epochs)
        self.use_sde, self.sde_sample_freq = False, -1
        self.ep_info_buffer = deque(maxlen=100)
        self.logger = Logger()
        self._custom_logger = False
        self.train_calls = 0
        self.trained_on = []

    def set_logger(self, logger):
        self.logger = logger

    def train(self):
        assert self.rollout_buffer.full, "SB3 asserts the buffer is full in RolloutBuffer.get"
        self.trained_on.append({k: getattr(self.rollout_buffer, k).copy() for k in (
            "rewards", "episode_starts", "advantages", "returns", "values")})
        orc.ppo_train(self.policy.net, self.rollout_buffer, self.hp)
        self.train_calls += 1
