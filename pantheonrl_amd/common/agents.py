"""Agent plugin API and the on-policy learner-as-callbacks (reference pantheonrl/common/agents.py:24-208).

`OnPolicyAgent` turns an on-policy algorithm's `learn()` loop inside-out: the environment drives it through
`get_action()` (policy forward + rollout-buffer row) and `update()` (late, additive rewards).  Here the model is the
gfx950-backed `pantheonrl_amd.PPO`, so those two callbacks are one fused kernel launch and one tiny `+=` launch; a
full buffer triggers GAE + the PPO update on the device before the next action is produced.

Reference quirks kept on purpose: the bootstrap value handed to GAE is V(o_{T-1}) cached from
the previous call (D-1); rows are written with reward 0 and rewards arrive additively (D-2); a full buffer is
trained on at the *next* recorded `get_action` (D-3); `record=False` still advances `n_steps` (D-4); the episode in
progress is excluded from the logged means (D-5); `_last_episode_starts` starts as `[True]` (D-6).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from collections import deque

import numpy as np
import torch as th

from ..logger import configure_logger, safe_mean
from .observation import Observation
from .trajsaver import TransitionsMinimal
from .util import action_from_policy, clip_actions, resample_noise


class Agent(ABC):
    """Anything that can sit in a seat of a MultiAgentEnv (agents.py:24-51)."""

    @abstractmethod
    def get_action(self, obs: Observation, record: bool = True) -> np.ndarray:
        """Return the action for `obs`; `record` asks a learner to keep the (obs, action) pair for training."""

    @abstractmethod
    def update(self, reward: float, done: bool) -> None:
        """Credit `reward` to the most recent recorded action.  Several updates for one action add their rewards
        up, and the `done` of the last one wins (agents.py:44-47)."""


class StaticPolicyAgent(Agent):
    """A frozen policy: acts, never learns (agents.py:54-79)."""

    def __init__(self, policy):
        self.policy = policy

    def get_action(self, obs: Observation, record: bool = True) -> np.ndarray:
        actions, _, _ = action_from_policy(obs.obs, self.policy)
        return clip_actions(actions, self.policy)[0]

    def update(self, reward: float, done: bool) -> None:
        return None


class OnPolicyAgent(Agent):
    """A learning partner built on an on-policy model (A2C/PPO style) -- agents.py:82-208.

    :param model: object with the SB3 `OnPolicyAlgorithm` surface (`pantheonrl_amd.PPO`)
    :param log_interval: log every this many updates (default: 1 if model.verbose else never)
    :param tensorboard_log / tb_log_name: where this agent's own logger writes
    """

    def __init__(self, model, log_interval=None, tensorboard_log=None, tb_log_name="OnPolicyAgent"):
        self.model = model
        self._last_episode_starts = [True]
        self.n_steps = 0
        self.values: th.Tensor = th.empty(0)
        self.model.set_logger(configure_logger(self.model.verbose, tensorboard_log, tb_log_name))
        self.name = tb_log_name
        self.num_timesteps = 0
        self.log_interval = log_interval or (1 if model.verbose else None)
        self.iteration = 0
        self.model.ep_info_buffer = deque([{"r": 0, "l": 0}], maxlen=100)

    # -- the two callbacks ---------------------------------------------------------------------------------------
    def get_action(self, obs: Observation, record: bool = True) -> np.ndarray:
        # The reference hands the policy obs.obs alone (agents.py:162: action_from_policy(obs.obs, ...)): a plain SB3 policy never
        # sees Observation.action_mask, the environment repairs illegal samples (pettingzoo.py:81-82).  `use_action_mask = True`
        # (an extension, off by default) applies ModularPolicy's -30 logit offset (modular/policies.py:330-333) instead.
        raw_obs, mask = obs.obs, (getattr(obs, "action_mask", None) if getattr(self, "use_action_mask", False) else None)
        model = self.model
        buf = model.rollout_buffer

        if record and self.n_steps >= model.n_steps:  # buffer full: learn before acting (agents.py:126-158)
            buf.compute_returns_and_advantage(last_values=self.values, dones=self._last_episode_starts[0])
            if self.log_interval is not None and self.iteration % self.log_interval == 0:
                self._log_rollout()
            model.train()
            self.iteration += 1
            buf.reset()
            self.n_steps = 0

        resample_noise(model, self.n_steps)

        fused = record and hasattr(model.policy, "forward_and_store")
        if (fused and mask is None and getattr(model.policy, "host_step_path", False) and not isinstance(raw_obs, th.Tensor)
                and isinstance(self._last_episode_starts, (list, tuple, np.ndarray))):
            # the environment lives on the host (the reference's situation): one native call per step -- stage in, forward + row
            # write, results out (ActorCriticPolicy.forward_and_store_host); same kernel and RNG stream as the general path
            actions, values, log_probs = model.policy.forward_and_store_host(self._shape_obs(raw_obs, buf), buf,
                                                                             self._last_episode_starts, as_numpy=True)
        elif fused:  # forward + RolloutBuffer.add(reward=0) in one launch
            shaped = self._shape_obs(raw_obs, buf)
            act_t, values, log_probs = model.policy.forward_and_store(
                shaped, buf, self._last_episode_starts,
                action_mask=None if mask is None else np.asarray(mask).reshape(shaped.shape[0], -1))
            actions = act_t.cpu().numpy()
        else:
            actions, values, log_probs = action_from_policy(raw_obs, model.policy)

        if record:
            self._bump_episode(length=1)
            if not fused:
                obs_shape = tuple(model.policy.observation_space.shape) or tuple(buf.obs_shape)
                act_shape = tuple(model.policy.action_space.shape)
                buf.add(np.reshape(raw_obs, (1,) + obs_shape), np.reshape(actions, (1,) + act_shape), [0],
                        self._last_episode_starts, values, log_probs)

        self.n_steps += 1
        self.num_timesteps += 1
        self.values = values
        return clip_actions(actions, model)[0]

    def _shape_obs(self, raw_obs, buf) -> np.ndarray:
        """the observation as rows for the policy (AdapAgent: rows without the context, which the policy appends)"""
        return np.asarray(raw_obs).reshape((-1,) + tuple(buf.obs_shape))

    def update(self, reward: float, done: bool) -> None:
        buf = self.model.rollout_buffer
        self._last_episode_starts = [done]
        if type(reward) is float and hasattr(buf, "add_reward_scalar"):
            buf.add_reward_scalar(reward)       # rewards[pos-1] += reward on the device (agents.py:198), the scalar as an argument
        elif hasattr(buf, "add_reward"):
            buf.add_reward(reward)
        else:
            buf.rewards[buf.pos - 1][0] += reward
        self._bump_episode(reward=reward)
        if done:
            self.model.ep_info_buffer.append({"r": 0, "l": 0})

    def learn(self, **kwargs) -> None:
        """Run the model's own learn loop (agents.py:205-208)."""
        self.model._custom_logger = False
        self.model.learn(**kwargs)

    # -- bookkeeping ------------------------------------------------------------------------------------------------
    def _bump_episode(self, reward: float = 0, length: int = 0) -> None:
        info = self.model.ep_info_buffer[-1]     # the running episode's record, updated in place
        info["r"] += reward
        info["l"] += length

    def _log_rollout(self) -> None:
        model, lg = self.model, self.model.logger
        lg.record("name", self.name, exclude="tensorboard")
        lg.record("time/iterations", self.iteration, exclude="tensorboard")
        eps = model.ep_info_buffer
        if len(eps) > 0 and len(eps[0]) > 0:
            running = eps.pop()  # the episode still in progress does not count (D-5)
            lg.record("rollout/ep_rew_mean", safe_mean([ep["r"] for ep in eps]))
            lg.record("rollout/ep_len_mean", safe_mean([ep["l"] for ep in eps]))
            eps.append(running)
        lg.record("time/total_timesteps", self.num_timesteps, exclude="tensorboard")
        lg.dump(step=self.num_timesteps)


class OffPolicyAgent(Agent):
    """The reference also wraps SB3's off-policy learners (agents.py:211-362).  They are outside the on-policy PPO path this
    engine implements (out of scope), so the name exists only to fail with a clear message."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("OffPolicyAgent (SB3 off-policy algorithms) is not part of the MI355X on-policy PPO "
                                  "engine; use OnPolicyAgent(pantheonrl_amd.PPO(...))")

    def get_action(self, obs: Observation, record: bool = True) -> np.ndarray:   # pragma: no cover
        raise NotImplementedError

    def update(self, reward: float, done: bool) -> None:                         # pragma: no cover
        raise NotImplementedError


class RecordingAgentWrapper(Agent):
    """An agent that behaves like `realagent` and keeps every (observation, action) pair it produced -- the data source of
    behaviour cloning (agents.py:365-413).  `get_transitions()` hands them over as a `TransitionsMinimal`."""

    def __init__(self, realagent: Agent):
        self.realagent = realagent
        self.allobs: list = []
        self.allacts: list = []

    def get_action(self, obs: Observation, record: bool = True) -> np.ndarray:
        action = self.realagent.get_action(obs, record)
        self.allobs.append(obs.obs)
        self.allacts.append(action)
        return action

    def update(self, reward: float, done: bool) -> None:
        self.realagent.update(reward, done)

    def get_transitions(self) -> TransitionsMinimal:
        return TransitionsMinimal(np.array(self.allobs), np.array(self.allacts))
