"""Vectorised (n_envs = E) generalisation of OnPolicyAgent, device-resident end to end.

The reference's partners are hard-wired to one environment (pantheonrl/common/agents.py:173-175,198).  Here the same
two callbacks take E-long device tensors, so a whole (E x T) rollout, its GAE pass and its PPO update never leave
HBM:

    get_action(obs[E,D])           <- agents.py:111-184  (train-before-act on a full buffer, fused forward + row write)
    update(reward[E], done[E])     <- agents.py:186-203  (rewards[pos-1] += reward ; last_episode_starts = done)

`SyntheticRollouts` provides the seeded synthetic (n_envs, n_steps, obs_dim) inputs of SURVEY.md 8(d), and
`IterationGraph` captures one whole PPO iteration (T steps + GAE + n_epochs of minibatch updates) into a hipGraph so
the launch-bound rollout loop costs one host call.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch as th

from . import _native as nat
from .ppo import PPO


class VecOnPolicyAgent:
    """OnPolicyAgent over E environments with device tensors in and out (see module docstring)."""

    def __init__(self, model: PPO):
        self.model = model
        pol, rb = model.policy, model.rollout_buffer
        E, lay, dev = rb.n_envs, pol.layout, pol.device
        self.E = E
        self._last_episode_starts = th.ones(E, dtype=th.float32, device=dev)   # D-6: starts True
        self.n_steps = 0
        self.num_timesteps = 0
        self.iteration = 0
        # preallocated outputs: no allocation on the per-step path (and stable pointers for graph capture)
        self.actions = th.zeros((E, lay.A), dtype=th.int32, device=dev)
        self.values = th.zeros((E,), dtype=th.float32, device=dev)
        self.log_probs = th.zeros((E,), dtype=th.float32, device=dev)
        self._lib, self._h = pol.ctx.lib, pol.ctx.handle
        self._spec, self._rb = C.byref(pol.spec), C.byref(rb.c_struct())
        self.sync_stats = False
        self._pending = None   # reward tensor of the previous update(), folded into the next step's launch

    # -- callbacks --------------------------------------------------------------------------------------------------
    def get_action(self, obs: th.Tensor, record: bool = True, action_mask: Optional[th.Tensor] = None,
                   episode_start: Optional[th.Tensor] = None) -> th.Tensor:
        model = self.model
        pol, rb = model.policy, model.rollout_buffer
        if record and self.n_steps >= model.n_steps:
            self.learn_from_buffer()
        es = self._last_episode_starts if episode_start is None else episode_start
        pending = self._pending if (record and rb.pos >= 1) else None
        pol._counter += 1
        nat.check(self._lib.ph_policy_forward(
            self._h, self._spec, pol.params.data_ptr(), obs.data_ptr(), self.E, nat.ptr(action_mask), None, None,
            pol._seed, pol._counter, 0, self.actions.data_ptr(), None, self.values.data_ptr(),
            self.log_probs.data_ptr(), None, None, self._rb if record else None, rb.pos if record else 0,
            es.data_ptr() if record else None, nat.ptr(pending), int(pol.gemm_mode)))
        if pending is not None:
            self._pending = None
        if record:
            rb.pos += 1
            rb.full = rb.pos == rb.buffer_size
        self.n_steps += 1
        self.num_timesteps += self.E
        return self.actions

    def update(self, reward: th.Tensor, done: th.Tensor, env_mask: Optional[th.Tensor] = None) -> None:
        if self._pending is not None or env_mask is not None:
            self.flush_rewards()              # a second update for the same action: rewards add up (agents.py:44-47)
        if env_mask is None:
            self._pending = reward            # applied by the next get_action's launch (or flush_rewards)
        else:
            rb = self.model.rollout_buffer
            nat.check(self._lib.ph_buffer_add_reward(self._h, self._rb, rb.pos - 1, reward.data_ptr(),
                                                     nat.ptr(env_mask)))
        self._last_episode_starts = done

    def flush_rewards(self) -> None:
        if self._pending is not None:
            rb = self.model.rollout_buffer
            nat.check(self._lib.ph_buffer_add_reward(self._h, self._rb, rb.pos - 1, self._pending.data_ptr(), None))
            self._pending = None

    def learn_from_buffer(self) -> None:
        """GAE with the cached V(o_{T-1}) (quirk D-1), PPO update, buffer reset (agents.py:126-158)."""
        model = self.model
        rb = model.rollout_buffer
        self.flush_rewards()
        nat.check(self._lib.ph_gae(self._h, self._rb, self.values.data_ptr(), self._last_episode_starts.data_ptr(),
                                   rb.gamma, rb.gae_lambda, int(rb.gae_mode)))
        model.train(sync_stats=self.sync_stats)
        self.iteration += 1
        rb.pos, rb.full = 0, False   # rows are fully overwritten by the next rollout; no memset needed on this path
        self.n_steps = 0

    def bind_stream(self) -> None:
        self.model.policy._bind()


class SyntheticRollouts:
    """Seeded synthetic rollout inputs resident in HBM: obs ~ N(0,1) for Box / uniform categories for
    the discrete family, rewards ~ N(0,1), dones ~ Bernoulli(1/horizon)."""

    def __init__(self, observation_space, n_envs: int, n_steps: int, horizon: int, seed: int, device):
        rng = np.random.default_rng(seed)
        T, E = n_steps, n_envs
        kind = type(observation_space).__name__
        if kind == "Box":
            D = int(np.prod(observation_space.shape))
            obs = rng.standard_normal((T, E, D), dtype=np.float32)
        elif kind == "Discrete":
            obs = rng.integers(0, observation_space.n, size=(T, E, 1)).astype(np.float32)
        else:
            nvec = np.asarray(observation_space.nvec)
            obs = (rng.random((T, E, len(nvec))) * nvec).astype(np.int64).astype(np.float32)
        self.obs = th.as_tensor(obs).to(device)
        self.rewards = th.as_tensor(rng.standard_normal((T, E), dtype=np.float32)).to(device)
        self.dones = th.as_tensor((rng.random((T, E)) < 1.0 / max(horizon, 1)).astype(np.float32)).to(device)
        self.T, self.E = T, E


def run_iteration_eager(agent: VecOnPolicyAgent, data: SyntheticRollouts) -> None:
    """one PPO iteration: T x (get_action, update), then GAE + train at the head of the next get_action -- here
    invoked explicitly so an iteration is self-contained."""
    agent.bind_stream()
    for t in range(data.T):
        agent.get_action(data.obs[t])
        agent.update(data.rewards[t], data.dones[t])
    agent.learn_from_buffer()


class IterationGraph:
    """One whole PPO iteration of one agent captured as a hipGraph on the agent's own stream."""

    def __init__(self, agent: VecOnPolicyAgent, data: SyntheticRollouts, stream: th.cuda.Stream):
        self.agent, self.data, self.stream = agent, data, stream
        pol = agent.model.policy
        self.epoch_word = th.zeros(1, dtype=th.int64, device=pol.device)
        nat.check(pol.ctx.lib.ph_ctx_set_rng_epoch(pol.ctx.handle, self.epoch_word.data_ptr()))
        agent.model.device_permutations = True   # in-kernel Feistel permutations: nothing host-generated per replay
        with th.cuda.stream(stream):
            run_iteration_eager(agent, data)     # warm-up outside capture: sizes the workspace, caches the spec
            run_iteration_eager(agent, data)
            stream.synchronize()
            agent.bind_stream()
            lib, h = pol.ctx.lib, pol.ctx.handle
            nat.check(lib.ph_graph_begin(h))
            try:
                run_iteration_eager(agent, data)
                nat.check(lib.ph_rng_epoch_advance(h))
            finally:
                gid = C.c_int(-1)
                nat.check(lib.ph_graph_end(h, C.byref(gid)))
            self.graph_id = gid.value

    def launch(self) -> None:
        pol = self.agent.model.policy
        nat.check(pol.ctx.lib.ph_graph_launch(pol.ctx.handle, self.graph_id))
        self.agent.iteration += 1
        self.agent.num_timesteps += self.data.T * self.data.E
