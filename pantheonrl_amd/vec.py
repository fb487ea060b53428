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
        self.actions = th.zeros((E, lay.A), dtype=th.int32, device=dev)   # may be re-pointed at an exchange buffer row
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

    def update_joint(self, base_reward: th.Tensor, done: th.Tensor, joint: th.Tensor, seat: int,
                     partner_seat: th.Tensor, bonus: float = 0.01) -> None:
        """update() of the agent-per-GPU simultaneous step: reward = base + bonus * [own action == partner's], computed
        from the all-gathered JOINT action (n_seats, E) in one launch.  `partner_seat` is a device int32 scalar."""
        self.flush_rewards()
        rb = self.model.rollout_buffer
        nat.check(self._lib.ph_buffer_add_reward_joint(self._h, self._rb, rb.pos - 1, base_reward.data_ptr(),
                                                       joint.data_ptr(), int(joint.shape[0]), int(seat),
                                                       partner_seat.data_ptr(), float(bonus)))
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


class VecFrameStack:
    """HistoryQueue (wrappers.py:37-71) for E environments as a ring buffer in HBM: `push(obs, reset_mask)` returns the
    (E, numframes*D) newest-first stacked observation that feeds the policy forward."""

    def __init__(self, n_envs: int, obs_dim: int, numframes: int, ctx: nat.Context, device, default_obs=None):
        self.E, self.D, self.nf, self.ctx, self.device = n_envs, obs_dim, numframes, ctx, device
        self.default = None if default_obs is None else th.as_tensor(
            np.asarray(default_obs, np.float32)).to(device).contiguous()
        self.stack = th.zeros((n_envs, numframes * obs_dim), dtype=th.float32, device=device)
        if self.default is not None:
            self.stack.copy_(self.default.repeat(numframes).expand(n_envs, -1))

    def push(self, obs: th.Tensor, reset_mask: Optional[th.Tensor] = None) -> th.Tensor:
        self.ctx.set_stream(th.cuda.current_stream(self.device).cuda_stream)
        nat.check(self.ctx.lib.ph_framestack_push(self.ctx.handle, self.stack.data_ptr(), obs.data_ptr(),
                                                  nat.ptr(reset_mask), nat.ptr(self.default), self.E, self.D, self.nf))
        return self.stack


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


class StepGraphs:
    """Agent-per-GPU rollout with a collective between the two halves of every environment step.

    For every step t two small hipGraphs are captured once: `act[t]` = the policy forward + row write of every local
    agent, `upd[t]` = every local agent's joint-action reward update.  An iteration then costs, per step, two graph
    launches plus ONE torch.distributed all-gather issued by the caller between them; GAE + PPO update of all local
    agents are a third graph.  Host overhead per step drops from ~10 calls to 3, which is what keeps the N>1 path
    GPU-bound (the collective itself is a KB-sized, latency-bound message)."""

    def __init__(self, agents, datas, exchange, stream: th.cuda.Stream, bonus: float = 0.01):
        self.agents, self.datas, self.exchange, self.stream, self.bonus = agents, datas, exchange, stream, bonus
        dev = agents[0].model.policy.device
        self.partner = [th.zeros(1, dtype=th.int32, device=dev) for _ in agents]
        self.T = datas[0].T
        lead = agents[0].model.policy
        self._lib, self._h = lead.ctx.lib, lead.ctx.handle
        self.epoch_word = th.zeros(1, dtype=th.int64, device=dev)
        self._side = [th.cuda.Stream(device=dev) for _ in agents[1:]]
        self._fork = th.cuda.Event()
        self._join = [th.cuda.Event() for _ in agents[1:]]
        for i, a in enumerate(agents):
            a.actions = exchange.local[i].view(a.E, 1)       # forward writes straight into the exchange buffer
            a.model.device_permutations = True
            nat.check(a.model.policy.ctx.lib.ph_ctx_set_rng_epoch(a.model.policy.ctx.handle, self.epoch_word.data_ptr()))
        with th.cuda.stream(stream):
            self._eager_iteration(gather=False)              # warm-up: sizes workspaces, caches specs
            stream.synchronize()
            self.act, self.upd = [], []
            for t in range(self.T):
                self.act.append(self._capture(lambda t=t: self._act(t)))
                self.upd.append(self._capture(lambda t=t: self._upd(t)))
            self.learn = self._capture(self._learn)
            for a in agents:                                  # captures do not execute: restore the Python-side counters
                a.model.rollout_buffer.pos, a.n_steps = 0, 0

    def _bind(self):
        for a in self.agents:
            a.bind_stream()

    def _act(self, t):
        """policy forwards of all local agents; agents 1.. run on side streams forked from / joined to the main
        stream, so inside the captured graph the (latency-bound, 64-workgroup) forwards execute concurrently"""
        main = th.cuda.current_stream()
        self._fork.record(main)
        for i, (a, d) in enumerate(zip(self.agents, self.datas)):
            if i == 0:
                a.bind_stream()
                a.get_action(d.obs[t])
                continue
            side = self._side[i - 1]
            side.wait_event(self._fork)
            with th.cuda.stream(side):
                a.bind_stream()
                a.get_action(d.obs[t])
                self._join[i - 1].record(side)
        for ev in self._join[:len(self.agents) - 1]:
            main.wait_event(ev)
        self._bind()

    def _upd(self, t):
        ex = self.exchange
        for i, (a, d) in enumerate(zip(self.agents, self.datas)):
            a.update_joint(d.rewards[t], d.dones[t], ex.joint, ex.seat(i), self.partner[i], self.bonus)

    def _learn(self):
        for a in self.agents:
            a.learn_from_buffer()
        nat.check(self._lib.ph_rng_epoch_advance(self._h))

    def _eager_iteration(self, gather: bool):
        self._bind()
        for t in range(self.T):
            self._act(t)
            if gather:
                self.exchange.gather_inplace()
            self._upd(t)
        self._learn()

    def _capture(self, fn) -> int:
        self._bind()
        nat.check(self._lib.ph_graph_begin(self._h))
        try:
            fn()
        finally:
            gid = C.c_int(-1)
            nat.check(self._lib.ph_graph_end(self._h, C.byref(gid)))
        return gid.value

    def set_pairing(self, pairing_round: int) -> None:
        ex = self.exchange
        for i, p in enumerate(self.partner):
            p.fill_(ex.partner_of(ex.seat(i), pairing_round))

    def run_iteration(self, pairing_round: int) -> None:
        """must be called with `self.stream` current"""
        self.set_pairing(pairing_round)
        lib, h, ex = self._lib, self._h, self.exchange
        for t in range(self.T):
            nat.check(lib.ph_graph_launch(h, self.act[t]))
            ex.gather_inplace()
            nat.check(lib.ph_graph_launch(h, self.upd[t]))
        nat.check(lib.ph_graph_launch(h, self.learn))
        for a in self.agents:
            a.iteration += 1
            a.num_timesteps += self.T * a.E
