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
from .ppo import PPO, require_mlp_kernels


class VecOnPolicyAgent:
    """OnPolicyAgent over E environments with device tensors in and out (see module docstring)."""

    def __init__(self, model: PPO):
        self.model = model
        pol, rb = model.policy, model.rollout_buffer
        require_mlp_kernels(pol, type(self).__name__)
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

    def rollout_scripted(self, data: "SyntheticRollouts") -> None:
        """data.T x (get_action(data.obs[t]); update(data.rewards[t], data.dones[t])) as ONE launch (ph_scripted_rollout): the
        environment is a script -- every step's observation, reward and done already sit in HBM -- so nothing has to return to
        the host, or even to a launch boundary, between steps.  Bitwise the per-step calls (test); needs an empty buffer of
        exactly data.T rows and the 16-row forward's shape class (raises otherwise)."""
        model = self.model
        pol, rb = model.policy, model.rollout_buffer
        if rb.pos != 0 or data.T != rb.buffer_size or data.E != self.E:
            raise nat.NativeError("rollout_scripted: needs an empty rollout buffer of data.T rows and data.E environments")
        self.flush_rewards()
        nat.check(self._lib.ph_scripted_rollout(
            self._h, self._spec, pol.params.data_ptr(), data.obs.data_ptr(), data.rewards.data_ptr(), data.dones.data_ptr(),
            self.E, data.T, self._last_episode_starts.data_ptr(), pol._seed, pol._counter + 1, self.actions.data_ptr(),
            self.values.data_ptr(), self.log_probs.data_ptr(), self._rb, 0, int(pol.gemm_mode)))
        pol._counter += data.T
        rb.pos, rb.full = data.T, True
        self.n_steps += data.T
        self.num_timesteps += data.T * self.E
        self._last_episode_starts = data.dones[data.T - 1]
        self._pending = None          # the last step's reward is already in its row

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

    def compute_returns(self) -> None:
        """late rewards + GAE with the cached V(o_{T-1}) (quirk D-1) <- agents.py:127-130"""
        rb = self.model.rollout_buffer
        self.flush_rewards()
        nat.check(self._lib.ph_gae(self._h, self._rb, self.values.data_ptr(), self._last_episode_starts.data_ptr(),
                                   rb.gamma, rb.gae_lambda, int(rb.gae_mode)))

    def finish_update(self) -> None:
        """bookkeeping after the PPO update: buffer reset (agents.py:157)"""
        rb = self.model.rollout_buffer
        self.iteration += 1
        rb.pos, rb.full = 0, False   # rows are fully overwritten by the next rollout; no memset needed on this path
        self.n_steps = 0

    def learn_from_buffer(self) -> None:
        """GAE, PPO update, buffer reset (agents.py:126-158)."""
        self.compute_returns()
        self.model.train(sync_stats=self.sync_stats)
        self.finish_update()

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


class RPSTables:
    """What RPSEnv (envs/rps.py; reference rpsgym/rps.py:8-48) hands two agents over n_steps rounds of n_envs tables, as the
    exchange rollouts' inputs: the constant observation [0], base reward 0 (the whole reward is the joint action's payoff:
    FusedSelfPlayRollout(..., reward_rule="rps", bonus=1.0)) and every round ends its episode."""

    def __init__(self, n_envs: int, n_steps: int, device):
        T, E = n_steps, n_envs
        self.obs = th.zeros((T, E, 1), dtype=th.float32, device=device)
        self.rewards = th.zeros((T, E), dtype=th.float32, device=device)
        self.dones = th.ones((T, E), dtype=th.float32, device=device)
        self.T, self.E = T, E


def run_iteration_eager(agent: VecOnPolicyAgent, data: SyntheticRollouts, scripted: bool = False) -> None:
    """one PPO iteration: T x (get_action, update), then GAE + train at the head of the next get_action -- here
    invoked explicitly so an iteration is self-contained.  scripted: the T steps as one launch (rollout_scripted)."""
    agent.bind_stream()
    if scripted:
        agent.rollout_scripted(data)
    else:
        for t in range(data.T):
            agent.get_action(data.obs[t])
            agent.update(data.rewards[t], data.dones[t])
    agent.learn_from_buffer()


class IterationGraph:
    """One whole PPO iteration of one agent captured as a hipGraph on the agent's own stream."""

    def __init__(self, agent: VecOnPolicyAgent, data: SyntheticRollouts, stream: th.cuda.Stream, scripted: bool = False):
        self.agent, self.data, self.stream, self.scripted = agent, data, stream, scripted
        pol = agent.model.policy
        self.epoch_word = th.zeros(1, dtype=th.int64, device=pol.device)
        nat.check(pol.ctx.lib.ph_ctx_set_rng_epoch(pol.ctx.handle, self.epoch_word.data_ptr()))
        agent.model.device_permutations = True   # in-kernel Feistel permutations: nothing host-generated per replay
        with th.cuda.stream(stream):
            run_iteration_eager(agent, data, scripted)     # warm-up outside capture: sizes the workspace, caches the spec
            run_iteration_eager(agent, data, scripted)
            stream.synchronize()
            agent.bind_stream()
            lib, h = pol.ctx.lib, pol.ctx.handle
            nat.check(lib.ph_graph_begin(h))
            try:
                run_iteration_eager(agent, data, scripted)
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


def run_joint_iteration_eager(agents, datas, streams) -> None:
    """one PPO iteration of several independent learners: every learner's rollout and GAE on its own stream, then ONE joint
    update call (PPO.train_joint) that chains the learners' gradient launches"""
    for agent, data, stream in zip(agents, datas, streams):
        with th.cuda.stream(stream):
            agent.bind_stream()
            for t in range(data.T):
                agent.get_action(data.obs[t])
                agent.update(data.rewards[t], data.dones[t])
            agent.compute_returns()
    PPO.train_joint([agent.model for agent in agents])
    for agent in agents:
        agent.finish_update()


class JointIterationGraph:
    """One whole PPO iteration of ALL local learners as a single hipGraph: the learners' streams fork from the first one at
    the start of the capture and join it at the end; the update is the chained joint call."""

    def __init__(self, agents, datas, streams):
        self.agents, self.datas, self.streams = list(agents), list(datas), list(streams)
        lead = self.agents[0].model.policy
        self.epoch_words = []
        for agent in self.agents:
            pol = agent.model.policy
            word = th.zeros(1, dtype=th.int64, device=pol.device)
            nat.check(pol.ctx.lib.ph_ctx_set_rng_epoch(pol.ctx.handle, word.data_ptr()))
            self.epoch_words.append(word)
            agent.model.device_permutations = True
        s0, others = self.streams[0], self.streams[1:]
        fork, joins = th.cuda.Event(), [th.cuda.Event() for _ in others]
        for _ in range(2):                    # warm-up outside capture: sizes the workspaces, creates the events
            run_joint_iteration_eager(self.agents, self.datas, self.streams)
            fork.record(s0)
            for s, j in zip(others, joins):
                s.wait_event(fork)
                j.record(s)
                s0.wait_event(j)
        th.cuda.synchronize(lead.device)
        with th.cuda.stream(s0):
            self.agents[0].bind_stream()
        lib, h = lead.ctx.lib, lead.ctx.handle
        nat.check(lib.ph_graph_begin(h))
        try:
            fork.record(s0)
            for s in others:
                s.wait_event(fork)            # the other learners' streams join the capture
            run_joint_iteration_eager(self.agents, self.datas, self.streams)
            for agent in self.agents:
                pol = agent.model.policy
                nat.check(pol.ctx.lib.ph_rng_epoch_advance(pol.ctx.handle))
            for s, j in zip(others, joins):
                j.record(s)
                s0.wait_event(j)
        finally:
            gid = C.c_int(-1)
            nat.check(lib.ph_graph_end(h, C.byref(gid)))
        self.graph_id = gid.value
        self._lib, self._h = lib, h

    def launch(self) -> None:
        nat.check(self._lib.ph_graph_launch(self._h, self.graph_id))
        for agent, data in zip(self.agents, self.datas):
            agent.iteration += 1
            agent.num_timesteps += data.T * data.E


class FusedSelfPlayRollout:
    """Agent-per-GPU rollout of the local agents with ONE kernel launch and ONE collective per environment step.

    Step t: `ph_policy_step_multi` runs the policy forward + rollout-buffer row write of every local agent (and applies
    step t-1's reward, including the shared coordination term computed from the joint action gathered at t-1), the
    actions land directly in the exchange buffer, then the caller-visible all-gather makes the joint action of step t
    available everywhere (multiagentenv.py:149-170 with the in-process hand-off replaced by RCCL).  The per-step launch
    records are prebuilt (pointers into the HBM-resident synthetic inputs), so the host does two calls per step; GAE +
    PPO update of the local agents then run concurrently on separate streams."""

    def __init__(self, agents, datas, exchange, stream: th.cuda.Stream, bonus: float = 0.01, update_graphs: bool = True,
                 masks=None, mask_mode: int = 2, persistent: Optional[bool] = None, reward_rule: str = "match"):
        """masks[i]: (T, E, L) uint8 action masks of local agent i's steps (SURVEY.md 8d, config 5 variant) or None.
        mask_mode 2 (default) is the reference's plain PPO partner: the policy never sees the mask (agents.py:162 hands it
        obs.obs), the environment replaces an illegal sample by the first legal index (pettingzoo.py:81-82) and the buffer row
        keeps the sample; 1 = ModularPolicy's logit offset (policies.py:330-333) plus that fix-up; 0 = the offset only.
        persistent: None = use the one-launch exchange rollout (ph_selfplay_rollout_persistent) whenever the peer-to-peer
        route is up and the launch fits the chip."""
        self.agents, self.datas, self.exchange, self.stream, self.bonus = agents, datas, exchange, stream, bonus
        # reward_rule "rps": the joint action pays rock-paper-scissors (bonus * payoff of (own, partner's), rps.py:41-45) instead of
        # the synthetic driver's match bonus -- with zero base rewards and every step a terminal one this IS RPSEnv for E tables
        for a in agents:
            a.model.policy.ctx.set_joint_reward_rule(reward_rule)
        self.masks, self.mask_mode, self.want_persistent = masks, int(mask_mode), persistent
        self.update_graphs, self._update_gid, self._iterations_run = update_graphs, None, 0
        dev = agents[0].model.policy.device
        n = len(agents)
        self.T = datas[0].T
        lead = agents[0].model.policy
        self._lib, self._h = lead.ctx.lib, lead.ctx.handle
        self.partner = [th.zeros(1, dtype=th.int32, device=dev) for _ in agents]
        self.epoch_word = th.zeros(1, dtype=th.int64, device=dev)
        self._side = [th.cuda.Stream(device=dev) for _ in agents[1:]]
        self._fork = th.cuda.Event()
        self._join = [th.cuda.Event() for _ in agents[1:]]
        self.first_start = th.ones(agents[0].E, dtype=th.float32, device=dev)
        if getattr(exchange, "want_p2p", False) or getattr(exchange, "requested_route", None):
            exchange.setup(lead.ctx, self.epoch_word, self.T)
        for i, a in enumerate(agents):
            a.actions = exchange.local[i].view(a.E, 1)       # forward writes straight into the exchange buffer
            a.model.device_permutations = True
            nat.check(a.model.policy.ctx.lib.ph_ctx_set_rng_epoch(a.model.policy.ctx.handle, self.epoch_word.data_ptr()))
        self.route_checked = False
        self._build_calls()

    def _build_calls(self) -> None:
        """prebuilt launch records: one contiguous [T][n] array (ph_selfplay_rollout walks it), `calls[t]` = step t's slice.
        They hold pointers into the exchange's joint-action buffers, so they are rebuilt when the exchange route changes."""
        agents, datas, exchange, bonus = self.agents, self.datas, self.exchange, self.bonus
        n = len(agents)
        self._all_calls = (nat.PhStepCall * (n * self.T))()
        self.calls = []
        for t in range(self.T):
            arr = (nat.PhStepCall * n).from_buffer(self._all_calls, t * n * C.sizeof(nat.PhStepCall))
            for i, (a, d) in enumerate(zip(agents, datas)):
                pol, rb, c = a.model.policy, a.model.rollout_buffer, arr[i]
                c.spec, c.params, c.obs, c.n = C.pointer(pol.spec), pol.params.data_ptr(), d.obs[t].data_ptr(), a.E
                c.action_mask = None if self.masks is None or self.masks[i] is None else self.masks[i][t].data_ptr()
                c.seed, c.counter = pol._seed, t + 1
                c.deterministic = 0 if not c.action_mask else (
                    0, nat.PH_STEP_FIX_ILLEGAL, nat.PH_STEP_FIX_ILLEGAL | nat.PH_STEP_MASK_ENV_ONLY)[self.mask_mode]
                c.actions_i32, c.values, c.log_probs = a.actions.data_ptr(), a.values.data_ptr(), a.log_probs.data_ptr()
                c.rb, c.pos = C.pointer(rb.c_struct()), t
                c.episode_start_in = (self.first_start if t == 0 else d.dones[t - 1]).data_ptr()
                if t > 0:
                    c.pending_reward, c.joint_actions = d.rewards[t - 1].data_ptr(), exchange.joint_slot(t - 1).data_ptr()
                    c.n_seats, c.seat, c.partner_seat, c.bonus = exchange.n_seats, exchange.seat(i), self.partner[i].data_ptr(), bonus
            self.calls.append(arr)
        # the same rollout as ONE launch (ph_selfplay_rollout_persistent): one record per local agent
        self._roll_calls = (nat.PhRolloutCall * n)()
        for i, (a, d) in enumerate(zip(agents, datas)):
            pol, rb, c = a.model.policy, a.model.rollout_buffer, self._roll_calls[i]
            c.spec, c.params, c.n = C.pointer(pol.spec), pol.params.data_ptr(), a.E
            c.obs_seq, c.rew_seq, c.done_seq = d.obs.data_ptr(), d.rewards.data_ptr(), d.dones.data_ptr()
            c.mask_seq = None if self.masks is None or self.masks[i] is None else self.masks[i].data_ptr()
            c.episode_start0, c.seed, c.counter0 = self.first_start.data_ptr(), pol._seed, 1
            c.mask_mode = self.mask_mode
            c.actions_i32, c.values, c.log_probs = a.actions.data_ptr(), a.values.data_ptr(), a.log_probs.data_ptr()
            c.rb = C.pointer(rb.c_struct())
            c.n_seats, c.seat, c.partner_seat, c.bonus = exchange.n_seats, exchange.seat(i), self.partner[i].data_ptr(), bonus

    def persistent_ok(self) -> bool:
        """can the rollouts run as the one-launch exchange rollout?  Needs the peer-to-peer words, the 16-row forward's shapes,
        2 T word slots, and every workgroup of the launch resident at once (value workgroups poll): the grid of all ranks sharing
        this device against what the runtime's occupancy query reports for the kernel.  The two rollout forms use different word
        slots, so the verdict is taken ONCE per exchange route and is the same on every rank (all-reduced): a rank whose local
        inputs differ (an environment variable, uneven ranks per GPU) takes the others to the per-step form with it."""
        import os
        ex = self.exchange
        # keyed on the exchange's attach generation (a counter the exchange advances at every (re-)attach on every rank), not on
        # the identity of the attached object: an id() can come back after a re-attach
        key = (getattr(ex, "attach_generation", 0), ex.p2p is not None, getattr(ex, "route", None))
        if getattr(self, "_persistent_verdict", None) is not None and self._persistent_verdict[0] == key:
            return self._persistent_verdict[1]
        ok = not (self.want_persistent is False or os.environ.get("PH_EXCHANGE_PERSISTENT", "1") == "0" or ex.p2p is None)
        if ok:
            # whatever goes wrong locally (a failing occupancy query included) becomes this rank's "no": every rank must reach
            # the collective below, a raise here would leave the others waiting in it
            try:
                lay = self.agents[0].model.policy.layout
                E = self.agents[0].E
                ok = lay.F <= 64 and lay.A == 1 and lay.L <= 8 and E < 16384
                ok = ok and not (ex.p2p.T < self.T or ex.p2p.ll_slots < 2 * ex.p2p.T)
                if ok:
                    cap = C.c_int(0)
                    nat.check(self._lib.ph_selfplay_rollout_persistent_capacity(self._h, C.byref(cap)))
                    ok = len(self.agents) * 2 * ((E + 15) // 16) * max(getattr(ex, "ranks_on_device", 1), 1) <= cap.value
            except Exception:  # noqa: BLE001 -- the verdict is "no", agreed with the other ranks below
                ok = False
        if getattr(ex, "world", 1) > 1 and hasattr(ex, "_everyone"):
            ok = ex._everyone(bool(ok))
        self._persistent_verdict = (key, bool(ok))
        return bool(ok)

    def set_pairing(self, pairing_round: int) -> None:
        ex = self.exchange
        for i, p in enumerate(self.partner):
            p.fill_(ex.partner_of(ex.seat(i), pairing_round))

    def run_iteration(self, pairing_round: int) -> None:
        """must be called with `self.stream` current"""
        self.set_pairing(pairing_round)
        agents, lib, h, ex, T = self.agents, self._lib, self._h, self.exchange, self.T
        agents[0].bind_stream()
        one_launch = self.persistent_ok()
        self.last_rollout_mode = "persistent" if one_launch else ("p2p" if ex.p2p is not None else "stepwise")
        if one_launch:
            # all T steps of every local agent in ONE launch, the per-step action hand-off done in-kernel over the stamp words
            nat.check(lib.ph_selfplay_rollout_persistent(h, len(agents), self._roll_calls, T, C.byref(ex.p2p),
                                                         int(max(getattr(ex, "ranks_on_device", 1), 1))))
        elif ex.p2p is not None:
            # peer-to-peer stores over xGMI: T x (fused launch, push, wait) enqueued by one native call
            nat.check(lib.ph_selfplay_rollout_p2p(h, len(agents), self._all_calls, T, ex.local.data_ptr(), C.byref(ex.p2p)))
        elif ex.native_ctx is not None and ex.native_ctx.handle.value == h.value:
            # engine-side exchange: the T x (fused launch, RCCL all-gather) chain is enqueued by one native call
            nat.check(lib.ph_selfplay_rollout(h, len(agents), self._all_calls, T, ex.local.data_ptr(),
                                              ex.joint.data_ptr(), ex.local.numel()))
        else:
            for t in range(T):
                nat.check(lib.ph_policy_step_multi(h, len(agents), self.calls[t]))
                ex.gather_inplace()
        if not self.route_checked and getattr(ex, "world", 1) > 1 and hasattr(ex, "verify_route"):
            # first real iteration on this node: the joint action consumed for the last step must be what torch.distributed
            # gathers from the same local actions, on every rank, with no poll timed out -- otherwise the native route is
            # dropped for good (one synchronising check, once)
            self.route_checked = True
            if not ex.verify_route(ex.joint_slot(T - 1)):
                import sys
                print(f"[pantheonrl_amd.vec] exchange route {ex.route!r} failed verification after the first iteration; "
                      f"falling back to {ex.demote()!r}", file=sys.stderr)
                ex.gather_inplace()          # the last step's joint action again, through the route now in use
                self._build_calls()
        # the last step's reward (no further forward to carry it)
        for i, (a, d) in enumerate(zip(agents, self.datas)):
            a.bind_stream()
            a.model.rollout_buffer.pos = T
            a._last_episode_starts = d.dones[T - 1]
            if one_launch:
                a._pending = None       # the one-launch rollout credited the last step's reward itself
            else:
                a.update_joint(d.rewards[T - 1], d.dones[T - 1], ex.joint_slot(T - 1), ex.seat(i), self.partner[i], self.bonus)
        # GAE + PPO update: local learners are independent -> concurrent on forked streams; each learner's 1 + 1 + 3*40
        # launches are replayed from a hipGraph captured on its stream at the second iteration (no collective inside)
        main = th.cuda.current_stream()
        self._fork.record(main)
        self._iterations_run += 1
        capture_now = (self.update_graphs and self._update_gid is None and self._iterations_run == 2
                       and not any(a.sync_stats for a in agents))   # a stats read-back cannot be captured
        gids = []
        for i, a in enumerate(agents):
            stream = main if i == 0 else self._side[i - 1]
            if i > 0:
                stream.wait_event(self._fork)
            with th.cuda.stream(stream):
                a.bind_stream()
                alib, ah = a.model.policy.ctx.lib, a.model.policy.ctx.handle
                if self._update_gid is not None:
                    nat.check(alib.ph_graph_launch(ah, self._update_gid[i]))
                    a.finish_update()
                elif capture_now:
                    stream.synchronize()
                    nat.check(alib.ph_graph_begin(ah))
                    try:
                        a.learn_from_buffer()
                    finally:
                        gid = C.c_int(-1)
                        nat.check(alib.ph_graph_end(ah, C.byref(gid)))
                    gids.append(gid.value)
                    nat.check(alib.ph_graph_launch(ah, gid.value))
                else:
                    a.learn_from_buffer()
                if i > 0:
                    self._join[i - 1].record(stream)
        if capture_now:
            self._update_gid = gids
        for ev in self._join:
            main.wait_event(ev)
        agents[0].bind_stream()
        nat.check(lib.ph_rng_epoch_advance(h))
        for a in agents:
            a.num_timesteps += T * a.E
