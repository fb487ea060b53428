"""Device-resident vectorised games: E independent copies stepped by one kernel launch.

`VecRPS` is the n_envs = E form of RPSEnv (reference pantheonrl/envs/rpsgym/rps.py:33-48): both seats act simultaneously
on the constant observation [0], payoffs follow the integer rule (ego - alt + 3) % 3, every episode lasts one step.
`selfplay_iteration` is the vectorised counterpart of `trainer.py RPS-v0 PPO PPO`: two learning agents, every callback
of the reference's step loop (multiagentenv.py:149-170) applied to E-long device tensors.

`VecLiarsDice` holds the state of E Liar's Dice tables on the device and applies `LiarEnv.player_step`
(liar.py:58-83) through `ph_liar_step`; dice are rolled on the host with the reference's draw order so a Python
`LiarEnv` fed the same dice is the bit-exact checker.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch as th

from .. import _native as nat
from ..spaces import Discrete, MultiDiscrete
from ..vec import VecOnPolicyAgent


class VecRPS:
    observation_space = Discrete(1)
    action_space = Discrete(3)

    def __init__(self, n_envs: int, ctx: nat.Context, device):
        self.E, self.ctx, self.device = n_envs, ctx, device
        self.obs = th.zeros((n_envs, 1), dtype=th.float32, device=device)      # NULL_OBS for every env, both seats
        self.dones = th.ones(n_envs, dtype=th.float32, device=device)          # one-step episodes: always done
        self.ego_rew = th.zeros(n_envs, dtype=th.float32, device=device)
        self.alt_rew = th.zeros(n_envs, dtype=th.float32, device=device)

    def step(self, ego_actions: th.Tensor, alt_actions: th.Tensor):
        """(E,) or (E,1) int32 device tensors -> (ego rewards, partner rewards, dones) device tensors"""
        self.ctx.set_stream(th.cuda.current_stream(self.device).cuda_stream)
        nat.check(self.ctx.lib.ph_rps_step(self.ctx.handle, ego_actions.data_ptr(), alt_actions.data_ptr(),
                                           self.ego_rew.data_ptr(), self.alt_rew.data_ptr(), self.E))
        return self.ego_rew, self.alt_rew, self.dones


def selfplay_iteration(env: VecRPS, ego: VecOnPolicyAgent, alt: VecOnPolicyAgent, n_steps: int) -> None:
    """n_steps simultaneous steps of E games, then both learners consume their rollouts."""
    ego.bind_stream()
    alt.bind_stream()
    for _ in range(n_steps):
        a0 = ego.get_action(env.obs)
        a1 = alt.get_action(env.obs)
        r0, r1, d = env.step(a0, a1)
        ego.update(r0, d)
        alt.update(r1, d)
        ego.flush_rewards()   # env.ego_rew / alt_rew are reused next step: apply now instead of deferring
        alt.flush_rewards()
    ego.learn_from_buffer()
    alt.learn_from_buffer()


class VecLiarsDice:
    N_SIDES, N_DICE, MAX_MOVES = 6, 6, 12
    observation_space = MultiDiscrete([7] * 6 + [7, 12] * 12)
    action_space = MultiDiscrete([7, 12])

    def __init__(self, n_envs: int, ctx: nat.Context, device):
        self.E, self.ctx, self.device = n_envs, ctx, device
        i32 = lambda *s: th.zeros(*s, dtype=th.int32, device=device)  # noqa: E731
        self.hands, self.history, self.nmoves = i32(n_envs, 12), i32(n_envs, 24), i32(n_envs)
        self.obs_next = th.zeros((n_envs, 30), dtype=th.float32, device=device)
        self.rewards = th.zeros((n_envs, 2), dtype=th.float32, device=device)
        self.done = th.zeros(n_envs, dtype=th.uint8, device=device)

    def reset(self, hands: np.ndarray) -> None:
        """hands (E, 12): ego histogram then partner histogram (roll them with envs.liar.roll_hand for reference order)"""
        self.hands.copy_(th.as_tensor(np.asarray(hands, np.int32)))
        self.history.zero_()
        self.nmoves.zero_()

    def player_step(self, actions: th.Tensor, is_ego: th.Tensor, active: th.Tensor = None):
        """actions (E,2) int32, is_ego (E) uint8, active (E) uint8 or None -> (obs of the other player, rewards (E,2), done)"""
        self.ctx.set_stream(th.cuda.current_stream(self.device).cuda_stream)
        nat.check(self.ctx.lib.ph_liar_step(self.ctx.handle, self.hands.data_ptr(), self.history.data_ptr(),
                                            self.nmoves.data_ptr(), actions.data_ptr(), is_ego.data_ptr(),
                                            nat.ptr(active), self.obs_next.data_ptr(), self.rewards.data_ptr(),
                                            self.done.data_ptr(), self.E))
        return self.obs_next, self.rewards, self.done


class RaggedVecOnPolicyAgent:
    """OnPolicyAgent for the PARTNER seat of a vectorised turn-based game.

    In a turn-based game the partner does not act in every environment at every vectorised step (a game may end on the
    ego's move, a new game may start with either player), so its rollout buffer cannot advance one row per step.  Each
    environment e owns column e of the (T, E) buffer and its own write row `pos[e]`; the learner trains
    when every column is full, which is the E-environment reading of "train before acting once the buffer is full"
    (agents.py:126).  Per environment the callbacks keep the reference's semantics:
      * a recorded action opens its row for late additive rewards until the agent is asked to act again (agents.py:198),
      * `episode_start` of a row = an episode ended since the previous recorded row (agents.py:176,197),
      * GAE bootstraps with V of the last recorded observation and dones = "the game ended before the next action"
        (agents.py:127-130, quirk D-1).
    `min_full` (default E = wait for every column) lowers the trigger: the learner trains as soon as that many columns are
    full, on exactly the full columns (compacted into a (T, n) buffer, `ph_buffer_compact_columns`), and only those columns
    start over.  Round-robin partner selection needs this: an environment spends whole episodes with other partners, so
    waiting for all columns starves the learner and drops the transitions that keep arriving in already-full columns.
    """

    def __init__(self, model):
        self.model = model
        pol, rb = model.policy, model.rollout_buffer
        E, lay, dev = rb.n_envs, pol.layout, pol.device
        self.E, self.T = E, rb.buffer_size
        u8 = lambda v: th.full((E,), v, dtype=th.uint8, device=dev)  # noqa: E731
        self.pos = th.zeros(E, dtype=th.int32, device=dev)
        self.actions = th.zeros((E, lay.A), dtype=th.int32, device=dev)
        self.values = th.zeros(E, dtype=th.float32, device=dev)
        self.log_probs = th.zeros(E, dtype=th.float32, device=dev)
        self.boundary, self.term, self.open = u8(1), u8(0), u8(0)
        self.iteration = 0
        self.num_timesteps = 0
        self.min_full = E
        self._compact = {}          # n -> RolloutBuffer (T, n) reused between partial updates
        self._lib, self._h = pol.ctx.lib, pol.ctx.handle
        self._spec, self._rb = C.byref(pol.spec), C.byref(rb.c_struct())

    def get_action(self, obs: th.Tensor, rec_mask: th.Tensor) -> th.Tensor:
        """forward in every env; record the transition where rec_mask is set and the column still has room"""
        pol = self.model.policy
        pol._bind()
        room = self.pos < self.T
        can = (rec_mask.bool() & room).to(th.uint8)
        blocked = rec_mask.bool() & ~room
        es = self.boundary.to(th.float32)
        pol._counter += 1
        nat.check(self._lib.ph_policy_forward_ragged(
            self._h, self._spec, pol.params.data_ptr(), obs.data_ptr(), None, pol._seed, pol._counter, 0,
            self.actions.data_ptr(), self.values.data_ptr(), self.log_probs.data_ptr(), self._rb, self.pos.data_ptr(),
            can.data_ptr(), es.data_ptr()))
        nat.check(self._lib.ph_ragged_advance(self._h, self._rb, self.pos.data_ptr(), can.data_ptr()))
        canb = can.bool()
        self.boundary = th.where(canb, th.zeros_like(self.boundary), self.boundary)
        self.term = th.where(canb, th.zeros_like(self.term), self.term)
        self.open = th.where(canb, th.ones_like(self.open), th.where(blocked, th.zeros_like(self.open), self.open))
        self.num_timesteps += int(self.E)
        return self.actions

    def update(self, reward: th.Tensor, done: th.Tensor, mask: th.Tensor) -> None:
        """credit `reward` to the last recorded action of the envs in `mask` (whose rows are still open); remember the
        episode boundaries"""
        self.model.policy._bind()
        m = (mask.bool() & self.open.bool())
        m8 = m.to(th.uint8)
        nat.check(self._lib.ph_buffer_add_reward_ragged(self._h, self._rb, self.pos.data_ptr(), reward.data_ptr(),
                                                        m8.data_ptr()))
        d = done.bool()
        self.boundary = (self.boundary.bool() | d).to(th.uint8)
        self.term = (self.term.bool() | (m & d)).to(th.uint8)

    def full(self) -> bool:
        return int((self.pos >= self.T).sum().item()) >= min(self.min_full, self.E)

    def _learn_from_columns(self, cols: th.Tensor) -> None:
        """GAE + PPO update on the full columns `cols` only; the other columns keep filling"""
        from ..ppo import RolloutBuffer
        model, rb = self.model, self.model.rollout_buffer
        pol = model.policy
        n = int(cols.numel())
        sub = self._compact.get(n)
        if sub is None:
            sub = self._compact[n] = RolloutBuffer(rb.buffer_size, rb.observation_space, rb.action_space, rb.device, pol.ctx,
                                                   pol.spec, gae_lambda=rb.gae_lambda, gamma=rb.gamma, n_envs=n)
        pol._bind()
        cols32 = cols.to(th.int32).contiguous()
        nat.check(self._lib.ph_buffer_compact_columns(self._h, self._spec, self._rb, C.byref(sub.c_struct()),
                                                      cols32.data_ptr(), n))
        last_v = self.values.index_select(0, cols).contiguous()
        dones = self.term.to(th.float32).index_select(0, cols).contiguous()
        nat.check(self._lib.ph_gae(self._h, C.byref(sub.c_struct()), last_v.data_ptr(), dones.data_ptr(), rb.gamma,
                                   rb.gae_lambda, int(rb.gae_mode)))
        sub.pos, sub.full = sub.buffer_size, True
        model.rollout_buffer = sub
        try:
            model.train(sync_stats=False)
        finally:
            model.rollout_buffer = rb
        self.pos[cols] = 0
        self.open[cols] = 0
        self.term[cols] = 0
        self.iteration += 1

    def learn_from_buffer(self) -> None:
        model, rb = self.model, self.model.rollout_buffer
        ready = self.pos >= self.T
        if not bool(ready.all().item()):
            self._learn_from_columns(th.nonzero(ready).reshape(-1))
            return
        model.policy._bind()
        dones = self.term.to(th.float32)
        nat.check(self._lib.ph_gae(self._h, self._rb, self.values.data_ptr(), dones.data_ptr(), rb.gamma, rb.gae_lambda,
                                   int(rb.gae_mode)))
        rb.pos, rb.full = rb.buffer_size, True
        model.train(sync_stats=False)
        rb.pos, rb.full = 0, False
        self.pos.zero_()
        self.open.zero_()
        self.term.zero_()
        self.iteration += 1


class VecLiarSelfPlay:
    """`trainer.py LiarsDice-v0 PPO PPO` (BASELINE config 2) with n_envs tables resident on the device.

    One vectorised step is one `MultiAgentEnv.step` of every table from the ego's point of view (multiagentenv.py:
    172-215): the ego moves, the partner replies in the tables that are still running, finished tables are re-dealt and --
    where the partner opens the new game -- the partner moves once more, so every table is back at the ego's turn.
    The ego is a rectangular VecOnPolicyAgent (one row per table per step); the partner is ragged.

    native=True (default): the whole step is ONE engine call (`ph_liar_selfplay_step`: the three policy forwards and ONE
    book-keeping launch after each -- a table's state is touched by its own lane only -- every mask stays on the device, no host
    synchronisation).  native=False walks the same step with the per-call entry points and torch masks -- the
    readable statement of the protocol, and the bit-exact cross-check of the native step (both use the same RNG counters:
    step c -> ego forward c, partner forwards 2c and 2c+1, dice c)."""

    def __init__(self, n_envs: int, ego: VecOnPolicyAgent, alt: RaggedVecOnPolicyAgent, seed: int = 0,
                 probegostart: float = 0.5, native: bool = True):
        self.E, self.ego, self.alt, self.native = n_envs, ego, alt, bool(native)
        pol = ego.model.policy
        self.dev = pol.device
        self.env = VecLiarsDice(n_envs, pol.ctx, self.dev)
        self.seed, self.counter, self.probegostart = int(seed), 0, float(probegostart)
        E, dev = n_envs, self.dev
        u8 = lambda v=0: th.full((E,), v, dtype=th.uint8, device=dev)  # noqa: E731
        self.ego_first = u8()
        self.obs_ego = th.zeros((E, 30), dtype=th.float32, device=dev)
        self.obs_alt = th.zeros((E, 30), dtype=th.float32, device=dev)
        self.alt_acted = u8()                                      # should_update of the partner seat, per table
        self.ones8, self.zeros8 = u8(1), u8(0)
        self._episodes_dev = th.zeros(1, dtype=th.int64, device=dev)
        self.steps_done = 0
        import os
        self.persistent = self.native and os.environ.get("LIAR_PERSISTENT", "1") != "0"   # whole rollouts as one launch
        if self.native:
            self._build_native()
            self._done.fill_(1)
            self._native_call(0, deal_only=True)
        else:
            self._deal(self.ones8, 0)

    @property
    def episodes(self) -> int:
        return int(self._episodes_dev.item())

    # -- helpers -----------------------------------------------------------------------------------------------------
    def _bind(self):
        stream = th.cuda.current_stream(self.dev).cuda_stream
        self.env.ctx.set_stream(stream)
        for agent in (self.ego, self.alt):
            agent.model.policy.ctx.set_stream(stream)

    def _build_native(self) -> None:
        E, dev, env, ego, alt = self.E, self.dev, self.env, self.ego, self.alt
        f32 = lambda *shape: th.zeros(shape, dtype=th.float32, device=dev)  # noqa: E731
        u8 = lambda: th.zeros(E, dtype=th.uint8, device=dev)               # noqa: E731
        self._obs_next, self._rew1, self._rew2, self._es_alt = f32(E, 30), f32(E, 2), f32(E, 2), f32(E)
        self._done1, self._done2, self._running, self._can = u8(), u8(), u8(), u8()
        self._alt_opens, self._ego_opens, self._done = u8(), u8(), u8()
        ego._last_episode_starts = ego._last_episode_starts.clone()    # updated in place by the step from here on
        s = nat.PhLiarSelfPlay()
        s.n, s.spec = E, C.pointer(ego.model.policy.spec)
        s.hands, s.history, s.nmoves = env.hands.data_ptr(), env.history.data_ptr(), env.nmoves.data_ptr()
        s.ego_first, s.dice_seed, s.probegostart = self.ego_first.data_ptr(), self.seed, self.probegostart
        pe, pa = ego.model.policy, alt.model.policy
        self._ego_rbc, self._alt_rbc = ego.model.rollout_buffer.c_struct(), alt.model.rollout_buffer.c_struct()
        s.ego_params, s.ego_rb, s.ego_actions = pe.params.data_ptr(), C.pointer(self._ego_rbc), ego.actions.data_ptr()
        s.ego_values, s.ego_log_probs = ego.values.data_ptr(), ego.log_probs.data_ptr()
        s.ego_episode_start, s.ego_seed = ego._last_episode_starts.data_ptr(), pe._seed
        s.alt_params, s.alt_rb, s.alt_actions = pa.params.data_ptr(), C.pointer(self._alt_rbc), alt.actions.data_ptr()
        s.alt_values, s.alt_log_probs, s.alt_pos = alt.values.data_ptr(), alt.log_probs.data_ptr(), alt.pos.data_ptr()
        s.alt_boundary, s.alt_term, s.alt_open = alt.boundary.data_ptr(), alt.term.data_ptr(), alt.open.data_ptr()
        s.alt_acted, s.alt_seed = self.alt_acted.data_ptr(), pa._seed
        s.obs_ego, s.obs_alt, s.episodes = self.obs_ego.data_ptr(), self.obs_alt.data_ptr(), self._episodes_dev.data_ptr()
        s.obs_next, s.rew1, s.rew2, s.es_alt = (t.data_ptr() for t in (self._obs_next, self._rew1, self._rew2, self._es_alt))
        s.done1, s.done2, s.running, s.can = (t.data_ptr() for t in (self._done1, self._done2, self._running, self._can))
        s.alt_opens, s.ego_opens, s.done = self._alt_opens.data_ptr(), self._ego_opens.data_ptr(), self._done.data_ptr()
        s.zeros8, s.ones8 = self.zeros8.data_ptr(), self.ones8.data_ptr()
        self._desc = s

    def _native_call(self, counter: int, deal_only: bool = False, ego_pos: int = -1) -> None:
        self._bind()
        ctx, rb = self.env.ctx, self.ego.model.rollout_buffer
        nat.check(ctx.lib.ph_liar_selfplay_step(ctx.handle, C.byref(self._desc), int(rb.pos if ego_pos < 0 else ego_pos),
                                                int(counter), int(deal_only)))

    def rollout_persistent(self, n_steps: int, first_counter: int, ego_pos: int = 0) -> None:
        """n_steps vectorised steps as ONE persistent launch (`ph_liar_selfplay_rollout`: one workgroup owns 16 tables for the
        whole rollout; bitwise the result of n_steps `_native_call`s with counters first_counter, first_counter + 1, ...)"""
        self._bind()
        ctx = self.env.ctx
        nat.check(ctx.lib.ph_liar_selfplay_rollout(ctx.handle, C.byref(self._desc), int(ego_pos), int(n_steps),
                                                   int(first_counter)))

    def _deal(self, reset_mask: th.Tensor, c: int) -> None:
        """(reference path) re-deal the tables in reset_mask; where the partner opens, it moves once"""
        env, lib, h = self.env, self.env.ctx.lib, self.env.ctx.handle
        self._bind()
        nat.check(lib.ph_liar_reset(h, env.hands.data_ptr(), env.history.data_ptr(), env.nmoves.data_ptr(),
                                    reset_mask.data_ptr(), self.ego_first.data_ptr(), self.seed, int(c),
                                    self.probegostart, self.E))
        rm = reset_mask.bool()
        self.alt_acted.copy_(th.where(rm, self.zeros8, self.alt_acted))
        alt_opens = (rm & ~self.ego_first.bool()).to(th.uint8)
        # partner's opening observation and move
        nat.check(lib.ph_liar_obs(h, env.hands.data_ptr(), env.history.data_ptr(), env.nmoves.data_ptr(),
                                  self.zeros8.data_ptr(), alt_opens.data_ptr(), self.obs_alt.data_ptr(), self.E))
        self.alt.model.policy._counter = 2 * c                      # the forward below draws with counter 2c + 1
        a_alt = self.alt.get_action(self.obs_alt, alt_opens)
        self.alt_acted.copy_((self.alt_acted.bool() | alt_opens.bool()).to(th.uint8))
        env.player_step(a_alt, self.zeros8, alt_opens)              # obs_next = ego's observation in those tables
        self.obs_ego.copy_(th.where(alt_opens.bool()[:, None], env.obs_next, self.obs_ego))
        ego_opens = (rm & self.ego_first.bool()).to(th.uint8)
        nat.check(lib.ph_liar_obs(h, env.hands.data_ptr(), env.history.data_ptr(), env.nmoves.data_ptr(),
                                  self.ones8.data_ptr(), ego_opens.data_ptr(), self.obs_ego.data_ptr(), self.E))

    # -- one vectorised MultiAgentEnv.step ---------------------------------------------------------------------------------
    def step(self):
        """-> (E,) uint8 device tensor: tables whose game ended in this step"""
        ego, alt = self.ego, self.alt
        self.steps_done += 1
        c = self.steps_done
        if self.native:
            model, rb = ego.model, ego.model.rollout_buffer
            if ego.n_steps >= model.n_steps:
                ego.learn_from_buffer()
            self._native_call(c)
            rb.pos += 1
            rb.full = rb.pos == rb.buffer_size
            ego.n_steps += 1
            ego.num_timesteps += self.E
            alt.num_timesteps += self.E
            return self._done
        env = self.env
        self._bind()
        ego.model.policy._counter = c - 1
        a_ego = ego.get_action(self.obs_ego)                                   # every table is at the ego's turn
        obs_alt, rew1, done1 = env.player_step(a_ego, self.ones8, None)
        rew1, done1 = rew1.clone(), done1.clone()
        running = (~done1.bool()).to(th.uint8)
        # partners that already acted this game are credited this transition (multiagentenv.py:163-170)
        alt.update(rew1[:, 1].contiguous(), done1, self.alt_acted)
        self.obs_alt.copy_(th.where(running.bool()[:, None], obs_alt, self.obs_alt))
        # partner replies where the game goes on
        alt.model.policy._counter = 2 * c - 1
        a_alt = alt.get_action(self.obs_alt, running)
        self.alt_acted.copy_((self.alt_acted.bool() | running.bool()).to(th.uint8))
        obs_ego, rew2, done2 = env.player_step(a_alt, self.zeros8, running)
        rew2 = th.where(running.bool()[:, None], rew2, th.zeros_like(rew2))
        done2 = (done2.bool() & running.bool())
        alt.update(rew2[:, 1].contiguous(), done2.to(th.uint8), running)
        done = done1.bool() | done2
        # the ego collects both transitions of the step; the last done wins (agents.py:44-47)
        ego.update((rew1[:, 0] + rew2[:, 0]).contiguous(), done.to(th.float32))
        ego.flush_rewards()
        self.obs_ego.copy_(th.where((running.bool() & ~done2)[:, None], obs_ego, self.obs_ego))
        done8 = done.to(th.uint8)
        self._episodes_dev += done8.sum()
        self._deal(done8, c)
        return done8

    def rollout_and_learn(self, n_steps: int) -> None:
        """n_steps vectorised steps, the ego's update, and the partner's whenever all its columns are full"""
        ego, rb = self.ego, self.ego.model.rollout_buffer
        if self.native and self.persistent and rb.pos == 0 and n_steps == rb.buffer_size and ego.n_steps == 0:
            self.rollout_persistent(n_steps, self.steps_done + 1, 0)
            self.steps_done += n_steps
            rb.pos, rb.full = n_steps, True
            ego.n_steps += n_steps
            ego.num_timesteps += n_steps * self.E
            self.alt.num_timesteps += n_steps * self.E
        else:
            for _ in range(n_steps):
                self.step()
        self.ego.learn_from_buffer()
        if self.alt.full():
            self.alt.learn_from_buffer()


class LiarIterationGraph:
    """One whole iteration of the device-resident Liar's Dice self-play with everything a replay must vary resident on the device:
    every random stream is keyed (RNG epoch word, counter) with the step-local counter baked in, and ONE epoch word -- shared by
    the forwards, the dice and the minibatch permutations of both learners -- is advanced once per iteration.

    Persistent mode (default): the rollout is ONE launch (`ph_liar_selfplay_rollout`); then the ego's GAE pass + PPO update replay
    from a hipGraph on this object's stream while -- whenever all its columns are full, the one host decision of the loop -- the
    partner's update runs beside it on a second stream (the general gradient kernel holds one workgroup per CU at this batch
    size, so two learners' launches fill the CUs instead of taking turns); the epoch word is advanced after both joined.
    LIAR_PERSISTENT=0: n_steps x 6 launches + the ego's update as one graph, the partner's update after it (round 1).
    `capture=False` runs the same body launch by launch (the graph's cross-check)."""

    def __init__(self, sp: VecLiarSelfPlay, n_steps: int, capture: bool = True, warmup: int = 2):
        assert sp.native, "the graph replays the engine-side step"
        self.sp, self.T = sp, int(n_steps)
        ego, alt = sp.ego, sp.alt
        assert self.T == ego.model.n_steps and ego.model.rollout_buffer.pos == 0
        self.stream = th.cuda.Stream(device=sp.dev)
        self.side = th.cuda.Stream(device=sp.dev)
        self._fork, self._join = th.cuda.Event(), th.cuda.Event()
        self.epoch_word = th.zeros(1, dtype=th.int64, device=sp.dev)
        for agent in (ego, alt):
            ctx = agent.model.policy.ctx
            nat.check(ctx.lib.ph_ctx_set_rng_epoch(ctx.handle, self.epoch_word.data_ptr()))
            agent.model.device_permutations = True
        self.graph_id = None
        self.split = bool(sp.persistent)          # rollout launch | ego-update graph || partner update | epoch advance
        th.cuda.synchronize(sp.dev)
        with th.cuda.stream(self.stream):
            self._perm_seed = ego.model.permutation_seed + 1
            for _ in range(warmup):          # outside capture: sizes the workspaces, allocates the statistics buffers
                self._iteration(eager=True)
            self.stream.synchronize()
            if capture:
                ctx = sp.env.ctx
                sp._bind()
                nat.check(ctx.lib.ph_graph_begin(ctx.handle))
                try:
                    self._update_body() if self.split else self._body()
                finally:
                    gid = C.c_int(-1)
                    nat.check(ctx.lib.ph_graph_end(ctx.handle, C.byref(gid)))
                self.graph_id = gid.value

    def _update_body(self) -> None:
        ego = self.sp.ego
        ego.compute_returns()
        ego.model.permutation_seed = self._perm_seed - 1     # train() pre-increments: the same baked seed in every replay
        ego.model.train(sync_stats=False)

    def _body(self) -> None:
        sp = self.sp
        for t in range(self.T):
            sp._native_call(t + 1, ego_pos=t)
        self._update_body()
        ctx = sp.env.ctx
        nat.check(ctx.lib.ph_rng_epoch_advance(ctx.handle))

    def _iteration_split(self, eager: bool) -> None:
        sp, ego, alt = self.sp, self.sp.ego, self.sp.alt
        main = th.cuda.current_stream(sp.dev)
        sp.rollout_persistent(self.T, 1, 0)                   # the T steps as ONE launch (counters 1..T)
        train_alt = alt.full()                                # host decision (synchronises on the rollout)
        if train_alt:
            self._fork.record(main)
            self.side.wait_event(self._fork)
            with th.cuda.stream(self.side):
                alt.learn_from_buffer()                       # beside the ego's update
                self._join.record(self.side)
        sp._bind()
        ctx = sp.env.ctx
        if eager:
            self._update_body()
        else:
            nat.check(ctx.lib.ph_graph_launch(ctx.handle, self.graph_id))
        if train_alt:
            main.wait_event(self._join)
        sp._bind()
        nat.check(ctx.lib.ph_rng_epoch_advance(ctx.handle))    # after BOTH learners drew their minibatch orders
        ego.finish_update()
        sp.steps_done += self.T
        ego.num_timesteps += self.T * sp.E
        alt.num_timesteps += self.T * sp.E

    def _iteration(self, eager: bool) -> None:
        if self.split:
            self._iteration_split(eager)
            return
        sp, ego, alt = self.sp, self.sp.ego, self.sp.alt
        if eager:
            self._body()
        else:
            sp._bind()
            ctx = sp.env.ctx
            nat.check(ctx.lib.ph_graph_launch(ctx.handle, self.graph_id))
        ego.finish_update()
        sp.steps_done += self.T
        ego.num_timesteps += self.T * sp.E
        alt.num_timesteps += self.T * sp.E
        if alt.full():
            alt.learn_from_buffer()

    def launch(self) -> None:
        with th.cuda.stream(self.stream):
            self._iteration(eager=self.graph_id is None)
