"""BASELINE config 4 -- Overcooked-simple shaped SimultaneousEnv, one ego against K on-policy partners with exactly one
partner active per episode, ONE AGENT PER GPU.

What the reference does in one process (pantheonrl/common/multiagentenv.py):
  * `reset()` picks the next partner round-robin -- `partnerids = [(id + 1) % K]`, at EVERY reset including the first
    (:118-125, :224) -- and the episode is played with that partner;
  * `step()` asks the current partner for its action on the partner-seat observation (`_get_actions`, :149-161), runs the
    game transition, and hands reward and done to that partner (`_update_players`, :163-170).

With E environments and the agents on different ranks, each environment e carries its own partner id, advanced at that
environment's own reset (the VecEnv auto-reset).  Rank 0 hosts the ego and the E (synthetic) environments, rank 1 + k
hosts partner k.  Per environment step:

  1. rank 0 sends the ROUTING BLOCK (E, 3 + D) f32 to the partner ranks: per environment
     [partner id of this step | reward of the previous step | done of the previous step | partner-seat observation];
  2. partner k consumes the rows whose id equals k: it first credits the previous step's reward / done to the rows it
     recorded then (`Agent.update`, agents.py:186-203), trains if every column of its ragged buffer is full (train before
     acting, agents.py:126), then runs its forward and records the transition in those environments' columns
     (`ph_policy_forward_ragged`: per-environment write rows);
  3. all ranks all-gather their actions (E int32 each); rank 0 takes, per environment, the action of that environment's
     partner, applies the transition (`ph_roundrobin_env_step`: shared reward, done, round-robin advance of the partner id
     where the episode ended) and credits the ego (`update`, folded into its next step's launch).

Partner selection stays a rule over small integers (the kernel applies the reference's `(id + 1) % K`); no gradient or
parameter ever crosses ranks.

Two carriers for steps 1 and 3.  The ENGINE-SIDE one (default on GPUs, `RoundRobinLink`): every rank owns a fine-grained receive
area that the others map through HIP IPC; rank 0 stores the routing block straight into the partners' areas and the partners
store their actions straight into rank 0's, each followed by a monotonic stamp that the consuming kernel polls (bounded) --
and the T steps of an iteration are enqueued by ONE native call per rank (`ph_roundrobin_ego_iteration` /
`ph_roundrobin_partner_iteration`): no host tensor op, collective call or synchronisation per environment step.  With it a
partner decides once per iteration (one host read) whether its full columns are trained on; the reference's partner checks
before every action (agents.py:126).  The HOST-DRIVEN one (`native=False`, CPU protocol tests): torch.distributed's broadcast
of E*(3+D)*4 B and all-gather of (K+1)*E*4 B per step, every step's book-keeping as torch ops.
"""
from __future__ import annotations

from typing import Optional

import torch as th
import torch.distributed as dist

from . import _native as nat

HEADER = 3   # routing-block columns before the observation: partner id, previous reward, previous done


class RoundRobinLink:
    """the engine-side carrier: this rank's receive area, every other rank's area mapped through HIP IPC (handles travel
    through the process group's store), and the ph_rr_link descriptor the native iterations take"""

    _generation = 0

    def __init__(self, ctx, n_partners: int, n_envs: int, obs_dim: int, device, timeout_s: Optional[float] = None):
        import ctypes as C
        import os
        if timeout_s is None:      # one bound for every in-kernel wait of the package, per device share (dist.ActionExchange.attach_p2p)
            from .dist import ranks_sharing_device
            timeout_s = float(os.environ.get("PH_P2P_TIMEOUT_S", "10")) * ranks_sharing_device(device)
        self.ctx, self.device = ctx, device
        rank, world = dist.get_rank(), dist.get_world_size()
        store = dist.distributed_c10d._get_default_store()
        # the generation advances BEFORE anything can fail, and a rank that fails says so under its key: the n-th attempt of
        # every rank meets under the same keys whatever happened to earlier attempts, and nobody waits out the store's timeout
        # for a handle that will never come
        RoundRobinLink._generation += 1
        gen = RoundRobinLink._generation
        key = f"pantheonrl_amd/rr/{gen}/{rank}"
        self._mapped, self._base = [], None
        try:
            size = C.c_size_t(0)
            nat.check(ctx.lib.ph_rr_area_bytes(n_partners, n_envs, HEADER + obs_dim, C.byref(size)))
            base, handle = C.c_void_p(), (C.c_ubyte * 64)()
            nat.check(ctx.lib.ph_p2p_alloc(ctx.handle, size.value + 64, C.byref(base), handle))
            self._base = base
        except Exception:
            store.set(key, b"failed")
            raise
        store.set(key, bytes(handle))
        link = nat.PhRRLink()
        link.n_partners, link.rank, link.n, link.block_ld = n_partners, rank, n_envs, HEADER + obs_dim
        try:
            for r in range(world):
                if r == rank:
                    link.area[r] = base.value
                    continue
                raw = store.get(f"pantheonrl_amd/rr/{gen}/{r}")
                if len(raw) != 64:
                    raise RuntimeError(f"rank {r} could not allocate its receive area")
                peer = (C.c_ubyte * 64).from_buffer_copy(raw)
                mapped = C.c_void_p()
                nat.check(ctx.lib.ph_p2p_open(ctx.handle, peer, C.byref(mapped)))
                link.area[r] = mapped.value
                self._mapped.append(mapped)
        except Exception:
            self.close()
            raise
        link.error = base.value + size.value          # the spare 64 bytes behind the area
        link.timeout_cycles = int(timeout_s * 1e8)
        self.link, self._size = link, size.value

        class _View:
            def __init__(self, ptr):
                self.__cuda_array_interface__ = {"shape": (8,), "typestr": "<i4", "data": (ptr, False), "version": 2}
        self._error = th.as_tensor(_View(link.error), device=device)

    def timeouts(self) -> int:
        return int(self._error[0].item())

    def timeout_record(self) -> Optional[dict]:
        """what the first timed-out wait was waiting for (csrc/ph_launch.h: p2p_note_timeout), or None"""
        if self.timeouts() == 0:
            return None
        w = [int(v) & 0xFFFFFFFF for v in self._error.cpu().tolist()]
        return {"rank": int(self.link.rank), "timeouts": w[0], "want_stamp": w[4], "seen_stamp": w[6]}

    def check(self, where: str) -> None:
        """a wait that timed out means the iteration went on with stale slots (the kernels never hang the device): the buffers it
        filled must not be trained on"""
        n = self.timeouts()
        if n:
            raise nat.NativeError(f"round-robin carrier: {n} in-kernel wait(s) timed out {where} ({self.timeout_record()}); "
                                  "a peer is lost or more than PH_P2P_TIMEOUT_S late -- the rollout buffers hold stale rows")

    def close(self) -> None:
        """unmap the peers' areas and free this rank's (idempotent)"""
        for m in self._mapped:
            try:
                self.ctx.lib.ph_p2p_close(self.ctx.handle, m)
            except Exception:  # noqa: BLE001
                pass
        self._mapped = []
        if self._base is not None:
            try:
                self.ctx.lib.ph_p2p_free(self.ctx.handle, self._base)
            except Exception:  # noqa: BLE001
                pass
            self._base = None


def _timing() -> bool:
    import os
    return os.environ.get("PH_RR_TIMING", "0") == "1"


def _now() -> float:
    import time
    return time.perf_counter()


def _native_wanted(device) -> bool:
    import os
    return th.device(device).type == "cuda" and os.environ.get("PH_RR_NATIVE", "1") != "0"


class _Collectives:
    """broadcast / all-gather of device tensors on the default group; gloo (CPU tests) stages through the host"""

    def __init__(self, device):
        self.device = device
        self.staged = dist.get_backend() == "gloo" and th.device(device).type == "cuda"

    def broadcast(self, t: th.Tensor, src: int = 0) -> None:
        if self.staged:
            h = t.cpu()
            dist.broadcast(h, src=src)
            if dist.get_rank() != src:
                t.copy_(h)
        else:
            dist.broadcast(t, src=src)

    def all_gather(self, out: th.Tensor, mine: th.Tensor) -> None:
        if self.staged:
            h = th.empty(out.numel(), dtype=out.dtype)
            dist.all_gather_into_tensor(h, mine.reshape(-1).cpu())
            out.view(-1).copy_(h)
        else:
            dist.all_gather_into_tensor(out.view(-1), mine.reshape(-1))


class RoundRobinEgoRank:
    """rank 0: the ego learner (a `vec.VecOnPolicyAgent`) and the E synthetic environments"""

    def __init__(self, ego, data_ego, obs_alt: th.Tensor, n_partners: int, bonus: float = 0.01, env_step=None):
        """`env_step(joint, partnerid, base, done, reward_out, alt_out, next_block, K, bonus)` replaces the device transition
        (`ph_roundrobin_env_step`) -- only the CPU protocol tests pass one"""
        self.ego, self.data, self.K, self.bonus = ego, data_ego, int(n_partners), float(bonus)
        self._env_step = env_step
        pol = ego.model.policy
        dev = pol.device
        T, E, D = data_ego.T, data_ego.E, int(obs_alt.shape[-1])
        self.T, self.E, self.D = T, E, D
        self.col = _Collectives(dev)
        # every step's routing block, the partner-seat observations filled in once (they are static synthetic inputs); the
        # header columns of block t+1 are written by step t's transition
        self.blocks = th.zeros((T, E, HEADER + D), dtype=th.float32, device=dev)
        self.blocks[:, :, HEADER:] = obs_alt
        self.partnerid = th.full((E,), 1 % self.K, dtype=th.int32, device=dev)   # the first reset already advances (D-9)
        self.blocks[0, :, 0] = float(1 % self.K)
        self.joint = th.zeros((self.K + 1, E), dtype=th.int32, device=dev)
        self.rewards = th.zeros((T, E), dtype=th.float32, device=dev)           # per step: ego.update keeps a reference
        self.alt_actions = th.zeros((T, E), dtype=th.int32, device=dev)
        self.partner_trace = th.zeros((T, E), dtype=th.int32, device=dev)        # who was partnered where (tests / info)
        self._lib, self._h = (pol.ctx.lib, pol.ctx.handle) if env_step is None else (None, None)
        self.link: Optional[RoundRobinLink] = None
        self.iteration = 0
        self._first_start = th.ones(E, dtype=th.float32, device=dev)

    def attach(self, link: RoundRobinLink) -> None:
        import ctypes as C
        self.link = link
        ego, d = self.ego, self.data
        pol, rb = ego.model.policy, ego.model.rollout_buffer
        c = nat.PhRREgo()
        c.spec, c.params = C.pointer(pol.spec), pol.params.data_ptr()
        c.obs_seq, c.base_reward_seq, c.done_seq = d.obs.data_ptr(), d.rewards.data_ptr(), d.dones.data_ptr()
        c.blocks, c.partnerid, c.rewards = self.blocks.data_ptr(), self.partnerid.data_ptr(), self.rewards.data_ptr()
        c.alt_actions, c.partner_trace = self.alt_actions.data_ptr(), self.partner_trace.data_ptr()
        c.episode_start0, c.seed = self._first_start.data_ptr(), pol._seed
        c.values, c.log_probs, c.rb, c.bonus = ego.values.data_ptr(), ego.log_probs.data_ptr(), C.pointer(rb.c_struct()), self.bonus
        self._c = c

    def _run_iteration_native(self) -> None:
        """the T steps as ONE native call: forward, routing block into the partners' areas, wait for their actions, transition"""
        import ctypes as C
        ego, d, T = self.ego, self.data, self.T
        pol, rb = ego.model.policy, ego.model.rollout_buffer
        ego.bind_stream()
        ego.flush_rewards()
        self._first_start.copy_(ego._last_episode_starts)
        self._c.counter0 = pol._counter + 1
        timing = _timing()
        if timing:
            th.cuda.synchronize()
            t0 = _now()
        nat.check(self._lib.ph_roundrobin_ego_iteration(self._h, C.byref(self.link.link), C.byref(self._c), T, self.iteration))
        if timing:
            th.cuda.synchronize()
            t1 = _now()
        pol._counter += T
        rb.pos, rb.full = T, True
        ego.n_steps += T
        ego.num_timesteps += T * self.E
        ego._pending = self.rewards[T - 1]            # the last step's reward: flushed by compute_returns
        ego._last_episode_starts = d.dones[T - 1]
        self.iteration += 1
        self.link.check(f"in the ego's iteration {self.iteration - 1}")   # one host read: never train on stale joint actions
        ego.learn_from_buffer()
        if timing:
            th.cuda.synchronize()
            print(f"[rr timing] ego iteration {self.iteration}: rollout {1e3 * (t1 - t0):.2f} ms, update {1e3 * (_now() - t1):.2f} ms",
                  flush=True)

    def run_iteration(self) -> None:
        if self.link is not None:
            return self._run_iteration_native()
        ego, d, T = self.ego, self.data, self.T
        ego.bind_stream()
        for t in range(T):
            actions = ego.get_action(d.obs[t])                   # forward + row write (+ the previous step's reward)
            self.partner_trace[t].copy_(self.partnerid)
            self.col.broadcast(self.blocks[t], src=0)
            self.col.all_gather(self.joint, actions.reshape(-1))
            if self._env_step is not None:
                self._env_step(self.joint, self.partnerid, d.rewards[t], d.dones[t], self.rewards[t], self.alt_actions[t],
                               self.blocks[(t + 1) % T], self.K, self.bonus)
                ego.update(self.rewards[t], d.dones[t])
                continue
            ego.bind_stream()
            nat.check(self._lib.ph_roundrobin_env_step(
                self._h, self.joint.data_ptr(), self.partnerid.data_ptr(), d.rewards[t].data_ptr(), d.dones[t].data_ptr(),
                self.rewards[t].data_ptr(), self.alt_actions[t].data_ptr(), self.blocks[(t + 1) % T].data_ptr(),
                HEADER + self.D, self.K, self.bonus, self.E))
            ego.update(self.rewards[t], d.dones[t])
        ego.learn_from_buffer()


class RoundRobinPartnerRank:
    """rank 1 + k: partner k (an `envs.vec.RaggedVecOnPolicyAgent`: per-environment write rows)"""

    def __init__(self, agent, partner_index: int, n_partners: int, n_envs: int, obs_dim: int, n_steps: int):
        self.agent, self.k, self.K = agent, int(partner_index), int(n_partners)
        dev = agent.model.policy.device
        self.T, self.E, self.D = int(n_steps), int(n_envs), int(obs_dim)
        self.col = _Collectives(dev)
        self.block = th.zeros((self.E, HEADER + self.D), dtype=th.float32, device=dev)
        self.joint = th.zeros((self.K + 1, self.E), dtype=th.int32, device=dev)
        self.prev_mask: Optional[th.Tensor] = None
        self.updates = 0
        self.link: Optional[RoundRobinLink] = None
        self.iteration = 0

    def attach(self, link: RoundRobinLink) -> None:
        self.link = link
        dev = self.agent.model.policy.device
        self._obs = th.zeros((self.E, self.D), dtype=th.float32, device=dev)
        self._es = th.zeros(self.E, dtype=th.float32, device=dev)
        self._can = th.zeros(self.E, dtype=th.uint8, device=dev)
        self._prev = th.zeros(self.E, dtype=th.uint8, device=dev)

    def _run_iteration_native(self) -> None:
        import ctypes as C
        agent = self.agent
        pol, rb = agent.model.policy, agent.model.rollout_buffer
        pol._bind()
        timing = _timing()
        if timing:
            th.cuda.synchronize()
            t0 = _now()
        if self.iteration > 0:                    # the previous iteration's waits, before its rows are trained on or extended
            self.link.check(f"in partner {self.k}'s iteration {self.iteration - 1}")
        if agent.full():                          # once per iteration (one host read): train on the full columns
            agent.learn_from_buffer()
            self.updates += 1
        if timing:
            th.cuda.synchronize()
            t1 = _now()
        c = nat.PhRRPartner()
        c.spec, c.params = C.pointer(pol.spec), pol.params.data_ptr()
        c.obs_scratch, c.es_scratch, c.can_scratch = self._obs.data_ptr(), self._es.data_ptr(), self._can.data_ptr()
        c.pos, c.boundary, c.term, c.open = agent.pos.data_ptr(), agent.boundary.data_ptr(), agent.term.data_ptr(), agent.open.data_ptr()
        c.prev_mask, c.seed, c.counter0 = self._prev.data_ptr(), pol._seed, pol._counter + 1
        c.actions, c.values, c.log_probs = agent.actions.data_ptr(), agent.values.data_ptr(), agent.log_probs.data_ptr()
        c.rb = C.pointer(rb.c_struct())
        pol._bind()
        nat.check(agent._lib.ph_roundrobin_partner_iteration(agent._h, C.byref(self.link.link), C.byref(c), self.T, self.iteration))
        pol._counter += self.T
        agent.num_timesteps += self.T * self.E
        self.iteration += 1
        if timing:
            th.cuda.synchronize()
            print(f"[rr timing] partner {self.k} iteration {self.iteration}: update {1e3 * (t1 - t0):.2f} ms, rollout "
                  f"{1e3 * (_now() - t1):.2f} ms", flush=True)

    def run_iteration(self) -> None:
        if self.link is not None:
            return self._run_iteration_native()
        agent = self.agent
        for _ in range(self.T):
            self.col.broadcast(self.block, src=0)
            blk = self.block
            if self.prev_mask is not None:
                # Agent.update of the previous step, for the environments this partner acted in (agents.py:186-203)
                m = self.prev_mask
                agent.update(blk[:, 1].contiguous(), (blk[:, 2] * m.to(th.float32)).contiguous(), m)
            if agent.full():                      # train before acting once every column is full (agents.py:126)
                agent.learn_from_buffer()
                self.updates += 1
            mask = (blk[:, 0] == float(self.k)).to(th.uint8)
            actions = agent.get_action(blk[:, HEADER:].contiguous(), mask)
            self.prev_mask = mask
            self.col.all_gather(self.joint, actions.reshape(-1).contiguous())


def make_rank(model, n_partners: int, steps_per_iteration: int, data_ego=None, obs_alt=None, bonus: float = 0.01,
              native: Optional[bool] = None):
    """this rank's half of the layout: rank 0 -> RoundRobinEgoRank (needs the synthetic inputs), rank 1 + k -> partner k.
    `steps_per_iteration` = the ego's n_steps (every rank walks that many environment steps per iteration; a partner's
    own buffer length may differ).  native: None = the engine-side carrier wherever the ranks' receive areas can be mapped
    into each other (every rank takes the same decision), False = torch.distributed collectives per step."""
    from .envs.vec import RaggedVecOnPolicyAgent
    from .vec import VecOnPolicyAgent
    rank, world = dist.get_rank(), dist.get_world_size()
    if world != n_partners + 1:
        raise ValueError(f"round-robin layout: {n_partners} partners need {n_partners + 1} ranks, the group has {world}")
    rb = model.rollout_buffer
    if rank == 0:
        side = RoundRobinEgoRank(VecOnPolicyAgent(model), data_ego, obs_alt, n_partners, bonus)
    else:
        agent = RaggedVecOnPolicyAgent(model)
        # an environment is with this partner one episode in K: train on the full columns once about half of the columns that
        # can be active at a time are full, instead of waiting K - 1 episodes for the rest
        agent.min_full = max(1, rb.n_envs // (2 * n_partners))
        side = RoundRobinPartnerRank(agent, rank - 1, n_partners, rb.n_envs, model.policy.layout.D, steps_per_iteration)
    want = _native_wanted(model.policy.device) if native is None else bool(native)
    link = None
    if want:
        try:
            link = RoundRobinLink(model.policy.ctx, n_partners, rb.n_envs, model.policy.layout.D, model.policy.device)
        except Exception as exc:  # noqa: BLE001 -- any failure: every rank falls back together (verdict below)
            import sys
            print(f"[pantheonrl_amd.roundrobin] engine-side carrier unavailable ({exc}); using torch.distributed per step",
                  file=sys.stderr)
            link = None
    vdev = model.policy.device if dist.get_backend() == "nccl" else "cpu"
    verdict = th.tensor([1.0 if link is not None else 0.0], device=vdev)
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
    if verdict.item() > 0.5:
        side.attach(link)
    side.native = side.link is not None
    return side
