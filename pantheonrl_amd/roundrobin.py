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
parameter ever crosses ranks.  The collectives are torch.distributed's (backend nccl = RCCL over xGMI; gloo with host
staging when several test ranks share one GPU): a broadcast of E*(3+D)*4 B (266 KB at E = 1024, D = 62) and an all-gather
of (K+1)*E*4 B per step, both latency-bound.
"""
from __future__ import annotations

from typing import Optional

import torch as th
import torch.distributed as dist

from . import _native as nat

HEADER = 3   # routing-block columns before the observation: partner id, previous reward, previous done


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

    def run_iteration(self) -> None:
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

    def run_iteration(self) -> None:
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


def make_rank(model, n_partners: int, steps_per_iteration: int, data_ego=None, obs_alt=None, bonus: float = 0.01):
    """this rank's half of the layout: rank 0 -> RoundRobinEgoRank (needs the synthetic inputs), rank 1 + k -> partner k.
    `steps_per_iteration` = the ego's n_steps (every rank walks that many environment steps per iteration; a partner's
    own buffer length may differ)"""
    from .envs.vec import RaggedVecOnPolicyAgent
    from .vec import VecOnPolicyAgent
    rank, world = dist.get_rank(), dist.get_world_size()
    if world != n_partners + 1:
        raise ValueError(f"round-robin layout: {n_partners} partners need {n_partners + 1} ranks, the group has {world}")
    rb = model.rollout_buffer
    if rank == 0:
        return RoundRobinEgoRank(VecOnPolicyAgent(model), data_ego, obs_alt, n_partners, bonus)
    agent = RaggedVecOnPolicyAgent(model)
    # an environment is with this partner one episode in K: train on the full columns once about half of the columns that can
    # be active at a time are full, instead of waiting K - 1 episodes for the rest
    agent.min_full = max(1, rb.n_envs // (2 * n_partners))
    return RoundRobinPartnerRank(agent, rank - 1, n_partners, rb.n_envs, model.policy.layout.D, steps_per_iteration)
