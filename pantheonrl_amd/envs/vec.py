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
