"""Rock-paper-scissors as a one-step simultaneous game (behaviour of reference pantheonrl/envs/rpsgym/rps.py:8-48).

Both seats always observe the constant [0]; the ego's payoff is +1 / 0 / -1 for win / draw / loss and the game is
zero-sum; every episode is a single step.  The payoff is an integer rule -- `PAYOFF[ego][alt]`, identical to the
reference's (ego - alt + 3) % 3 with 2 mapped to -1 -- and is what `ph_rps_step` evaluates for n_envs tables at once.
"""
from __future__ import annotations

import numpy as np

from ..common.agents import Agent
from ..common.multiagentenv import SimultaneousEnv
from ..spaces import Discrete

ACTION_NAMES = ("ROCK", "PAPER", "SCISSORS")
N_ACTIONS = len(ACTION_NAMES)
# rows: ego action, columns: partner action
PAYOFF = np.array([[0, -1, 1],
                   [1, 0, -1],
                   [-1, 1, 0]], dtype=np.int64)


def rps_payoff(ego_action, alt_action):
    """ego payoff for scalars or equally shaped integer arrays"""
    return PAYOFF[np.asarray(ego_action), np.asarray(alt_action)]


class RPSWeightedAgent(Agent):
    """scripted partner: draws rock / paper / scissors in proportion r : p : s (uniform when all are zero)"""

    def __init__(self, r=1, p=1, s=1, np_random=np.random):
        weights = np.array([r, p, s], dtype=float)
        total = weights.sum()
        cuts = np.cumsum(weights / total) if total else np.array([1, 2, 3]) / 3.0
        self.c0, self.c1 = float(cuts[0]), float(cuts[1])
        self.np_random = np_random

    def get_action(self, obs, record=True):
        roll = self.np_random.rand()
        return int(roll >= self.c0) + int(roll >= self.c1)

    def update(self, reward, done):
        return None


class RPSEnv(SimultaneousEnv):
    observation_space = Discrete(1)
    action_space = Discrete(N_ACTIONS)

    def __init__(self):
        super().__init__()
        self.history = []
        self._blank = np.zeros(1, dtype=np.int64)

    def multi_reset(self):
        return self._blank, self._blank

    def multi_step(self, ego_action, alt_action):
        ego_gain = int(rps_payoff(ego_action, alt_action))
        return (self._blank, self._blank), (ego_gain, -ego_gain), True, {}
