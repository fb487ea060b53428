"""Rock-paper-scissors: a one-step SimultaneousEnv (reference pantheonrl/envs/rpsgym/rps.py:8-48).

Payoff for the ego: (ego - alt) mod 3 mapped {0: draw 0, 1: win +1, 2: loss -1}; zero-sum; every episode lasts one
step; both players always observe the constant [0].
"""
from __future__ import annotations

import numpy as np

from ..common.agents import Agent
from ..common.multiagentenv import SimultaneousEnv
from ..spaces import Discrete

ACTION_NAMES = ("ROCK", "PAPER", "SCISSORS")
N_ACTIONS = 3


def rps_payoff(ego_action, alt_action):
    """integer payoff of the ego; works on scalars and arrays (bit-exact integer rule, rps.py:42-43)."""
    outcome = (np.asarray(ego_action) - np.asarray(alt_action) + N_ACTIONS) % N_ACTIONS
    return np.where(outcome == 2, -1, outcome)


class RPSWeightedAgent(Agent):
    """plays rock/paper/scissors with fixed weights r:p:s (rps.py:14-30)."""

    def __init__(self, r=1, p=1, s=1, np_random=np.random):
        total = r + p + s
        self.c0, self.c1 = (1. / 3, 2. / 3) if total == 0 else (r / total, (r + p) / total)
        self.np_random = np_random

    def get_action(self, obs, record=True):
        roll = self.np_random.rand()
        if roll < self.c0:
            return 0
        return 1 if roll < self.c1 else 2

    def update(self, reward, done):
        return None


class RPSEnv(SimultaneousEnv):
    def __init__(self):
        super().__init__()
        self.history = []
        self.observation_space = Discrete(1)
        self.action_space = Discrete(N_ACTIONS)
        self._null = np.array([0])

    def multi_step(self, ego_action, alt_action):
        outcome = int(rps_payoff(ego_action, alt_action))
        return (self._null, self._null), (outcome, -outcome), True, {}

    def multi_reset(self):
        return self._null, self._null
