"""Liar's Dice: a TurnBasedEnv (reference pantheonrl/envs/liargym/liar.py:7-102).

Each player rolls M=6 dice with N=6 sides.  A move is (side, count-1); a legal raise strictly increases the count
component; side == N means "call the bluff".  Observation = own hand histogram (N entries in 0..M) followed by the
last MAX_MOVES=12 moves, newest first, padded with the null move [N, 0].  Integer rules throughout.
"""
from __future__ import annotations

import numpy as np

from ..common.agents import Agent
from ..common.multiagentenv import TurnBasedEnv
from ..spaces import MultiDiscrete

N_SIDES = 6
N_DICE = 6
MAX_MOVES = 2 * N_DICE
CALL = [N_SIDES, 2 * N_DICE - 1]   # the "bluff!" move
NULL_MOVE = [N_SIDES, 0]
EGO_WINS, EGO_LOSES = (1, -1), (-1, 1)


def roll_hand(rng=np.random):
    """histogram of N_DICE dice (liar.py:22-26); consumes N_DICE randint draws like the reference."""
    faces = [rng.randint(N_SIDES) for _ in range(N_DICE)]
    return [faces.count(side) for side in range(N_SIDES)]


class LiarDefaultAgent(Agent):
    """bids its most common face, calls when the standing bid exceeds it (liar.py:29-42)."""

    def get_action(self, obs, record=True):
        vec = obs.obs.tolist()
        hand = vec[:N_SIDES]
        best = max(hand)
        if vec[N_SIDES] != N_SIDES and vec[N_SIDES + 1] > best:
            return np.array(CALL)
        return np.array([hand.index(best), best])

    def update(self, reward, done):
        return None


class LiarEnv(TurnBasedEnv):
    def __init__(self, probegostart=0.5):
        super().__init__(probegostart=probegostart)
        self.history = []
        self.observation_space = MultiDiscrete([N_DICE + 1] * N_SIDES + [N_SIDES + 1, 2 * N_DICE] * MAX_MOVES)
        self.action_space = MultiDiscrete([N_SIDES + 1, 2 * N_DICE])
        self.egohand, self.althand = [0] * N_SIDES, [0] * N_SIDES

    def getObs(self, isego):
        moves = self.history + NULL_MOVE * (MAX_MOVES - len(self.history) // 2)
        return np.array((self.egohand if isego else self.althand) + moves)

    def sanitize_action(self, action):
        """map an arbitrary (side, count) pair onto a legal move (liar.py:58-67)."""
        if self.history:
            if action[1] <= self.history[1] or action[0] == N_SIDES:
                return CALL
        elif action[0] == N_SIDES:
            return [0, 0]
        return np.asarray(action).tolist()

    def eval_bluff(self):
        """was the standing bid a bluff? (liar.py:69-75)"""
        if not self.history:
            return False
        side, bid = self.history[0], self.history[1]
        return bid > self.egohand[side] + self.althand[side] - 1

    def player_step(self, action, isego):
        move = self.sanitize_action(action)
        if move == CALL:
            caller_wins = self.eval_bluff()
            ego_won = (caller_wins == isego)
            return self.getObs(not isego), EGO_WINS if ego_won else EGO_LOSES, True, {}
        self.history = move + self.history
        return self.getObs(not isego), (0, 0), False, {}

    def ego_step(self, action):
        return self.player_step(action, True)

    def alt_step(self, action):
        return self.player_step(action, False)

    def multi_reset(self, egofirst):
        self.history = []
        self.egohand = roll_hand()
        self.althand = roll_hand()
        return self.getObs(egofirst)
