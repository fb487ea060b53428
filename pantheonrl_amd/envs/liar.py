"""Liar's Dice for two players, turn based (behaviour of reference pantheonrl/envs/liargym/liar.py:7-102).

Each player rolls N_DICE six-sided dice and sees only the histogram of its own hand.  A move is a pair
(side, count - 1); a raise must strictly increase the count component; side == N_SIDES means "I call the previous
bid a bluff", which ends the game: the caller wins iff fewer dice than bid show that side across both hands.  An
observation is the mover's hand histogram followed by the last MAX_MOVES moves, newest first, padded with the null move
(N_SIDES, 0) -- 30 integers in total.

The three rules are pure functions of (hands, history) so the same statements serve the Python game below, the bit-exact
checks of the vectorised device kernel `ph_liar_step`, and anyone who wants a batched NumPy restatement.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from ..common.agents import Agent
from ..common.multiagentenv import TurnBasedEnv
from ..spaces import MultiDiscrete

N_SIDES = 6
N_DICE = 6
MAX_MOVES = 2 * N_DICE
CALL = [N_SIDES, 2 * N_DICE - 1]
NULL_MOVE = [N_SIDES, 0]
EGO_WINS, EGO_LOSES = (1, -1), (-1, 1)


# ---- rules -----------------------------------------------------------------------------------------------------------
def legalise(move: Sequence[int], history: Sequence[int]) -> List[int]:
    """map any (side, count-1) pair onto a legal move: a non-raising bid or a call against a standing bid is a call;
    calling with nothing on the table becomes the lowest bid"""
    side, count = int(move[0]), int(move[1])
    if len(history) == 0:
        return [0, 0] if side == N_SIDES else [side, count]
    if side == N_SIDES or count <= history[1]:
        return list(CALL)
    return [side, count]


def is_bluff(history: Sequence[int], egohand: Sequence[int], althand: Sequence[int]) -> bool:
    """True when the standing bid (newest entry of `history`) claims more dice than the two hands hold"""
    if len(history) == 0:
        return False
    side, bid = history[0], history[1]
    return bid > egohand[side] + althand[side] - 1


def encode_obs(hand: Sequence[int], history: Sequence[int]) -> np.ndarray:
    """hand histogram + history (newest first) padded to MAX_MOVES with the null move"""
    missing = MAX_MOVES - len(history) // 2
    return np.array(list(hand) + list(history) + NULL_MOVE * missing)


def roll_hand(rng=np.random) -> List[int]:
    """histogram of N_DICE dice, one randint draw per die"""
    hist = [0] * N_SIDES
    for _ in range(N_DICE):
        hist[rng.randint(N_SIDES)] += 1
    return hist


# ---- scripted partner -------------------------------------------------------------------------------------------------
class LiarDefaultAgent(Agent):
    """bids its most frequent face at its own count; calls as soon as the standing bid exceeds that count"""

    def get_action(self, obs, record=True):
        vec = obs.obs.tolist()
        hand, last_side, last_count = vec[:N_SIDES], vec[N_SIDES], vec[N_SIDES + 1]
        best = max(hand)
        if last_side != N_SIDES and last_count > best:
            return np.array(CALL)
        return np.array([hand.index(best), best])

    def update(self, reward, done):
        return None


# ---- the game -----------------------------------------------------------------------------------------------------------
class LiarEnv(TurnBasedEnv):
    observation_space = MultiDiscrete([N_DICE + 1] * N_SIDES + [N_SIDES + 1, 2 * N_DICE] * MAX_MOVES)
    action_space = MultiDiscrete([N_SIDES + 1, 2 * N_DICE])

    def __init__(self, probegostart=0.5):
        super().__init__(probegostart=probegostart)
        self.history: List[int] = []
        self.egohand: List[int] = [0] * N_SIDES
        self.althand: List[int] = [0] * N_SIDES

    # thin views on the rules, named as in the reference
    def getObs(self, isego: bool) -> np.ndarray:
        return encode_obs(self.egohand if isego else self.althand, self.history)

    def sanitize_action(self, action) -> List[int]:
        return legalise(action, self.history)

    def eval_bluff(self) -> bool:
        return is_bluff(self.history, self.egohand, self.althand)

    def player_step(self, action, isego: bool) -> Tuple[np.ndarray, Tuple[int, int], bool, dict]:
        move = self.sanitize_action(action)
        if move == CALL:
            caller_right = self.eval_bluff()
            payoff = EGO_WINS if caller_right == isego else EGO_LOSES
            return self.getObs(not isego), payoff, True, {}
        self.history = move + self.history
        return self.getObs(not isego), (0, 0), False, {}

    def ego_step(self, action):
        return self.player_step(action, True)

    def alt_step(self, action):
        return self.player_step(action, False)

    def multi_reset(self, egofirst: bool):
        self.history = []
        self.egohand, self.althand = roll_hand(), roll_hand()
        return self.getObs(egofirst)
