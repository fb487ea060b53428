"""Ego-centric multi-agent environments (reference pantheonrl/common/multiagentenv.py:12-442).

An N-player game is presented to the *ego* learner as an ordinary single-agent env: `step(ego_action)` plays every
partner that is due to move (asking its `Agent` for an action, crediting it rewards) until the ego is due again.
This is the driver of the partners' OnPolicyAgent callbacks; partner selection (round-robin / random) stays in
Python exactly as in the reference.

Behaviour kept from the reference, with the SURVEY.md Appendix D item in brackets:
  * `info['_partnerid']` is injected on every step (:197);
  * the first action of an episode hands a partner the reward it accrued before it moved (:158-159) [D-2];
  * on `done`, `step` returns the ego's *previous* observation (:206-208) [D-8];
  * round-robin advances at every `reset`, including the first (:224) [D-9].
Deliberate deviation [D-7]: the reference builds the default partner lists as `[[]] * (n_players-1)`, which aliases
ONE list across all seats; here every seat gets its own list, so adding a partner to seat 2 does not add it to
seat 1.  For 2-player games (every BASELINE config that uses this class) the two are identical.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

from .agents import Agent
from .observation import Observation, extract_obs


class PlayerException(Exception):
    """Raise when players in the environment are incorrectly set (multiagentenv.py:12)."""


class DummyEnv:
    """Carrier of a partner's observation/action spaces so it can build its policy (multiagentenv.py:16-22)."""

    _is_dummy_space_env = True

    def __init__(self, observation_space, action_space):
        self.observation_space = observation_space
        self.action_space = action_space


class _Seats:
    """Book-keeping of the non-ego seats: who may sit there, who currently does, who already acted this episode."""

    def __init__(self, n_seats: int, ego_ind: int, candidates: Optional[List[List[Agent]]]):
        self.ego_ind = ego_ind
        self.candidates: List[List[Agent]] = candidates if candidates else [[] for _ in range(n_seats)]
        self.current: List[int] = [0] * n_seats
        self.acted: List[bool] = [False] * n_seats

    def index_of(self, player_num: int) -> int:
        if player_num == self.ego_ind:
            raise PlayerException("Ego agent is not set by the environment")
        return player_num - (player_num > self.ego_ind)

    def player_of(self, seat: int) -> int:
        return seat + (seat >= self.ego_ind)

    def agent(self, seat: int) -> Agent:
        return self.candidates[seat][self.current[seat]]

    def new_episode(self) -> None:
        self.acted = [False] * len(self.acted)


class MultiAgentEnv(ABC):
    """Base of all multi-agent games (multiagentenv.py:25-284).

    :param ego_ind: seat of the ego
    :param n_players: number of seats
    :param resample_policy: "default" | "robin" | "random"  (see set_resample_policy)
    :param partners: per non-ego seat, the list of agents that may occupy it
    :param ego_extractor: maps the ego's Observation to what the ego learner consumes
    """

    def __init__(self, ego_ind: int = 0, n_players: int = 2, resample_policy: str = "default",
                 partners: Optional[List[List[Agent]]] = None,
                 ego_extractor: Callable[[Observation], Any] = extract_obs):
        if partners is not None:
            if len(partners) != n_players - 1:
                raise PlayerException("The number of partners needs to equal the number of non-ego players")
            if not all(isinstance(seat, list) and len(seat) > 0 for seat in partners):
                raise PlayerException("Sublist for each partner must be nonempty list")
        self.ego_ind, self.n_players = ego_ind, n_players
        self._seats = _Seats(n_players - 1, ego_ind, partners)
        self._players: Tuple[int, ...] = ()
        self._obs: Tuple[Optional[Observation], ...] = ()
        self._old_ego_obs: Optional[Observation] = None
        self.total_rews = [0] * n_players
        self.ego_moved = False
        self.ego_extractor = ego_extractor
        self.set_resample_policy(resample_policy)

    # the reference exposes these three as plain attributes; keep them readable / assignable under the same names
    @property
    def partners(self) -> List[List[Agent]]:
        return self._seats.candidates

    @property
    def partnerids(self) -> List[int]:
        return self._seats.current

    @partnerids.setter
    def partnerids(self, ids: List[int]) -> None:
        self._seats.current = list(ids)

    @property
    def should_update(self) -> List[bool]:
        return self._seats.acted

    # -- partner management (multiagentenv.py:72-147) ---------------------------------------------------------------
    def getDummyEnv(self, player_num: int):
        """Environment-shaped object whose spaces are those seat `player_num` sees; default: the env itself."""
        return self

    def set_ego_extractor(self, ego_extractor: Callable[[Observation], Any]) -> None:
        self.ego_extractor = ego_extractor

    def _get_partner_num(self, player_num: int) -> int:
        return self._seats.index_of(player_num)

    def add_partner_agent(self, agent: Agent, player_num: int = 1) -> None:
        """Register `agent` as a candidate for seat `player_num`; one candidate is drawn per episode."""
        self.partners[self._get_partner_num(player_num)].append(agent)

    def set_partnerid(self, agent_id: int, player_num: int = 1) -> None:
        seat = self._get_partner_num(player_num)
        assert 0 <= agent_id < len(self.partners[seat])
        self._seats.current[seat] = agent_id

    def resample_random(self) -> None:
        self.partnerids = [np.random.randint(len(options)) for options in self.partners]

    def resample_round_robin(self) -> None:
        """advance the single partner seat to its next candidate (2-player games only)"""
        self.partnerids = [(self.partnerids[0] + 1) % len(self.partners[0])]

    def set_resample_policy(self, resample_policy: str) -> None:
        choice = resample_policy
        if choice == "default":
            choice = "robin" if self.n_players == 2 else "random"
        if choice == "robin" and self.n_players != 2:
            raise PlayerException("Cannot do round robin resampling for >2 players")
        table = {"robin": self.resample_round_robin, "random": self.resample_random}
        if choice not in table:
            raise PlayerException(f"Invalid resampling policy: {resample_policy}")
        self.resample_partner = table[choice]

    # -- driving the partners (multiagentenv.py:149-170) ----------------------------------------------------------------
    def _get_actions(self, players, obs, ego_act=None):
        """actions of everybody due to move: the ego's is given, partners are asked (and, on their first move of the
        episode, handed the reward they accrued before moving -- never a terminal one)"""
        seats = self._seats
        actions = []
        for player, ob in zip(players, obs):
            if player == self.ego_ind:
                actions.append(ego_act)
                continue
            seat = seats.index_of(player)
            agent = seats.agent(seat)
            actions.append(agent.get_action(ob))
            if not seats.acted[seat]:
                agent.update(self.total_rews[player], False)
                seats.acted[seat] = True
        return np.array(actions)

    def _update_players(self, rews, done) -> None:
        """credit this transition to every partner that has acted, then to the running totals"""
        seats = self._seats
        for seat, acted in enumerate(seats.acted):
            if acted:
                seats.agent(seat).update(rews[seats.player_of(seat)], done)
        self.total_rews = [total + r for total, r in zip(self.total_rews, rews)]

    def _advance(self, ego_act=None):
        """one n_step with everybody due to move"""
        acts = self._get_actions(self._players, self._obs, ego_act)
        self._players, self._obs, rews, done, info = self.n_step(acts)
        return rews, done, info

    def _ego_view(self):
        ego_obs = self._obs[self._players.index(self.ego_ind)]
        self._old_ego_obs = ego_obs
        return self.ego_extractor(ego_obs)

    # -- gym.Env surface (multiagentenv.py:172-243) --------------------------------------------------------------------
    def step(self, action: np.ndarray):
        """One ego timestep -> (ego observation, ego reward, done, info): the ego's move, then every partner move until
        the ego is due again or the game ends (in which case the previous ego observation is returned)."""
        ego_rew = 0.0
        while True:
            rews, done, info = self._advance(action)
            info["_partnerid"] = self.partnerids
            self._update_players(rews, done)
            # the first time the ego is credited in an episode it also collects what accrued before it moved
            ego_rew += rews[self.ego_ind] if self.ego_moved else self.total_rews[self.ego_ind]
            self.ego_moved = True
            if done:
                return self.ego_extractor(self._old_ego_obs), ego_rew, done, info
            if self.ego_ind in self._players:
                return self._ego_view(), ego_rew, done, info

    def reset(self):
        """New episode: resample partners, play partners forward until the ego is due, return its first obs."""
        self.resample_partner()
        self._players, self._obs = self.n_reset()
        self._seats.new_episode()
        self.total_rews = [0] * self.n_players
        self.ego_moved = False
        while self.ego_ind not in self._players:
            rews, done, _ = self._advance()
            if done:
                raise PlayerException("Game ended before ego moved")
            self._update_players(rews, done)
        assert self._obs[self._players.index(self.ego_ind)] is not None
        return self._ego_view()

    # -- what a concrete game implements (multiagentenv.py:245-284) -------------------------------------------------------
    @abstractmethod
    def n_step(self, actions: List[np.ndarray]) -> Tuple[Tuple[int, ...], Tuple[Optional[Observation], ...],
                                                         Tuple[float, ...], bool, Dict]:
        """apply the moving players' actions -> (next players, their observations, all rewards, done, info)."""

    @abstractmethod
    def n_reset(self) -> Tuple[Tuple[int, ...], Tuple[Optional[Observation], ...]]:
        """start a game -> (players that move first, their observations)."""


class TurnBasedEnv(MultiAgentEnv, ABC):
    """2-player alternating-move games (multiagentenv.py:287-380); ego is seat 0."""

    def __init__(self, probegostart: float = 0.5, partners: Optional[List[Agent]] = None):
        super().__init__(ego_ind=0, n_players=2, partners=[partners] if partners else None)
        self.probegostart = probegostart
        self.ego_next = True

    def n_step(self, actions):
        mover_is_ego = self.ego_next
        obs, rews, done, info = (self.ego_step if mover_is_ego else self.alt_step)(actions[0])
        self.ego_next = not mover_is_ego
        return (1 if mover_is_ego else 0,), (Observation(obs),), rews, done, info

    def n_reset(self):
        self.ego_next = bool(np.random.rand() < self.probegostart)
        first_obs = self.multi_reset(self.ego_next)
        return (0 if self.ego_next else 1,), (Observation(first_obs),)

    @abstractmethod
    def ego_step(self, action):
        """ego moves -> (partner's observation, (ego reward, partner reward), done, info)."""

    @abstractmethod
    def alt_step(self, action):
        """partner moves -> (ego's observation, (ego reward, partner reward), done, info)."""

    @abstractmethod
    def multi_reset(self, egofirst: bool):
        """new game -> observation of whoever moves first."""


class SimultaneousEnv(MultiAgentEnv, ABC):
    """2-player simultaneous-move games (multiagentenv.py:383-442); ego is seat 0."""

    def __init__(self, partners: Optional[List[Agent]] = None):
        super().__init__(ego_ind=0, n_players=2, partners=[partners] if partners else None)

    def n_step(self, actions):
        (obs0, obs1), rews, done, info = self.multi_step(actions[0], actions[1])
        return (0, 1), (Observation(obs0), Observation(obs1)), rews, done, info

    def n_reset(self):
        obs0, obs1 = self.multi_reset()
        return (0, 1), (Observation(obs0), Observation(obs1))

    @abstractmethod
    def multi_step(self, ego_action, alt_action):
        """both move -> ((ego obs, partner obs), (ego reward, partner reward), done, info)."""

    @abstractmethod
    def multi_reset(self):
        """new game -> (ego obs, partner obs)."""
