"""Frame-stack wrappers (reference pantheonrl/common/wrappers.py:23-27,37-71,233-349).

"Frame stacking" is the reference's stand-in for recurrence (README.md:120): the observation handed to an agent is the
concatenation of its last `numframes` observations, MOST RECENT FIRST, with a default observation filling the slots an
episode has not reached yet.  It widens obs_dim by `numframes` and adds no arithmetic to the PPO path; the vectorised,
device-resident form is `ph_framestack_push` (pantheonrl_amd.vec.VecFrameStack).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from .multiagentenv import MultiAgentEnv, SimultaneousEnv, TurnBasedEnv
from .trajsaver import MultiTransitions, SimultaneousTransitions, TurnBasedTransitions
from .util import calculate_space, get_default_obs

# move flags of the recorders (wrappers.py:12-20)
EGO_NOT_DONE, ALT_NOT_DONE, EGO_DONE, ALT_DONE = 0, 1, 2, 3
NOT_DONE, DONE = 0, 1


class HistoryQueue:
    """the last `size` elements, newest first; unfilled slots hold `defaultelem` (wrappers.py:37-71)"""

    def __init__(self, defaultelem, size: int):
        self.defaultelem, self.size = defaultelem, size
        self.reset()

    def add(self, toadd) -> np.ndarray:
        """push `toadd`, return the flat newest-first view"""
        self.frames.insert(0, toadd)
        self.frames.pop()
        return np.array([v for frame in self.frames for v in frame])

    def reset(self) -> None:
        self.frames: List = [self.defaultelem] * self.size


def frame_wrap(env: MultiAgentEnv, numframes: int):
    """the right frame-stack wrapper for the env's move structure (wrappers.py:23-27)"""
    return (TurnBasedFrameStack if isinstance(env, TurnBasedEnv) else SimultaneousFrameStack)(env, numframes)


class TurnBasedFrameStack(TurnBasedEnv):
    """frame-stacks both players' observation streams of a turn-based game (wrappers.py:233-302)"""

    def __init__(self, env, numframes: int, defaultobs: Optional[np.ndarray] = None, altenv=None,
                 defaultaltobs: Optional[np.ndarray] = None):
        super().__init__(probegostart=env.probegostart, partners=env.partners[0])
        self.env, self.numframes = env, numframes
        self.action_space = env.action_space
        self.observation_space = calculate_space(env.observation_space, numframes)
        ego_default = defaultobs if defaultobs is not None else get_default_obs(env)
        alt_default = defaultaltobs if defaultaltobs is not None else get_default_obs(altenv if altenv is not None else env)
        self.egohistory = HistoryQueue(ego_default, numframes)
        self.althistory = HistoryQueue(alt_default, numframes)

    def ego_step(self, action):
        altobs, rews, done, info = self.env.ego_step(action)
        return self.althistory.add(altobs), rews, done, info

    def alt_step(self, action):
        egoobs, rews, done, info = self.env.alt_step(action)
        return self.egohistory.add(egoobs), rews, done, info

    def multi_reset(self, egofirst: bool):
        first = self.env.multi_reset(egofirst)
        self.egohistory.reset()
        self.althistory.reset()
        return (self.egohistory if egofirst else self.althistory).add(first)


class SimultaneousFrameStack(SimultaneousEnv):
    """frame-stacks both players' observation streams of a simultaneous game (wrappers.py:305-349)"""

    def __init__(self, env, numframes: int, defaultobs: Optional[np.ndarray] = None):
        super().__init__(partners=env.partners[0])
        self.env, self.numframes = env, numframes
        self.action_space = env.action_space
        self.observation_space = calculate_space(env.observation_space, numframes)
        self.defaultobs = get_default_obs(env) if defaultobs is None else list(defaultobs)
        self.egohistory = HistoryQueue(self.defaultobs, numframes)
        self.althistory = HistoryQueue(self.defaultobs, numframes)

    def multi_step(self, ego_action, alt_action):
        (o0, o1), rews, done, info = self.env.multi_step(ego_action, alt_action)
        return (self.egohistory.add(o0), self.althistory.add(o1)), rews, done, info

    def multi_reset(self):
        o0, o1 = self.env.multi_reset()
        self.egohistory.reset()
        self.althistory.reset()
        return self.egohistory.add(o0), self.althistory.add(o1)


def recorder_wrap(env: MultiAgentEnv):
    """the right trajectory recorder for the env's move structure (wrappers.py:30-34)"""
    return (TurnBasedRecorder if isinstance(env, TurnBasedEnv) else SimultaneousRecorder)(env)


class MultiRecorder:
    """a wrapper that can return everything it saw as MultiTransitions"""

    def get_transitions(self) -> MultiTransitions:
        raise NotImplementedError


class TurnBasedRecorder(TurnBasedEnv, MultiRecorder):
    """records (observation seen by the mover, its action, who moved / whether it ended the game) for every move of a
    turn-based game (wrappers.py:82-160).  The observation of a move is the one returned by the PREVIOUS call (the reset
    or the other player's step), so observations are logged when they are produced and a trailing one that nobody acted on
    yet is dropped at export."""

    def __init__(self, env):
        super().__init__(probegostart=env.probegostart, partners=env.partners[0])
        self.env = env
        self.action_space, self.observation_space = env.action_space, env.observation_space
        self.allobs: List[np.ndarray] = []
        self.allacts: List[np.ndarray] = []
        self.flags: List[int] = []
        self.incomplete = False   # True while the newest logged observation has no action yet

    def _log_move(self, step_fn, action, not_done_flag: int):
        nextobs, rews, done, info = step_fn(action)
        self.allacts.append(action)
        if done:
            self.flags.append(not_done_flag + 2)
            self.incomplete = False
        else:
            self.allobs.append(nextobs)
            self.flags.append(not_done_flag)
        return nextobs, rews, done, info

    def ego_step(self, action):
        return self._log_move(self.env.ego_step, action, EGO_NOT_DONE)

    def alt_step(self, action):
        return self._log_move(self.env.alt_step, action, ALT_NOT_DONE)

    def multi_reset(self, egofirst: bool):
        first = self.env.multi_reset(egofirst)
        if self.incomplete:
            self.allobs[-1] = first      # the previous episode's dangling observation is replaced
        else:
            self.allobs.append(first)
        self.incomplete = True
        return first

    def get_transitions(self) -> TurnBasedTransitions:
        obs = np.array(self.allobs)
        if self.incomplete:
            obs = obs[:-1]
        return TurnBasedTransitions(obs, np.array(self.allacts), np.array(self.flags))


class SimultaneousRecorder(SimultaneousEnv, MultiRecorder):
    """records both players' (observation, action) pairs and a done flag for every joint move (wrappers.py:163-230)"""

    def __init__(self, env):
        super().__init__(partners=env.partners[0])
        self.env = env
        self.action_space, self.observation_space = env.action_space, env.observation_space
        self.allegoobs, self.allegoacts, self.allaltobs, self.allaltacts, self.allflags = [], [], [], [], []
        self.incomplete = False

    def multi_step(self, ego_action, alt_action):
        obs, rews, done, info = self.env.multi_step(ego_action, alt_action)
        self.allegoacts.append(ego_action)
        self.allaltacts.append(alt_action)
        if done:
            self.allflags.append(DONE)
            self.incomplete = False
        else:
            self.allegoobs.append(obs[0])
            self.allaltobs.append(obs[1])
            self.allflags.append(NOT_DONE)
        return obs, rews, done, info

    def multi_reset(self):
        obs = self.env.multi_reset()
        self.allegoobs.append(obs[0])
        self.allaltobs.append(obs[1])
        self.incomplete = True
        return obs

    def get_transitions(self) -> SimultaneousTransitions:
        egoobs, altobs = np.array(self.allegoobs), np.array(self.allaltobs)
        if self.incomplete:
            egoobs, altobs = egoobs[:-1], altobs[:-1]
        return SimultaneousTransitions(egoobs, np.array(self.allegoacts), altobs, np.array(self.allaltacts),
                                       np.array(self.allflags))
