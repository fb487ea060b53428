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
from .util import calculate_space, get_default_obs


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
