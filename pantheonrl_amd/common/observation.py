"""`Observation` record and the two extractors (reference pantheonrl/common/observation.py:7-43).

`action_mask` is the integer 0/1 legality vector over the discrete actions (None = everything legal); it is carried
untouched from the environment to the policy (reference pettingzoo.py:89-91,121-127 -> modular/policies.py:330-333).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


class Observation:
    """obs: what the agent sees; state: full state (defaults to obs); action_mask: legality of each action."""

    __slots__ = ("obs", "state", "action_mask")

    def __init__(self, obs: np.ndarray, state: Optional[np.ndarray] = None,
                 action_mask: Optional[np.ndarray] = None):
        self.obs = obs
        self.state = obs if state is None else state
        self.action_mask = action_mask

    def __repr__(self) -> str:
        return f"Observation(obs={self.obs!r}, state={self.state!r}, action_mask={self.action_mask!r})"

    def __eq__(self, other) -> bool:
        if not isinstance(other, Observation):
            return NotImplemented
        same = lambda a, b: (a is None and b is None) or (  # noqa: E731
            a is not None and b is not None and np.array_equal(a, b))
        return same(self.obs, other.obs) and same(self.state, other.state) and same(self.action_mask,
                                                                                     other.action_mask)


def extract_obs(observation: Observation) -> np.ndarray:
    """what an SB3-style ego consumes: the bare (partial) observation."""
    return observation.obs


def extract_partial_obs(observation: Observation) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """(obs, action_mask) pair for mask-aware learners."""
    return observation.obs, observation.action_mask
