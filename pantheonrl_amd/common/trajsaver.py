"""Transition datasets and their `.npy` wire format (behaviour of reference pantheonrl/common/trajsaver.py:53-232).

On disk a recording is ONE 2-D float array, one row per recorded move:

    single agent        [ obs | acts ]
    turn-based game     [ obs | acts | flag ]                      flag: 0 ego moved, 1 partner moved, +2 if that move ended
                                                                   the game (wrappers.py:12-16)
    simultaneous game   [ egoobs | egoacts | altobs | altacts | flag ]   flag: 0 not done, 1 done (wrappers.py:18-20)

`obs` is what the mover saw, `acts` what it played.  Column widths come from the spaces (`get_space_size`), so a file
can only be read back with the spaces it was written for -- exactly like the reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import numpy as np

from .util import get_space_size


def _rows(a: np.ndarray, n: int) -> np.ndarray:
    return np.reshape(np.asarray(a), (n, -1))


@dataclass(frozen=True)
class TransitionsMinimal:
    """(obs, acts) pairs of one agent; indexable like a torch Dataset (returns {"obs", "acts"} rows)"""
    obs: np.ndarray
    acts: np.ndarray

    def __post_init__(self):
        if len(self.obs) != len(self.acts):
            raise ValueError(f"obs and acts must have the same number of rows ({len(self.obs)} != {len(self.acts)})")

    def __len__(self) -> int:
        return len(self.obs)

    def __getitem__(self, idx) -> Dict[str, np.ndarray]:
        if isinstance(idx, slice):
            return TransitionsMinimal(self.obs[idx], self.acts[idx])
        return {"obs": self.obs[idx], "acts": self.acts[idx]}

    def write_transition(self, file) -> None:
        n = len(self)
        np.save(file, np.concatenate((_rows(self.obs, n), _rows(self.acts, n)), axis=1))

    @classmethod
    def read_transition(cls, file, obs_space, act_space) -> "TransitionsMinimal":
        table = np.load(file)
        k = get_space_size(obs_space)
        return cls(table[:, :k], table[:, k:])


class MultiTransitions:
    """anything that can hand out the ego's and the partner's transitions"""

    def get_ego_transitions(self) -> TransitionsMinimal:
        raise NotImplementedError

    def get_alt_transitions(self) -> TransitionsMinimal:
        raise NotImplementedError


@dataclass(frozen=True)
class TurnBasedTransitions(MultiTransitions):
    obs: np.ndarray
    acts: np.ndarray
    flags: np.ndarray

    def _of(self, parity: int) -> TransitionsMinimal:
        sel = (np.asarray(self.flags) % 2 == parity)
        return TransitionsMinimal(self.obs[sel], self.acts[sel])

    def get_ego_transitions(self) -> TransitionsMinimal:
        return self._of(0)

    def get_alt_transitions(self) -> TransitionsMinimal:
        return self._of(1)

    def write_transition(self, file) -> None:
        n = len(self.flags)
        np.save(file, np.concatenate((_rows(self.obs, n), _rows(self.acts, n), _rows(self.flags, n)), axis=1))

    @classmethod
    def read_transition(cls, file, obs_space, act_space) -> "TurnBasedTransitions":
        table = np.load(file)
        k = get_space_size(obs_space)
        return cls(table[:, :k], table[:, k:-1], table[:, -1])


@dataclass(frozen=True)
class SimultaneousTransitions(MultiTransitions):
    egoobs: np.ndarray
    egoacts: np.ndarray
    altobs: np.ndarray
    altacts: np.ndarray
    flags: np.ndarray

    def get_ego_transitions(self) -> TransitionsMinimal:
        return TransitionsMinimal(self.egoobs, self.egoacts)

    def get_alt_transitions(self) -> TransitionsMinimal:
        return TransitionsMinimal(self.altobs, self.altacts)

    def write_transition(self, file) -> None:
        n = len(self.flags)
        cols = (self.egoobs, self.egoacts, self.altobs, self.altacts, self.flags)
        np.save(file, np.concatenate([_rows(c, n) for c in cols], axis=1))

    @classmethod
    def read_transition(cls, file, obs_space, act_space) -> "SimultaneousTransitions":
        table = np.load(file)
        k, m = get_space_size(obs_space), get_space_size(act_space)
        cuts = np.cumsum([k, m, k])
        return cls(table[:, :cuts[0]], table[:, cuts[0]:cuts[1]], table[:, cuts[1]:cuts[2]], table[:, cuts[2]:-1],
                   table[:, -1])
