"""Minimal observation/action space descriptions.

`gym` is not importable in this environment, and the engine only needs shapes: these classes carry the same public
attributes the reference reads from gym spaces (`.shape`, `.n`, `.nvec`, `.low`, `.high`, `.dtype`; reference
pantheonrl/common/util.py:18-60) so environments written against gym translate one-to-one.  A real gym space can be
passed wherever a space is expected -- `to_native_space` duck-types on those attributes.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

from . import _native as nat


class Space:
    shape: Tuple[int, ...] = ()
    dtype = np.float32

    def sample(self, rng: np.random.Generator = None):  # pragma: no cover - convenience only
        raise NotImplementedError


class Box(Space):
    def __init__(self, low, high, shape: Sequence[int] = None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(int(s) for s in shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()
        self.dtype = np.dtype(dtype)

    def sample(self, rng=None):
        rng = rng or np.random.default_rng()
        return rng.standard_normal(self.shape).astype(self.dtype)

    def __repr__(self):
        return f"Box{self.shape}"


class Discrete(Space):
    def __init__(self, n: int):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def sample(self, rng=None):
        rng = rng or np.random.default_rng()
        return int(rng.integers(self.n))

    def __repr__(self):
        return f"Discrete({self.n})"


class MultiDiscrete(Space):
    def __init__(self, nvec: Sequence[int]):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = (len(self.nvec),)
        self.dtype = np.dtype(np.int64)

    def sample(self, rng=None):
        rng = rng or np.random.default_rng()
        return (rng.random(len(self.nvec)) * self.nvec).astype(np.int64)

    def __repr__(self):
        return f"MultiDiscrete({self.nvec.tolist()})"


class MultiBinary(Space):
    def __init__(self, n: int):
        self.n = int(n)
        self.shape = (self.n,)
        self.dtype = np.dtype(np.int8)

    def __repr__(self):
        return f"MultiBinary({self.n})"


class SpaceException(Exception):
    """Raise when an illegal Space is used (reference pantheonrl/common/util.py:14)."""


def _kind(space) -> str:
    name = type(space).__name__
    if name in ("Box", "Discrete", "MultiDiscrete", "MultiBinary"):
        return name
    raise SpaceException(f"unsupported space {space!r}")


def obs_stored_shape(space) -> Tuple[int, ...]:
    """SB3 get_obs_shape: Box -> shape; Discrete -> (1,); MultiDiscrete -> (len(nvec),); MultiBinary -> (n,)."""
    k = _kind(space)
    if k == "Box":
        return tuple(space.shape)
    if k == "Discrete":
        return (1,)
    if k == "MultiDiscrete":
        return (len(space.nvec),)
    return (space.n,)


def action_dim(space) -> int:
    k = _kind(space)
    if k == "Discrete":
        return 1
    if k == "MultiDiscrete":
        return len(space.nvec)
    if k == "Box" and len(space.shape) == 1:
        return int(space.shape[0])        # SB3 get_action_dim: Box -> int(np.prod(shape)); one-dimensional shapes only here
    raise SpaceException("the MI355X PPO path implements Discrete / MultiDiscrete and one-dimensional Box action spaces")


def to_native_space(space, role: str) -> nat.PhSpace:
    k = _kind(space)
    if k == "Box":
        if role == "act" and (len(space.shape) != 1 or int(space.shape[0]) > nat.PH_MAX_BOX_ACT):
            raise SpaceException(f"Box action spaces: one dimension of at most {nat.PH_MAX_BOX_ACT} components (DiagGaussian head)")
        return nat.make_space(nat.PH_SPACE_BOX, int(np.prod(space.shape)))
    if k == "Discrete":
        return nat.make_space(nat.PH_SPACE_DISCRETE, 1, [space.n])
    if k == "MultiDiscrete":
        return nat.make_space(nat.PH_SPACE_DISCRETE, len(space.nvec), [int(v) for v in space.nvec])
    if role == "obs":  # MultiBinary observation: SB3 preprocess_obs is .float(), i.e. Box-like
        return nat.make_space(nat.PH_SPACE_BOX, space.n)
    raise SpaceException("MultiBinary action spaces are not on the categorical PPO path")


def make_spec(observation_space, action_space) -> nat.PhSpec:
    spec = nat.PhSpec()
    spec.obs = to_native_space(observation_space, "obs")
    spec.act = to_native_space(action_space, "act")
    return spec
