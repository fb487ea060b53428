"""Policy call shim and space helpers (reference pantheonrl/common/util.py:14-111)."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch as th

from ..spaces import Box, Discrete, MultiBinary, MultiDiscrete, SpaceException  # noqa: F401


def _kind(space) -> str:
    return type(space).__name__


def get_space_size(space) -> int:
    """number of stored scalars of one sample (util.py:18-29)."""
    k = _kind(space)
    if k == "Box":
        return len(space.low)
    if k == "Discrete":
        return 1
    if k == "MultiBinary":
        return space.n
    if k == "MultiDiscrete":
        return len(space.nvec)
    raise SpaceException


def calculate_space(space, numframes: int):
    """the observation space after stacking `numframes` frames (util.py:32-45)."""
    k = _kind(space)
    if k == "Box":
        return Box(np.tile(space.low, numframes), np.tile(space.high, numframes), dtype=space.dtype)
    if k == "Discrete":
        return MultiDiscrete([space.n] * numframes)
    if k == "MultiBinary":
        return MultiBinary(space.n * numframes)
    if k == "MultiDiscrete":
        return MultiDiscrete(list(space.nvec) * numframes)
    raise SpaceException


def get_default_obs(env):
    """the filler observation used before an episode has enough history (util.py:48-60)."""
    space = env.observation_space
    k = _kind(space)
    if k == "Box":
        return space.low
    if k == "Discrete":
        return [0]
    if k == "MultiBinary":
        return [0] * space.n
    if k == "MultiDiscrete":
        return [0] * len(space.nvec)
    raise SpaceException


def action_from_policy(obs: np.ndarray, policy, action_mask=None) -> Tuple[np.ndarray, th.Tensor, th.Tensor]:
    """(actions as numpy, values tensor, log_probs tensor) from one policy forward (util.py:63-81).

    The reshape to (-1,)+obs_shape and the host copy of the actions are kept; the forward itself is one fused
    launch on the GPU."""
    obs = np.asarray(obs).reshape((-1,) + tuple(_obs_shape(policy.observation_space)))
    if action_mask is not None:
        actions, values, log_probs = policy.forward(obs, action_mask=np.asarray(action_mask).reshape(obs.shape[0], -1))
    else:
        actions, values, log_probs = policy.forward(obs)
    return actions.cpu().numpy(), values, log_probs


def _obs_shape(space):
    from ..spaces import obs_stored_shape
    shape = getattr(space, "shape", None)
    return shape if shape else obs_stored_shape(space)


def clip_actions(actions: np.ndarray, policy) -> np.ndarray:
    """clip to a Box action space, identity otherwise (util.py:84-99)."""
    space = policy.action_space
    if _kind(space) == "Box":
        actions = np.clip(actions, space.low, space.high)
    return actions


def resample_noise(model, n_steps: int) -> None:
    """gSDE noise tick (util.py:102-111); the MlpPolicy default never uses it."""
    if getattr(model, "use_sde", False) and model.sde_sample_freq > 0 and n_steps % model.sde_sample_freq == 0:
        model.policy.reset_noise(model.env.num_envs)
