"""The two games whose rules live entirely in the reference tree plus a tiny registry
standing in for `gym.make` (reference pantheonrl/envs/__init__.py:3-21)."""
from .rps import RPSEnv, RPSWeightedAgent  # noqa: F401
from .liar import LiarEnv, LiarDefaultAgent  # noqa: F401

REGISTRY = {"RPS-v0": RPSEnv, "LiarsDice-v0": LiarEnv}


def make(env_id: str, **kwargs):
    if env_id not in REGISTRY:
        raise KeyError(f"unknown environment id {env_id!r}; known: {sorted(REGISTRY)}")
    return REGISTRY[env_id](**kwargs)
