"""pantheonrl_amd -- MI355X-native multi-agent PPO rollout + update engine behind PantheonRL's
OnPolicyAgent / MultiAgentEnv / trainer.py surface.

Importing the package does not need a GPU; creating a PPO model or a native context does, and fails loudly
without one (there is no CPU fallback).
"""
from . import _native  # noqa: F401
from .common import (Agent, DummyEnv, MultiAgentEnv, Observation, OnPolicyAgent, PlayerException,  # noqa: F401
                     SimultaneousEnv, StaticPolicyAgent, TurnBasedEnv)
from .ppo import PPO, ActorCriticPolicy, RolloutBuffer  # noqa: F401
from .adap import ADAP, AdapAgent, AdapPolicy  # noqa: F401
from .modular import ModularAlgorithm, ModularPolicy  # noqa: F401

__version__ = "0.1.0"
