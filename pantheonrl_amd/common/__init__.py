"""Host-side mirror of `pantheonrl.common` for the OnPolicyAgent / MultiAgentEnv hot path (same names, argument
meaning and error behaviour as the reference; the arithmetic runs in libpantheon_hip.so)."""
from .observation import Observation, extract_obs, extract_partial_obs  # noqa: F401
from .agents import (Agent, OffPolicyAgent, OnPolicyAgent, RecordingAgentWrapper,  # noqa: F401
                     StaticPolicyAgent)
from .multiagentenv import (DummyEnv, MultiAgentEnv, PlayerException, SimultaneousEnv,  # noqa: F401
                            TurnBasedEnv)
from .wrappers import HistoryQueue, SimultaneousFrameStack, TurnBasedFrameStack, frame_wrap  # noqa: F401,E402
from .wrappers import SimultaneousRecorder, TurnBasedRecorder, recorder_wrap  # noqa: F401,E402
from .trajsaver import SimultaneousTransitions, TransitionsMinimal, TurnBasedTransitions  # noqa: F401,E402
