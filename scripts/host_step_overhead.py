#!/usr/bin/env python
"""How much of an n_envs = 1 environment step is the Python around the native calls?  Runs WITHOUT a GPU: the policy and the
rollout buffer are built around stub native functions that return at once (the objects are assembled field by field -- the real
constructors need a device), then `OnPolicyAgent.get_action` + `update` are timed exactly as bench.py's gpu_reference_semantics_E1
drives them.  What this prints is the floor the host step path adds to one launch + one wait."""
import cProfile
import ctypes as C
import os
import pstats
import sys
import time
from collections import deque

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pantheonrl_amd import _native as nat, ppo as P, spaces as sp  # noqa: E402
from pantheonrl_amd.common.agents import OnPolicyAgent  # noqa: E402
from pantheonrl_amd.common.observation import Observation  # noqa: E402


class _Lib:
    """every native entry point returns success at once; `calls` keeps the names in order (tests/test_host_logic.py reads it)"""

    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        def stub(*a):
            self.calls.append(name)
            return 0
        return stub


class _Ctx:
    handle = None
    _bound_stream = None

    def __init__(self):
        self.lib = _Lib()

    def set_stream(self, s):
        if s != self._bound_stream:
            self._bound_stream = s


def build():
    obs_space, act_space = sp.Box(-np.inf, np.inf, (62,)), sp.Discrete(6)
    pol = P.ActorCriticPolicy.__new__(P.ActorCriticPolicy)
    pol.observation_space, pol.action_space = obs_space, act_space
    pol.ctx, pol.spec, pol.device = _Ctx(), nat.PhSpec(), th.device("cpu")
    pol.layout = type("L", (), dict(D=62, A=1, L=6))()
    pol.params = th.zeros(16839)
    pol._host_out, pol._counter, pol._seed, pol.gemm_mode = {}, 0, 1, 2
    pol._bind = lambda: None
    rb = P.RolloutBuffer.__new__(P.RolloutBuffer)
    rb.buffer_size, rb.n_envs, rb.obs_shape, rb.ctx = 1 << 30, 1, (62,), pol.ctx
    rb._c, rb.pos, rb.full = nat.PhRollout(), 0, False
    rb._bind = lambda: None
    model = type("M", (), {})()
    model.policy, model.rollout_buffer, model.n_steps, model.verbose = pol, rb, 1 << 30, 0
    model.set_logger = lambda *_: None
    model.action_space = act_space
    agent = OnPolicyAgent.__new__(OnPolicyAgent)
    agent.model, agent._last_episode_starts, agent.n_steps, agent.values = model, [True], 0, th.empty(0)
    agent.num_timesteps, agent.log_interval, agent.iteration, agent.name = 0, None, 0, "x"
    model.ep_info_buffer = deque([{"r": 0, "l": 0}], maxlen=100)
    return agent


def main():
    agent = build()
    obs = [Observation(np.random.randn(62).astype(np.float32)) for _ in range(64)]
    def run(n):
        for i in range(n):
            agent.get_action(obs[i & 63])
            agent.update(0.5, False)
    run(2000)
    agent.model.policy.ctx.lib.calls.clear()
    t0 = time.perf_counter()
    n = 20000
    run(n)
    dt = (time.perf_counter() - t0) / n
    assert len(agent.model.policy.ctx.lib.calls) == 2 * n      # one forward and one reward add per environment step, nothing else
    print(f"Python around the native calls: {dt * 1e6:.2f} us per get_action + update pair (native calls stubbed)")
    if "--profile" in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        run(5000)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
