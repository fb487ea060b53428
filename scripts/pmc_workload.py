#!/usr/bin/env python
"""Workload for the PMC passes: (a) calibration kernels with an exactly known byte count in this engine's own access
patterns -- GAE serial / scan at E=16384, T=2048 read 12 B and write 8 B per element with 4-byte-per-lane coalesced
accesses -- and (b) the dominant kernel, ppo_grad at the bench shape.  Run under
    rocprofv3 --pmc FETCH_SIZE -- python scripts/pmc_workload.py     (and again with WRITE_SIZE, in its own pass)
"""
import ctypes as C
import os
import sys

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pantheonrl_amd import PPO, _native as nat, spaces as sp  # noqa: E402
from pantheonrl_amd.vec import SyntheticRollouts, VecOnPolicyAgent, run_iteration_eager  # noqa: E402

E, T = 1024, 128
obs_space, act_space = sp.Box(-np.inf, np.inf, (62,)), sp.Discrete(6)
env = type("S", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()
model = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=E * T // 4, n_epochs=1, seed=0)
model.device_permutations = True
agent = VecOnPolicyAgent(model)
data = SyntheticRollouts(obs_space, E, T, 400, 0, model.device)
run_iteration_eager(agent, data)
th.cuda.synchronize()
pol, rb = model.policy, model.rollout_buffer
lib, h = pol.ctx.lib, pol.ctx.handle
ms = C.c_float(0)
hp = model.hyper()
pol._bind()
for gm in (int(m) for m in os.environ.get("PMC_GEMM_MODES", "0,2").split(",")):   # 0: ppo_grad_fast_kernel, 2: ppo_grad_split_kernel
    nat.check(lib.ph_bench_ppo_grad(h, C.byref(pol.spec), pol.params.data_ptr(), C.byref(rb.c_struct()), C.byref(hp),
                                    int(model.batch_size), 5, gm, C.byref(ms)))
Tb, Eb = 2048, 16384
big = nat.PhRollout()
big.T, big.E = Tb, Eb
keep = []
for name in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns"):
    t = th.zeros(1, device="cuda") if name in ("observations", "actions", "log_probs") else th.randn((Tb, Eb), device="cuda")
    keep.append(t)
    setattr(big, name, t.data_ptr())
lvb = th.zeros(Eb, device="cuda")
for mode in (1, 2):
    nat.check(lib.ph_bench_gae(h, C.byref(big), lvb.data_ptr(), lvb.data_ptr(), 0.99, 0.95, mode, 3, C.byref(ms)))
th.cuda.synchronize()
print("pmc workload done")
