#!/usr/bin/env python
"""Per-phase shader-clock breakdown of ONE step (step 8) inside the one-launch scripted rollout (policy_fwd16_rollout_kernel),
from the debug stamps 8..15 the kernel writes for that step: X rows in, layer 1, layer 2, head product, transpose, row tail."""
import ctypes as C
import os
import sys

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pantheonrl_amd import PPO, _native as nat, spaces as sp  # noqa: E402
from pantheonrl_amd.vec import SyntheticRollouts, VecOnPolicyAgent, run_iteration_eager  # noqa: E402

E, T = int(os.environ.get("E", 1024)), int(os.environ.get("T", 128))
obs_space, act_space = sp.Box(-np.inf, np.inf, (62,)), sp.Discrete(6)
env = type("S", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()
model = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=E * T // 4, n_epochs=1, seed=0)
model.device_permutations = True
agent = VecOnPolicyAgent(model)
data = SyntheticRollouts(obs_space, E, T, 400, 0, model.device)
run_iteration_eager(agent, data, True)
th.cuda.synchronize()
pol = model.policy
stamps = th.zeros(16 * 1024, dtype=th.int64, device="cuda")
nat.check(pol.ctx.lib.ph_debug_set_profile_buffer(pol.ctx.handle, stamps.data_ptr()))
agent.bind_stream()
e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
e0.record(th.cuda.current_stream())
agent.rollout_scripted(data)
e1.record(th.cuda.current_stream())
th.cuda.synchronize()
nat.check(pol.ctx.lib.ph_debug_set_profile_buffer(pol.ctx.handle, None))
print(f"scripted rollout of {T} steps x {E} environments: {e0.elapsed_time(e1) * 1e3:.1f} us = {e0.elapsed_time(e1) * 1e3 / T:.2f} us per step")
nwg = (E + 15) // 16
st = stamps.cpu().numpy().reshape(-1, 16)[: 2 * nwg]
labels = ["X rows: load, commit, 2 barriers", "layer 1 (product, tanh, barrier)", "layer 2", "head product (wave 0)",
          "head -> LDS -> row lanes", "row tail (softmax, sample, stores | value, reward)", "end of step (obs copy, next uniforms)"]
for net in range(2):
    blk = st[net * nwg:(net + 1) * nwg, 8:16].astype(np.float64)
    blk = blk[blk[:, 0] > 0]
    d = np.diff(blk, axis=1)
    print(f"net {net}: step 8 of {len(blk)} workgroups, {np.median(blk[:, -1] - blk[:, 0]):.0f} cycles (median)")
    for lab, colv in zip(labels, d.T):
        print(f"    {lab:<58} median {np.median(colv):>7.0f}   max {colv.max():>7.0f}")
    if net == 0 and os.environ.get("TAIL_STAMPS"):   # library built with -DPH_TAIL_STAMPS: slots 2 / 4 / 6 inside the row tail
        raw = st[:nwg].astype(np.float64)
        raw = raw[raw[:, 8] > 0]
        pts = [raw[:, 13], raw[:, 2], raw[:, 4], raw[:, 6], raw[:, 14]]
        for lab, a0, a1 in zip(["tail: mask / max / exp / sum / log", "tail: uniform + inverse CDF", "tail: log-prob, entropy, env fix-up",
                                "tail: stores (+ exchange push)"], pts[:-1], pts[1:]):
            print(f"        {lab:<54} median {np.median(a1 - a0):>7.0f}   max {(a1 - a0).max():>7.0f}")
