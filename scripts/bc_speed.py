#!/usr/bin/env python
"""Behavioural-cloning throughput: Adam steps per second of the persistent-workgroup kernel (one launch for the whole run) next to
the oracle's BC.train on the host (torch CPU, the way the reference runs it: DataLoader batches of 32, eager autograd)."""
import os
import sys
import time

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sb3_oracle as orc  # noqa: E402  (measurement script: the oracle is the CPU side of the comparison)
from pantheonrl_amd.bc import BC  # noqa: E402
from pantheonrl_amd.common import TransitionsMinimal  # noqa: E402
from pantheonrl_amd.spaces import Box, Discrete  # noqa: E402

N, D, L = 32768, 62, 6
rng = np.random.default_rng(0)
obs = rng.standard_normal((N, D)).astype(np.float32)
acts = rng.integers(0, L, N).astype(np.float32)
clone = BC(Box(-np.inf, np.inf, (D,)), Discrete(L), expert_data=TransitionsMinimal(obs, acts))
clone.train(n_epochs=1)
th.cuda.synchronize()
for epochs in (1, 10):
    t0 = time.perf_counter()
    st = clone.train(n_epochs=epochs)
    th.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"device: {st.shape[0]} Adam steps (batch 32) in {dt * 1e3:.1f} ms = {st.shape[0] / dt / 1e3:.1f} k steps/s, "
          f"{dt / st.shape[0] * 1e6:.2f} us per step, {32 * st.shape[0] / dt / 1e6:.2f} M samples/s")
th.set_num_threads(int(os.environ.get("BC_CPU_THREADS", "1")))
orac = orc.FeedForward32Oracle(orc.SpaceSpec("box", dim=D), orc.SpaceSpec("discrete", nvec=(L,)))
order = [np.random.default_rng(1).permutation(4096)]
t0 = time.perf_counter()
ref = orc.bc_train(orac, obs, acts.reshape(-1, 1), order, 32)
dt = time.perf_counter() - t0
print(f"host oracle ({th.get_num_threads()} thread): {len(ref)} Adam steps in {dt * 1e3:.0f} ms = {len(ref) / dt / 1e3:.2f} k steps/s")
