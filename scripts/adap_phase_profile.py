#!/usr/bin/env python
"""Per-phase shader-clock breakdown of adap_context_kernel (debug stamps) at the bench shape + 3 context components."""
import ctypes as C
import os
import sys

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pantheonrl_amd import _native as nat, spaces as sp  # noqa: E402
from pantheonrl_amd.adap import ADAP                     # noqa: E402
from scripts.adap_speed import fill                      # noqa: E402

E, T, D = 1024, 128, int(os.environ.get("D", 59))
env = type("E", (), dict(observation_space=sp.Box(-np.inf, np.inf, (D,)), action_space=sp.Discrete(6),
                         _is_dummy_space_env=True))()
model = ADAP("AdapPolicy", env, n_steps=T, n_envs=E, batch_size=32768, n_epochs=1, seed=0)
model.device_permutations = True
fill(model, np.random.default_rng(0))
model.train(sync_stats=False)
th.cuda.synchronize()
stamps = th.zeros(16 * 1024, dtype=th.int64, device="cuda")
pol = model.policy
nat.check(pol.ctx.lib.ph_debug_set_profile_buffer(pol.ctx.handle, stamps.data_ptr()))
pol.gemm_mode = 0
N, hp = T * E, model.hyper()
idx = th.arange(32768, dtype=th.int32, device="cuda")
g, st = th.zeros(pol.layout.P, device="cuda"), th.zeros(nat.PH_NSTAT, device="cuda")
ad = model.adap_struct(1)
for rep in range(3):
    stamps.zero_()
    nat.check(pol.ctx.lib.ph_adap_minibatch_grad(pol.ctx.handle, C.byref(pol.spec), pol.params.data_ptr(),
                                                 C.byref(model.rollout_buffer.c_struct()), C.byref(hp), idx.data_ptr(), 32768,
                                                 g.data_ptr(), st.data_ptr(), 0, C.byref(ad)))
    th.cuda.synchronize()
nwg = 11
blk = stamps.cpu().numpy().reshape(-1, 16)[:nwg].astype(np.float64)
phases = [("weights || samples", 0, 2), ("X gather", 2, 3), ("H1, H2 (MFMA)", 3, 4), ("logits (MFMA)", 4, 5),
          ("softmax, pairwise KL, dlogits", 5, 7), ("dZ2, dZ1 (MFMA)", 7, 8), ("dW1, dW2, d act_W (MFMA)", 8, 9), ("biases", 9, 10)]
print(f"adap_context_kernel: {nwg} workgroups, total {np.median(blk[:, 10] - blk[:, 0]):.0f} cycles (median)")
for lab, i, j in phases:
    col = blk[:, j] - blk[:, i]
    print(f"    {lab:<32} median {np.median(col):>9.0f}   max {col.max():>9.0f}")
