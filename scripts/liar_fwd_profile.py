#!/usr/bin/env python
"""Per-phase shader-clock breakdown of policy_fwd16h_kernel on the Liar's Dice shape (256 rows; debug stamps)."""
import os
import sys

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pantheonrl_amd import PPO, _native as nat  # noqa: E402
from pantheonrl_amd.envs.vec import VecLiarsDice  # noqa: E402
from pantheonrl_amd.vec import VecOnPolicyAgent  # noqa: E402

E, T = 256, 128
spaces = type("S", (), dict(observation_space=VecLiarsDice.observation_space, action_space=VecLiarsDice.action_space,
                            _is_dummy_space_env=True))()
model = PPO("MlpPolicy", spaces, n_steps=T, n_envs=E, batch_size=E * T // 4, n_epochs=10, seed=0)
agent = VecOnPolicyAgent(model)
pol = model.policy
rng = np.random.default_rng(0)
nvec = np.asarray(VecLiarsDice.observation_space.nvec)
obs = th.as_tensor((rng.random((E, len(nvec))) * nvec).astype(np.float32)).to(pol.device)
stamps = th.zeros(16 * 1024, dtype=th.int64, device="cuda")
lib, h = pol.ctx.lib, pol.ctx.handle
nat.check(lib.ph_debug_set_profile_buffer(h, stamps.data_ptr()))
agent.bind_stream()
for rep in range(3):
    stamps.zero_()
    agent.get_action(obs)
    th.cuda.synchronize()
nx = (E + 15) // 16
st = stamps.cpu().numpy().reshape(-1, 16)[: nx * 2]
labels = {0: "start", 1: "feature rows + staging issue", 3: "gather-sum + tanh", 5: "layer 2", 6: "head tiles", 7: "tail"}
for by, name in ((0, "policy"), (1, "value")):
    blk = st[by * nx:(by + 1) * nx]
    blk = blk[blk[:, 0] > 0]
    slots = [k for k in (0, 1, 3, 5, 6, 7) if blk[:, k].max() > 0]
    print(f"{name} net: {len(blk)} workgroups, total {np.median(blk[:, 7] - blk[:, 0]):.0f} ticks (median); grid spread "
          f"{blk[:, 7].max() - blk[:, 0].min()} ticks")
    for p, q in zip(slots[:-1], slots[1:]):
        d = blk[:, q] - blk[:, p]
        print(f"    {labels[q]:<32} median {np.median(d):>8.0f}   max {d.max():>8.0f}")
