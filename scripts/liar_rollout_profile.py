"""Per-phase shader-clock breakdown of one vectorised step (the third) inside liar_rollout_kernel (debug stamps)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

from pantheonrl_amd import PPO, _native as nat  # noqa: E402
from pantheonrl_amd.envs.vec import RaggedVecOnPolicyAgent, VecLiarsDice, VecLiarSelfPlay  # noqa: E402
from pantheonrl_amd.vec import VecOnPolicyAgent  # noqa: E402

E, T = 256, 128
spaces = type("S", (), dict(observation_space=VecLiarsDice.observation_space, action_space=VecLiarsDice.action_space,
                            _is_dummy_space_env=True))()
models = [PPO("MlpPolicy", spaces, n_steps=T, n_envs=E, batch_size=E * T // 4, n_epochs=10, seed=s) for s in (0, 1)]
ego, alt = VecOnPolicyAgent(models[0]), RaggedVecOnPolicyAgent(models[1])
sp = VecLiarSelfPlay(E, ego, alt, seed=3, native=True)
sp.rollout_and_learn(T)
th.cuda.synchronize()
stamps = th.zeros(16 * 1024, dtype=th.int64, device="cuda")
pol = models[0].policy
nat.check(pol.ctx.lib.ph_debug_set_profile_buffer(pol.ctx.handle, stamps.data_ptr()))
alt.pos.zero_()
models[0].rollout_buffer.pos = 0
ego.n_steps = 0
sp.rollout_persistent(T, 1, 0)
th.cuda.synchronize()
st = stamps.cpu().numpy().reshape(-1, 16).astype(np.float64)
st = st[st[:, 8] > 0]                    # the workgroups of the launch (tables per workgroup: launch_liar_rollout)
med = lambda a, b: np.median(st[:, b] - st[:, a])   # noqa: E731
print(f"step 2 of the rollout, {len(st)} workgroups (cycles, median)")
print(f"  ego forward                {med(8, 9):8.0f}")
print(f"  book-keeping after ego     {med(9, 10):8.0f}")
print(f"  reply forward              {med(10, 11):8.0f}")
print(f"  book-keeping after reply   {med(11, 12):8.0f}")
print(f"  opening forward            {med(12, 13):8.0f}")
print(f"  book-keeping after opening {med(13, 14):8.0f}")
print(f"  whole step                 {med(8, 14):8.0f}")
print(f"  a step without stamps (steps 0 and 1) {med(15, 8) / 2:8.0f}; average over the {T} steps {med(15, 2) / T:8.0f}")
print("ego forward of that step, by phase (policy half):")
print(f"  step top -> forward body (argument record)     {med(8, 0):8.0f}")
print(f"  loads issued, hot positions -> LDS + barrier   {med(0, 1):8.0f}")
print(f"  W1 row gather + staging commits + tanh + barrier {med(1, 3):6.0f}")
print(f"  layer 2 + barrier                              {med(3, 5):8.0f}")
print(f"  head products + barrier                        {med(5, 6):8.0f}")
print(f"  row tails (softmax, sample, buffer rows)       {med(6, 7):8.0f}")
print(f"  tails done -> behind the closing barrier       {med(7, 9):8.0f}")
