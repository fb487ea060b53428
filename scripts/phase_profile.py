#!/usr/bin/env python
"""Per-phase shader-clock breakdown of policy_fwd_kernel and ppo_grad_kernel at the bench sizes (debug stamps)."""
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
LIAR = os.environ.get("SHAPE", "overcooked") == "liar"   # Liar's Dice shapes: general gradient kernel, 5 feature chunks
if LIAR:
    E = int(os.environ.get("E", 256))
    obs_space, act_space = sp.MultiDiscrete([7] * 6 + [7, 12] * 12), sp.MultiDiscrete([7, 12])
env = type("S", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()
model = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=E * T // 4, n_epochs=2, seed=0)
model.device_permutations = True
agent = VecOnPolicyAgent(model)
data = SyntheticRollouts(obs_space, E, T, 400, 0, model.device)
run_iteration_eager(agent, data)
th.cuda.synchronize()
pol, rb = model.policy, model.rollout_buffer
lib, h = pol.ctx.lib, pol.ctx.handle
stamps = th.zeros(16 * 1024, dtype=th.int64, device="cuda")


def report(name, nblk_x, nblk_y, labels, slots=None):
    st = stamps.cpu().numpy().reshape(-1, 16)[: nblk_x * nblk_y]
    if slots is not None:
        st = st[:, slots]
    for by in range(nblk_y):
        blk = st[by * nblk_x:(by + 1) * nblk_x]
        blk = blk[blk[:, 0] > 0]
        if len(blk) == 0:
            continue
        d = np.diff(blk[:, :len(labels) + 1].astype(np.float64), axis=1)
        print(f"{name} net={by}: {len(blk)} workgroups, total {np.median(blk[:, len(labels)] - blk[:, 0]):.0f} cycles (median)")
        for lab, col in zip(labels, d.T):
            print(f"    {lab:<34} median {np.median(col):>9.0f}   max {col.max():>9.0f}")


nat.check(lib.ph_debug_set_profile_buffer(h, stamps.data_ptr()))
# forward
agent.bind_stream()
for rep in range(2):
    stamps.zero_()
    rb.pos = 0
    agent.n_steps = 0
    agent.get_action(data.obs[0])
    th.cuda.synchronize()
if LIAR:
    pass   # the one-hot forward has its own script (scripts/liar_fwd_profile.py)
elif os.environ.get("PH_FWD16", "1") != "0":
    report("policy_fwd16", (E + 15) // 16, 2, ["staging (W1, W2, X, head) + barriers", "L1 mma+tanh", "L2 mma+tanh",
                                              "head + tail"], slots=[0, 1, 3, 5, 7])
else:
    report("policy_fwd", (E + 31) // 32, 2, ["all staging loads+commit", "(chunk loop entry)", "L1 mma", "H1 tanh",
                                            "L2 mma+tanh", "head mma", "sampling / value+obs copy"])
st = stamps.cpu().numpy().reshape(-1, 16)[:(E + 31) // 32]
for lab, a0, a1 in () if LIAR else (("issue W2/W1/Wo/bias", 0, 8), ("barrier(rowphys)", 8, 9), ("X issue + W2 commit", 9, 10),
                    ("W1/Wo/bias commit", 10, 11), ("X commit + barrier", 11, 1)):
    print(f"    fwd prologue  {lab:<24} median {np.median(st[:, a1] - st[:, a0]):>8.0f}")
# grad
hp = model.hyper()
ms = C.c_float(0)
stamps.zero_()
rb.pos = T
nat.check(lib.ph_bench_ppo_grad(h, C.byref(pol.spec), pol.params.data_ptr(), C.byref(rb.c_struct()), C.byref(hp),
                                int(model.batch_size), 1, 0, C.byref(ms)))
th.cuda.synchronize()
nwg = min((model.batch_size + 63) // 64, 256)
print(f"ppo_grad launch {ms.value * 1e3:.1f} us, {nwg} workgroups per net")
if os.environ.get("PH_GRAD_FAST", "1") != "0" and not LIAR:
    report("ppo_grad_fast", nwg, 2, ["prologue + T0 (rows, X, W1, W2)", "S1 mma+tanh", "S2 mma+tanh", "SH head (VALU)",
                                     "S6a fetch_rows", "S6a W1 issue", "S6a side work", "S6a dW2 mma", "S6a dH1 mma", "S6b dZ1, X, W1", "S7 dW1", "remaining tiles",
                                     "epilogue"], slots=[0, 1, 2, 3, 4, 8, 9, 10, 11, 5, 6, 7, 12, 13])
else:
    report("ppo_grad", nwg, 2, ["prologue+S0 meta", "S1 X+W1 load", "S1 mma", "Wo load+H1 tanh", "S2 mma+tanh",
                                "S3 head/value", "S4 loss", "S5a dWo", "S5b dH2/dZ2", "S6a dW2+dH1", "S6b dZ1",
                                "S7 dW1 (+rest of tiles)", "stats"])
if LIAR:
    for rep in range(2):
        stamps.zero_()
        nat.check(lib.ph_bench_ppo_grad(h, C.byref(pol.spec), pol.params.data_ptr(), C.byref(rb.c_struct()), C.byref(hp),
                                        int(model.batch_size), 20 if rep == 0 else 1, 2, C.byref(ms)))
        th.cuda.synchronize()
        if rep == 0:
            print(f"ppo_grad_split_oh: {ms.value * 1e3:.1f} us per launch (20 back-to-back launches)")
    report("ppo_grad_split_oh prologue", nwg, 2, ["kernel entry -> W1 fragment loads issued", "row gather (order -> observations, scalars) arrived",
                                                  "biases, action ranges, accumulators", "X zero + hot features + scalars to LDS", "barrier"],
           slots=[0, 8, 9, 10, 11, 1])
    report("ppo_grad_split_oh", nwg, 2, ["prologue + T0 (gather, one-hot X)", "S1 X W1 (chunks), tanh, split", "S2 mma+tanh+split",
                                         "HZ head forward, loss, dz", "HD d head, dH2 -> dZ2", "S6a dW2, dH1", "S6b dZ1 + S7 dW1 (chunks)",
                                         "remaining tiles", "epilogue"], slots=[0, 1, 2, 3, 4, 5, 6, 7, 12, 13])
if not LIAR:
    stamps.zero_()
    nat.check(lib.ph_bench_ppo_grad(h, C.byref(pol.spec), pol.params.data_ptr(), C.byref(rb.c_struct()), C.byref(hp),
                                    int(model.batch_size), 1, 2, C.byref(ms)))
    th.cuda.synchronize()
    print(f"ppo_grad_split launch {ms.value * 1e3:.1f} us")
    report("ppo_grad_split prologue", nwg, 2, ["issue idx + weight loads", "row scalars issued (idx arrived)", "X loads issued", "weight fragments arrived",
                                               "head state to LDS, T0 scalars", "X commit (rows arrived, split, stores)", "barrier"],
           slots=[0, 8, 9, 10, 11, 14, 15, 1])
    report("ppo_grad_split", nwg, 2, ["prologue + T0", "S1 mma+tanh+split", "S2 mma+tanh", "SH-a head (VALU)",
                                      "SH-b d head W + SH-c dZ2 split", "S6a dW2, dH1 mma", "S6b dZ1 split + S7 dW1", "remaining tiles",
                                      "epilogue"], slots=[0, 1, 2, 3, 4, 5, 6, 7, 12, 13])
nat.check(lib.ph_debug_set_profile_buffer(h, None))
