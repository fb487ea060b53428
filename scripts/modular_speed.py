"""ModularAlgorithm.train on the device at the bench shape (Overcooked-simple shapes, n_envs = 1024, n_steps = 128, batch = 32 768,
10 epochs) for K = 1, 2, 3 partners: ms per train() call and us per minibatch, next to PPO.train on the plain MlpPolicy; and the
rollout side: us per ModularPolicy.forward + RolloutBuffer.add of 1024 environments (three launches) vs the fused MlpPolicy step."""
import os
import sys
import time

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pantheonrl_amd import PPO, ModularAlgorithm, spaces as sp  # noqa: E402

E, T = 1024, 128
obs_space, act_space = sp.Box(-np.inf, np.inf, (62,)), sp.Discrete(6)
env = type("S", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()
rng = np.random.default_rng(0)


def fill(rb):
    rb.observations.copy_(th.as_tensor(rng.standard_normal((T, E, 62)).astype(np.float32)))
    rb.actions.copy_(th.as_tensor(rng.integers(0, 6, (T, E, 1)).astype(np.float32)))
    for k in ("rewards", "values", "advantages", "returns"):
        getattr(rb, k).copy_(th.as_tensor(rng.standard_normal((T, E)).astype(np.float32)))
    rb.log_probs.fill_(-1.79)
    rb.pos, rb.full = T, True


def timed(fn, reps=3):
    fn()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    th.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


ppo = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=E * T // 4, n_epochs=10, seed=0)
ppo.device_permutations = True
fill(ppo.rollout_buffer)
dt = timed(lambda: ppo.train(sync_stats=False))
print(f"PPO.train (MlpPolicy): {dt * 1e3:.2f} ms per call = {dt / 40 * 1e6:.1f} us per 32768-row minibatch")
obs = th.as_tensor(rng.standard_normal((E, 62)).astype(np.float32)).cuda()
es = th.zeros(E, device="cuda")


def step_plain():
    ppo.rollout_buffer.pos = 0
    for _ in range(32):
        ppo.policy.forward_and_store(obs, ppo.rollout_buffer, es)


dt = timed(step_plain)
print(f"MlpPolicy forward + add, {E} environments: {dt / 32 * 1e6:.1f} us per step (host call included)")
KS = tuple(int(k) for k in os.environ.get("MOD_K", "1,2,3").split(","))
for K in KS:
    m = ModularAlgorithm("ModularPolicy", env, n_steps=T, n_envs=E, batch_size=E * T // 4, n_epochs=10, seed=0,
                         marginal_reg_coef=0.5, policy_kwargs=dict(num_partners=K))
    m.device_permutations = True
    for rb in m.rollout_buffer:
        fill(rb)
    dt = timed(lambda: m.train(sync_stats=False), reps=2)
    print(f"ModularAlgorithm.train, {K} partner(s): {dt * 1e3:.2f} ms per call = {dt / (40 * K) * 1e6:.1f} us per 32768-row minibatch "
          f"({2 * K + 6} launches each: {K + 1} tower forwards, loss, {K + 1} tower backwards, reduce, finalize, Adam)")

    def step_mod():
        m.rollout_buffer[0].pos = 0
        for _ in range(32):
            m.policy.forward_and_store(obs, m.rollout_buffer[0], es, partner_idx=K - 1)
    dt = timed(step_mod)
    print(f"ModularPolicy forward + add, {K} partner(s): {dt / 32 * 1e6:.1f} us per step (two tower launches + the action kernel)")
