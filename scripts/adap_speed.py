"""Cost of ADAP's context term per PPO update: ph_ppo_train vs ph_adap_train on the same buffer (device permutations, in-kernel
samples, statistics left on the device), at the bench shape with the observation widened by the context (62 + 3 components:
two feature chunks, general gradient kernel) and at a 59 + 3 shape that stays on the 64-row fast gradient kernel.

    python scripts/adap_speed.py            # prints one line per (shape, learner)
"""
import os
import sys
import time

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pantheonrl_amd import PPO, spaces as sp          # noqa: E402
from pantheonrl_amd.adap import ADAP                   # noqa: E402


def fill(model, rng):
    rb = model.rollout_buffer
    T, E = rb.buffer_size, rb.n_envs
    rb.observations.copy_(th.as_tensor(rng.standard_normal(tuple(rb.observations.shape)).astype(np.float32)))
    rb.actions.copy_(th.as_tensor(rng.integers(0, 6, size=tuple(rb.actions.shape)).astype(np.float32)))
    for k in ("values", "log_probs", "advantages", "returns"):
        getattr(rb, k).copy_(th.as_tensor(rng.standard_normal((T, E)).astype(np.float32)) * (0.1 if k == "log_probs" else 1.0))
    rb.log_probs.sub_(1.7)
    rb.pos, rb.full = T, True


def timed(model, reps=5):
    model.device_permutations = True
    model.train(sync_stats=False)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model.train(sync_stats=False)
    th.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    E, T, batch, epochs = 1024, 128, 32768, 10
    for d_env in (59, 62):
        env = type("E", (), dict(observation_space=sp.Box(-np.inf, np.inf, (d_env,)), action_space=sp.Discrete(6),
                                 _is_dummy_space_env=True))()
        wide = type("E", (), dict(observation_space=sp.Box(-np.inf, np.inf, (d_env + 3,)), action_space=sp.Discrete(6),
                                  _is_dummy_space_env=True))()
        kw = dict(n_steps=T, n_envs=E, batch_size=batch, n_epochs=epochs, seed=0)
        ppo, adap = PPO("MlpPolicy", wide, **kw), ADAP("AdapPolicy", env, **kw)
        rng = np.random.default_rng(0)
        fill(ppo, rng)
        fill(adap, rng)
        n_mb = epochs * ((T * E + batch - 1) // batch)
        a, b = timed(ppo), timed(adap)
        print(f"obs {d_env}+3: PPO.train {a:.3f} ms ({a / n_mb * 1e3:.1f} us / minibatch)   ADAP.train {b:.3f} ms "
              f"({b / n_mb * 1e3:.1f} us / minibatch)   context term +{(b - a) / n_mb * 1e3:.1f} us / minibatch", flush=True)
        # AdapPolicyMult: another network (x, x_a = tanh(Ws x + bs) as (64, C), latent = tanh(W2 (x + x_a ctx) + b2)), run as a chain
        # of small launches over dense intermediates (csrc/ph_adapmult.hip), at the bench-size minibatch and at the reference's own
        mult = ADAP("AdapPolicyMult", env, **kw)
        fill(mult, rng)
        c = timed(mult, reps=2)
        print(f"obs {d_env}+3: ADAP(AdapPolicyMult).train {c:.3f} ms ({c / n_mb * 1e3:.1f} us / 32768-row minibatch)", flush=True)
    env = type("E", (), dict(observation_space=sp.Box(-np.inf, np.inf, (62,)), action_space=sp.Discrete(6), _is_dummy_space_env=True))()
    for pol in ("AdapPolicy", "AdapPolicyMult"):     # the reference's sizes: n_envs 1, n_steps 2048, batch 64, 10 epochs = 320 steps
        m = ADAP(pol, env, n_steps=2048, n_envs=1, batch_size=64, n_epochs=10, seed=0)
        fill(m, np.random.default_rng(1))
        t = timed(m, reps=2)
        print(f"reference sizes (2048 rows, batch 64, 10 epochs): ADAP({pol}).train {t:.2f} ms = {t / 320 * 1e3:.1f} us per Adam step",
              flush=True)


if __name__ == "__main__":
    main()
