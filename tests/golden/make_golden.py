"""Regenerates the committed golden vectors from the CPU oracle (oracle/sb3_oracle.py).

    python tests/golden/make_golden.py

The reference itself cannot produce vectors (its arithmetic lives in stable-baselines3==1.7.0, which is absent from
/root/reference and from this image -- SURVEY.md 0.3), so these fixtures pin the ORACLE: a drift in the restatement
shows up as a CPU test failure, and the GPU tests compare the HIP path with the same committed numbers.
Everything is seeded; the files are a few tens of KB.
"""
import os
import sys

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import sb3_oracle as orc  # noqa: E402
from tests import helpers as H  # noqa: E402


def gae_cases():
    out = {}
    for i, (T, E) in enumerate([(4, 1), (16, 3), (64, 8), (130, 5)]):
        rng = np.random.default_rng(100 + i)
        r = rng.standard_normal((T, E)).astype(np.float32)
        v = rng.standard_normal((T, E)).astype(np.float32)
        s = (rng.random((T, E)) < 0.1).astype(np.float32)
        lv = rng.standard_normal(E).astype(np.float32)
        dn = (rng.random(E) < 0.3).astype(np.float32)
        a, ret = orc.gae_reference(r, v, s, lv, dn)
        for k, arr in dict(r=r, v=v, s=s, lv=lv, dn=dn, adv=a, ret=ret).items():
            out[f"c{i}_{k}"] = arr
    out["n_cases"] = np.int64(4)
    return out


def forward_cases():
    out = {}
    for name in ("rps", "liar", "overcooked", "mpe8"):
        orac = H.oracle_policy(name, seed=42)
        obs_s, act_s = H.CONFIGS[name]
        rng = np.random.default_rng(7)
        n = 24
        obs = H.sample_obs(obs_s, n, rng)
        u = rng.random((n, act_s.stored_len)).astype(np.float32)
        with th.no_grad():
            z = orac.logits(th.as_tensor(obs)).numpy()
            a, v, lp = orac.forward(th.as_tensor(obs), uniforms=th.as_tensor(u))
            _, lp_e, ent = orac.evaluate_actions(th.as_tensor(obs), a)
        for k, arr in dict(params=orac.flat_params(), obs=obs, u=u, logits=z, actions=a.numpy().astype(np.int32),
                           values=v.numpy().reshape(-1), logp=lp.numpy(), entropy=ent.numpy()).items():
            out[f"{name}_{k}"] = arr
    return out


def ppo_case():
    name, T, E = "overcooked", 16, 4
    orac = H.oracle_policy(name, seed=43)
    ob = H.filled_oracle_buffer(name, orac, T, E, seed=43)
    hp = orc.PPOHyper(batch_size=24, n_epochs=2)   # 64 rows -> minibatches of 24, 24, 16 (ragged last one)
    perms = np.stack([np.random.default_rng(ep).permutation(T * E) for ep in range(hp.n_epochs)]).astype(np.int32)
    out = dict(params0=orac.flat_params(), perms=perms)
    for k in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns"):
        out["rb_" + k] = getattr(ob, k).copy()
    stats = orc.ppo_train(orac, ob, hp, perms)
    keys = ("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss", "grad_norm")
    out["stats"] = np.array([[s[k] for k in keys] for s in stats], np.float32)
    out["params1"] = orac.flat_params()
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "gae.npz"), **gae_cases())
    np.savez_compressed(os.path.join(HERE, "forward.npz"), **forward_cases())
    np.savez_compressed(os.path.join(HERE, "ppo_step.npz"), **ppo_case())
    for f in ("gae.npz", "forward.npz", "ppo_step.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
