"""Shared test plumbing: matched (oracle, device) policy pairs and synthetic rollouts.  The oracle is the checker."""
from __future__ import annotations

import numpy as np
import torch as th

from oracle.sb3_oracle import GaussianMlpPolicyOracle, MlpPolicyOracle, PPOHyper, RolloutBufferOracle, SpaceSpec

# BASELINE.json configs: name -> (obs SpaceSpec, act SpaceSpec)
CONFIGS = {
    "rps": (SpaceSpec("discrete", nvec=(1,)), SpaceSpec("discrete", nvec=(3,))),
    "liar": (SpaceSpec("multidiscrete", nvec=tuple([7] * 6 + [7, 12] * 12)), SpaceSpec("multidiscrete", nvec=(7, 12))),
    "overcooked": (SpaceSpec("box", dim=62), SpaceSpec("discrete", nvec=(6,))),
    "mpe8": (SpaceSpec("box", dim=48), SpaceSpec("discrete", nvec=(5,))),
    "wide": (SpaceSpec("box", dim=130), SpaceSpec("multidiscrete", nvec=(3, 30, 7))),  # 3 feature chunks, Lp=64
    # one-hot observations with heads the 8-logit fast path does not take (policy_fwd16h_kernel): three action components
    # filling all 32 logit lanes (the middle one straddles the 16-lane DPP row boundary); one 20-way Discrete head
    "onehot32": (SpaceSpec("multidiscrete", nvec=(3, 4, 5, 2, 6)), SpaceSpec("multidiscrete", nvec=(5, 16, 11))),
    "discrete20": (SpaceSpec("discrete", nvec=(5,)), SpaceSpec("discrete", nvec=(20,))),
    # every instantiation of the general gradient kernel (observation kind x head class x logit padding): four 16-way action
    # components = 64 logits through the per-component head phase; one-hot observations of exactly two 64-feature chunks with
    # a 20-way head (one lane per row); a 17-way component (too wide for the per-component phase) beside a small one
    "quad16": (SpaceSpec("box", dim=20), SpaceSpec("multidiscrete", nvec=(16, 16, 16, 16))),
    "onehot128": (SpaceSpec("multidiscrete", nvec=(32, 32, 32, 32)), SpaceSpec("discrete", nvec=(20,))),
    "onehot17": (SpaceSpec("multidiscrete", nvec=(9, 40, 30)), SpaceSpec("multidiscrete", nvec=(17, 3))),
    # ADAP: the stored observation is (environment observation ++ context) -- adap_learn.py:448-452
    "adap_oc": (SpaceSpec("box", dim=62 + 3), SpaceSpec("discrete", nvec=(6,))),            # two feature chunks
    "adap_small": (SpaceSpec("box", dim=35 + 3), SpaceSpec("discrete", nvec=(5,))),         # the 64-row fast gradient kernel
    "adap_multi": (SpaceSpec("box", dim=20 + 4), SpaceSpec("multidiscrete", nvec=(3, 9, 4))),
    # corners of the split gradient kernel's shape class: one feature; all 64 features (no free column for the folded bias) with
    # all 8 logits; 63 features (the bias column is the last free one) with a 2-logit head
    "box1": (SpaceSpec("box", dim=1), SpaceSpec("discrete", nvec=(2,))),
    "box64": (SpaceSpec("box", dim=64), SpaceSpec("discrete", nvec=(8,))),
    "box63": (SpaceSpec("box", dim=63), SpaceSpec("discrete", nvec=(2,))),
    # Box observations of three and four feature chunks with heads inside the wide split kernel's class (<= 32 logits)
    "box130": (SpaceSpec("box", dim=130), SpaceSpec("multidiscrete", nvec=(5, 16, 11))),
    "box200": (SpaceSpec("box", dim=200), SpaceSpec("discrete", nvec=(20,))),
    # Box (continuous) action spaces: SB3's DiagGaussian head on the general kernels -- one dimension on the shape the categorical
    # fast kernels would take, MPE's continuous 5-vector, two feature chunks, one-hot observations, the widest head (16)
    "gauss1": (SpaceSpec("box", dim=62), SpaceSpec("box", dim=1)),
    "gauss5": (SpaceSpec("box", dim=48), SpaceSpec("box", dim=5)),
    "gauss_wide": (SpaceSpec("box", dim=70), SpaceSpec("box", dim=3)),
    "gauss_onehot": (SpaceSpec("multidiscrete", nvec=(3, 4, 5, 2, 6)), SpaceSpec("box", dim=2)),
    "gauss16": (SpaceSpec("box", dim=8), SpaceSpec("box", dim=16)),
}


def to_space(spec: SpaceSpec, role: str = "obs"):
    from pantheonrl_amd import spaces as sp
    if spec.kind == "box":
        return sp.Box(-np.inf, np.inf, (spec.dim,)) if role == "obs" else sp.Box(-1.0, 1.0, (spec.dim,))
    if spec.kind == "discrete":
        return sp.Discrete(spec.nvec[0])
    return sp.MultiDiscrete(list(spec.nvec))


def sample_obs(spec: SpaceSpec, n: int, rng: np.random.Generator) -> np.ndarray:
    if spec.kind == "box":
        return rng.standard_normal((n, spec.dim)).astype(np.float32)
    return np.stack([rng.integers(0, k, size=n) for k in spec.nvec], axis=1).astype(np.float32)


def oracle_policy(name: str, seed: int = 0, perturb: float = 0.3) -> MlpPolicyOracle:
    """seeded oracle policy; biases and the 0.01-gain action head are perturbed so logits are not ~uniform."""
    th.manual_seed(seed)
    obs_s, act_s = CONFIGS[name]
    pol = (GaussianMlpPolicyOracle if act_s.kind == "box" else MlpPolicyOracle)(obs_s, act_s)
    g = th.Generator().manual_seed(seed + 1)
    with th.no_grad():
        for p in pol.parameters():
            if p.ndim == 1:
                p.add_(perturb * th.randn(p.shape, generator=g))
        pol.action_net.weight.add_(perturb * th.randn(pol.action_net.weight.shape, generator=g))
    return pol


def device_policy(name: str, oracle: MlpPolicyOracle):
    from pantheonrl_amd.ppo import ActorCriticPolicy, GaussianActorCriticPolicy
    obs_s, act_s = CONFIGS[name]
    cls = GaussianActorCriticPolicy if act_s.kind == "box" else ActorCriticPolicy
    pol = cls(to_space(obs_s), to_space(act_s, "act"), device="cuda", seed=0)
    pol.set_flat_params(oracle.flat_params())
    return pol


def filled_oracle_buffer(name: str, oracle: MlpPolicyOracle, T: int, E: int, seed: int = 0,
                         p_done: float = 0.05, obs_fn=None) -> RolloutBufferOracle:
    """a full rollout buffer produced by the oracle policy on synthetic inputs (SURVEY.md 8d generators).
    obs_fn(obs, rng) -> obs replaces the N(0, 1) Box observations (integer-valued / rescaled features)."""
    rng = np.random.default_rng(seed)
    obs_s, act_s = CONFIGS[name]
    buf = RolloutBufferOracle(T, E, obs_s.stored_len, act_s.stored_len)
    starts = np.ones(E, np.float32)
    values = None
    for _ in range(T):
        obs = sample_obs(obs_s, E, rng)
        if obs_fn is not None:
            obs = np.ascontiguousarray(obs_fn(obs, rng), dtype=np.float32)
        with th.no_grad():
            actions, values, logp = oracle.forward(th.as_tensor(obs))
        buf.add(obs, actions.numpy(), rng.standard_normal(E).astype(np.float32), starts, values, logp)
        starts = (rng.random(E) < p_done).astype(np.float32)
    buf.compute_returns_and_advantage(values, starts)
    return buf


def upload_buffer(dev_buf, ob: RolloutBufferOracle) -> None:
    """copy every array of an oracle buffer into a device RolloutBuffer (marks it full)."""
    for k in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns"):
        getattr(dev_buf, k).copy_(th.as_tensor(getattr(ob, k)))
    dev_buf.pos, dev_buf.full = dev_buf.buffer_size, True


def make_device_buffer(name: str, pol, T: int, E: int):
    from pantheonrl_amd.ppo import RolloutBuffer
    obs_s, act_s = CONFIGS[name]
    return RolloutBuffer(T, to_space(obs_s), to_space(act_s, "act"), pol.device, pol.ctx, pol.spec, n_envs=E)


__all__ = ["CONFIGS", "PPOHyper", "to_space", "sample_obs", "oracle_policy", "device_policy",
           "filled_oracle_buffer", "upload_buffer", "make_device_buffer"]
