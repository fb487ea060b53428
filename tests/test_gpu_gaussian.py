"""-m gpu: Box (continuous) action spaces -- SB3's DiagGaussian head on the general forward / gradient kernels -- against the CPU
oracle (`GaussianMlpPolicyOracle`: torch.distributions.Normal summed over the action dimensions) on the same seeded inputs.

Reference rows: `action_from_policy` + `clip_actions` (pantheonrl/common/util.py:63-99), `OnPolicyAgent.get_action`
(agents.py:111-184: the buffer row keeps the raw sample, the environment gets the clipped one), PPO.train()'s arithmetic
(pantheonrl/algos/adap/adap_learn.py:253-344).  Tolerances are test_gpu_parity.py's: means / values / log-probs / entropy atol 2e-5,
gradients 1e-6 + 2e-4 of the largest entry, post-Adam weights 2e-6 per optimizer step."""
import ctypes as C

import numpy as np
import pytest
import torch as th

from oracle import sb3_oracle as orc
from tests import helpers as H
from tests.test_gpu_parity import _assert_grads, _assert_train_stats, _grad_pair, _train_pair

pytestmark = pytest.mark.gpu

GAUSS = [c for c in H.CONFIGS if H.CONFIGS[c][1].kind == "box"]


@pytest.mark.parametrize("name", GAUSS)
@pytest.mark.parametrize("n", [1, 33, 256, 1000])
def test_gaussian_forward_and_evaluate_match_oracle(name, n):
    orac = H.oracle_policy(name, seed=3)
    pol = H.device_policy(name, orac)
    obs_s, act_s = H.CONFIGS[name]
    rng = np.random.default_rng(n)
    obs = H.sample_obs(obs_s, n, rng)
    eps = rng.standard_normal((n, act_s.dim)).astype(np.float32)
    with th.no_grad():
        a_ref, v_ref, lp_ref = orac.forward(th.as_tensor(obs), uniforms=th.as_tensor(eps))
        mu_ref = orac.forward(th.as_tensor(obs), deterministic=True)[0]
    acts, values, logp = pol.forward(obs, uniforms=eps)     # `uniforms` teacher-forces the standard-normal draws
    assert acts.dtype == th.float32 and tuple(acts.shape) == (n, act_s.dim)
    np.testing.assert_allclose(acts.cpu().numpy(), a_ref.numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(values.cpu().numpy(), v_ref.numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(logp.cpu().numpy(), lp_ref.numpy(), atol=2e-5 * act_s.dim + 1e-5, rtol=0)
    np.testing.assert_allclose(pol.get_logits(obs).cpu().numpy(), mu_ref.numpy(), atol=2e-5, rtol=0)     # the head's outputs: the means
    d = pol.forward(obs, deterministic=True)[0].cpu().numpy()
    np.testing.assert_allclose(d, mu_ref.numpy(), atol=2e-5, rtol=0)
    # evaluate_actions on other actions than the policy's own
    given = rng.standard_normal((n, act_s.dim)).astype(np.float32)
    with th.no_grad():
        v2_ref, lp2_ref, ent_ref = orac.evaluate_actions(th.as_tensor(obs), th.as_tensor(given))
    v2, lp2, ent = pol.evaluate_actions(obs, given)
    np.testing.assert_allclose(v2.cpu().numpy(), v2_ref.numpy(), atol=2e-5, rtol=0)
    scale = 1.0 + np.abs(lp2_ref.numpy()).max()
    np.testing.assert_allclose(lp2.cpu().numpy(), lp2_ref.numpy(), atol=2e-5 * scale, rtol=0)
    np.testing.assert_allclose(ent.cpu().numpy(), ent_ref.numpy(), atol=2e-5, rtol=0)


def test_gaussian_sampling_without_teacher_forcing_is_standard_normal_and_replayable():
    """Philox + Box-Muller draws: same (seed, counter) -> same sample; mean and variance of (a - mu) / sigma over 64 k draws"""
    name = "gauss5"
    orac = H.oracle_policy(name, seed=1)
    pol = H.device_policy(name, orac)
    obs = H.sample_obs(H.CONFIGS[name][0], 16384, np.random.default_rng(0))
    a1, _, lp1 = pol.forward(obs)
    pol._counter -= 1
    a2, _, lp2 = pol.forward(obs)
    assert th.equal(a1, a2) and th.equal(lp1, lp2)
    a3 = pol.forward(obs)[0]
    assert not th.equal(a1, a3)
    mu = pol.forward(obs, deterministic=True)[0]
    z = ((a1 - mu) / pol.log_std().exp()).cpu().numpy()
    assert abs(z.mean()) < 0.02 and abs(z.var() - 1.0) < 0.03 and np.abs(z).max() < 7.0
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.03


@pytest.mark.parametrize("name,T,E,nb", [("gauss1", 16, 8, 64), ("gauss5", 16, 8, 100), ("gauss_wide", 8, 12, 96),
                                          ("gauss_onehot", 16, 6, 77), ("gauss16", 8, 8, 37), ("gauss5", 64, 64, 4096)])
def test_gaussian_minibatch_gradient_matches_autograd(name, T, E, nb):
    rng = np.random.default_rng(nb)
    idx = rng.permutation(T * E)[:nb]
    hp = orc.PPOHyper(ent_coef=0.01)          # the entropy term's only gradient is d log_std = 1 per dimension
    g, g_ref, st, st_ref, lay = _grad_pair(name, T, E, idx, hp)
    assert lay.P == lay.val_b + 1 + lay.A
    _assert_grads(g, g_ref, lay)
    # log_std's own entries against their own scale (they are few and small beside the weight gradients)
    gl, gl_ref = g[-lay.A:], g_ref[-lay.A:]
    assert np.abs(gl - gl_ref).max() <= 1e-6 + 2e-4 * np.abs(gl_ref).max(), (gl, gl_ref)
    for i, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss")):
        assert abs(st[i] - st_ref[k]) <= 1e-5 + 1e-4 * abs(st_ref[k]), (k, st[i], st_ref[k])
    # gemm_mode 2 asks for the split kernels, which do not take the spec: the same general kernel must answer, bit for bit
    g2 = _grad_pair(name, T, E, idx, hp, gemm_mode=2)[0]
    assert np.array_equal(g, g2)
    g1 = _grad_pair(name, T, E, idx, hp, gemm_mode=1)[0]      # MFMA tiles == fmaf chains, bitwise
    assert np.array_equal(g, g1)


def test_gaussian_gradient_options():
    idx = np.random.default_rng(0).permutation(16 * 8)[:100]
    hp = orc.PPOHyper(clip_range=0.1, clip_range_vf=0.3, ent_coef=0.02, vf_coef=0.7, normalize_advantage=False)
    g, g_ref, _, _, lay = _grad_pair("gauss5", 16, 8, idx, hp)
    _assert_grads(g, g_ref, lay)


@pytest.mark.parametrize("name,T,E,batch,epochs", [("gauss5", 32, 8, 64, 3), ("gauss1", 25, 5, 64, 2), ("gauss_onehot", 16, 6, 32, 2)])
def test_gaussian_train_matches_oracle(name, T, E, batch, epochs):
    hp = orc.PPOHyper(batch_size=batch, n_epochs=epochs, ent_coef=0.01)
    model, orac, stats_ref = _train_pair(name, T, E, hp)
    from pantheonrl_amd.ppo import GaussianActorCriticPolicy
    assert isinstance(model.policy, GaussianActorCriticPolicy)
    st = model.last_train_stats
    steps = len(stats_ref)
    assert steps == st.shape[0]
    p, p_ref = model.policy.get_flat_params(), orac.flat_params()
    assert np.abs(p - p_ref).max() <= 2e-6 * steps + 1e-6, np.abs(p - p_ref).max()
    A = H.CONFIGS[name][1].dim
    assert np.abs(p[-A:] - p_ref[-A:]).max() <= 2e-6 * steps + 1e-6 and np.abs(p_ref[-A:]).max() > 0      # log_std moved, the same way
    assert int(model.policy.opt_step.item()) == steps
    N = T * E
    for i, s in enumerate(stats_ref):
        nb_i = min(batch, N - (i % (-(-N // batch))) * batch)
        _assert_train_stats(st[i], s, nb_i, (name, i))


def test_gaussian_agent_stores_the_raw_sample_and_hands_the_environment_the_clipped_one():
    """agents.py:172-184: buf.add(obs, actions, ...) then `return clip_actions(actions, self.model)[0]` (util.py:84-99)"""
    from pantheonrl_amd import PPO, spaces as sp
    from pantheonrl_amd.common.agents import OnPolicyAgent
    from pantheonrl_amd.common.observation import Observation
    env = type("E", (), dict(observation_space=sp.Box(-np.inf, np.inf, (6,)), action_space=sp.Box(-0.05, 0.05, (3,)),
                             _is_dummy_space_env=True))()
    model = PPO("MlpPolicy", env, n_steps=8, batch_size=4, n_epochs=2, seed=0)
    agent = OnPolicyAgent(model)
    rng = np.random.default_rng(0)
    outs = []
    for t in range(8):
        a = agent.get_action(Observation(rng.standard_normal(6).astype(np.float32)))
        agent.update(float(rng.standard_normal()), bool(t == 5))
        assert a.shape == (3,) and a.dtype == np.float32
        outs.append(a)
    outs = np.stack(outs)
    x = rng.standard_normal((5, 6)).astype(np.float32)
    pa, _ = model.predict(x)                                         # SB3 BasePolicy.predict clips Box actions
    assert pa.shape == (5, 3) and np.abs(pa).max() <= 0.05
    stored = model.rollout_buffer.actions.cpu().numpy().reshape(8, 3)
    assert np.abs(stored).max() > 0.05                              # log_std = 0: the raw samples leave the tiny box ...
    assert np.array_equal(outs, np.clip(stored, -0.05, 0.05))       # ... the environment's copy never does
    before = model.policy.get_flat_params()
    agent.get_action(Observation(rng.standard_normal(6).astype(np.float32)))   # buffer full: GAE + train first (agents.py:126)
    after = model.policy.get_flat_params()
    assert not np.array_equal(before, after) and np.all(np.isfinite(after))
    assert int(model.policy.opt_step.item()) == 2 * 2


def test_gaussian_learn_on_a_host_environment_and_checkpoint_round_trip(tmp_path):
    """PPO.learn() on a one-dimensional target-tracking environment: reward -(a - target)^2; the mean moves toward the target and
    save / load restore log_std under SB3's state_dict name"""
    from pantheonrl_amd import PPO, spaces as sp

    class Track:
        observation_space, action_space = sp.Box(-1, 1, (2,)), sp.Box(-1, 1, (1,))

        def __init__(self):
            self.rng, self.t = np.random.default_rng(0), 0

        def reset(self):
            self.t = 0
            self.x = self.rng.uniform(-1, 1, 2).astype(np.float32)
            return self.x

        def step(self, a):
            assert np.all(np.abs(a) <= 1.0)                          # clipped before it gets here
            r = -float((a[0] - 0.5 * self.x[0]) ** 2)
            self.t += 1
            self.x = self.rng.uniform(-1, 1, 2).astype(np.float32)
            return self.x, r, self.t >= 16, {}

    model = PPO("MlpPolicy", Track(), n_steps=256, batch_size=64, n_epochs=4, learning_rate=3e-3, seed=1)
    obs = np.random.default_rng(5).uniform(-1, 1, (512, 2)).astype(np.float32)

    def err():
        mu = model.policy.forward(obs, deterministic=True)[0].cpu().numpy()[:, 0]
        return float(np.mean((mu - 0.5 * obs[:, 0]) ** 2))
    e0 = err()
    model.learn(total_timesteps=256 * 12)
    assert err() < 0.5 * e0, (e0, err())
    path = str(tmp_path / "gauss")
    model.save(path)
    sd = model.policy.state_dict()
    assert tuple(sd["log_std"].shape) == (1,) and tuple(sd["action_net.weight"].shape) == (1, 64)
    clone = PPO.load(path, env=Track())
    assert np.array_equal(clone.policy.get_flat_params(), model.policy.get_flat_params())


def test_entry_points_written_for_the_categorical_heads_refuse_a_box_action_spec():
    from pantheonrl_amd import _native as nat, spaces as sp
    from pantheonrl_amd.ppo import GaussianActorCriticPolicy, RolloutBuffer
    obs_sp, act_sp = sp.Box(-np.inf, np.inf, (62,)), sp.Box(-1, 1, (1,))
    pol = GaussianActorCriticPolicy(obs_sp, act_sp, device="cuda", seed=0)
    rb = RolloutBuffer(4, obs_sp, act_sp, pol.device, pol.ctx, pol.spec, n_envs=16)
    lib, h = pol.ctx.lib, pol.ctx.handle
    x = th.zeros(16 * 62, device="cuda")
    acts = th.zeros(16, dtype=th.int32, device="cuda")
    out = th.zeros(16, device="cuda")
    rec = th.ones(16, dtype=th.uint8, device="cuda")
    pol._bind()
    # the ragged forward (turn-based partners) is written for the categorical heads: the library refuses the spec by name
    with pytest.raises(nat.NativeError, match=r"Box \(continuous\) action spaces"):
        nat.check(lib.ph_policy_forward_ragged(h, C.byref(pol.spec), pol.params.data_ptr(), x.data_ptr(), None, 1, 1, 0, acts.data_ptr(),
                                               out.data_ptr(), out.data_ptr(), C.byref(rb.c_struct()), acts.data_ptr(), rec.data_ptr(),
                                               out.data_ptr()))
    from pantheonrl_amd import PPO
    from pantheonrl_amd.adap import AdapPolicy
    from pantheonrl_amd.bc import FeedForward32Policy
    from pantheonrl_amd.ppo import UnsupportedPolicyConfig
    from pantheonrl_amd.vec import VecOnPolicyAgent
    env = type("E", (), dict(observation_space=obs_sp, action_space=act_sp, _is_dummy_space_env=True))()
    with pytest.raises(nat.NativeError, match="fused MLP kernels"):
        VecOnPolicyAgent(PPO("MlpPolicy", env, n_steps=4, n_envs=16, batch_size=16))
    with pytest.raises(UnsupportedPolicyConfig, match="categorical heads"):
        AdapPolicy(obs_sp, act_sp, device="cuda")
    with pytest.raises(ValueError, match="categorical"):
        FeedForward32Policy(obs_sp, act_sp)
    with pytest.raises(nat.NativeError, match="host step"):
        pol.forward_and_store_host(np.zeros((16, 62), np.float32), rb, np.zeros(16, np.float32))
    with pytest.raises(nat.NativeError, match="categorical"):
        pol.forward(np.zeros((16, 62), np.float32), action_mask=np.ones((16, 1), np.uint8))
