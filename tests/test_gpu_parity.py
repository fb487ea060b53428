"""-m gpu: parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (float32 path, stated per test):
  * GAE serial mode: bit-exact vs the numpy loop (same operation order, contraction off).
  * GAE scan mode: |d| <= 2e-5 * (1 + max|A|) -- the associative re-ordering changes rounding only.
  * logits / values / log-probs / entropy: atol 2e-5 (tanh/exp/log differ by <= 2 ulp between ocml and torch-CPU).
  * gradients: atol 1e-6 + rtol 2e-4 ;  post-Adam weights: atol 2e-6 per optimizer step taken.
  * everything integer (actions under a mask, illegal-action fix-up, permutation indices): bit-exact.
The update runs with the engine's default product arithmetic (gemm_mode 2: float32 operands as three bf16 planes, six matrix-pipe
terms per product, float32 accumulation) wherever the split gradient kernel takes the shape; the tolerances above were stated for
the exact-float32 kernels and none was changed for it.  Tests that pin a mode pass gemm_mode explicitly (0 = exact-f32 MFMA, 1 = its
bitwise VALU restatement, 2 = split).
"""
import copy
import os

import numpy as np
import pytest
import torch as th

from oracle import sb3_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _dev(x):
    return th.as_tensor(x).cuda()


# ----------------------------------------------------------------------------------------------------------------
# K2 GAE
# ----------------------------------------------------------------------------------------------------------------
def _gae_inputs(T, E, seed, p_start=0.05):
    rng = np.random.default_rng(seed)
    r = rng.standard_normal((T, E)).astype(np.float32)
    v = rng.standard_normal((T, E)).astype(np.float32)
    s = (rng.random((T, E)) < p_start).astype(np.float32)
    lv = rng.standard_normal(E).astype(np.float32)
    dn = (rng.random(E) < 0.3).astype(np.float32)
    return r, v, s, lv, dn


def _run_gae(r, v, s, lv, dn, mode, gamma=0.99, lam=0.95):
    from pantheonrl_amd import spaces as sp
    from pantheonrl_amd.ppo import ActorCriticPolicy, RolloutBuffer
    T, E = r.shape
    pol = _run_gae.pol = getattr(_run_gae, "pol", None) or ActorCriticPolicy(sp.Box(-1, 1, (2,)), sp.Discrete(2))
    buf = RolloutBuffer(T, sp.Box(-1, 1, (2,)), sp.Discrete(2), pol.device, pol.ctx, pol.spec, gae_lambda=lam,
                        gamma=gamma, n_envs=E)
    buf.rewards.copy_(_dev(r)); buf.values.copy_(_dev(v)); buf.episode_starts.copy_(_dev(s))
    buf.gae_mode = mode
    buf.compute_returns_and_advantage(_dev(lv), _dev(dn))
    th.cuda.synchronize()
    return buf.advantages.cpu().numpy(), buf.returns.cpu().numpy()


def test_gae_known_answers_appendix_c():
    r = np.array([1, 0, 2, -1], np.float32)[:, None]
    v = np.array([.5, .4, .3, .2], np.float32)[:, None]
    s = np.array([1, 0, 1, 0], np.float32)[:, None]
    for lv, dn, adv in ((0.1, 0, [0.5198, -0.4, 0.86250937, -1.1010001]), (0.2, 0, [0.5198, -0.4, 0.955619, -1.002]),
                        (0.2, 1, [0.5198, -0.4, 0.7693999, -1.2])):
        for mode in (1, 2):
            a, ret = _run_gae(r, v, s, np.float32([lv]), np.float32([dn]), mode)
            np.testing.assert_allclose(a.ravel(), adv, rtol=0, atol=1e-6)
            np.testing.assert_allclose(ret.ravel(), np.float32(adv) + v.ravel(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("T,E", [(1, 1), (4, 1), (7, 3), (8, 64), (37, 5), (128, 1024), (2048, 1), (2048, 70), (300, 33)])
def test_gae_serial_is_bit_exact(T, E):
    r, v, s, lv, dn = _gae_inputs(T, E, seed=T * 1000 + E)
    a_ref, ret_ref = orc.gae_reference(r, v, s, lv, dn)
    a, ret = _run_gae(r, v, s, lv, dn, mode=1)
    assert np.array_equal(a, a_ref), f"max diff {np.abs(a - a_ref).max()}"
    assert np.array_equal(ret, ret_ref)


@pytest.mark.parametrize("T,E", [(1, 1), (4, 1), (7, 3), (8, 64), (37, 5), (128, 1024), (129, 40), (257, 33), (512, 16),
                                 (1000, 31), (1025, 2), (2048, 1), (2048, 70), (3000, 7), (4097, 33)])
def test_gae_scan_within_fp32_tolerance(T, E):
    r, v, s, lv, dn = _gae_inputs(T, E, seed=T * 1000 + E + 1)
    a_ref, ret_ref = orc.gae_reference(r, v, s, lv, dn)
    a64, _ = orc.gae_float64(r, v, s, lv, dn)
    for mode in (2, 0):
        a, ret = _run_gae(r, v, s, lv, dn, mode=mode)
        tol = 2e-5 * (1 + np.abs(a_ref).max())
        assert np.abs(a - a_ref).max() <= tol
        assert np.abs(ret - ret_ref).max() <= tol
        # the scan is no less accurate against float64 than the serial float32 loop is (same order of magnitude)
        assert np.abs(a - a64).max() <= 4 * max(np.abs(a_ref - a64).max(), 1e-6)


@pytest.mark.parametrize("T,E", [(257, 16384), (300, 16397), (1030, 16384 + 64 + 5)])
def test_gae_scan_wide_workgroups_match_the_serial_kernel(T, E):
    """At E >= 16384 and T > 256 the scan walks 64 environments and sixteen chunks per workgroup (csrc/ph_gae.hip: launch_gae);
    the checker is the serial kernel, which the tests above pin bit for bit to the numpy loop (SB3 buffers.py GAE,
    SURVEY.md A.2).  Ragged E (a last workgroup with 13 / 5 live environments) and a first super-chunk that starts below 0."""
    r, v, s, lv, dn = _gae_inputs(T, E, seed=T + E)
    a1, r1 = _run_gae(r, v, s, lv, dn, mode=1)
    idx = np.random.default_rng(0).choice(E, 24, replace=False)
    a_ref, ret_ref = orc.gae_reference(r[:, idx], v[:, idx], s[:, idx], lv[idx], dn[idx])
    assert np.array_equal(a1[:, idx], a_ref) and np.array_equal(r1[:, idx], ret_ref)
    a2, r2 = _run_gae(r, v, s, lv, dn, mode=2)
    tol = 2e-5 * (1 + np.abs(a1).max())
    assert np.abs(a2 - a1).max() <= tol
    assert np.abs(r2 - r1).max() <= tol
    a32, _ = _run_gae(r[:, :96], v[:, :96], s[:, :96], lv[:96], dn[:96], mode=2)   # the 32-environment form on the same columns
    assert np.abs(a32 - a1[:, :96]).max() <= tol


def test_gae_no_episode_boundaries_and_all_boundaries():
    T, E = 256, 48
    r, v, s, lv, dn = _gae_inputs(T, E, seed=5)
    for fill in (0.0, 1.0):
        s[:] = fill
        a_ref, _ = orc.gae_reference(r, v, s, lv, dn)
        assert np.array_equal(_run_gae(r, v, s, lv, dn, 1)[0], a_ref)
        assert np.abs(_run_gae(r, v, s, lv, dn, 2)[0] - a_ref).max() <= 2e-5 * (1 + np.abs(a_ref).max())


def test_gae_full_size_closed_forms():
    """BASELINE full size (E=1024, T=128): size-independent properties instead of the slow oracle loop."""
    T, E, g, lam = 128, 1024, 0.99, 0.95
    ones, zeros = np.ones((T, E), np.float32), np.zeros((T, E), np.float32)
    a, _ = _run_gae(ones, zeros, zeros, np.zeros(E, np.float32), np.zeros(E, np.float32), 2, g, lam)
    k = np.arange(T, 0, -1, dtype=np.float64)[:, None]
    np.testing.assert_allclose(a, np.broadcast_to((1 - (g * lam) ** k) / (1 - g * lam), (T, E)), rtol=2e-5)
    # lambda = 0  =>  A_t = r_t + gamma V_{t+1} nnt - V_t
    r, v, s, lv, dn = _gae_inputs(T, E, seed=9)
    a0, _ = _run_gae(r, v, s, lv, dn, 2, g, 0.0)
    nv = np.vstack([v[1:], lv[None]])
    nnt = 1 - np.vstack([s[1:], dn[None]])
    np.testing.assert_allclose(a0, r + np.float32(g) * nv * nnt - v, atol=1e-5)
    # linearity in the rewards at V = 0
    r2 = np.random.default_rng(3).standard_normal((T, E)).astype(np.float32)
    f = lambda rew: _run_gae(rew, zeros, s, np.zeros(E, np.float32), dn, 2, g, lam)[0]  # noqa: E731
    np.testing.assert_allclose(f(r + r2), f(r) + f(r2), atol=5e-5)


# ----------------------------------------------------------------------------------------------------------------
# K4 forward / evaluate
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", [c for c in H.CONFIGS if H.CONFIGS[c][1].kind != "box"])   # (Box actions: test_gpu_gaussian.py)
@pytest.mark.parametrize("n", [1, 31, 33, 256, 1000])
def test_forward_matches_oracle(name, n):
    orac = H.oracle_policy(name, seed=3)
    pol = H.device_policy(name, orac)
    obs_s, act_s = H.CONFIGS[name]
    rng = np.random.default_rng(n)
    obs = H.sample_obs(obs_s, n, rng)
    u = rng.random((n, act_s.stored_len)).astype(np.float32)
    with th.no_grad():
        z_ref = orac.logits(th.as_tensor(obs)).numpy()
        a_ref, v_ref, lp_ref = orac.forward(th.as_tensor(obs), uniforms=th.as_tensor(u))
    z = pol.get_logits(obs).cpu().numpy()
    np.testing.assert_allclose(z, z_ref, atol=2e-5, rtol=0)
    acts, values, logp = pol.forward(obs, uniforms=u)
    np.testing.assert_allclose(values.cpu().numpy(), v_ref.numpy(), atol=2e-5, rtol=0)
    # teacher-forced inverse-CDF sampling: identical actions except where u sits within 1e-5 of a CDF edge
    acts = acts.cpu().numpy().reshape(n, -1)
    probs = [th.softmax(zc, 1).numpy() for zc in th.split(th.as_tensor(z_ref), list(act_s.nvec), dim=1)]
    near = np.zeros(n, bool)
    for c, p in enumerate(probs):
        near |= (np.abs(np.cumsum(p, 1) - u[:, c:c + 1]) < 1e-5).any(1)
    assert np.array_equal(acts[~near], a_ref.numpy()[~near])
    assert near.mean() < 0.01
    np.testing.assert_allclose(logp.cpu().numpy()[~near], lp_ref.numpy()[~near], atol=2e-5, rtol=0)
    # deterministic = argmax, bit-exact wherever the top-2 gap is not a rounding tie
    with th.no_grad():
        d_ref = orac.forward(th.as_tensor(obs), deterministic=True)[0].numpy()
    d = pol.forward(obs, deterministic=True)[0].cpu().numpy().reshape(n, -1)
    gap_ok = np.ones(n, bool)
    for zc in np.split(z_ref, np.cumsum(act_s.nvec)[:-1], axis=1):
        if zc.shape[1] > 1:
            top = np.sort(zc, 1)
            gap_ok &= (top[:, -1] - top[:, -2]) > 1e-4
    assert np.array_equal(d[gap_ok], d_ref[gap_ok])


@pytest.mark.parametrize("name", ["overcooked", "liar", "wide", "onehot32", "discrete20"])
def test_evaluate_actions_matches_oracle(name):
    orac = H.oracle_policy(name, seed=4)
    pol = H.device_policy(name, orac)
    obs_s, act_s = H.CONFIGS[name]
    rng = np.random.default_rng(0)
    n = 333
    obs = H.sample_obs(obs_s, n, rng)
    acts = H.sample_obs(act_s, n, rng)
    with th.no_grad():
        v_ref, lp_ref, e_ref = orac.evaluate_actions(th.as_tensor(obs), th.as_tensor(acts))
    v, lp, e = pol.evaluate_actions(obs, acts)
    np.testing.assert_allclose(v.cpu().numpy(), v_ref.numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref.numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(e.cpu().numpy(), e_ref.numpy(), atol=2e-5, rtol=0)


def test_mfma_and_valu_tiles_agree_bitwise():
    """v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain: the VALU restatement of the same tiles must match exactly."""
    for name in ("overcooked", "liar"):
        orac = H.oracle_policy(name, seed=5)
        pol = H.device_policy(name, orac)
        obs = H.sample_obs(H.CONFIGS[name][0], 200, np.random.default_rng(1))
        pol.gemm_mode = 0
        z0 = pol.get_logits(obs).cpu().numpy()
        v0 = pol.predict_values(obs).cpu().numpy()
        pol.gemm_mode = 1
        z1 = pol.get_logits(obs).cpu().numpy()
        v1 = pol.predict_values(obs).cpu().numpy()
        assert np.array_equal(z0, z1), np.abs(z0 - z1).max()
        assert np.array_equal(v0, v1)


@pytest.mark.parametrize("name,L", [("mpe8", 5), ("discrete20", 20)])
def test_action_mask_is_integer_exact(name, L):
    """mask path (observation.py action_mask -> modular/policies.py:330-333 -> pettingzoo.py:81-82)."""
    orac = H.oracle_policy(name, seed=6)
    pol = H.device_policy(name, orac)
    rng = np.random.default_rng(2)
    n = 4096
    obs = H.sample_obs(H.CONFIGS[name][0], n, rng)
    mask = (rng.random((n, L)) < 0.6)
    mask[np.arange(n), rng.integers(0, L, n)] = True  # at least one legal action
    # logits are exactly z - 30*(1-m) in float32
    z = pol.get_logits(obs).cpu().numpy()
    zm = pol.get_logits(obs, action_mask=mask.astype(np.uint8)).cpu().numpy()
    assert np.array_equal(zm, (z - np.float32(30.0) * (1 - mask.astype(np.float32))).astype(np.float32))
    with th.no_grad():
        zm_ref = orac.logits(th.as_tensor(obs), th.as_tensor(mask)).numpy()
    np.testing.assert_allclose(zm, zm_ref, atol=2e-5)
    # greedy actions under the mask are always legal and equal the oracle's
    acts = pol.forward(obs, deterministic=True, action_mask=mask.astype(np.uint8))[0].cpu().numpy().ravel()
    assert mask[np.arange(n), acts].all()
    with th.no_grad():
        a_ref = orac.forward(th.as_tensor(obs), deterministic=True, action_mask=th.as_tensor(mask))[0].numpy().ravel()
    top = np.sort(zm_ref, 1)
    ok = (top[:, -1] - top[:, -2]) > 1e-4
    assert np.array_equal(acts[ok], a_ref[ok])
    # env-side fix-up of illegal actions: first legal index, bit-exact
    raw = rng.integers(0, L, n).astype(np.int32)
    fixed = th.as_tensor(raw).cuda()
    import ctypes as C
    from pantheonrl_amd import _native as nat
    m_dev = th.as_tensor(mask.astype(np.uint8)).cuda()
    pol._bind()
    nat.check(pol.ctx.lib.ph_fix_illegal_actions(pol.ctx.handle, fixed.data_ptr(), m_dev.data_ptr(), n, L))
    assert np.array_equal(fixed.cpu().numpy(), orc.fix_illegal_actions(raw, mask))


# ----------------------------------------------------------------------------------------------------------------
# K1 buffer
# ----------------------------------------------------------------------------------------------------------------
def test_buffer_add_reward_reset_and_fused_step():
    name, T, E = "liar", 6, 37
    orac = H.oracle_policy(name, seed=7)
    pol = H.device_policy(name, orac)
    obs_s, act_s = H.CONFIGS[name]
    buf = H.make_device_buffer(name, pol, T, E)
    ref = orc.RolloutBufferOracle(T, E, obs_s.stored_len, act_s.stored_len)
    rng = np.random.default_rng(3)
    starts = np.ones(E, np.float32)
    for t in range(T):
        obs = H.sample_obs(obs_s, E, rng)
        u = rng.random((E, act_s.stored_len)).astype(np.float32)
        if t % 2 == 0:   # explicit RolloutBuffer.add
            acts, values, logp = pol.forward(obs, uniforms=u)
            buf.add(obs, acts.cpu().numpy(), np.zeros(E, np.float32), starts, values, logp)
        else:            # forward fused with the row write
            acts, values, logp = pol.forward_and_store(obs, buf, starts, uniforms=u)
        ref.add(obs, acts.cpu().numpy(), np.zeros(E, np.float32), starts, values.cpu(), logp.cpu())
        for _ in range(2):  # late, additive rewards (agents.py:198), once masked
            rew = rng.standard_normal(E).astype(np.float32)
            buf.add_reward(rew)
            ref.rewards[ref.pos - 1] += rew
        m = rng.random(E) < 0.5
        rew = rng.standard_normal(E).astype(np.float32)
        buf.add_reward(rew, env_mask=m)
        ref.rewards[ref.pos - 1] += np.where(m, rew, 0).astype(np.float32)
        starts = (rng.random(E) < 0.2).astype(np.float32)
    assert buf.full and buf.pos == T
    got = buf.host()
    for k in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs"):
        assert np.array_equal(got[k].reshape(getattr(ref, k).shape), getattr(ref, k)), k
    with pytest.raises(Exception):
        buf.add(obs, acts.cpu().numpy(), np.zeros(E), starts, values, logp)
    buf.reset()
    assert buf.pos == 0 and all(float(np.abs(a).max()) == 0.0 for a in buf.host().values())


# ----------------------------------------------------------------------------------------------------------------
# K3/K5/K6 PPO update
# ----------------------------------------------------------------------------------------------------------------
def _grad_pair(name, T, E, idx, hp: orc.PPOHyper, seed=11, gemm_mode=0, f64=False, obs_fn=None, w1_scale=1.0):
    """device minibatch gradient, the oracle's autograd gradient (float32 as the reference computes it; f64: the same graph in
    float64 -- the yardstick for "which float32 path is closer to the true gradient").  obs_fn: other observation
    distributions than N(0, 1); w1_scale multiplies both first-layer weight matrices (keeps pre-activations in tanh's live range
    when the observations are rescaled)"""
    import ctypes as C
    from pantheonrl_amd import _native as nat
    from pantheonrl_amd.ppo import PPO
    orac = H.oracle_policy(name, seed=seed)
    if w1_scale != 1.0:
        with th.no_grad():
            orac.policy_net[0].weight.mul_(w1_scale)
            orac.value_net_mlp[0].weight.mul_(w1_scale)
    ob = H.filled_oracle_buffer(name, orac, T, E, seed=seed, obs_fn=obs_fn)
    pol = H.device_policy(name, orac)
    pol.gemm_mode = gemm_mode
    buf = H.make_device_buffer(name, pol, T, E)
    H.upload_buffer(buf, ob)
    # oracle gradient
    flat = ob.flat()
    mb = {k: th.as_tensor(v[idx]) for k, v in flat.items()}
    orac.optimizer.zero_grad()
    loss, stats_ref = orc.ppo_minibatch_loss(orac, mb, hp)
    loss.backward()
    g_ref = orac.flat_grads()
    if f64:
        o64 = copy.deepcopy(orac).double()
        mb64 = {k: (v.double() if v.is_floating_point() else v) for k, v in mb.items()}
        o64.optimizer = None
        for q in o64.parameters():
            q.grad = None
        # the loss of orc.ppo_minibatch_loss written out on the float64 modules (Box observations, one Discrete head; the
        # oracle's own entry point casts observations to float32)
        lat_pi, lat_vf = o64.policy_net(mb64["observations"]), o64.value_net_mlp(mb64["observations"])
        logp_all = th.log_softmax(o64.action_net(lat_pi), dim=-1)
        act64 = mb64["actions"].long().flatten()
        log_prob = logp_all.gather(1, act64[:, None])[:, 0]
        entropy = -(logp_all.exp() * logp_all).sum(-1)
        values = o64.value_net(lat_vf).flatten()
        adv = mb64["advantages"]
        if hp.normalize_advantage and len(adv) > 1:
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        ratio = th.exp(log_prob - mb64["old_log_prob"])
        pl = -th.min(adv * ratio, adv * th.clamp(ratio, 1 - hp.clip_range, 1 + hp.clip_range)).mean()
        vp = values if hp.clip_range_vf is None else mb64["old_values"] + th.clamp(values - mb64["old_values"], -hp.clip_range_vf,
                                                                                  hp.clip_range_vf)
        loss64 = pl + hp.ent_coef * (-entropy.mean()) + hp.vf_coef * ((mb64["returns"] - vp) ** 2).mean()
        loss64.backward()
        parts = []
        for seq in (o64.policy_net, o64.value_net_mlp):       # the flat parameter order of MlpPolicyOracle.flat_grads, kept in float64
            for li in (0, 2):
                parts += [seq[li].weight.grad.t().contiguous().reshape(-1), seq[li].bias.grad]
        parts += [o64.action_net.weight.grad.t().contiguous().reshape(-1), o64.action_net.bias.grad,
                  o64.value_net.weight.grad.reshape(-1), o64.value_net.bias.grad]
        g_ref = th.cat(parts).numpy().copy()
    # device gradient
    model = PPO.__new__(PPO)
    for k in ("learning_rate", "clip_range", "clip_range_vf", "ent_coef", "vf_coef", "max_grad_norm", "target_kl",
              "normalize_advantage"):
        setattr(model, k, getattr(hp, k))
    h = PPO.hyper(model)
    idx_t = th.as_tensor(np.asarray(idx, np.int32)).cuda()
    g = th.zeros(pol.layout.P, device="cuda")
    st = th.zeros(nat.PH_NSTAT, device="cuda")
    pol._bind()
    nat.check(pol.ctx.lib.ph_ppo_minibatch_grad(pol.ctx.handle, C.byref(pol.spec), pol.params.data_ptr(),
                                                C.byref(buf.c_struct()), C.byref(h), idx_t.data_ptr(), len(idx),
                                                g.data_ptr(), st.data_ptr(), gemm_mode))
    th.cuda.synchronize()
    return g.cpu().numpy(), g_ref, st.cpu().numpy(), stats_ref, pol.layout


def _assert_grads(g, g_ref, lay):
    scale = np.abs(g_ref).max()
    err = np.abs(g - g_ref)
    assert err.max() <= 1e-6 + 2e-4 * scale, (err.max(), scale, int(err.argmax()), lay.P)


@pytest.mark.parametrize("name,T,E,nb", [("overcooked", 16, 8, 64), ("overcooked", 32, 16, 200), ("mpe8", 8, 8, 37),
                                          ("rps", 64, 1, 64), ("liar", 16, 6, 77), ("wide", 8, 12, 96),
                                          ("overcooked", 64, 64, 4096)])
def test_minibatch_gradient_matches_autograd(name, T, E, nb):
    rng = np.random.default_rng(nb)
    idx = rng.permutation(T * E)[:nb]
    g, g_ref, st, st_ref, lay = _grad_pair(name, T, E, idx, orc.PPOHyper())
    _assert_grads(g, g_ref, lay)
    for i, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss")):
        assert abs(st[i] - st_ref[k]) <= 1e-5 + 1e-4 * abs(st_ref[k]), (k, st[i], st_ref[k])


def test_minibatch_gradient_options_and_valu_cross_check():
    idx = np.random.default_rng(0).permutation(16 * 8)[:100]
    hp = orc.PPOHyper(clip_range=0.1, clip_range_vf=0.3, ent_coef=0.01, vf_coef=0.7, normalize_advantage=False)
    g, g_ref, st, st_ref, lay = _grad_pair("overcooked", 16, 8, idx, hp)
    _assert_grads(g, g_ref, lay)
    g1 = _grad_pair("overcooked", 16, 8, idx, hp, gemm_mode=1)[0]
    assert np.array_equal(g, g1), np.abs(g - g1).max()   # MFMA == fmaf chain, bitwise
    g2, g2_ref, _, _, lay2 = _grad_pair("liar", 16, 6, idx[idx < 96][:70], hp)
    _assert_grads(g2, g2_ref, lay2)


@pytest.mark.parametrize("name,T,E,nb", [("overcooked", 16, 8, 64), ("overcooked", 32, 16, 200), ("mpe8", 8, 8, 37),
                                          ("overcooked", 64, 64, 4096)])
def test_split_bf16_gradient_matches_autograd(name, T, E, nb):
    """gemm_mode 2 (ppo_grad_split_kernel: every product as six bf16 MFMA terms over three-plane operands) against the same
    autograd gradient and at the same tolerance as the exact-float32 kernel"""
    rng = np.random.default_rng(nb)
    idx = rng.permutation(T * E)[:nb]
    g, g_ref, st, st_ref, lay = _grad_pair(name, T, E, idx, orc.PPOHyper(), gemm_mode=2)
    _assert_grads(g, g_ref, lay)
    for i, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss")):
        assert abs(st[i] - st_ref[k]) <= 1e-5 + 1e-4 * abs(st_ref[k]), (k, st[i], st_ref[k])
    hp = orc.PPOHyper(clip_range=0.1, clip_range_vf=0.3, ent_coef=0.01, vf_coef=0.7, normalize_advantage=False)
    g, g_ref, _, _, lay = _grad_pair(name, T, E, idx, hp, gemm_mode=2)
    _assert_grads(g, g_ref, lay)


@pytest.mark.parametrize("name,T,E,nb", [("liar", 16, 6, 77), ("liar", 16, 8, 128), ("liar", 8, 8, 1), ("onehot32", 16, 8, 90),
                                          ("discrete20", 16, 8, 64), ("onehot128", 16, 8, 128), ("onehot17", 16, 6, 77),
                                          ("liar", 64, 8, 321), ("adap_oc", 16, 8, 100), ("adap_multi", 16, 6, 70),
                                          ("box130", 16, 8, 128), ("box200", 8, 8, 37), ("adap_oc", 64, 8, 321)])
def test_split_bf16_one_hot_gradient_matches_autograd(name, T, E, nb):
    """gemm_mode 2 on one-hot observations (ppo_grad_split_oh_kernel: single-plane X, weight fragments from the image, the head as
    MFMA tiles): Liar's Dice (F = 270 in five chunks, components 7 + 12), three components filling all 32 logit slots, a 20-way
    Discrete head (two 16-logit blocks), two feature chunks, a 17-way component across the block boundary; partial tiles, a
    single row, workgroups with a second tile -- against autograd at the exact-f32 kernels' tolerance, and against the exact-f32
    general kernel at float32 rounding level.  The same kernel in its Box form (three-plane X split per tile): the reference's
    Overcooked + ADAP row (65 features: two chunks), a 24-feature row with a three-component head, three and four chunks."""
    rng = np.random.default_rng(nb)
    idx = rng.permutation(T * E)[:nb]
    # (non-default hyper-parameters -- value clipping, entropy and value coefficients, raw advantages -- on every second case)
    hp = orc.PPOHyper() if nb % 2 else orc.PPOHyper(clip_range=0.1, clip_range_vf=0.3, ent_coef=0.01, vf_coef=0.7, normalize_advantage=False)
    g2, g_ref, st, st_ref, lay = _grad_pair(name, T, E, idx, hp, gemm_mode=2)
    _assert_grads(g2, g_ref, lay)
    for i, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss")):
        assert abs(st[i] - st_ref[k]) <= 1e-5 + 1e-4 * abs(st_ref[k]), (k, st[i], st_ref[k])
    g0 = _grad_pair(name, T, E, idx, hp, gemm_mode=0)[0]
    assert np.abs(g2 - g0).max() <= 2e-6 * max(np.abs(g0).max(), 1e-3), (np.abs(g2 - g0).max(), np.abs(g0).max())


def test_split_one_hot_kernel_weight_image_tracks_the_parameters_through_adam_steps():
    """the one-hot kernel's weight fragment image (W1 in five chunks, W2 twice, the head twice) after a train() of several Adam
    steps equals what the parameters split to"""
    import ctypes as C
    from pantheonrl_amd import _native as nat
    hp = orc.PPOHyper(batch_size=40, n_epochs=2)
    model, orac, stats_ref = _train_pair("liar", 16, 6, hp)
    pol = model.policy
    assert pol.gemm_mode == 2
    bad = C.c_int(-2)
    nat.check(pol.ctx.lib.ph_debug_weight_image_mismatches(pol.ctx.handle, C.byref(pol.spec), pol.params.data_ptr(), C.byref(bad)))
    assert bad.value == 0, bad.value
    p, p_ref = pol.get_flat_params(), orac.flat_params()
    assert np.abs(p - p_ref).max() <= 2e-6 * len(stats_ref) + 1e-6, np.abs(p - p_ref).max()


@pytest.mark.parametrize("name,T,E,nb", [("box1", 8, 8, 64), ("box64", 16, 8, 128), ("box64", 8, 8, 1), ("box63", 16, 8, 65),
                                          ("box63", 8, 4, 31), ("mpe8", 64, 8, 257)])
def test_split_bf16_gradient_corner_shapes(name, T, E, nb):
    """feature counts 1 / 63 / 64 (folded and unfolded first-layer bias), 2 and 8 logits, minibatches of 1, 31, 65, 257 rows
    (partial tiles, a tile with a single live row, workgroups with and without a second tile): against autograd, and against the
    exact-f32 kernel at float32 rounding level"""
    rng = np.random.default_rng(nb)
    idx = rng.permutation(T * E)[:nb]
    hp = orc.PPOHyper(clip_range_vf=0.2, ent_coef=0.01)
    g2, g_ref, st, st_ref, lay = _grad_pair(name, T, E, idx, hp, gemm_mode=2)
    _assert_grads(g2, g_ref, lay)
    g0 = _grad_pair(name, T, E, idx, hp, gemm_mode=0)[0]
    assert np.abs(g2 - g0).max() <= 2e-6 * max(np.abs(g0).max(), 1e-3), (np.abs(g2 - g0).max(), np.abs(g0).max())
    for i, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss")):
        assert abs(st[i] - st_ref[k]) <= 1e-5 + 1e-4 * abs(st_ref[k]), (k, st[i], st_ref[k])


@pytest.mark.parametrize("T,E,nb", [(64, 64, 4096), (128, 1024, 32768)])
def test_split_bf16_gradient_is_as_close_to_float64_as_the_float32_kernel(T, E, nb):
    """The accuracy claim behind gemm_mode 2, measured: against the float64 gradient of the same minibatch the split kernel's
    error is no larger than 1.5x the exact-float32 MFMA kernel's (in practice it is smaller: the matrix pipe adds the 32
    products of an instruction before it rounds) -- and both are at float32 rounding level.  At 4 096 rows and at the bench's
    32 768-row minibatch (every workgroup walks two tiles, 512 gradient slabs are reduced)."""
    idx = np.random.default_rng(3).permutation(T * E)[:nb]
    hp = orc.PPOHyper()
    g0, g64, _, _, _ = _grad_pair("overcooked", T, E, idx, hp, gemm_mode=0, f64=True)
    g2 = _grad_pair("overcooked", T, E, idx, hp, gemm_mode=2)[0]
    assert not np.array_equal(g0, g2)                      # it IS the other kernel
    scale = np.abs(g64).max()
    e0, e2 = np.abs(g0 - g64), np.abs(g2 - g64)
    rms0, rms2 = float(np.sqrt((e0 ** 2).mean())), float(np.sqrt((e2 ** 2).mean()))
    print(f"vs float64: f32 kernel max {e0.max():.3e} rms {rms0:.3e} | split kernel max {e2.max():.3e} rms {rms2:.3e} | scale {scale:.3e}")
    assert e2.max() <= 1.5 * e0.max() + 1e-9 and rms2 <= 1.5 * rms0 + 1e-10, (e0.max(), e2.max(), rms0, rms2)
    assert e2.max() <= 2e-6 * scale + 1e-9, (e2.max(), scale)


@pytest.mark.parametrize("kind", ["counts_0_400", "counts_0_400_rescaled_w1", "x1e4", "x1e4_rescaled_w1", "x1e-4",
                                  "x1e-4_rescaled_w1"])
def test_split_bf16_gradient_on_integer_valued_large_and_tiny_observations(kind):
    """Overcooked's real features are counts and distances, not N(0, 1): the split kernel (three bf16 planes per float32 operand)
    on integer-valued observations in [0, 400] and on observations scaled by 1e4 / 1e-4, against the exact-f32 kernel at
    <= 2e-6 of the largest gradient entry and against autograd at the usual tolerance.  Raw, such inputs drive SB3's
    orthogonal first layer deep into tanh's flat region (most first-layer gradients vanish); the `_rescaled_w1` variants divide
    the first-layer weights by the input scale so that every layer's gradient is live while the X planes still carry the
    large / tiny magnitudes."""
    T, E, nb = 64, 64, 4096
    idx = np.random.default_rng(5).permutation(T * E)[:nb]
    scale = {"counts": 400.0, "x1e4": 1e4, "x1e-4": 1e-4}[kind.split("_")[0]]
    if kind.startswith("counts"):
        obs_fn = lambda obs, rng: rng.integers(0, 401, size=obs.shape).astype(np.float32)   # noqa: E731
    else:
        obs_fn = lambda obs, rng: obs * np.float32(scale)                                   # noqa: E731
    w1 = 1.0 / scale if kind.endswith("rescaled_w1") else 1.0
    hp = orc.PPOHyper(ent_coef=0.01)
    g2, g_ref, st, st_ref, lay = _grad_pair("overcooked", T, E, idx, hp, gemm_mode=2, obs_fn=obs_fn, w1_scale=w1)
    g0, g64, _, _, _ = _grad_pair("overcooked", T, E, idx, hp, gemm_mode=0, obs_fn=obs_fn, w1_scale=w1, f64=True)
    top = max(np.abs(g64).max(), 1e-6)
    e0, e2, d20 = np.abs(g0 - g64).max(), np.abs(g2 - g64).max(), np.abs(g2 - g0).max()
    print(f"{kind}: |g|max {top:.3e}; vs float64: exact-f32 kernel {e0:.3e}, split kernel {e2:.3e}; split vs exact-f32 {d20:.3e}")
    assert np.isfinite(g2).all()
    if kind.endswith("rescaled_w1") or kind == "x1e-4":
        assert d20 <= 2e-6 * top, (kind, d20, top)
    # Raw counts / 1e4-scaled inputs put most hidden units deep in tanh's flat region, where 1 - H^2 turns the float32 rounding of
    # a pre-activation of magnitude ~1e2..1e5 into a RELATIVE change of the gradient: the problem is ill-conditioned for any
    # float32 kernel, so the yardstick is the float64 gradient -- the split kernel must not be farther from it than the exact-f32
    # kernel is (the claim of test_split_bf16_gradient_is_as_close_to_float64_as_the_float32_kernel, on these inputs)
    assert e2 <= 1.5 * e0 + 2e-6 * top, (kind, e0, e2, top)
    if kind != "x1e4":
        _assert_grads(g2, g_ref, lay)
    else:
        # pre-activations of magnitude 1e4..1e5: the float32 AUTOGRAD gradient is itself only good to ~1e-3 here (it differs from
        # the float64 one by that much), so "2e-4 of the largest entry against float32 autograd" is not a statement about the
        # kernel; what is asserted is the float64 comparison above and that both kernels sit as close to the float32 autograd
        # gradient as that one sits to float64
        ref_err = np.abs(g_ref - g64).max()
        assert np.abs(g2 - g_ref).max() <= 2.0 * ref_err + e0 + 2e-6 * top, (np.abs(g2 - g_ref).max(), ref_err, e0)
    if kind.endswith("rescaled_w1"):     # the first-layer gradient is really there (not a comparison of zeros)
        assert np.abs(g_ref[lay.pi_W1:lay.pi_b1]).max() > 1e-4 * top * min(scale, 1.0)
    for i, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss")):
        assert abs(st[i] - st_ref[k]) <= 1e-5 + 1e-4 * abs(st_ref[k]), (k, st[i], st_ref[k])


def test_split_kernel_weight_image_tracks_the_parameters_through_adam_steps_and_early_stops():
    """The fragments ppo_grad_split_kernel loads come from an image the Adam kernel updates in place.  After a train() of many
    minibatches -- and after one that the KL test cuts short -- the image must equal what the CURRENT parameters split to,
    element for element; overwriting the parameters from outside must be picked up by the next gradient call."""
    import ctypes as C
    from pantheonrl_amd import _native as nat
    from pantheonrl_amd.ppo import PPO
    name, T, E = "overcooked", 16, 8
    orac = H.oracle_policy(name, seed=4)
    ob = H.filled_oracle_buffer(name, orac, T, E, seed=4)
    obs_s, act_s = H.CONFIGS[name]
    env = type("E", (), dict(observation_space=H.to_space(obs_s), action_space=H.to_space(act_s), _is_dummy_space_env=True))()

    def mismatches(model):
        n = C.c_int(-2)
        pol = model.policy
        nat.check(pol.ctx.lib.ph_debug_weight_image_mismatches(pol.ctx.handle, C.byref(pol.spec), pol.params.data_ptr(), C.byref(n)))
        return n.value

    for target_kl in (None, 1e-6):
        model = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=32, n_epochs=5, learning_rate=3e-3, target_kl=target_kl, seed=0)
        assert model.policy.gemm_mode == 2
        model.policy.set_flat_params(orac.flat_params())
        H.upload_buffer(model.rollout_buffer, ob)
        assert mismatches(model) == -1                     # no image before the first update
        before = model.policy.params.clone()
        model.train()
        th.cuda.synchronize()
        assert not th.equal(before, model.policy.params)
        assert mismatches(model) == 0, (target_kl, mismatches(model))
        model.policy.set_flat_params(orac.flat_params() * 1.01)      # from outside: the image is now stale ...
        assert mismatches(model) > 0
        H.upload_buffer(model.rollout_buffer, ob)
        model.train()                                                 # ... and rebuilt at the start of the next call
        th.cuda.synchronize()
        assert mismatches(model) == 0


def _train_pair(name, T, E, hp: orc.PPOHyper, seed=21, device_perms=False):
    from pantheonrl_amd import _native as nat
    from pantheonrl_amd.ppo import PPO
    orac = H.oracle_policy(name, seed=seed)
    ob = H.filled_oracle_buffer(name, orac, T, E, seed=seed)
    obs_s, act_s = H.CONFIGS[name]
    env = type("E", (), dict(observation_space=H.to_space(obs_s), action_space=H.to_space(act_s),
                             _is_dummy_space_env=True))()
    model = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=hp.batch_size, n_epochs=hp.n_epochs,
                learning_rate=hp.learning_rate, clip_range=hp.clip_range, clip_range_vf=hp.clip_range_vf,
                normalize_advantage=hp.normalize_advantage, ent_coef=hp.ent_coef, vf_coef=hp.vf_coef,
                max_grad_norm=hp.max_grad_norm, target_kl=hp.target_kl, seed=0)
    model.policy.set_flat_params(orac.flat_params())
    H.upload_buffer(model.rollout_buffer, ob)
    N = T * E
    if device_perms:
        model.device_permutations = True
        perms = np.stack([nat.feistel_indices(N, model.permutation_seed + 1, ep) for ep in range(hp.n_epochs)])
        model.train()
    else:
        perms = np.stack([np.random.default_rng(seed + ep).permutation(N) for ep in range(hp.n_epochs)])
        model.train(perms=perms)
    stats_ref = orc.ppo_train(orac, ob, hp, perms)
    return model, orac, stats_ref


# Per-statistic tolerances of the train() comparisons (device statistics vs the oracle's, minibatch by minibatch).  The loss
# terms are means of f32 row terms summed in another order (relative 2e-4, absolute floor 2e-5); approx_kl = mean((ratio - 1) -
# log ratio) cancels to ~1e-7 per row, so it gets an absolute 3e-6; clip_fraction is a COUNT / nb -- a row whose ratio sits on
# the clip boundary may fall either way, so one row of slack; grad_norm relative 2e-4.
def _assert_train_stats(row, ref, nb, where=()):
    tol = {"policy_loss": lambda x: 2e-5 + 2e-4 * abs(x), "value_loss": lambda x: 2e-5 + 2e-4 * abs(x),
           "entropy_loss": lambda x: 2e-5 + 2e-4 * abs(x), "loss": lambda x: 3e-5 + 2e-4 * abs(x),
           "approx_kl": lambda x: 3e-6 + 2e-4 * abs(x), "clip_fraction": lambda x: 1.0 / nb + 1e-7,
           "grad_norm": lambda x: 1e-5 + 2e-4 * abs(x)}
    for j, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss", "grad_norm")):
        assert abs(row[j] - ref[k]) <= tol[k](ref[k]), (where, k, row[j], ref[k])


@pytest.mark.parametrize("name,T,E,batch,epochs", [("overcooked", 32, 8, 64, 3), ("overcooked", 25, 5, 64, 2),
                                                   ("liar", 16, 6, 32, 2), ("rps", 128, 1, 64, 2),
                                                   ("mpe8", 16, 16, 100, 2)])
def test_train_matches_oracle(name, T, E, batch, epochs):
    hp = orc.PPOHyper(batch_size=batch, n_epochs=epochs)
    model, orac, stats_ref = _train_pair(name, T, E, hp)
    st = model.last_train_stats
    assert len(stats_ref) == st.shape[0]
    steps = len(stats_ref)
    p, p_ref = model.policy.get_flat_params(), orac.flat_params()
    assert np.abs(p - p_ref).max() <= 2e-6 * steps + 1e-6, np.abs(p - p_ref).max()
    assert int(model.policy.opt_step.item()) == steps
    N = T * E
    for i, s in enumerate(stats_ref):
        nb_i = min(batch, N - (i % (-(-N // batch))) * batch)
        _assert_train_stats(st[i], s, nb_i, (name, i))
    # first minibatch of the first epoch: ratio == 1
    assert st[0, 3] == 0.0 and abs(st[0, 4]) < 1e-6 and abs(st[0, 0]) < 1e-5


def test_train_target_kl_early_stop_matches_oracle():
    hp = orc.PPOHyper(batch_size=32, n_epochs=6, learning_rate=3e-2, target_kl=0.01)
    model, orac, stats_ref = _train_pair("overcooked", 32, 8, hp, seed=5)
    st = model.last_train_stats
    applied_ref = sum(0 if s.get("stopped") else 1 for s in stats_ref)
    assert any(s.get("stopped") for s in stats_ref), "test must exercise the early stop"
    assert int(model.policy.opt_step.item()) == applied_ref
    assert int((st[:, 7] > 0).sum()) == applied_ref
    p, p_ref = model.policy.get_flat_params(), orac.flat_params()
    assert np.abs(p - p_ref).max() <= 5e-5


def test_device_feistel_permutation_equals_host_statement():
    """perms == NULL: the in-kernel permutation is the integer function ph_feistel_indices evaluates on the host."""
    hp = orc.PPOHyper(batch_size=64, n_epochs=2)
    model, orac, _ = _train_pair("overcooked", 32, 8, hp, seed=9, device_perms=True)
    p, p_ref = model.policy.get_flat_params(), orac.flat_params()
    assert np.abs(p - p_ref).max() <= 2e-6 * 8 + 1e-6


def test_train_full_size_properties():
    """Overcooked-simple throughput shape (E=1024, T=128, batch=E*T/4): first-minibatch closed forms and determinism."""
    from pantheonrl_amd.ppo import PPO
    name, T, E = "overcooked", 128, 1024
    orac = H.oracle_policy(name, seed=1)
    obs_s, act_s = H.CONFIGS[name]
    env = type("E", (), dict(observation_space=H.to_space(obs_s), action_space=H.to_space(act_s),
                             _is_dummy_space_env=True))()

    def run(exclusive=False):
        model = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=T * E // 4, n_epochs=2, seed=0)
        model.policy.ctx.set_exclusive_device(exclusive)
        model.policy.set_flat_params(orac.flat_params())
        rng = np.random.default_rng(0)
        rb, pol = model.rollout_buffer, model.policy
        starts = th.ones(E, device="cuda")
        for _ in range(T):
            obs = _dev(rng.standard_normal((E, 62)).astype(np.float32))
            pol.forward_and_store(obs, rb, starts, uniforms=_dev(rng.random((E, 1)).astype(np.float32)))
            rb.add_reward(_dev(rng.standard_normal(E).astype(np.float32)))
            starts = _dev((rng.random(E) < 1 / 400).astype(np.float32))
        rb.compute_returns_and_advantage(th.zeros(E, device="cuda"), starts)
        model.device_permutations = True
        model.train()
        return model
    m1, m2 = run(), run()
    st = m1.last_train_stats
    assert st.shape[0] == 8 and (st[:, 7] == 1).all()
    assert st[0, 3] == 0.0 and abs(st[0, 4]) < 1e-6 and abs(st[0, 0]) < 2e-5   # ratio == 1 on the first minibatch
    assert np.isfinite(st).all() and np.isfinite(m1.policy.get_flat_params()).all()
    assert np.array_equal(m1.policy.get_flat_params(), m2.policy.get_flat_params())  # fixed-order reductions
    assert not np.array_equal(m1.policy.get_flat_params(), orac.flat_params())
    # a learner that has the device to itself runs reduce + clip + Adam as ONE launch (ppo_step_kernel: every block publishes its
    # sum of squares as a stamped word, sweeps all words, updates its own 64 entries): bitwise the two-launch path -- parameters,
    # Adam moments, step counter, weight image, per-minibatch statistics
    m3 = run(exclusive=True)
    assert np.array_equal(m1.policy.get_flat_params(), m3.policy.get_flat_params())
    assert th.equal(m1.policy.adam_m, m3.policy.adam_m) and th.equal(m1.policy.adam_v, m3.policy.adam_v)
    assert int(m3.policy.opt_step.item()) == int(m1.policy.opt_step.item()) == 8
    assert np.array_equal(m1.last_train_stats, m3.last_train_stats)


def test_fused_step_launch_is_bitwise_the_two_launches_through_a_kl_early_stop():
    """the one-launch minibatch step (exclusive device) on the early-stop path: the KL test stops the update before its step, the
    later minibatches of the call do nothing, statistics rows and optimizer step count equal the two-launch path's"""
    from pantheonrl_amd.ppo import PPO
    name, T, E = "overcooked", 32, 8
    orac = H.oracle_policy(name, seed=5)
    ob = H.filled_oracle_buffer(name, orac, T, E, seed=5)
    obs_s, act_s = H.CONFIGS[name]
    env = type("E", (), dict(observation_space=H.to_space(obs_s), action_space=H.to_space(act_s), _is_dummy_space_env=True))()
    out = []
    for exclusive in (False, True):
        model = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=32, n_epochs=6, learning_rate=3e-2, target_kl=0.01, seed=0)
        model.policy.ctx.set_exclusive_device(exclusive)
        model.policy.set_flat_params(orac.flat_params())
        H.upload_buffer(model.rollout_buffer, ob)
        model.device_permutations = True
        model.train()
        th.cuda.synchronize()
        out.append((model.policy.get_flat_params(), model.last_train_stats.copy(), int(model.policy.opt_step.item())))
    (p0, st0, n0), (p1, st1, n1) = out
    assert 0 < n0 < 6 * 8, n0                                   # the test exercises the early stop
    assert n0 == n1 and np.array_equal(p0, p1) and np.array_equal(st0, st1)


def test_host_step_path_is_bitwise_the_general_path():
    """OnPolicyAgent.get_action for a host environment goes through ph_policy_act_host (stage in, forward + row write, results
    out: ONE native call and one synchronisation per environment step, agents.py:111-184) and Agent.update's scalar reward
    through ph_buffer_add_reward_const; the tensor path it replaces gives the same actions, the same buffer, the same values and,
    after two train() calls on those buffers, the same parameters -- bit for bit (same kernel, same (seed, counter))."""
    from pantheonrl_amd import OnPolicyAgent, PPO
    from pantheonrl_amd.common import Observation
    obs_s, act_s = H.CONFIGS["overcooked"]
    env = type("E", (), dict(observation_space=H.to_space(obs_s), action_space=H.to_space(act_s), _is_dummy_space_env=True))()
    rng = np.random.default_rng(3)
    obs = rng.standard_normal((40, 62)).astype(np.float32)
    rew = rng.standard_normal(40).astype(np.float32)
    done = rng.random(40) < 0.2
    outs = []
    for host in (True, False):
        model = PPO("MlpPolicy", env, n_steps=16, n_envs=1, batch_size=8, n_epochs=2, seed=7)
        model.policy.host_step_path = host          # the instance attribute shadows the class flag
        agent = OnPolicyAgent(model)
        acts = []
        for t in range(40):
            acts.append(np.asarray(agent.get_action(Observation(obs[t]))).copy())
            agent.update(float(rew[t]), bool(done[t]))
        th.cuda.synchronize()
        outs.append((np.array(acts), model.rollout_buffer.host(), model.policy.get_flat_params(), agent.n_steps, agent.iteration))
    (a0, b0, p0, n0, i0), (a1, b1, p1, n1, i1) = outs
    assert i0 == i1 == 2 and n0 == n1                      # two updates happened in both runs
    assert np.array_equal(a0, a1)
    for k in b0:
        assert np.array_equal(b0[k], b1[k]), k
    assert np.array_equal(p0, p1)
    assert len(set(a0.reshape(-1).tolist())) > 1           # the policy did sample


# ----------------------------------------------------------------------------------------------------------------
# the drop-in surface end to end: trainer.py preset-1 object graph on RPS (BASELINE config 1)
# ----------------------------------------------------------------------------------------------------------------
def test_rps_ppo_vs_ppo_plumbing():
    from pantheonrl_amd import OnPolicyAgent, PPO
    from pantheonrl_amd.envs import make
    env = make("RPS-v0")
    altenv = env.getDummyEnv(1)
    ego = PPO("MlpPolicy", env, n_steps=256, seed=0)
    partner = OnPolicyAgent(PPO("MlpPolicy", altenv, n_steps=256, seed=1))
    env.add_partner_agent(partner)
    ego.learn(total_timesteps=1000)
    assert ego.num_timesteps == 1024                      # 4 rollouts of 256
    assert partner.num_timesteps == 1024                  # one partner action per ego step (simultaneous game)
    assert partner.iteration == 3                          # trains at the NEXT get_action after its buffer fills (D-3)
    assert int(ego.policy.opt_step.item()) == 4 * 10 * 4  # 4 updates x 10 epochs x 4 minibatches of 64
    assert len(ego.ep_info_buffer) == 100 and all(e["l"] == 1 for e in ego.ep_info_buffer)


# ----------------------------------------------------------------------------------------------------------------
# vectorised agent (n_envs = E) and the captured iteration graph
# ----------------------------------------------------------------------------------------------------------------
def _vec_setup(T=12, E=96, seed=0, n_epochs=2, name="overcooked"):
    from pantheonrl_amd import PPO
    from pantheonrl_amd.vec import SyntheticRollouts, VecOnPolicyAgent
    orac = H.oracle_policy(name, seed=seed)
    obs_s, act_s = H.CONFIGS[name]
    env = type("E", (), dict(observation_space=H.to_space(obs_s), action_space=H.to_space(act_s),
                             _is_dummy_space_env=True))()
    model = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=T * E // 4, n_epochs=n_epochs, seed=seed)
    model.policy.set_flat_params(orac.flat_params())
    agent = VecOnPolicyAgent(model)
    data = SyntheticRollouts(H.to_space(obs_s), E, T, horizon=5, seed=seed, device=model.device)
    return orac, model, agent, data


def test_vec_agent_rollout_buffer_contents():
    """T x (get_action, update) on device tensors leaves exactly the buffer the reference's callbacks would: obs copied,
    rewards = the late additive rewards (folded into the next step's launch), episode_starts = previous dones (first
    row True), values / log-probs = the policy's at the stored actions, GAE with V(o_{T-1}) (quirk D-1)."""
    orac, model, agent, data = _vec_setup()
    rb = model.rollout_buffer
    agent.bind_stream()
    for t in range(data.T):
        acts = agent.get_action(data.obs[t])
        assert acts.shape == (data.E, 1) and int(acts.min()) >= 0 and int(acts.max()) < 6
        agent.update(data.rewards[t] * 0.5, data.dones[t] * 0)       # two updates for one action: rewards add up,
        agent.update(data.rewards[t] * 0.5, data.dones[t])           # the last done wins (agents.py:44-47)
    agent.flush_rewards()
    got = rb.host()
    obs, rew, dones = data.obs.cpu().numpy(), data.rewards.cpu().numpy(), data.dones.cpu().numpy()
    assert np.array_equal(got["observations"], obs)
    np.testing.assert_allclose(got["rewards"], rew, atol=1e-7)
    assert np.array_equal(got["episode_starts"], np.vstack([np.ones((1, data.E), np.float32), dones[:-1]]))
    with th.no_grad():
        v_ref, lp_ref, _ = orac.evaluate_actions(th.as_tensor(obs.reshape(-1, 62)),
                                                 th.as_tensor(got["actions"].reshape(-1, 1)))
    np.testing.assert_allclose(got["values"].ravel(), v_ref.numpy().ravel(), atol=2e-5)
    np.testing.assert_allclose(got["log_probs"].ravel(), lp_ref.numpy(), atol=2e-5)
    # sampled actions follow the policy distribution (chi-square-ish sanity on the pooled histogram)
    with th.no_grad():
        p = th.softmax(orac.logits(th.as_tensor(obs.reshape(-1, 62))), 1).mean(0).numpy()
    hist = np.bincount(got["actions"].astype(int).ravel(), minlength=6) / got["actions"].size
    assert np.abs(hist - p).max() < 0.05
    # learn_from_buffer: GAE bootstraps with the cached V(o_{T-1}) and the last dones
    values_last = agent.values.clone()
    rb.gae_mode = 1
    agent.sync_stats = True
    agent.learn_from_buffer()
    a_ref, _ = orc.gae_reference(got["rewards"], got["values"], got["episode_starts"], values_last.cpu().numpy(),
                                 dones[-1])
    assert np.array_equal(rb.advantages.cpu().numpy(), a_ref)
    assert agent.iteration == 1 and rb.pos == 0 and agent.n_steps == 0
    assert model.last_train_stats.shape == (8, 8) and (model.last_train_stats[:, 7] == 1).all()


def test_iteration_graph_replays_with_fresh_randomness_and_is_deterministic():
    from pantheonrl_amd.vec import IterationGraph
    _, model_g, agent_g, data = _vec_setup(seed=3)
    stream = th.cuda.Stream()
    graph = IterationGraph(agent_g, data, stream)        # 2 eager warm-up iterations + capture
    with th.cuda.stream(stream):
        graph.launch()
        stream.synchronize()
        acts1 = model_g.rollout_buffer.actions.clone()
        p1 = model_g.policy.get_flat_params()
        graph.launch()
        stream.synchronize()
        acts2 = model_g.rollout_buffer.actions.clone()
        p2 = model_g.policy.get_flat_params()
    assert int(graph.epoch_word.item()) == 2
    assert not th.equal(acts1, acts2), "each replay must draw new actions (device RNG epoch)"
    assert np.isfinite(p2).all() and not np.array_equal(p1, p2)
    # a second, independently built and captured agent reproduces the parameters bit for bit
    _, model_h, agent_h, data_h = _vec_setup(seed=3)
    graph_h = IterationGraph(agent_h, data_h, th.cuda.Stream())
    graph_h.launch(); graph_h.launch()
    th.cuda.synchronize()
    assert np.array_equal(model_h.policy.get_flat_params(), p2)   # run-to-run deterministic


# ----------------------------------------------------------------------------------------------------------------
# vectorised integer game rules: bit-exact against the Python games
# ----------------------------------------------------------------------------------------------------------------
def test_vec_rps_and_liars_dice_rules_are_bit_exact():
    from pantheonrl_amd import _native as nat
    from pantheonrl_amd.envs.liar import LiarEnv
    from pantheonrl_amd.envs.rps import rps_payoff
    from pantheonrl_amd.envs.vec import VecLiarsDice, VecRPS
    ctx = nat.Context(0)
    rng = np.random.default_rng(0)
    E = 1000
    env = VecRPS(E, ctx, th.device("cuda", 0))
    a0, a1 = rng.integers(0, 3, E).astype(np.int32), rng.integers(0, 3, E).astype(np.int32)
    r0, r1, d = env.step(_dev(a0), _dev(a1))
    assert np.array_equal(r0.cpu().numpy(), rps_payoff(a0, a1).astype(np.float32))
    assert np.array_equal(r1.cpu().numpy(), -rps_payoff(a0, a1).astype(np.float32)) and bool((d == 1).all())

    # Liar's Dice: E Python tables and the device tables are fed the same dice and the same (mostly illegal) raw moves
    E = 256
    tables = [LiarEnv() for _ in range(E)]
    hands = np.zeros((E, 12), np.int32)
    for e, t in enumerate(tables):
        t.multi_reset(True)
        hands[e, :6], hands[e, 6:] = t.egohand, t.althand
    vec = VecLiarsDice(E, ctx, th.device("cuda", 0))
    vec.reset(hands)
    alive = np.ones(E, bool)
    ego_turn = rng.random(E) < 0.5
    finished = 0
    for step in range(14):
        acts = np.stack([rng.integers(0, 7, E), rng.integers(0, 12, E)], 1).astype(np.int32)
        if step < 3:
            acts[:, 1] = np.minimum(acts[:, 1], 3 * step + 2)   # keep some games going for a few raises
        obs, rew, done = vec.player_step(_dev(acts), _dev(ego_turn.astype(np.uint8)), _dev(alive.astype(np.uint8)))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        for e in np.nonzero(alive)[0]:
            o_ref, r_ref, d_ref, _ = tables[e].player_step(acts[e], bool(ego_turn[e]))
            assert np.array_equal(obs[e], np.asarray(o_ref, np.float32)), (step, e)
            assert tuple(rew[e]) == tuple(float(x) for x in r_ref) and bool(done[e]) == d_ref
            if d_ref:
                alive[e] = False
                finished += 1
        ego_turn = ~ego_turn
    assert finished == E   # every table reached a call within 12 raises + 1


def test_vec_rps_selfplay_on_device():
    """trainer.py RPS-v0 PPO PPO with n_envs = 256, entirely on the device."""
    from pantheonrl_amd import PPO
    from pantheonrl_amd.envs.rps import rps_payoff
    from pantheonrl_amd.envs.vec import VecRPS, selfplay_iteration
    from pantheonrl_amd.vec import VecOnPolicyAgent
    E, T = 256, 16
    spaces = type("S", (), dict(observation_space=VecRPS.observation_space, action_space=VecRPS.action_space,
                                _is_dummy_space_env=True))()
    agents = []
    for seed in (0, 1):
        m = PPO("MlpPolicy", spaces, n_steps=T, n_envs=E, batch_size=E * T // 4, n_epochs=2, seed=seed)
        m.device_permutations = True
        agents.append(VecOnPolicyAgent(m))
    ego, alt = agents
    env = VecRPS(E, ego.model.policy.ctx, ego.model.device)
    p0 = ego.model.policy.get_flat_params()
    selfplay_iteration(env, ego, alt, T)
    th.cuda.synchronize()
    be, ba = ego.model.rollout_buffer.host(), alt.model.rollout_buffer.host()
    pay = rps_payoff(be["actions"][..., 0].astype(int), ba["actions"][..., 0].astype(int)).astype(np.float32)
    assert np.array_equal(be["rewards"], pay) and np.array_equal(ba["rewards"], -pay)      # zero-sum, integer exact
    assert (be["episode_starts"] == 1).all() and (be["observations"] == 0).all()
    assert ego.iteration == 1 and alt.iteration == 1
    assert not np.array_equal(ego.model.policy.get_flat_params(), p0)
    # one-step episodes with V bootstrapped by a terminal: A_t = r_t - V_t exactly (GAE with every step an episode end)
    np.testing.assert_allclose(be["advantages"][:-1], (be["rewards"] - be["values"])[:-1], atol=1e-6)


# ----------------------------------------------------------------------------------------------------------------
# checkpoint + FIXED / LOAD partners (trainer.py:140-162, 419-432; agents.py:54-79)
# ----------------------------------------------------------------------------------------------------------------
def test_save_load_and_static_policy_agent(tmp_path):
    from pantheonrl_amd import OnPolicyAgent, PPO, StaticPolicyAgent
    from pantheonrl_amd.common import Observation
    from pantheonrl_amd.envs import make
    env = make("LiarsDice-v0")
    model = PPO("MlpPolicy", env, n_steps=32, batch_size=16, n_epochs=2, seed=3)
    partner = OnPolicyAgent(PPO("MlpPolicy", env.getDummyEnv(1), n_steps=32, batch_size=16, n_epochs=2, seed=4))
    env.add_partner_agent(partner)
    np.random.seed(0)
    model.learn(total_timesteps=64)
    path = str(tmp_path / "models" / "liar-ego")
    model.save(path)                                            # trainer.py:420
    partner.model.save(str(tmp_path / "models" / "liar-alt"))    # trainer.py:424
    loaded = PPO.load(path)                                      # trainer.py:149 (gen_load)
    obs = H.sample_obs(H.CONFIGS["liar"][0], 50, np.random.default_rng(0))
    assert th.equal(loaded.policy.get_logits(obs), model.policy.get_logits(obs))
    assert np.array_equal(loaded.policy.adam_m.cpu().numpy(), model.policy.adam_m.cpu().numpy())
    assert int(loaded.policy.opt_step.item()) == int(model.policy.opt_step.item()) > 0
    assert loaded.num_timesteps == model.num_timesteps and loaded.n_steps == 32
    sd = model.policy.state_dict()                               # SB3 module names / nn.Linear shapes
    assert sd["mlp_extractor.policy_net.0.weight"].shape == (64, 270) and sd["action_net.weight"].shape == (19, 64)
    assert sd["value_net.weight"].shape == (1, 64) and len(sd) == 12
    # FIXED partner: a frozen policy in a seat (gen_fixed, trainer.py:160-162)
    fixed = StaticPolicyAgent(PPO.load(str(tmp_path / "models" / "liar-alt")).policy)
    env2 = make("LiarsDice-v0")
    env2.add_partner_agent(fixed)
    env2.reset()
    done, steps = False, 0
    while not done and steps < 20:
        _, r, done, _ = env2.step(env2.action_space.sample())
        steps += 1
    assert done and r in (1, -1)
    act = fixed.get_action(Observation(np.zeros(30)))
    assert act.shape == (2,)
    # LOAD ego: continue training a loaded model on a fresh env (trainer.py:116-124)
    loaded.set_env(env2)
    loaded.learn(total_timesteps=32, reset_num_timesteps=False)
    assert loaded.num_timesteps == model.num_timesteps + 32


def test_vec_frame_stack_matches_history_queue():
    from pantheonrl_amd import _native as nat
    from pantheonrl_amd.common.wrappers import HistoryQueue
    from pantheonrl_amd.vec import VecFrameStack
    E, D, nf = 37, 5, 4
    rng = np.random.default_rng(0)
    default = rng.standard_normal(D).astype(np.float32)
    ctx = nat.Context(0)
    vec = VecFrameStack(E, D, nf, ctx, th.device("cuda", 0), default_obs=default)
    queues = [HistoryQueue(default, nf) for _ in range(E)]
    for step in range(9):
        obs = rng.standard_normal((E, D)).astype(np.float32)
        reset = rng.random(E) < (1.0 if step == 0 else 0.2)
        got = vec.push(_dev(obs), _dev(reset.astype(np.uint8))).cpu().numpy()
        for e, q in enumerate(queues):
            if reset[e]:
                q.reset()
            assert np.array_equal(got[e], q.add(obs[e]).astype(np.float32)), (step, e)


def test_trainer_preset_object_graph_end_to_end(tmp_path, monkeypatch):
    """`trainer.py RPS-v0 PPO PPO --preset 1` (BASELINE config 1) through the mirrored CLI, at a reduced step count."""
    import os
    from pantheonrl_amd import trainer
    monkeypatch.chdir(tmp_path)
    ego, partners, env = trainer.run(["RPS-v0", "PPO", "PPO", "--preset", "1", "--seed", "0", "-t", "600",
                                      "--ego-config", '{"n_steps": 256}', "--alt-config", '{"n_steps": 256}',
                                      "--framestack", "2"])
    assert ego.num_timesteps == 768 and partners[0].iteration == 2
    assert env.observation_space.nvec.tolist() == [1, 1]
    assert os.path.exists("models/RPS-v0-PPO-ego-0.zip") and os.path.exists("models/RPS-v0-PPO-alt-0.zip")
    assert os.path.isdir("logs") and any(n.startswith("RPS-v0-PPOPPO-0") for n in os.listdir("logs"))


def test_trainer_rps_preset_1_exactly_as_baseline_config_1_states_it(tmp_path, monkeypatch):
    """BASELINE config 1 AS WRITTEN: `trainer.py RPS-v0 PPO PPO --preset 1 -t 10000` on SB3's defaults (n_envs 1, n_steps 2048,
    batch 64, 10 epochs; trainer.py:231-256,404-413).  SURVEY.md 8(c): the ego collects 5 x 2048 = 10 240 steps (learn() stops at
    the first rollout boundary past 10 000) and updates after each; the partner trains at the NEXT get_action after its buffer
    fills (agents.py:126), so its fifth full buffer is never trained on: 4 updates.  Every update is 10 epochs x 32 minibatches."""
    from pantheonrl_amd import trainer
    monkeypatch.chdir(tmp_path)
    ego, partners, env = trainer.run(["RPS-v0", "PPO", "PPO", "--preset", "1", "--seed", "0", "-t", "10000"])
    assert ego.n_steps == 2048 and ego.batch_size == 64 and ego.n_epochs == 10 and ego.n_envs == 1
    assert ego.num_timesteps == 10240
    assert int(ego.policy.opt_step.item()) == 5 * 320
    partner = partners[0]
    assert partner.model.n_steps == 2048 and partner.num_timesteps == 10240
    assert partner.iteration == 4 and int(partner.model.policy.opt_step.item()) == 4 * 320
    assert len(ego.ep_info_buffer) == 100 and all(e["l"] == 1 for e in ego.ep_info_buffer)    # one-step episodes (rps.py:45)
    assert np.isfinite(ego.policy.get_flat_params()).all() and np.isfinite(partner.model.policy.get_flat_params()).all()


def test_fused_multi_agent_step_matches_per_agent_calls():
    """ph_policy_step_multi (all local agents in one launch, previous step's joint-action reward folded in) leaves the
    same rollout buffers as the per-agent forward + ph_buffer_add_reward_joint sequence."""
    from pantheonrl_amd import dist as pdist
    from pantheonrl_amd.vec import FusedSelfPlayRollout
    built = [_vec_setup(T=6, E=64, seed=s) for s in (0, 1)]
    agents, datas = [b[2] for b in built], [b[3] for b in built]
    ex = pdist.ActionExchange(2, 64, agents[0].model.device)
    stream = th.cuda.Stream()
    with th.cuda.stream(stream):
        roll = FusedSelfPlayRollout(agents, datas, ex, stream, bonus=0.25)
        for a in agents:
            a.sync_stats = True
        roll.run_iteration(0)
        stream.synchronize()
    bufs = [a.model.rollout_buffer.host() for a in agents]
    acts = [b["actions"][..., 0] for b in bufs]
    for i, (b, d) in enumerate(zip(bufs, datas)):
        expect = d.rewards.cpu().numpy() + 0.25 * (acts[i] == acts[1 - i])
        assert np.array_equal(b["rewards"], expect.astype(np.float32)), i
        assert np.array_equal(b["observations"], d.obs.cpu().numpy())
        assert np.array_equal(b["episode_starts"], np.vstack([np.ones((1, 64), np.float32), d.dones.cpu().numpy()[:-1]]))
        assert agents[i].iteration == 1 and agents[i].model.last_train_stats[:, 7].all()
    assert ex.partner_of(0, 0) == 1 and int(roll.epoch_word.item()) == 1


# ----------------------------------------------------------------------------------------------------------------
# edge cases: single-row minibatch, batch larger than the buffer, maximum logit width, ABI misuse
# ----------------------------------------------------------------------------------------------------------------
def test_train_single_row_last_minibatch_and_oversized_batch():
    # N = 65 rows, batch 64 -> second minibatch has ONE row: SB3 skips advantage normalisation there (len > 1 guard)
    model, orac, stats_ref = _train_pair("overcooked", 13, 5, orc.PPOHyper(batch_size=64, n_epochs=2), seed=31)
    assert model.last_train_stats.shape[0] == len(stats_ref) == 4
    p, p_ref = model.policy.get_flat_params(), orac.flat_params()
    assert np.abs(p - p_ref).max() <= 1e-5
    # batch_size larger than the buffer: one minibatch per epoch with every row
    model, orac, stats_ref = _train_pair("mpe8", 8, 4, orc.PPOHyper(batch_size=1000, n_epochs=3), seed=32)
    assert model.last_train_stats.shape[0] == len(stats_ref) == 3
    assert np.abs(model.policy.get_flat_params() - orac.flat_params()).max() <= 1e-5


def test_maximum_logit_width_and_many_feature_chunks():
    """L = 64 logits (PH_MAX_LOGITS) over 4 components, 5 feature chunks of one-hot observations."""
    from pantheonrl_amd.ppo import ActorCriticPolicy
    obs_s = orc.SpaceSpec("multidiscrete", nvec=(100, 90, 80, 30))        # F = 300
    act_s = orc.SpaceSpec("multidiscrete", nvec=(30, 20, 10, 4))           # L = 64
    H.CONFIGS["_max"] = (obs_s, act_s)
    try:
        orac = H.oracle_policy("_max", seed=8)
        pol = H.device_policy("_max", orac)
        rng = np.random.default_rng(0)
        obs = H.sample_obs(obs_s, 77, rng)
        acts = H.sample_obs(act_s, 77, rng)
        with th.no_grad():
            v_ref, lp_ref, e_ref = orac.evaluate_actions(th.as_tensor(obs), th.as_tensor(acts))
        v, lp, e = pol.evaluate_actions(obs, acts)
        np.testing.assert_allclose(v.cpu().numpy(), v_ref.numpy(), atol=3e-5)
        np.testing.assert_allclose(lp.cpu().numpy(), lp_ref.numpy(), atol=3e-5)
        np.testing.assert_allclose(e.cpu().numpy(), e_ref.numpy(), atol=3e-5)
        idx = rng.permutation(8 * 16)[:100]
        g, g_ref, st, st_ref, lay = _grad_pair("_max", 8, 16, idx, orc.PPOHyper(ent_coef=0.01))
        _assert_grads(g, g_ref, lay)
    finally:
        del H.CONFIGS["_max"]
    with pytest.raises(Exception, match="logits"):
        from pantheonrl_amd import spaces as sp
        ActorCriticPolicy(sp.Box(-1, 1, (3,)), sp.MultiDiscrete([33, 32]))   # 65 logits > PH_MAX_LOGITS


def test_abi_misuse_is_reported_not_fatal():
    import ctypes as C
    from pantheonrl_amd import _native as nat
    orac = H.oracle_policy("mpe8", seed=1)
    pol = H.device_policy("mpe8", orac)
    buf = H.make_device_buffer("mpe8", pol, 4, 8)
    lib, h = pol.ctx.lib, pol.ctx.handle
    obs = th.zeros((8, 48), device="cuda")
    acts = th.zeros((8, 1), dtype=th.int32, device="cuda")
    v = th.zeros(8, device="cuda")
    # n = 0, negative pos, pos beyond the buffer, fused add with n != E, misaligned params, bad GAE mode
    assert lib.ph_policy_forward(h, C.byref(pol.spec), pol.params.data_ptr(), obs.data_ptr(), 0, None, None, None, 0, 0, 0,
                                 acts.data_ptr(), None, v.data_ptr(), v.data_ptr(), None, None, None, 0, None, None, 0) != 0
    assert b"positive" in lib.ph_last_error()
    for pos in (-1, 4):
        assert lib.ph_policy_forward(h, C.byref(pol.spec), pol.params.data_ptr(), obs.data_ptr(), 8, None, None, None, 0, 0,
                                     0, acts.data_ptr(), None, v.data_ptr(), v.data_ptr(), None, None,
                                     C.byref(buf.c_struct()), pos, v.data_ptr(), None, 0) != 0
        assert b"pos" in lib.ph_last_error()
    assert lib.ph_policy_forward(h, C.byref(pol.spec), pol.params.data_ptr(), obs.data_ptr(), 4, None, None, None, 0, 0, 0,
                                 acts.data_ptr(), None, v.data_ptr(), v.data_ptr(), None, None, C.byref(buf.c_struct()), 0,
                                 v.data_ptr(), None, 0) != 0
    assert lib.ph_policy_forward(h, C.byref(pol.spec), pol.params.data_ptr() + 4, obs.data_ptr(), 8, None, None, None, 0, 0,
                                 0, acts.data_ptr(), None, v.data_ptr(), v.data_ptr(), None, None, None, 0, None, None, 0) != 0
    assert b"aligned" in lib.ph_last_error()
    assert lib.ph_gae(h, C.byref(buf.c_struct()), v.data_ptr(), v.data_ptr(), 0.99, 0.95, 7) != 0
    assert lib.ph_buffer_add_reward(h, C.byref(buf.c_struct()), 9, v.data_ptr(), None) != 0
    assert lib.ph_graph_launch(h, 3) != 0 and lib.ph_rng_epoch_advance(h) != 0
    # the context is still usable afterwards
    assert pol.forward(np.zeros((8, 48), np.float32))[0].shape == (8,)
    with pytest.raises(nat.NativeError):
        buf.add_reward(np.zeros(8), pos=99)


# ----------------------------------------------------------------------------------------------------------------
# committed golden vectors (tests/golden/*.npz, made by make_golden.py from the oracle): a file-based expectation
# ----------------------------------------------------------------------------------------------------------------
def test_device_reproduces_committed_golden_vectors():
    import os
    from pantheonrl_amd.ppo import PPO
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(gold, "gae.npz"))
    for i in range(int(g["n_cases"])):
        a, ret = _run_gae(g[f"c{i}_r"], g[f"c{i}_v"], g[f"c{i}_s"], g[f"c{i}_lv"], g[f"c{i}_dn"], mode=1)
        assert np.array_equal(a, g[f"c{i}_adv"]) and np.array_equal(ret, g[f"c{i}_ret"])
    f = np.load(os.path.join(gold, "forward.npz"))
    for name in ("rps", "liar", "overcooked", "mpe8"):
        obs_s, act_s = H.CONFIGS[name]
        from pantheonrl_amd.ppo import ActorCriticPolicy
        pol = ActorCriticPolicy(H.to_space(obs_s), H.to_space(act_s), seed=0)
        pol.set_flat_params(f[f"{name}_params"])
        np.testing.assert_allclose(pol.get_logits(f[f"{name}_obs"]).cpu().numpy(), f[f"{name}_logits"], atol=2e-5)
        v, lp, ent = pol.evaluate_actions(f[f"{name}_obs"], f[f"{name}_actions"])
        np.testing.assert_allclose(v.cpu().numpy().ravel(), f[f"{name}_values"], atol=2e-5)
        np.testing.assert_allclose(lp.cpu().numpy(), f[f"{name}_logp"], atol=2e-5)
        np.testing.assert_allclose(ent.cpu().numpy(), f[f"{name}_entropy"], atol=2e-5)
    s = np.load(os.path.join(gold, "ppo_step.npz"))
    obs_s, act_s = H.CONFIGS["overcooked"]
    env = type("E", (), dict(observation_space=H.to_space(obs_s), action_space=H.to_space(act_s),
                             _is_dummy_space_env=True))()
    model = PPO("MlpPolicy", env, n_steps=16, n_envs=4, batch_size=24, n_epochs=2, seed=0)
    model.policy.set_flat_params(s["params0"])
    rb = model.rollout_buffer
    for k in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns"):
        getattr(rb, k).copy_(th.as_tensor(s["rb_" + k]).reshape(getattr(rb, k).shape))
    rb.pos, rb.full = 16, True
    model.train(perms=s["perms"])
    np.testing.assert_allclose(model.last_train_stats[:, :7], s["stats"], atol=2e-4, rtol=2e-3)
    assert np.abs(model.policy.get_flat_params() - s["params1"]).max() <= 1.5e-5


# ----------------------------------------------------------------------------------------------------------------
# vectorised turn-based self-play (BASELINE config 2: LiarsDice-v0 PPO PPO) with a ragged partner buffer
# ----------------------------------------------------------------------------------------------------------------
def _liar_selfplay(E, T_ego, T_alt, seed=0, native=True):
    from pantheonrl_amd import PPO
    from pantheonrl_amd.envs.vec import RaggedVecOnPolicyAgent, VecLiarsDice, VecLiarSelfPlay
    from pantheonrl_amd.vec import VecOnPolicyAgent
    spaces = type("S", (), dict(observation_space=VecLiarsDice.observation_space,
                                action_space=VecLiarsDice.action_space, _is_dummy_space_env=True))()
    me = PPO("MlpPolicy", spaces, n_steps=T_ego, n_envs=E, batch_size=E * T_ego // 2, n_epochs=2, seed=seed)
    ma = PPO("MlpPolicy", spaces, n_steps=T_alt, n_envs=E, batch_size=E * T_alt // 2, n_epochs=2, seed=seed + 1)
    me.device_permutations = ma.device_permutations = True
    ego, alt = VecOnPolicyAgent(me), RaggedVecOnPolicyAgent(ma)
    calls = []
    inner = alt.get_action

    def logged(obs, rec_mask):
        acts = inner(obs, rec_mask)
        calls.append((acts.cpu().numpy().copy(), rec_mask.cpu().numpy().astype(bool)))
        return acts
    alt.get_action = logged
    return VecLiarSelfPlay(E, ego, alt, seed=seed + 7, native=native), ego, alt, calls


def test_vec_liars_dice_selfplay_matches_the_python_step_loop():
    """Every table of the device self-play is replayed through the Python MultiAgentEnv step loop (same dice, same first
    mover, same sampled moves): the ego's and the partner's recorded transitions must be identical, row by row."""
    from collections import deque
    from pantheonrl_amd.common import Agent, Observation
    from pantheonrl_amd.envs.liar import LiarEnv

    class Shadow(LiarEnv):
        def __init__(self):
            super().__init__()
            self.deals = deque()

        def n_reset(self):
            ego_first, hands = self.deals.popleft()
            self.ego_next = bool(ego_first)
            self.history, self.egohand, self.althand = [], [int(x) for x in hands[:6]], [int(x) for x in hands[6:]]
            return (0 if self.ego_next else 1,), (Observation(self.getObs(self.ego_next)),)

    class Replay(Agent):
        """partner that plays the moves the device sampled and keeps OnPolicyAgent's book (agents.py:172-198)"""
        def __init__(self):
            self.moves, self.rows, self.last_done = deque(), [], True

        def get_action(self, obs, record=True):
            act = self.moves.popleft()
            self.rows.append(dict(obs=np.asarray(obs.obs, np.float32), act=act, rew=0.0, start=float(self.last_done)))
            return act

        def update(self, reward, done):
            self.rows[-1]["rew"] += float(reward)
            self.last_done = bool(done)

    E, T, steps = 24, 20, 20
    sp, ego, alt, calls = _liar_selfplay(E, T, 64, native=False)   # the per-call statement of the step (logs the partner's moves)
    shadows, partners = [Shadow() for _ in range(E)], [Replay() for _ in range(E)]
    for s, p in zip(shadows, partners):
        s.add_partner_agent(p)

    def feed(reset_mask):
        hands, first = sp.env.hands.cpu().numpy(), sp.ego_first.cpu().numpy()
        for acts, mask in calls:
            for e in np.nonzero(mask)[0]:
                partners[e].moves.append(acts[e].copy())
        calls.clear()
        for e in np.nonzero(reset_mask)[0]:
            shadows[e].deals.append((first[e], hands[e].copy()))

    feed(np.ones(E, bool))
    cur = [s.reset() for s in shadows]
    ego_rows = []
    n_games = 0
    for t in range(steps):
        before = sp.obs_ego.cpu().numpy().copy()
        done = sp.step().cpu().numpy().astype(bool)
        a_ego = ego.actions.cpu().numpy().copy()
        feed(done)
        after = sp.obs_ego.cpu().numpy()
        for e in range(E):
            assert np.array_equal(before[e], np.asarray(cur[e], np.float32)), (t, e)
            o, r, d, _ = shadows[e].step(a_ego[e])
            assert bool(d) == bool(done[e]), (t, e)
            ego_rows.append((t, e, float(r), bool(d)))
            if d:
                n_games += 1
                o = shadows[e].reset()
            cur[e] = o
            assert np.array_equal(after[e], np.asarray(o, np.float32)), (t, e)
            assert not partners[e].moves            # the Python loop consumed exactly the moves the device made
    assert n_games == sp.episodes and n_games > E   # several games per table
    th.cuda.synchronize()
    be, ba = ego.model.rollout_buffer.host(), alt.model.rollout_buffer.host()
    for t, e, r, d in ego_rows:
        assert be["rewards"][t, e] == r
        if t + 1 < T:
            assert be["episode_starts"][t + 1, e] == float(d)
    pos = alt.pos.cpu().numpy()
    term, opened = alt.term.cpu().numpy(), alt.open.cpu().numpy()
    assert pos.min() >= 1 and len(set(pos.tolist())) > 1           # the columns really are ragged
    for e in range(E):
        rows = partners[e].rows
        assert pos[e] == len(rows)
        for k, row in enumerate(rows):
            assert np.array_equal(ba["observations"][k, e], row["obs"]), (e, k)
            assert np.array_equal(ba["actions"][k, e], row["act"].astype(np.float32))
            assert ba["rewards"][k, e] == row["rew"] and ba["episode_starts"][k, e] == row["start"], (e, k)
            assert np.isfinite(ba["values"][k, e]) and ba["log_probs"][k, e] < 0
        assert opened[e] == 1 and bool(term[e]) == partners[e].last_done


def test_vec_liars_dice_native_step_is_bitwise_the_per_call_step():
    """ph_liar_selfplay_step (one engine call per vectorised step, masks on the device) against the per-call / torch-mask walk
    of the same step with the same RNG counters: identical game state, observations, both rollout buffers, partner
    book-keeping and -- after both learners have trained -- identical parameters."""
    E, T_ego, T_alt = 48, 8, 6
    runs = []
    for native in (True, False):
        sp, ego, alt, _ = _liar_selfplay(E, T_ego, T_alt, seed=11, native=native)
        alt.model.rollout_buffer.gae_mode = ego.model.rollout_buffer.gae_mode = 1
        trained = 0
        for _ in range(3 * T_ego):
            sp.step()
            if alt.full():
                alt.learn_from_buffer()
                trained += 1
        th.cuda.synchronize()
        be, ba = ego.model.rollout_buffer.host(), alt.model.rollout_buffer.host()
        runs.append(dict(hands=sp.env.hands.cpu().numpy(), hist=sp.env.history.cpu().numpy(), obs=sp.obs_ego.cpu().numpy(),
                         pos=alt.pos.cpu().numpy(), flags=np.stack([t.cpu().numpy() for t in (alt.boundary, alt.term, alt.open)]),
                         acted=sp.alt_acted.cpu().numpy(), episodes=sp.episodes, trained=trained, ego_it=ego.iteration,
                         pe=ego.model.policy.get_flat_params(), pa=alt.model.policy.get_flat_params(),
                         **{"e_" + k: v for k, v in be.items() if k in ("observations", "actions", "rewards", "episode_starts")},
                         **{"a_" + k: v for k, v in ba.items()}))
    a, b = runs
    assert a["trained"] >= 1 and a["ego_it"] >= 2 and a["episodes"] > E
    pos = a["pos"]
    for key in a:
        x, y = a[key], b[key]
        if key.startswith("a_") and getattr(x, "ndim", 0) >= 2:      # only the recorded rows of the ragged buffer are defined
            rows = np.arange(x.shape[0])[:, None] < pos[None, :]
            x, y = x[rows], y[rows]
        assert np.array_equal(x, y), key


def test_liar_iteration_graph_replays_the_launch_by_launch_iteration_bitwise():
    """LiarIterationGraph: n_steps vectorised steps + the ego's GAE and update as ONE hipGraph (step-local RNG counters baked
    in, one device epoch word advanced per replay) against the same body executed launch by launch: identical game state,
    buffers and parameters of both learners after several iterations, and the iterations differ from each other."""
    from pantheonrl_amd.envs.vec import LiarIterationGraph
    E, T_ego, T_alt = 48, 8, 6
    runs = []
    for capture in (True, False):
        sp, ego, alt, _ = _liar_selfplay(E, T_ego, T_alt, seed=5)
        g = LiarIterationGraph(sp, T_ego, capture=capture)
        assert (g.graph_id is not None) == capture
        snaps = []
        for _ in range(4):
            g.launch()
            snaps.append(ego.model.rollout_buffer.host()["actions"].copy())
        th.cuda.synchronize()
        runs.append(dict(hands=sp.env.hands.cpu().numpy(), hist=sp.env.history.cpu().numpy(), obs=sp.obs_ego.cpu().numpy(),
                         pos=alt.pos.cpu().numpy(), episodes=sp.episodes, alt_it=alt.iteration, ego_it=ego.iteration,
                         steps=sp.steps_done, pe=ego.model.policy.get_flat_params(), pa=alt.model.policy.get_flat_params(),
                         snaps=np.stack(snaps), epoch=int(g.epoch_word.item())))
    a, b = runs
    assert a["ego_it"] == 6 and a["alt_it"] >= 1 and a["episodes"] > E and a["epoch"] == 6 and a["steps"] == 6 * T_ego
    assert not np.array_equal(a["snaps"][0], a["snaps"][1])           # a replay draws fresh random numbers
    for key in a:
        assert np.array_equal(a[key], b[key]), key


def test_vec_liars_dice_partner_trains_when_every_column_is_full():
    E, T_ego, T_alt = 64, 8, 4
    sp, ego, alt, _ = _liar_selfplay(E, T_ego, T_alt, seed=3)
    alt.model.rollout_buffer.gae_mode = 1                              # serial-in-T: bit-exact with the numpy loop
    p_ego, p_alt = ego.model.policy.get_flat_params(), alt.model.policy.get_flat_params()
    while not alt.full():
        sp.step()
        if ego.n_steps >= T_ego:
            ego.learn_from_buffer()
    assert int(alt.pos.max().item()) == T_alt                         # a full column stops recording
    buf = alt.model.rollout_buffer.host()
    last_values, dones = alt.values.cpu().numpy().copy(), alt.term.cpu().numpy().astype(np.float32)
    alt.learn_from_buffer()
    th.cuda.synchronize()
    adv = alt.model.rollout_buffer.host()["advantages"]
    rb = alt.model.rollout_buffer
    adv_ref, _ = orc.gae_reference(buf["rewards"], buf["values"], buf["episode_starts"], last_values, dones,
                                   rb.gamma, rb.gae_lambda)
    assert np.array_equal(adv, adv_ref)                                # serial GAE is bit-exact
    assert set(np.unique(buf["rewards"]).tolist()) <= {-1.0, 0.0, 1.0} and np.abs(buf["rewards"]).sum() > 0
    assert alt.iteration == 1 and int(alt.pos.max().item()) == 0
    for _ in range(2 * T_ego):                                         # keeps running after both learners updated
        sp.step()
        if ego.n_steps >= T_ego:
            ego.learn_from_buffer()
    assert int(alt.pos.min().item()) >= 1 and ego.iteration >= 1
    q_ego, q_alt = ego.model.policy.get_flat_params(), alt.model.policy.get_flat_params()
    assert np.isfinite(q_ego).all() and np.isfinite(q_alt).all()
    assert not np.array_equal(p_ego, q_ego) and not np.array_equal(p_alt, q_alt)


@pytest.mark.parametrize("game", ["RPS-v0", "LiarsDice-v0"])
def test_trainer_n_envs_runs_device_selfplay(game, tmp_path):
    from pantheonrl_amd import PPO
    from pantheonrl_amd.trainer import run
    ego, partners, _ = run([game, "PPO", "PPO", "--n-envs", "32", "-t", "2048", "--seed", "1",
                            "--ego-config", '{"n_steps": 16, "n_epochs": 2}',
                            "--alt-config", '{"n_steps": 8, "n_epochs": 2}' if game != "RPS-v0" else
                            '{"n_steps": 16, "n_epochs": 2}',
                            "--ego-save", str(tmp_path / "ego"), "--alt-save", str(tmp_path / "alt")])
    assert partners[0].iteration >= 1
    again = PPO.load(str(tmp_path / "ego"))
    assert np.array_equal(again.policy.get_flat_params(), ego.policy.get_flat_params())


@pytest.mark.parametrize("name,T,E,nb", [("quad16", 16, 8, 100), ("onehot128", 16, 8, 128), ("onehot17", 16, 6, 77),
                                          ("onehot32", 16, 8, 90), ("discrete20", 16, 8, 64), ("adap_multi", 16, 6, 70)])
def test_every_general_gradient_kernel_instantiation_matches_autograd(name, T, E, nb):
    """ppo_grad_kernel<64, LP, ., OH, H16>: Box / one-hot observations x per-component / one-lane-per-row head x 32 / 64 padded
    logits: gradient and statistics against autograd, MFMA against the fmaf-chain restatement bit for bit, and a two-epoch
    update chain."""
    idx = np.random.default_rng(nb).permutation(T * E)[:nb]
    hp = orc.PPOHyper(ent_coef=0.01)
    g, g_ref, st, st_ref, lay = _grad_pair(name, T, E, idx, hp)
    _assert_grads(g, g_ref, lay)
    for i, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss")):
        assert abs(st[i] - st_ref[k]) <= 1e-5 + 1e-4 * abs(st_ref[k]), (k, st[i], st_ref[k])
    g1 = _grad_pair(name, T, E, idx, hp, gemm_mode=1)[0]
    assert np.array_equal(g, g1), np.abs(g - g1).max()
    hp2 = orc.PPOHyper(batch_size=nb // 2 + 3, n_epochs=2, ent_coef=0.01)
    model, orac, stats_ref = _train_pair(name, T, E, hp2)
    p, p_ref = model.policy.get_flat_params(), orac.flat_params()
    assert np.abs(p - p_ref).max() <= 2e-6 * len(stats_ref) + 1e-6, np.abs(p - p_ref).max()


def test_general_kernels_pass_the_gradient_and_forward_parity_tests_on_the_small_shapes():
    """PH_GRAD_FAST=0 / PH_FWD16=0 / PH_FWD16H=0 route the small and the one-hot shapes through the general kernels (the ones
    the wide shapes always use): the switches are read once per process, so the parity tests run in their own interpreter."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env in ({"PH_GRAD_FAST": "0", "PH_FWD16": "0", "PH_FWD16H": "0"},):
        out = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-m", "gpu", "-k",
                              "minibatch_gradient or forward_matches_oracle or mfma_and_valu or train_matches_oracle"],
                             cwd=root, env={**os.environ, **env}, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (env, out.stdout[-2000:], out.stderr[-2000:])


def test_joint_update_of_two_learners_is_bitwise_the_two_separate_updates():
    """ph_ppo_train_multi only re-orders the learners' launches relative to each other (event-chained gradient launches on the
    learners' own streams): parameters, Adam moments and statistics must equal two plain ph_ppo_train calls bit for bit --
    eagerly and replayed from the single two-stream hipGraph."""
    from pantheonrl_amd import PPO, spaces as sp
    from pantheonrl_amd.vec import (JointIterationGraph, SyntheticRollouts, VecOnPolicyAgent, run_iteration_eager,
                                    run_joint_iteration_eager)
    E, T = 64, 16
    obs_space, act_space = sp.Box(-np.inf, np.inf, (62,)), sp.Discrete(6)
    env = type("S", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()

    def make():
        agents, datas = [], []
        for seed in (3, 4):
            m = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=E * T // 4, n_epochs=3, seed=seed)
            m.device_permutations = True
            agents.append(VecOnPolicyAgent(m))
            datas.append(SyntheticRollouts(obs_space, E, T, 400, seed, m.device))
        return agents, datas

    ref_agents, ref_datas = make()
    for _ in range(3):
        for a, d in zip(ref_agents, ref_datas):
            run_iteration_eager(a, d)
    th.cuda.synchronize()
    want = [a.model.policy.get_flat_params() for a in ref_agents]

    agents, datas = make()
    streams = [th.cuda.Stream() for _ in agents]
    for _ in range(3):
        run_joint_iteration_eager(agents, datas, streams)
    th.cuda.synchronize()
    for a, w, r in zip(agents, want, ref_agents):
        assert np.array_equal(a.model.policy.get_flat_params(), w)
        assert th.equal(a.model.policy.adam_v, r.model.policy.adam_v)
        assert th.equal(a.model._stats_dev, r.model._stats_dev)

    # the same three iterations out of the captured graph (its constructor runs two warm-up iterations + the capture pass:
    # rebuild the reference with the same count)
    agents, datas = make()
    streams = [th.cuda.Stream() for _ in agents]
    joint = JointIterationGraph(agents, datas, streams)
    joint.launch()
    th.cuda.synchronize()
    ref_agents, ref_datas = make()
    for a, d in zip(ref_agents, ref_datas):
        from pantheonrl_amd.vec import IterationGraph
    graphs = [IterationGraph(a, d, th.cuda.Stream()) for a, d in zip(ref_agents, ref_datas)]
    for gph in graphs:
        gph.launch()
    th.cuda.synchronize()
    for a, r in zip(agents, ref_agents):
        assert np.isfinite(a.model.policy.get_flat_params()).all()
        assert np.array_equal(a.model.policy.get_flat_params(), r.model.policy.get_flat_params())


def test_engine_side_exchange_single_rank_and_native_rollout_loop():
    """ph_all_gather_i32 / ph_selfplay_rollout: without a communicator the gather is a device copy; with a one-rank RCCL
    communicator (PANTHEON_FORCE_RCCL=1: dlopen of librccl, ncclCommInitRank, ncclAllGather on the engine's stream) the result
    is the same; the native T-step loop leaves the same buffers and parameters as the per-step Python loop."""
    import os
    from pantheonrl_amd import PPO, spaces as sp
    from pantheonrl_amd import dist as pdist
    from pantheonrl_amd.vec import FusedSelfPlayRollout, SyntheticRollouts, VecOnPolicyAgent
    E, T = 64, 8
    obs_space, act_space = sp.Box(-np.inf, np.inf, (62,)), sp.Discrete(6)
    env = type("S", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()

    def run(native, force_rccl):
        os.environ["PANTHEON_FORCE_RCCL"] = "1" if force_rccl else "0"
        agents, datas = [], []
        for seed in (5, 6):
            m = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=E * T // 2, n_epochs=2, seed=seed)
            agents.append(VecOnPolicyAgent(m))
            datas.append(SyntheticRollouts(obs_space, E, T, 400, seed, m.device))
        ex = pdist.ActionExchange(len(agents), E, agents[0].model.device)
        if native:
            assert ex.attach_native(agents[0].model.policy.ctx)
        stream = th.cuda.Stream()
        with th.cuda.stream(stream):
            steps = FusedSelfPlayRollout(agents, datas, ex, stream)
            for it in range(2):
                steps.run_iteration(it)
        th.cuda.synchronize()
        out = [a.model.policy.get_flat_params() for a in agents] + [ex.joint.cpu().numpy().copy()]
        del steps, ex, agents
        return out

    base = run(False, False)
    for native, force in ((True, False), (True, True)):
        got = run(native, force)
        for x, y in zip(base, got):
            assert np.array_equal(x, y), (native, force)
    os.environ["PANTHEON_FORCE_RCCL"] = "0"


def test_peer_to_peer_exchange_single_rank_matches_the_copy_route():
    """ph_selfplay_rollout_p2p with one rank (receive area mapped onto itself): same buffers and parameters as the device-copy
    route, no timeouts, stamps advance with the iteration word."""
    from pantheonrl_amd import PPO, spaces as sp
    from pantheonrl_amd import dist as pdist
    from pantheonrl_amd.vec import FusedSelfPlayRollout, SyntheticRollouts, VecOnPolicyAgent
    E, T = 64, 8
    obs_space, act_space = sp.Box(-np.inf, np.inf, (62,)), sp.Discrete(6)
    env = type("S", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()

    def run(p2p):
        agents, datas = [], []
        for seed in (5, 6):
            m = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=E * T // 2, n_epochs=2, seed=seed)
            agents.append(VecOnPolicyAgent(m))
            datas.append(SyntheticRollouts(obs_space, E, T, 400, seed, m.device))
        ex = pdist.ActionExchange(len(agents), E, agents[0].model.device)
        ex.want_p2p = p2p
        stream = th.cuda.Stream()
        with th.cuda.stream(stream):
            steps = FusedSelfPlayRollout(agents, datas, ex, stream)
            assert (ex.p2p is not None) == p2p
            for it in range(3):
                steps.run_iteration(it)
        th.cuda.synchronize()
        assert ex.p2p_timeouts() == 0
        return ([a.model.policy.get_flat_params() for a in agents] +
                [a.model.rollout_buffer.host()["rewards"] for a in agents] + [ex.joint_slot(T - 1).cpu().numpy().copy()])

    for x, y in zip(run(False), run(True)):
        assert np.array_equal(x, y)


def _synthetic_masks(T, E, L, seed, device):
    """SURVEY.md 8(d), config 5 variant: action masks Bernoulli(0.8) with at least one legal action forced"""
    rng = np.random.default_rng(seed)
    m = (rng.random((T, E, L)) < 0.8).astype(np.uint8)
    dead = m.sum(-1) == 0
    m[dead, rng.integers(0, L, size=int(dead.sum()))] = 1
    return th.as_tensor(m).to(device)


@pytest.mark.parametrize("E,T,n_agents,mask_mode", [(64, 8, 2, None), (40, 5, 2, None), (64, 8, 1, None), (96, 6, 2, 2),
                                                     (96, 6, 2, 1), (1024, 128, 2, None)])
def test_persistent_exchange_rollout_is_bitwise_the_per_step_walk(E, T, n_agents, mask_mode):
    """ph_selfplay_rollout_persistent (all T steps of every local agent in ONE launch, the per-step action hand-off done
    in-kernel over the stamp-in-band words) against ph_selfplay_rollout_p2p (one launch per step) + the last step's
    ph_buffer_add_reward_joint: every array of both rollout buffers after each of three iterations (the word slots alternate
    between two halves; the pairing changes), the trained parameters, the unpacked last joint action; no poll timed out.
    mask_mode: the config-5 variant -- Bernoulli(0.8) action masks; 2 = the reference's plain PPO partner (the
    policy never sees the mask, the environment repairs an illegal sample with the first legal index, pettingzoo.py:81-82,
    and the buffer keeps the sample), 1 = ModularPolicy's logit offset as well.  The repaired actions are checked against
    numpy bit for bit, and every one of them is legal."""
    from pantheonrl_amd import PPO, spaces as sp
    from pantheonrl_amd import dist as pdist
    from pantheonrl_amd.vec import FusedSelfPlayRollout, SyntheticRollouts, VecOnPolicyAgent
    masked = mask_mode is not None
    obs_space, act_space = sp.Box(-np.inf, np.inf, (48 if masked else 62,)), sp.Discrete(5 if masked else 6)
    env = type("S", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()

    def run(persistent):
        agents, datas, masks = [], [], []
        for seed in range(5, 5 + n_agents):
            m = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=E * T // 2, n_epochs=2, seed=seed)
            agents.append(VecOnPolicyAgent(m))
            datas.append(SyntheticRollouts(obs_space, E, T, 25, seed, m.device))
            masks.append(_synthetic_masks(T, E, act_space.n, seed, m.device) if masked else None)
        ex = pdist.ActionExchange(len(agents), E, agents[0].model.device)
        ex.want_p2p = True
        stream = th.cuda.Stream()
        snaps = []
        with th.cuda.stream(stream):
            steps = FusedSelfPlayRollout(agents, datas, ex, stream, masks=masks if masked else None,
                                         mask_mode=mask_mode if masked else 2, persistent=persistent)
            assert ex.p2p is not None
            for it in range(3):
                steps.run_iteration(it)
                assert steps.last_rollout_mode == ("persistent" if persistent else "p2p")
                th.cuda.synchronize()
                snaps.append({"joint": ex.joint_slot(T - 1).cpu().numpy().copy(), "local": ex.local.cpu().numpy().copy(),
                              **{f"rb{i}_{k}": v.copy() for i, a in enumerate(agents)
                                 for k, v in a.model.rollout_buffer.host().items()}})
        assert ex.p2p_timeouts() == 0
        return snaps, [a.model.policy.get_flat_params() for a in agents], masks

    walk, w_params, masks = run(False)
    one, o_params, _ = run(True)
    for x, y in zip(walk, one):
        for k in x:
            assert np.array_equal(x[k], y[k]), k
    for x, y in zip(w_params, o_params):
        assert np.array_equal(x, y)
    if masked:
        last = one[-1]
        n_illegal = 0
        for i in range(n_agents):
            raw = last[f"rb{i}_actions"][..., 0].astype(np.int64)       # (T, E) sampled actions as the agent recorded them
            mk = masks[i].cpu().numpy()
            legal = np.take_along_axis(mk, raw[..., None], axis=-1)[..., 0] != 0
            n_illegal += int((~legal).sum())
            fixed = np.where(legal, raw, mk.argmax(-1))                 # first legal index where the sample is illegal
            env_last = last["local"][i]                                 # what the environment / the exchange got at step T-1
            assert np.array_equal(env_last, fixed[T - 1])
            assert (np.take_along_axis(mk[T - 1], env_last[:, None].astype(np.int64), axis=-1) != 0).all()
            assert np.array_equal(last["joint"][i], env_last)           # single rank: seat i of the joint action
        # an unmasked policy (mode 2) samples illegal actions about one time in five; with the -30 offset (mode 1) never
        assert (n_illegal > T * E * n_agents // 20) if mask_mode == 2 else (n_illegal == 0)


@pytest.mark.parametrize("E,T", [(64, 8), (1024, 128)])
def test_device_resident_rps_self_play_is_one_launch_and_pays_the_reference_payoff(E, T):
    """BASELINE config 1's game as a device-resident self-play (rpsgym/rps.py:41-45; multiagentenv.py:395-409): two PPO learners on
    E tables, the exchange rollout under PH_JOINT_RPS -- every round's reward is the payoff of the joint action, +1 / 0 / -1 for the
    one and its negative for the other, every round ends its episode.  The one-launch rollout (ph_selfplay_rollout_persistent) is
    bitwise the launch-per-step walk; the rewards are envs/rps.py's PAYOFF of the recorded actions, exactly."""
    from pantheonrl_amd import PPO, spaces as sp
    from pantheonrl_amd import dist as pdist
    from pantheonrl_amd.envs.rps import rps_payoff
    from pantheonrl_amd.vec import FusedSelfPlayRollout, RPSTables, VecOnPolicyAgent
    env = type("S", (), dict(observation_space=sp.Discrete(1), action_space=sp.Discrete(3), _is_dummy_space_env=True))()

    def run(persistent):
        agents, datas = [], []
        for seed in (5, 6):
            m = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=E * T // 2, n_epochs=2, seed=seed)
            agents.append(VecOnPolicyAgent(m))
            datas.append(RPSTables(E, T, m.device))
        ex = pdist.ActionExchange(len(agents), E, agents[0].model.device)
        ex.want_p2p = True
        stream = th.cuda.Stream()
        snaps = []
        with th.cuda.stream(stream):
            steps = FusedSelfPlayRollout(agents, datas, ex, stream, bonus=1.0, persistent=persistent, reward_rule="rps")
            for it in range(2):
                steps.run_iteration(it)
                assert steps.last_rollout_mode == ("persistent" if persistent else "p2p")
                th.cuda.synchronize()
                snaps.append({f"rb{i}_{k}": v.copy() for i, a in enumerate(agents) for k, v in a.model.rollout_buffer.host().items()})
        assert ex.p2p_timeouts() == 0
        return snaps, [a.model.policy.get_flat_params() for a in agents]

    walk, w_params = run(False)
    one, o_params = run(True)
    for x, y in zip(walk, one):
        for k in x:
            assert np.array_equal(x[k], y[k]), k
    for x, y in zip(w_params, o_params):
        assert np.array_equal(x, y)
    for snap in one:
        a0, a1 = snap["rb0_actions"][..., 0].astype(np.int64), snap["rb1_actions"][..., 0].astype(np.int64)
        assert set(np.unique(a0)) <= {0, 1, 2} and len(np.unique(a0)) == 3
        assert np.array_equal(snap["rb0_rewards"], rps_payoff(a0, a1).astype(np.float32))          # the ego's gain ...
        assert np.array_equal(snap["rb1_rewards"], -snap["rb0_rewards"])                             # ... and its negative
        assert np.array_equal(snap["rb0_episode_starts"], np.ones((T, E), np.float32))               # one-step episodes
    assert not np.array_equal(o_params[0], o_params[1])


@pytest.mark.parametrize("world", [2, 4])
def test_peer_to_peer_exchange_between_processes(world):
    """ranks (sharing this GPU) map each other's receive areas through HIP IPC and exchange 3 x 8 steps, then drive the fused
    rollout.  Four ranks time-slicing one GPU skew by several steps along the partner chain: with fewer stamp-in-band slots
    than steps a fast rank overwrote words a slow one had not read (polls timed out); two ranks cannot show that."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(29531 + world), os.path.join(root, "tests", "scripts", "p2p_two_rank.py")]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.count("P2P_OK") == world, (out.stdout[-1500:], out.stderr[-3000:])


def test_rccl_first_contact_with_two_ranks_on_this_box():
    """ncclCommInitRank with nranks = 2 is attempted for real (tests/scripts/rccl_two_rank.py): on a box with two GPUs the engine-side
    all-gather must verify against torch.distributed's; on a one-GPU box RCCL refuses the duplicate device and BOTH ranks must come back
    (no hang), agree that the native route is out and land on the same fallback, which then carries a real exchange.  Until round 6
    the communicator had only ever been created with one rank."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "tests", "scripts", "rccl_two_rank.py")]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, out.stdout[-1500:]
    d = json.loads(line[0])
    if d["ranks_on_device"] > 1:
        assert d["rccl_verified"] is False and d["route"] in ("p2p", "torch")
        assert "native RCCL exchange unavailable" in out.stderr            # the refusal was reported, by name
    else:
        assert d["rccl_verified"] is True and d["route"] in ("rccl", "p2p")


# ----------------------------------------------------------------------------------------------------------------
# round 2: sizes and chains the round-1 suite only property-tested, reference-semantics run, truncation bootstrap
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,T,E,nb,gemm_mode", [("overcooked", 128, 1024, 32768, 0), ("overcooked", 128, 1024, 32768, 2),
                                                    ("liar", 128, 256, 8192, 0), ("adap_oc", 128, 256, 32768, 0),
                                                    ("liar", 128, 256, 32768, 0), ("liar", 128, 256, 8192, 2),
                                                    ("liar", 128, 256, 32768, 2), ("adap_oc", 128, 256, 32768, 2)])
def test_full_size_minibatch_gradient_matches_autograd(name, T, E, nb, gemm_mode):
    """One whole minibatch of BASELINE configs 3 and 2 at their real sizes (32 768 rows of Overcooked-simple; 8 192 rows of
    Liar's Dice with F = 270 one-hot features and two action components) against autograd on the oracle.  A sum over nb
    rows in another order: tolerance 2e-4 of the largest gradient entry, as for the small shapes.  The 32 768-row cases of
    the 65-feature Box shape (Overcooked + ADAP context) and of Liar's Dice make every workgroup of the general kernel walk
    two tiles (slab read-modify-write, row metadata of the next tile staged behind the current one).  The Overcooked case runs on
    both arithmetics: the exact-f32 kernels (gemm_mode 0) and the split-bf16 kernel PPO.train() and bench.py use by default (2)."""
    idx = np.random.default_rng(nb).permutation(T * E)[:nb]
    g, g_ref, st, st_ref, lay = _grad_pair(name, T, E, idx, orc.PPOHyper(), gemm_mode=gemm_mode)
    _assert_grads(g, g_ref, lay)
    for i, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss")):
        assert abs(st[i] - st_ref[k]) <= 1e-5 + 1e-4 * abs(st_ref[k]), (k, st[i], st_ref[k])


@pytest.mark.parametrize("name,T,E", [("overcooked", 128, 1024), ("liar", 128, 256)])
def test_full_size_one_epoch_train_matches_oracle(name, T, E):
    """train() for one epoch at the bench sizes (batch = E*T/4: four dependent Adam steps over the whole buffer), teacher-forced
    permutation, against the oracle's PPO.train(): per-minibatch statistics and the parameters after the fourth step."""
    hp = orc.PPOHyper(batch_size=T * E // 4, n_epochs=1)
    model, orac, stats_ref = _train_pair(name, T, E, hp, seed=33)
    st = model.last_train_stats
    assert len(stats_ref) == st.shape[0] == 4 and int(model.policy.opt_step.item()) == 4
    p, p_ref = model.policy.get_flat_params(), orac.flat_params()
    assert np.abs(p - p_ref).max() <= 2e-6 * 4 + 1e-6, np.abs(p - p_ref).max()
    for i, s in enumerate(stats_ref):
        _assert_train_stats(st[i], s, T * E // 4, (name, i))


def test_reference_semantics_320_step_chain():
    """How the reference actually runs a partner (agents.py:111-203 on SB3's defaults): n_envs = 1, n_steps = 2048, batch 64,
    10 epochs = 320 dependent Adam steps per rollout, np.random.permutation order teacher-forced.  A priori one Adam step
    moves a parameter by at most ~lr = 3e-4 and the two normalised updates disagree by the f32 noise of the gradients, so a
    drift of up to ~1e-5 after 320 steps would be unremarkable; measured on MI355X: max |dW| = 1.2e-7 (one ulp of a weight of
    magnitude 1) while the chain moves the weights by up to 6.2e-2.  Asserted: <= 2e-6 absolute, and every minibatch's
    statistics within _assert_train_stats' per-statistic tolerances."""
    hp = orc.PPOHyper(batch_size=64, n_epochs=10)
    model, orac, stats_ref = _train_pair("overcooked", 2048, 1, hp, seed=77)
    st = model.last_train_stats
    assert st.shape[0] == len(stats_ref) == 320 and int(model.policy.opt_step.item()) == 320
    p, p_ref = model.policy.get_flat_params(), orac.flat_params()
    drift = np.abs(p - p_ref).max()
    moved = np.abs(p_ref - H.oracle_policy("overcooked", seed=77).flat_params()).max()
    print(f"320-step chain: max |dW| = {drift:.3e}, the chain itself moved the weights by up to {moved:.3e}")
    assert moved > 5e-3 and drift <= 2e-6, (drift, moved)
    for i in range(320):        # every one of the 320 minibatches, the per-statistic tolerances of the short chains
        _assert_train_stats(st[i], stats_ref[i], 64, ("320-step chain", i))


def test_models_built_with_the_same_seed_sample_independently():
    """trainer.py RPS-v0 PPO PPO --seed S: ego and partner start from the same weights (as in the reference) but must not draw
    the same uniforms step after step -- the partner's model gets sampling_stream = index + 1."""
    from pantheonrl_amd.ppo import PPO
    from pantheonrl_amd.spaces import Discrete
    env = type("E", (), dict(observation_space=Discrete(1), action_space=Discrete(3), _is_dummy_space_env=True))()
    ego = PPO("MlpPolicy", env, n_steps=8, seed=5)
    alt = PPO("MlpPolicy", env, n_steps=8, seed=5, sampling_stream=1)
    twin = PPO("MlpPolicy", env, n_steps=8, seed=5)
    assert np.array_equal(ego.policy.get_flat_params(), alt.policy.get_flat_params())
    obs = np.zeros((512, 1), np.float32)
    a_ego = ego.policy.forward(obs)[0].cpu().numpy()
    a_alt = alt.policy.forward(obs)[0].cpu().numpy()
    a_twin = twin.policy.forward(obs)[0].cpu().numpy()
    assert np.array_equal(a_ego, a_twin)                       # the bare seed still reproduces
    assert 0.5 < (a_ego != a_alt).mean() < 0.8                 # independent uniform 3-way draws differ 2/3 of the time


def test_device_games_reproduce_the_committed_traces():
    """ph_rps_step / ph_liar_step against tests/golden/liar_hand_worked.json (worked by hand from the reference's liar.py) and
    tests/golden/game_traces.npz (seeded restatement output, see make_game_traces.py) -- no Python game runs here."""
    import json
    import os
    from pantheonrl_amd import _native as nat
    from pantheonrl_amd.envs.vec import VecLiarsDice, VecRPS
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ctx, dev = nat.Context(0), th.device("cuda", 0)
    games = json.load(open(os.path.join(here, "liar_hand_worked.json")))["games"]
    G, S = len(games), max(len(g["steps"]) for g in games)
    hands = np.array([g["egohand"] + g["althand"] for g in games], np.int32)
    vec = VecLiarsDice(G, ctx, dev)
    vec.reset(hands)
    for s in range(S):
        alive = np.array([s < len(g["steps"]) for g in games])
        acts = np.array([g["steps"][s]["raw"] if a else [0, 0] for g, a in zip(games, alive)], np.int32)
        is_ego = np.array([g["steps"][s]["is_ego"] if a else False for g, a in zip(games, alive)])
        obs, rew, done = vec.player_step(_dev(acts), _dev(is_ego.astype(np.uint8)), _dev(alive.astype(np.uint8)))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        for i in np.nonzero(alive)[0]:
            want = games[i]["steps"][s]
            assert obs[i].tolist() == want["obs"] and rew[i].tolist() == want["rew"] and bool(done[i]) == want["done"], (i, s)
    z = np.load(os.path.join(here, "game_traces.npz"))
    env = VecRPS(len(z["rps_ego"]), ctx, dev)
    r0, r1, d = env.step(_dev(z["rps_ego"]), _dev(z["rps_alt"]))
    assert np.array_equal(r0.cpu().numpy(), z["rps_ego_reward"]) and np.array_equal(r1.cpu().numpy(), -z["rps_ego_reward"])
    E = z["liar_hands"].shape[0]
    vec = VecLiarsDice(E, ctx, dev)
    vec.reset(z["liar_hands"])
    turn = z["liar_ego_first"].astype(bool)
    for s in range(z["liar_acts"].shape[0]):
        alive = z["liar_alive"][s].astype(bool)
        obs, rew, done = vec.player_step(_dev(z["liar_acts"][s]), _dev(turn.astype(np.uint8)), _dev(alive.astype(np.uint8)))
        assert np.array_equal(obs.cpu().numpy()[alive], z["liar_obs"][s][alive])
        assert np.array_equal(rew.cpu().numpy()[alive], z["liar_rew"][s][alive])
        assert np.array_equal(done.cpu().numpy()[alive].astype(bool), z["liar_done"][s][alive].astype(bool))
        turn = ~turn


class _TimeLimitVec:
    """scripted 3-env VecEnv: observations are a function of (env, step) only, episodes end by time limit every `horizon`
    steps in env 0 (truncated), by a true terminal every 5 steps in env 1, never in env 2."""
    num_envs = 3

    def __init__(self, D=62, horizon=4):
        from pantheonrl_amd.spaces import Box, Discrete
        self.observation_space, self.action_space = Box(-np.inf, np.inf, (D,)), Discrete(6)
        self.D, self.horizon, self.t = D, horizon, 0

    def _obs(self, t):
        return np.stack([np.sin(0.37 * (t + 1) * (e + 1) + 0.11 * np.arange(self.D)) for e in range(3)]).astype(np.float32)

    def reset(self):
        self.t = 0
        return self._obs(0)

    def step(self, actions):
        self.t += 1
        obs = self._obs(self.t)
        rew = np.array([0.5, -0.25, 1.0], np.float32) * (1 + (self.t % 3))
        dones = np.array([self.t % self.horizon == 0, self.t % 5 == 0, False])
        infos = [{}, {}, {}]
        for e in range(3):
            if dones[e]:
                infos[e]["terminal_observation"] = obs[e].copy()
                obs[e] = 0.1 * (e + 1)                          # the auto-reset observation
        if dones[0]:
            infos[0]["TimeLimit.truncated"] = True
        return obs, rew, dones, infos


def test_collect_rollouts_bootstraps_time_limit_truncations():
    """SB3 1.7.0's ego loop: rewards[i] += gamma * V(terminal_observation) when an episode is cut by a time limit, and only
    then.  Device PPO.collect_rollouts against the oracle's restatement on a scripted VecEnv, actions teacher-forced."""
    from pantheonrl_amd.ppo import PPO
    T, E = 12, 3
    orac = H.oracle_policy("overcooked", seed=3)
    env_d, env_o = _TimeLimitVec(), _TimeLimitVec()
    model = PPO("MlpPolicy", env_d, n_steps=T, n_envs=E, batch_size=12, n_epochs=1, seed=0)
    model.policy.set_flat_params(orac.flat_params())
    model._last_obs, model._last_episode_starts = env_d.reset(), np.ones(E, np.float32)
    u = np.random.default_rng(0).random((T, E, 1)).astype(np.float32)
    model.collect_rollouts(forced_uniforms=u)
    dev = model.rollout_buffer.host()
    ob = orc.RolloutBufferOracle(T, E, 62, 1)
    orc.collect_rollouts(orac, ob, env_o, env_o.reset(), np.ones(E, np.float32), forced_actions=dev["actions"])
    plain = np.stack([np.array([0.5, -0.25, 1.0], np.float32) * (1 + (t % 3)) for t in range(1, T + 1)])
    boosted = np.abs(ob.rewards - plain) > 0
    assert boosted[:, 0].tolist() == [(t % 4 == 0) for t in range(1, T + 1)] and not boosted[:, 1:].any()
    assert np.abs(dev["rewards"] - ob.rewards).max() <= 2e-6
    assert np.array_equal(dev["episode_starts"], ob.episode_starts)
    assert np.abs(dev["values"] - ob.values).max() <= 2e-5 and np.abs(dev["advantages"] - ob.advantages).max() <= 1e-4


def test_round_robin_partners_one_agent_per_rank_replayed_through_multiagentenv():
    """BASELINE config 4 (ego vs 3 OnPolicy partners, one agent per rank; here 4 ranks share the test box's GPU over gloo):
    every environment of the device run is replayed through the Python MultiAgentEnv loop with round-robin partner
    selection and must give the ego's and all three partners' buffers row by row; then a run with short partner buffers
    checks that a partner trains once all its columns are full."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # second run: 4-row partner buffers and 5-step episodes: environments change partner at different times, so a partner's
    # columns fill at different rates -- it trains on its full columns (min_full) instead of waiting for all of them
    # fourth run: the replay through oracle/multiagent_oracle.py -- an independent restatement of the reference's step / reset control
    # flow that shares no code with the product's SimultaneousEnv; fifth: BASELINE config 4 at its written size (1 024 environments,
    # 62 features, 3 partners, 4 ranks), a 64-environment sample spread over the vector replayed through that oracle
    for extra, port in (({}, 29561), ({"RR_T_PARTNER": "4"}, 29562), ({"PH_RR_NATIVE": "0"}, 29563),
                        ({"RR_ENV_IMPL": "oracle"}, 29564),
                        ({"RR_ENV_IMPL": "oracle", "RR_E": "1024", "RR_T": "32", "RR_ITER": "2", "RR_SAMPLE": "64"}, 29565)):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", "scripts", "roundrobin_ranks.py")]
        out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env={**os.environ, **extra})
        assert out.returncode == 0 and out.stdout.count("RR_OK") == 4, (out.stdout[-1500:], out.stderr[-3000:])
        assert ("RR_REPLAY_OK" in out.stdout) == ("RR_T_PARTNER" not in extra)
        assert out.stdout.count("native=True") == (0 if "PH_RR_NATIVE" in extra else 4)


def test_learn_schedules_and_callback():
    """learning_rate / clip_range as SB3 schedules of progress_remaining (adap_learn.py:233-244) and learn(callback=...)"""
    from pantheonrl_amd.ppo import PPO
    env = _TimeLimitVec(horizon=1000)
    seen = []
    model = PPO("MlpPolicy", env, n_steps=8, n_envs=3, batch_size=12, n_epochs=1, seed=0,
                learning_rate=lambda p: 1e-3 * p, clip_range=lambda p: 0.1 + 0.1 * p)
    model.learn(total_timesteps=4 * 24, callback=lambda loc, glob: seen.append(loc["self"].num_timesteps) or True)
    assert seen == [24, 48, 72, 96]
    assert model._current_progress_remaining == 0.0
    h = model.hyper()
    assert h.learning_rate == 0.0 and abs(h.clip_range - 0.1) < 1e-7
    model._current_progress_remaining = 0.5
    assert abs(model.hyper().learning_rate - 5e-4) < 1e-9 and abs(model.hyper().clip_range - 0.15) < 1e-7
    stop = PPO("MlpPolicy", _TimeLimitVec(horizon=1000), n_steps=8, n_envs=3, batch_size=12, n_epochs=1, seed=0)
    stop.learn(total_timesteps=10 ** 6, callback=lambda loc, glob: loc["self"].num_timesteps < 48)
    assert stop.num_timesteps == 48                       # a callback returning False ends training


def test_learn_drives_an_sb3_style_callback_object_per_step():
    """SB3's BaseCallback protocol on learn(): on_rollout_start before every rollout, update_locals + on_step after every
    vectorised environment step (the in-tree witness of SB3's loop: modular/learn.py:173,195-196), a False return from
    on_step ends the rollout and the training without an update on the cut rollout."""
    from pantheonrl_amd.ppo import PPO

    class Cb:
        def __init__(self, stop_at=None):
            self.events, self.stop_at, self.n_calls, self.locals = [], stop_at, 0, None
        def init_callback(self, model): self.model = model
        def on_training_start(self, loc, glob): self.events.append("train_start")
        def on_rollout_start(self): self.events.append("rollout_start")
        def update_locals(self, loc): self.locals = loc
        def on_step(self):
            self.n_calls += 1
            assert "new_obs" in self.locals and "rewards" in self.locals and "dones" in self.locals
            return not (self.stop_at is not None and self.n_calls >= self.stop_at)
        def on_rollout_end(self): self.events.append("rollout_end")
        def on_training_end(self): self.events.append("train_end")

    cb = Cb()
    model = PPO("MlpPolicy", _TimeLimitVec(horizon=1000), n_steps=8, n_envs=3, batch_size=12, n_epochs=1, seed=0)
    model.learn(total_timesteps=2 * 24, callback=cb)
    assert cb.n_calls == 16 and cb.model is model
    assert cb.events == ["train_start", "rollout_start", "rollout_end", "rollout_start", "rollout_end", "train_end"]
    cut = Cb(stop_at=11)
    m2 = PPO("MlpPolicy", _TimeLimitVec(horizon=1000), n_steps=8, n_envs=3, batch_size=12, n_epochs=1, seed=0)
    m2.learn(total_timesteps=10 ** 6, callback=cut)
    assert cut.n_calls == 11 and m2.num_timesteps == 33 and m2._n_updates == 1      # one full rollout trained, the second cut
    assert cut.events == ["train_start", "rollout_start", "rollout_end", "rollout_start", "train_end"]


def test_ragged_partner_trains_on_its_full_columns_only():
    """RaggedVecOnPolicyAgent.min_full < E: the update runs on exactly the full columns (compacted, ph_buffer_compact_columns)
    and equals -- bitwise -- the update of a fresh model whose (T, n) buffer holds those columns; the other columns keep
    their rows and positions.  (Round-robin partner selection needs this: see DESIGN.md section 5.)"""
    from pantheonrl_amd.envs.vec import RaggedVecOnPolicyAgent
    from pantheonrl_amd.ppo import PPO
    name, T, E = "overcooked", 6, 10
    obs_s, act_s = H.CONFIGS[name]
    env = type("E", (), dict(observation_space=H.to_space(obs_s), action_space=H.to_space(act_s), _is_dummy_space_env=True))()
    orac = H.oracle_policy(name, seed=2)

    def model(n_envs):
        m = PPO("MlpPolicy", env, n_steps=T, n_envs=n_envs, batch_size=8, n_epochs=2, seed=0)
        m.policy.set_flat_params(orac.flat_params())
        m.device_permutations = True
        return m
    big = model(E)
    agent = RaggedVecOnPolicyAgent(big)
    agent.min_full = 3
    rng = np.random.default_rng(0)
    # environments 1, 4, 7 and 8 are asked to act at every step, the others at every third step
    often = np.zeros(E, bool)
    often[[1, 4, 7, 8]] = True
    step = 0
    while not agent.full():
        mask = often | (step % 3 == 0)
        agent.get_action(_dev(rng.standard_normal((E, 62)).astype(np.float32)), _dev(mask.astype(np.uint8)))
        agent.update(_dev(rng.standard_normal(E).astype(np.float32)), _dev((rng.random(E) < 0.2).astype(np.float32)),
                     _dev(mask.astype(np.uint8)))
        step += 1
    th.cuda.synchronize()
    pos = agent.pos.cpu().numpy().copy()
    cols = np.nonzero(pos >= T)[0]
    assert cols.tolist() == [1, 4, 7, 8] and pos[0] < T
    before = big.rollout_buffer.host()
    last_v, term = agent.values.cpu().numpy().copy(), agent.term.cpu().numpy().astype(np.float32)
    # the same update on a fresh (T, 4) model fed those columns
    small = model(len(cols))
    for k in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs"):
        getattr(small.rollout_buffer, k).copy_(th.as_tensor(before[k][:, cols]))
    small.rollout_buffer.pos, small.rollout_buffer.full = T, True
    small.rollout_buffer.compute_returns_and_advantage(last_v[cols], term[cols])
    small.permutation_seed = big.permutation_seed
    small.train(sync_stats=False)
    agent.learn_from_buffer()
    th.cuda.synchronize()
    assert np.array_equal(big.policy.get_flat_params(), small.policy.get_flat_params())
    assert not np.array_equal(big.policy.get_flat_params(), orac.flat_params())
    after, pos2 = big.rollout_buffer.host(), agent.pos.cpu().numpy()
    assert (pos2[cols] == 0).all() and np.array_equal(np.delete(pos2, cols), np.delete(pos, cols)) and agent.iteration == 1
    rest = np.setdiff1d(np.arange(E), cols)
    for k in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs"):
        assert np.array_equal(after[k][:, rest], before[k][:, rest]), k
    agent.min_full = E                                          # default trigger: every column
    assert not agent.full()


@pytest.mark.parametrize("rpw,skip,size", [(None, None, (40, 8, 6)), (16, None, (40, 8, 6)), (3, None, (40, 8, 6)),
                                           (1, "0", (40, 8, 6)), (16, "0", (40, 8, 6)), (None, None, (256, 128, 128))])
def test_liar_persistent_rollout_is_bitwise_the_step_by_step_walk(monkeypatch, rpw, skip, size):
    """ph_liar_selfplay_rollout (ONE launch: a workgroup owns up to 16 tables for all n_steps) against n_steps calls of
    ph_liar_selfplay_step with the same counters: identical game state, observations, both rollout buffers, the partner's
    book-keeping and -- after both learners trained on them -- identical parameters.  Default: one table per workgroup
    (40 tables on 256 CUs) and partner forwards no table of the workgroup asks for skipped; PH_LIAR_RPW = 16 leaves the last
    workgroup with 8 live tables, 3 with one; PH_LIAR_SKIP = 0 runs every forward.  The last case is BASELINE config 2 at its
    full size (LiarsDice-v0 PPO-vs-PPO, n_envs = 256, 128 steps per rollout, both learners training on 16 384-row minibatches)."""
    if rpw is not None:
        monkeypatch.setenv("PH_LIAR_RPW", str(rpw))
    if skip is not None:
        monkeypatch.setenv("PH_LIAR_SKIP", skip)
    E, T_ego, T_alt = size
    runs = []
    for persistent in (True, False):
        sp, ego, alt, _ = _liar_selfplay(E, T_ego, T_alt, seed=23)
        sp.persistent = persistent
        alt.model.rollout_buffer.gae_mode = ego.model.rollout_buffer.gae_mode = 1
        trained = 0
        for _ in range(4):
            sp.rollout_and_learn(T_ego)          # T_ego steps (one launch or 6 * T_ego), ego update, partner update when full
            trained += alt.iteration
        th.cuda.synchronize()
        be, ba = ego.model.rollout_buffer.host(), alt.model.rollout_buffer.host()
        runs.append(dict(hands=sp.env.hands.cpu().numpy(), hist=sp.env.history.cpu().numpy(), obs=sp.obs_ego.cpu().numpy(),
                         pos=alt.pos.cpu().numpy(), flags=np.stack([t.cpu().numpy() for t in (alt.boundary, alt.term, alt.open)]),
                         acted=sp.alt_acted.cpu().numpy(), episodes=sp.episodes, trained=alt.iteration, ego_it=ego.iteration,
                         pe=ego.model.policy.get_flat_params(), pa=alt.model.policy.get_flat_params(),
                         ev=ego.values.cpu().numpy(), av=alt.values.cpu().numpy(),
                         **{"e_" + k: v for k, v in be.items()}, **{"a_" + k: v for k, v in ba.items()}))
    a, b = runs
    assert a["ego_it"] == 4 and a["trained"] >= 1 and a["episodes"] > E
    pos = a["pos"]
    for key in a:
        x, y = a[key], b[key]
        if key.startswith("a_") and getattr(x, "ndim", 0) >= 2:      # only the recorded rows of the ragged buffer are defined
            rows = np.arange(x.shape[0])[:, None] < pos[None, :]
            x, y = x[rows], y[rows]
        assert np.array_equal(x, y), key


@pytest.mark.parametrize("E,T,name", [(96, 12, "overcooked"), (40, 7, "overcooked"), (1024, 128, "overcooked"),
                                      (64, 9, "rps"), (48, 10, "mpe8"), (32, 1, "overcooked"), (17, 2, "overcooked")])
def test_scripted_rollout_is_bitwise_the_per_step_walk(E, T, name):
    """ph_scripted_rollout (ONE launch: a workgroup stages the network once and walks the T steps of its 16 environments)
    against T x (get_action, update): identical rollout-buffer rows (observations, actions, values, log-probs, episode starts,
    rewards incl. the last step's), identical cached outputs of the last step, and -- after GAE and the update -- identical
    advantages and parameters, over two iterations (the second starts from the first's last dones).  E = 40 leaves the last
    workgroup with 8 live rows; (1024, 128) is the bench size; "rps" has one-hot observations."""
    runs = []
    for scripted in (False, True):
        _, model, agent, data = _vec_setup(T=T, E=E, seed=5, n_epochs=1, name=name)
        model.device_permutations = True
        snaps = []
        for it in range(2):
            agent.bind_stream()
            if scripted:
                agent.rollout_scripted(data)
            else:
                for t in range(data.T):
                    agent.get_action(data.obs[t])
                    agent.update(data.rewards[t], data.dones[t])
                agent.flush_rewards()
            th.cuda.synchronize()
            snap = {k: v.copy() for k, v in model.rollout_buffer.host().items() if k not in ("advantages", "returns")}
            snap.update(act=agent.actions.cpu().numpy(), val=agent.values.cpu().numpy(), lp=agent.log_probs.cpu().numpy(),
                        es=agent._last_episode_starts.cpu().numpy(), counter=model.policy._counter)
            agent.learn_from_buffer()
            th.cuda.synchronize()
            snap.update(adv=model.rollout_buffer.advantages.cpu().numpy(), params=model.policy.get_flat_params())
            snaps.append(snap)
        runs.append(snaps)
    for a, b in zip(*runs):
        for key in a:
            assert np.array_equal(a[key], b[key]), key
    assert not np.array_equal(runs[0][0]["params"], runs[0][1]["params"])


def test_scripted_rollout_general_form_is_still_bitwise_the_per_step_walk():
    """The one-launch rollout takes its LEAN form by default (csrc/ph_policy.hip: fwd_args_lean -- the optional fields of the argument
    record pinned to constants); the general form of the same kernel, which masks / debug stamps select, is checked by the same
    bitwise test in a process that switches the lean form off."""
    import subprocess, sys
    env = dict(os.environ, PH_ROLLOUT_LEAN="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_scripted_rollout_is_bitwise_the_per_step_walk and (96-12 or 40-7 or 64-9)"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "3 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_scripted_rollout_refuses_other_shapes_and_a_used_buffer():
    from pantheonrl_amd import _native as nat
    _, model, agent, data = _vec_setup(T=8, E=32, seed=1)
    agent.bind_stream()
    agent.get_action(data.obs[0])
    with pytest.raises(nat.NativeError):
        agent.rollout_scripted(data)                        # buffer not empty
    from pantheonrl_amd import PPO
    from pantheonrl_amd.vec import SyntheticRollouts, VecOnPolicyAgent
    obs_s, act_s = H.CONFIGS["liar"]
    env = type("E", (), dict(observation_space=H.to_space(obs_s), action_space=H.to_space(act_s), _is_dummy_space_env=True))()
    liar = PPO("MlpPolicy", env, n_steps=8, n_envs=32, batch_size=64, n_epochs=1, seed=0)
    ag = VecOnPolicyAgent(liar)
    d = SyntheticRollouts(env.observation_space, 32, 8, 6, 0, liar.device)
    ag.bind_stream()
    with pytest.raises(nat.NativeError, match="16-row forward"):
        ag.rollout_scripted(d)                              # one-hot observations / two action components: no silent slow path
