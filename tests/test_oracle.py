"""not-gpu: the oracle against the closed-form known answers, against the properties SB3's
algorithm must have, and against the committed golden vectors (tests/golden, made by make_golden.py)."""
import os

import numpy as np
import pytest
import torch as th

from oracle import sb3_oracle as orc
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_gae_appendix_c_known_answers():
    r = np.array([1, 0, 2, -1], np.float32)[:, None]
    v = np.array([.5, .4, .3, .2], np.float32)[:, None]
    s = np.array([1, 0, 1, 0], np.float32)[:, None]
    cases = {  # (last_values, dones) -> (advantages, returns)
        (0.1, 0): ([0.5198, -0.4, 0.86250937, -1.1010001], [1.0198, 0.0, 1.1625094, -0.9010001]),   # C-A  SB3 ego
        (0.2, 0): ([0.5198, -0.4, 0.955619, -1.002], [1.0198, 0.0, 1.255619, -0.802]),              # C-B  quirk D-1
        (0.2, 1): ([0.5198, -0.4, 0.7693999, -1.2], [1.0198, 0.0, 1.0693998, -1.0]),                # C-C  terminal
    }
    for (lv, dn), (adv, ret) in cases.items():
        a, R = orc.gae_reference(r, v, s, [lv], [dn])
        np.testing.assert_allclose(a.ravel(), adv, rtol=0, atol=1e-6)
        np.testing.assert_allclose(R.ravel(), ret, rtol=0, atol=1e-6)
    # C-D: gamma = lambda = 1, V = 0, r = 1 -> reward-to-go
    a, R = orc.gae_reference(np.ones((5, 1), np.float32), np.zeros((5, 1), np.float32), np.zeros((5, 1), np.float32),
                             [0], [0], 1.0, 1.0)
    assert a.ravel().tolist() == [5, 4, 3, 2, 1] and np.array_equal(a, R)


def test_gae_python_bool_dones_like_onpolicyagent():
    """OnPolicyAgent passes dones=self._last_episode_starts[0], a Python bool (agents.py:129)."""
    rng = np.random.default_rng(0)
    r, v = rng.standard_normal((9, 1)).astype(np.float32), rng.standard_normal((9, 1)).astype(np.float32)
    s = np.zeros((9, 1), np.float32)
    for flag in (True, False):
        buf = orc.RolloutBufferOracle(9, 1, 1, 1)
        buf.rewards[:], buf.values[:], buf.episode_starts[:] = r, v, s
        buf.compute_returns_and_advantage(th.tensor([[0.3]]), flag)
        a_ref, _ = orc.gae_reference(r, v, s, [0.3], [float(flag)])
        assert np.array_equal(buf.advantages, a_ref)


def test_gae_closed_forms():
    g, lam, T = 0.99, 0.95, 50
    a, _ = orc.gae_reference(np.ones((T, 2), np.float32), np.zeros((T, 2), np.float32), np.zeros((T, 2), np.float32),
                             [0, 0], [0, 0], g, lam)
    k = np.arange(T, 0, -1)
    np.testing.assert_allclose(a[:, 0], (1 - (g * lam) ** k) / (1 - g * lam), rtol=1e-5)
    a64, _ = orc.gae_float64(np.ones((T, 2)), np.zeros((T, 2)), np.zeros((T, 2)), [0, 0], [0, 0], g, lam)
    np.testing.assert_allclose(a, a64, rtol=1e-5)


def test_rollout_buffer_semantics():
    buf = orc.RolloutBufferOracle(3, 2, 4, 1)
    obs = np.arange(8, dtype=np.float32).reshape(2, 4)
    buf.add(obs, np.array([[1], [2]]), [0, 0], [1, 1], th.tensor([[.5], [.25]]), th.tensor([-1., -2.]))
    obs[:] = -1  # add() copied its inputs
    assert buf.observations[0, 1, 3] == 7 and buf.pos == 1 and not buf.full
    buf.rewards[buf.pos - 1][0] += 2.5   # Agent.update (agents.py:198)
    buf.rewards[buf.pos - 1][0] += 1.0
    assert buf.rewards[0, 0] == 3.5 and buf.rewards[0, 1] == 0
    for _ in range(2):
        buf.add(obs, np.array([[0], [0]]), [0, 0], [0, 0], th.zeros(2, 1), th.zeros(2))
    assert buf.full
    flat = buf.flat()   # env-major: row e*T + t
    assert flat["observations"].shape == (6, 4) and flat["observations"][3, 3] == 7  # e=1, t=0
    seen = np.concatenate([mb["old_values"].numpy() for mb in buf.get(4, np.arange(6))])
    assert seen.shape == (6,)
    sizes = [len(mb["returns"]) for mb in buf.get(4, np.arange(6))]
    assert sizes == [4, 2]   # last minibatch may be short
    buf.reset()
    assert buf.pos == 0 and not buf.full and buf.rewards.sum() == 0


@pytest.mark.parametrize("name", list(H.CONFIGS))
def test_policy_shapes_param_counts_and_roundtrip(name):
    obs_s, act_s = H.CONFIGS[name]
    pol = H.oracle_policy(name, seed=1)
    F, L = obs_s.flat_len, act_s.flat_len
    expected = 2 * (F * 64 + 64 + 64 * 64 + 64) + 64 * L + L + 65
    if act_s.kind == "box":
        expected += act_s.dim          # DiagGaussian: log_std[A] beside the A-output action_net
    assert sum(p.numel() for p in pol.parameters()) == expected == len(pol.flat_params())
    if name == "overcooked":
        assert expected == 16839
    if name == "liar":
        assert expected == 44308
    other = H.oracle_policy(name, seed=2)
    other.load_flat_params(pol.flat_params())
    obs = th.as_tensor(H.sample_obs(obs_s, 5, np.random.default_rng(0)))
    with th.no_grad():
        assert th.equal(other.logits(obs), pol.logits(obs))
        a, v, lp = pol.forward(obs, uniforms=th.rand(5, act_s.stored_len))
        v2, lp2, ent = pol.evaluate_actions(obs, a)
    assert a.shape == (5, act_s.stored_len) and v.shape == (5, 1) and lp.shape == (5,)
    assert th.allclose(lp, lp2) and th.equal(v, v2) and (act_s.kind == "box" or (ent > 0).all())


def test_orthogonal_init_gains():
    th.manual_seed(0)
    pol = orc.MlpPolicyOracle(*H.CONFIGS["overcooked"])
    w = pol.policy_net[2].weight.detach()             # 64x64 orthogonal * sqrt(2)
    assert th.allclose(w @ w.t(), 2 * th.eye(64), atol=1e-4)
    wa = pol.action_net.weight.detach()               # 6x64, gain 0.01
    assert th.allclose(wa @ wa.t(), 1e-4 * th.eye(6), atol=1e-7)
    assert all(float(m.bias.detach().abs().max()) == 0 for m in pol.modules() if isinstance(m, th.nn.Linear))
    assert pol.optimizer.defaults["eps"] == 1e-5 and pol.optimizer.defaults["betas"] == (0.9, 0.999)


def test_ppo_first_minibatch_closed_forms():
    """first minibatch of the first epoch: policy unchanged => ratio == 1, clip_fraction 0, approx_kl 0, policy loss
    -mean(A_hat) == 0."""
    name = "mpe8"
    pol = H.oracle_policy(name, seed=3)
    buf = H.filled_oracle_buffer(name, pol, 16, 4, seed=3)
    hp = orc.PPOHyper(batch_size=32, n_epochs=1)
    stats = orc.ppo_train(pol, buf, hp, [np.arange(64)])
    assert stats[0]["clip_fraction"] == 0.0 and abs(stats[0]["approx_kl"]) < 1e-7
    assert abs(stats[0]["policy_loss"]) < 1e-6
    assert stats[0]["grad_norm"] > 0 and len(stats) == 2


def test_ppo_matches_hand_written_gradient():
    """the autograd oracle equals the closed-form gradient the kernels implement (min/clamp tie rule included)."""
    name = "overcooked"
    pol = H.oracle_policy(name, seed=4)
    buf = H.filled_oracle_buffer(name, pol, 8, 4, seed=4)
    hp = orc.PPOHyper(batch_size=32, ent_coef=0.02, clip_range=0.15)
    # move the policy off the behaviour policy so some ratios leave the clip range
    with th.no_grad():
        pol.action_net.weight.add_(0.5 * th.randn(pol.action_net.weight.shape, generator=th.Generator().manual_seed(0)))
    flat = buf.flat()
    mb = {k: th.as_tensor(v) for k, v in flat.items()}
    z = pol.logits(mb["observations"]).detach().requires_grad_(True)
    dist = th.distributions.Categorical(logits=z)
    acts = mb["actions"].long().flatten()
    logp, ent = dist.log_prob(acts), dist.entropy()
    adv = mb["advantages"]
    adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    ratio = th.exp(logp - mb["old_log_prob"])
    loss = -th.min(adv * ratio, adv * th.clamp(ratio, 1 - hp.clip_range, 1 + hp.clip_range)).mean() \
        - hp.ent_coef * ent.mean()
    loss.backward()
    with th.no_grad():
        B = len(adv)
        p = th.softmax(z, 1)
        lo, hi = 1 - hp.clip_range, 1 + hp.clip_range
        pl1, pl2 = adv * ratio, adv * th.clamp(ratio, lo, hi)
        inr = ((ratio >= lo) & (ratio <= hi)).float()
        gate = th.where(pl1 < pl2, th.ones_like(inr), th.where(pl1 > pl2, inr, 0.5 + 0.5 * inr))
        g_lp = -(adv * ratio * gate) / B
        onehot = th.nn.functional.one_hot(acts, z.shape[1]).float()
        lpa = th.log_softmax(z, 1)
        H_ = -(p * lpa).sum(1, keepdim=True)
        dz = g_lp[:, None] * (onehot - p) + (-hp.ent_coef / B) * (-p * (lpa + H_))
    assert ((ratio < lo) | (ratio > hi)).any(), "test must exercise clipped rows"
    assert th.allclose(z.grad, dz, atol=1e-8, rtol=1e-5)


def test_inverse_cdf_and_illegal_fixup():
    probs = th.tensor([[0.2, 0.3, 0.5], [1.0, 0.0, 0.0]])
    assert orc.inverse_cdf_sample(probs, th.tensor([0.1, 0.99])).tolist() == [0, 0]
    assert orc.inverse_cdf_sample(probs, th.tensor([0.2, 0.0])).tolist() == [1, 0]
    assert orc.inverse_cdf_sample(probs, th.tensor([0.75, 0.5])).tolist() == [2, 0]
    acts = orc.fix_illegal_actions(np.array([0, 2, 1]), np.array([[0, 1, 1], [1, 0, 1], [0, 0, 1]]))
    assert acts.tolist() == [1, 2, 2]   # first legal index (pettingzoo.py:81-82)


# ---- golden vectors --------------------------------------------------------------------------------------------------
def test_oracle_reproduces_golden_gae():
    g = np.load(os.path.join(GOLD, "gae.npz"))
    for i in range(int(g["n_cases"])):
        a, ret = orc.gae_reference(g[f"c{i}_r"], g[f"c{i}_v"], g[f"c{i}_s"], g[f"c{i}_lv"], g[f"c{i}_dn"])
        assert np.array_equal(a, g[f"c{i}_adv"]) and np.array_equal(ret, g[f"c{i}_ret"])


def test_oracle_reproduces_golden_forward():
    g = np.load(os.path.join(GOLD, "forward.npz"))
    for name in ("rps", "liar", "overcooked", "mpe8"):
        pol = orc.MlpPolicyOracle(*H.CONFIGS[name])
        pol.load_flat_params(g[f"{name}_params"])
        obs = th.as_tensor(g[f"{name}_obs"])
        with th.no_grad():
            z = pol.logits(obs).numpy()
            a, v, lp = pol.forward(obs, uniforms=th.as_tensor(g[f"{name}_u"]))
        np.testing.assert_allclose(z, g[f"{name}_logits"], atol=2e-6)
        np.testing.assert_allclose(v.numpy().ravel(), g[f"{name}_values"], atol=2e-6)
        assert np.array_equal(a.numpy(), g[f"{name}_actions"])
        np.testing.assert_allclose(lp.numpy(), g[f"{name}_logp"], atol=2e-6)


def test_oracle_reproduces_golden_ppo_step():
    g = np.load(os.path.join(GOLD, "ppo_step.npz"))
    pol = orc.MlpPolicyOracle(*H.CONFIGS["overcooked"])
    pol.load_flat_params(g["params0"])
    buf = orc.RolloutBufferOracle(16, 4, 62, 1)
    for k in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns"):
        getattr(buf, k)[...] = g["rb_" + k]
    buf.full = True
    stats = orc.ppo_train(pol, buf, orc.PPOHyper(batch_size=24, n_epochs=2), g["perms"])
    assert len(stats) == 6
    got = np.array([[s[k] for k in ("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss",
                                     "grad_norm")] for s in stats], np.float32)
    np.testing.assert_allclose(got, g["stats"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(pol.flat_params(), g["params1"], atol=1e-6)


# ----------------------------------------------------------------------------------------------------------------
# ADAP's context term (adap/util.py:97-131)
# ----------------------------------------------------------------------------------------------------------------
def test_adap_context_loss_known_answers():
    """identical contexts: every KL is 0, the term is exactly 1 and pulls on nothing; two states, two contexts, one
    2-way head worked by hand from the definition  mean_s exp(-sum_a p_a log(p_a / q_a))"""
    pol = H.oracle_policy("adap_small", seed=2)
    obs = th.as_tensor(np.random.default_rng(0).standard_normal((12, 38)).astype(np.float32))
    same = np.tile(np.array([[0.6, -0.8, 0.0]], np.float32), (4, 1))
    pol.optimizer.zero_grad()
    cl = orc.adap_context_loss(pol, obs, 3, np.arange(6), same)
    cl.backward()
    assert cl.item() == 1.0 and all(float(p.grad.abs().max()) < 1e-7 for p in pol.policy_net.parameters())
    # hand-worked: a policy whose logits are a linear read-out of the context only
    spec_o, spec_a = orc.SpaceSpec("box", dim=1 + 2), orc.SpaceSpec("discrete", nvec=(2,))
    tiny = orc.MlpPolicyOracle(spec_o, spec_a)
    c = np.array([[0.3, -0.2], [-0.5, 0.4]], np.float32)
    o = th.as_tensor(np.array([[0.1, 9, 9], [0.7, 9, 9], [-0.4, 9, 9]], np.float32))   # stored contexts are ignored
    val = orc.adap_context_loss(tiny, o, 2, np.array([2, 0]), c).item()
    want = []
    for s in (2, 0):
        z = [tiny.logits(th.as_tensor(np.r_[o[s, :1].numpy(), ci][None])).detach().numpy()[0].astype(np.float64) for ci in c]
        p, q = np.exp(z[0]) / np.exp(z[0]).sum(), np.exp(z[1]) / np.exp(z[1]).sum()
        want.append(np.exp(-np.sum(p * np.log(p / q))))
    assert abs(val - np.mean(want)) < 1e-6


def test_adap_context_term_matches_the_hand_written_gradient():
    """autograd through torch's kl_divergence equals the closed form the context kernel implements:
    d KL(i||j) / d z_i = p_i ((lp_i - lp_j) - KL),  d KL(i||j) / d z_j = p_j - p_i,  per action component"""
    rng = np.random.default_rng(5)
    nvec, C, S = (3, 4), 4, 6
    L = sum(nvec)
    z = th.as_tensor(rng.standard_normal((C, S, L)), dtype=th.float64).requires_grad_(True)
    from itertools import combinations
    cls = []
    for i, j in combinations(range(C), 2):
        kl = 0
        for zi, zj in zip(th.split(z[i], list(nvec), dim=1), th.split(z[j], list(nvec), dim=1)):
            kl = kl + th.distributions.kl.kl_divergence(th.distributions.Categorical(logits=zi),
                                                        th.distributions.Categorical(logits=zj))
        cls.append(th.mean(th.exp(-kl)))
    loss = sum(cls) / len(cls)
    loss.backward()
    zn = z.detach().numpy()
    lp = np.concatenate([zc - np.log(np.exp(zc).sum(-1, keepdims=True)) for zc in np.split(zn, np.cumsum(nvec)[:-1], axis=-1)], -1)
    p = np.exp(lp)
    seg = np.repeat(np.arange(len(nvec)), nvec)
    dz = np.zeros_like(zn)
    w = 1.0 / (C * (C - 1) / 2 * S)
    for i, j in combinations(range(C), 2):
        klc = np.stack([(p[i] * (lp[i] - lp[j]))[:, seg == g].sum(1) for g in range(len(nvec))], 1)   # (S, components)
        T = np.exp(-klc.sum(1))[:, None]
        dz[i] -= w * T * p[i] * ((lp[i] - lp[j]) - klc[:, seg])
        dz[j] -= w * T * (p[j] - p[i])
    assert np.abs(z.grad.numpy() - dz).max() < 1e-12


def test_adap_context_samplers_have_the_reference_shapes():
    u = np.random.default_rng(1).random((64, 3))
    l2 = orc.adap_sample_contexts("l2", 3, 64, u)
    assert np.abs(np.linalg.norm(l2, axis=1) - 1).max() < 1e-6 and np.allclose(l2 * np.linalg.norm(u * 2 - 1, axis=1)[:, None], u * 2 - 1, atol=1e-6)
    assert np.allclose(orc.adap_sample_contexts("unit_square", 3, 64, u), u * 2 - 1)
    assert np.allclose(orc.adap_sample_contexts("positive_square", 3, 64, u), u)
    cat = orc.adap_sample_contexts("categorical", 3, 64, u)
    assert np.array_equal(cat.sum(1), np.ones(64)) and np.array_equal(cat.argmax(1), np.floor(u[:, 0] * 3))
    from pantheonrl_amd.adap import SAMPLERS
    rng = np.random.default_rng(0)
    for name, fn in SAMPLERS.items():
        c = fn(3, 16, rng)
        # util.py:80-89: "natural_numbers" is (num, 1) integers in [0, ctx_size) whatever ctx_size is
        assert c.shape == ((16, 1) if name == "natural_numbers" else (16, 3)) and c.dtype == np.float32
    nat_num = orc.adap_sample_contexts("natural_numbers", 3, 64, u)
    assert nat_num.shape == (64, 1) and np.array_equal(nat_num[:, 0], np.floor(u[:, 0] * 3))
    assert np.abs(np.linalg.norm(SAMPLERS["l2"](3, 16, rng), axis=1) - 1).max() < 1e-6


# ----------------------------------------------------------------------------------------------------------------
# ModularPolicy / ModularAlgorithm (pantheonrl/algos/modular) -- oracle only: the engine has no device path for it yet
# ----------------------------------------------------------------------------------------------------------------
def _modular(num_partners=2, seed=3, **kw):
    th.manual_seed(seed)
    return orc.ModularPolicyOracle(orc.SpaceSpec("box", dim=6), orc.SpaceSpec("discrete", nvec=(5,)),
                                   num_partners=num_partners, **kw)


def _filled_buffer(pol, partner, T=8, E=4, seed=0):
    rng = np.random.default_rng(seed)
    buf = orc.RolloutBufferOracle(T, E, 6, 1)
    es = np.ones(E, np.float32)
    for t in range(T):
        obs = rng.standard_normal((E, 6)).astype(np.float32)
        with th.no_grad():
            a, v, lp = pol.forward(th.as_tensor(obs), partner_idx=partner,
                                   uniforms=th.as_tensor(rng.random((E, 1)).astype(np.float32)))
        buf.add(obs, a.numpy().astype(np.float32), rng.standard_normal(E).astype(np.float32), es, v.flatten(), lp)
        es = (rng.random(E) < 0.2).astype(np.float32)
    buf.compute_returns_and_advantage(np.zeros(E, np.float32), es)
    return buf


def test_modular_policy_reduces_to_the_main_policy_when_the_partner_heads_are_silent():
    """logits = main + partner, value = main + partner (modular/policies.py:286,325-328): with a partner's two heads zeroed
    the network IS the main MlpPolicy -- same values / log-probs / entropy, marginal regulariser exactly 0 -- and the
    partner towers still read the main policy latent (policies.py:254,281), not the features."""
    pol = _modular()
    plain = orc.MlpPolicyOracle(pol.obs_space, pol.act_space)
    plain.load_flat_params(pol.flat_params()[:plain.flat_params().size])
    with th.no_grad():
        for pm in pol.partners:
            for head in (pm["act"], pm["val"]):
                head.weight.zero_()
                head.bias.zero_()
    rng = np.random.default_rng(1)
    obs = th.as_tensor(rng.standard_normal((9, 6)).astype(np.float32))
    acts = th.as_tensor(rng.integers(0, 5, size=(9, 1)))
    want = plain.evaluate_actions(obs, acts)
    for k in range(2):
        got = pol.evaluate_actions(obs, acts, partner_idx=k)
        for g, w in zip(got, want):
            np.testing.assert_allclose(g.detach().numpy(), w.detach().numpy(), atol=1e-6)
    assert orc.modular_marginal_regularization(pol, obs).item() == 0.0
    # the partner towers' input is the 64-wide policy latent whatever the observation width is
    assert pol.partners[0]["pi"][0].in_features == orc.HIDDEN and pol.partners[0]["vf"][0].in_features == orc.HIDDEN


def test_modular_marginal_regularization_known_answer():
    """modular/learn.py:309-316 by hand: main logits 0 (uniform over 4 actions), two partners whose logits are constants d1, d2
    -> mean_rows sum_a | 1/4 - (softmax(d1) + softmax(d2)) / 2 |"""
    th.manual_seed(0)
    pol = orc.ModularPolicyOracle(orc.SpaceSpec("box", dim=3), orc.SpaceSpec("discrete", nvec=(4,)), num_partners=2)
    d = [np.array([0.5, -0.2, 0.1, 0.0], np.float32), np.array([-1.0, 0.3, 0.0, 0.7], np.float32)]
    with th.no_grad():
        pol.action_net.weight.zero_()
        pol.action_net.bias.zero_()
        for pm, dk in zip(pol.partners, d):
            pm["act"].weight.zero_()
            pm["act"].bias.copy_(th.as_tensor(dk))
    obs = th.as_tensor(np.random.default_rng(0).standard_normal((5, 3)).astype(np.float32))
    sm = [np.exp(x.astype(np.float64)) / np.exp(x.astype(np.float64)).sum() for x in d]
    want = np.abs(0.25 - (sm[0] + sm[1]) / 2).sum()
    assert abs(orc.modular_marginal_regularization(pol, obs).item() - want) < 1e-6


def test_modular_train_first_step_is_ppo_on_the_main_network_and_reaches_it_through_the_partner():
    """(i) silent partner heads + marginal_reg_coef 0 + no gradient clipping: the first optimiser step moves the main network
    exactly as PPO.train moves the plain MlpPolicy on the same minibatch (Adam is per-parameter) and leaves the partner TOWERS where they were
    (their gradient is head^T dlogits = 0) while the partner heads move.  (ii) with live partner heads the main policy
    trunk receives gradient through the partner module: d(main pi trunk) differs from the plain policy's."""
    pol = _modular(num_partners=1)
    plain = orc.MlpPolicyOracle(pol.obs_space, pol.act_space)
    n_main = plain.flat_params().size
    plain.load_flat_params(pol.flat_params()[:n_main])
    with th.no_grad():
        for head in (pol.partners[0]["act"], pol.partners[0]["val"]):
            head.weight.zero_()
            head.bias.zero_()
    buf = _filled_buffer(pol, 0)
    # no clipping: clip_grad_norm_ runs over ALL parameters (learn.py:324), so the moving partner heads would rescale the main
    # network's gradient through the shared norm
    hp = orc.PPOHyper(n_epochs=1, batch_size=32, max_grad_norm=1e9)
    perm = [np.random.default_rng(5).permutation(32)]
    before = pol.flat_params()
    st_m = orc.modular_train(pol, [buf], hp, 0.0, perms=[perm])
    st_p = orc.ppo_train(plain, buf, hp, perms=perm)
    assert len(st_m) == 1 and abs(st_m[0]["loss"] - st_p[0]["loss"]) < 1e-6 and st_m[0]["marginal_reg"] == 0.0
    after = pol.flat_params()
    np.testing.assert_allclose(after[:n_main], plain.flat_params(), atol=1e-7)
    towers = 2 * (2 * (64 * 64 + 64))
    assert np.array_equal(after[n_main:n_main + towers], before[n_main:n_main + towers])
    assert not np.array_equal(after[n_main + towers:], before[n_main + towers:])
    # (ii)
    pol2 = _modular(num_partners=1, seed=4)
    plain2 = orc.MlpPolicyOracle(pol2.obs_space, pol2.act_space)
    plain2.load_flat_params(pol2.flat_params()[:n_main])
    buf2 = _filled_buffer(pol2, 0, seed=2)
    mb = next(iter(buf2.get(32, np.arange(32))))
    pol2.optimizer.zero_grad()
    orc.modular_minibatch_loss(pol2, mb, hp, 0, 0.5)[0].backward()
    g = pol2.policy_net[0].weight.grad
    assert g is not None and float(g.abs().max()) > 0
    assert float(pol2.partners[0]["vf"][0].weight.grad.abs().max()) > 0     # value loss reaches the partner's vf tower
    assert float(pol2.value_net_mlp[0].weight.grad.abs().max()) > 0


def test_modular_train_walks_partner_by_partner_with_one_buffer_each():
    """modular/learn.py:237-243: partner 0's epochs over buffer 0, then partner 1's over buffer 1; a partner's minibatches never
    touch the other partner's module unless the marginal regulariser (which evaluates EVERY partner, :307) is on."""
    pol = _modular(num_partners=2, seed=6)
    bufs = [_filled_buffer(pol, k, seed=10 + k) for k in range(2)]
    hp = orc.PPOHyper(n_epochs=2, batch_size=16)
    n_main = orc.MlpPolicyOracle(pol.obs_space, pol.act_space).flat_params().size
    per = (pol.flat_params().size - n_main) // 2
    before = pol.flat_params()
    stats = orc.modular_train(pol, bufs[:1], hp, 0.0)
    assert [s["partner"] for s in stats] == [0] * 4
    after = pol.flat_params()
    assert np.array_equal(after[n_main + per:], before[n_main + per:]) and not np.array_equal(after[n_main:n_main + per],
                                                                                              before[n_main:n_main + per])
    stats = orc.modular_train(pol, bufs, hp, 0.3)
    assert [s["partner"] for s in stats] == [0] * 4 + [1] * 4 and all(s["marginal_reg"] > 0 for s in stats)
    # with the regulariser on, partner 1's module moves during partner 0's minibatches too
    pol3 = _modular(num_partners=2, seed=6)
    b3 = pol3.flat_params()
    orc.modular_train(pol3, bufs[:1], hp, 0.3)
    assert not np.array_equal(pol3.flat_params()[n_main + per:], b3[n_main + per:])


# ----------------------------------------------------------------------------------------------------------------
# AdapPolicyMult (adap/policies.py:136-283): the restatement the device path (csrc/ph_adapmult.hip) is tested against
# ----------------------------------------------------------------------------------------------------------------
def _mult(F=7, C=3, L=4, seed=2):
    th.manual_seed(seed)
    return orc.AdapMultPolicyOracle(orc.SpaceSpec("box", dim=F + C), orc.SpaceSpec("discrete", nvec=(L,)), context_size=C)


def test_adap_mult_forward_is_the_written_out_arithmetic():
    """x = tanh(W1 o + b1); x_a = tanh(Ws x + bs) viewed (64, C) ROW-major (output row j * C + c pairs hidden unit j with context
    component c: policies.py:245 `.view(batch, hidden, context)`); latent = tanh(W2 (x + x_a ctx) + b2) -- in float64 numpy"""
    F, C, L = 7, 3, 4
    pol = _mult(F, C, L)
    with th.no_grad():                      # biases off zero so that a dropped bias would show
        for prm in pol.parameters():
            if prm.ndim == 1:
                prm.add_(0.1 * th.randn_like(prm))
    rng = np.random.default_rng(0)
    obs = rng.standard_normal((9, F + C)).astype(np.float32)
    o, ctx = obs[:, :F].astype(np.float64), obs[:, F:].astype(np.float64)

    def lin(seq, x):
        return x @ seq[0].weight.detach().numpy().astype(np.float64).T + seq[0].bias.detach().numpy().astype(np.float64)

    def branch(b1, sc, b2):
        x = np.tanh(lin(b1, o))
        s = np.tanh(lin(sc, x))                                             # (n, 64 * C)
        xa = np.stack([sum(s[:, j * C + c] * ctx[:, c] for c in range(C)) for j in range(64)], axis=1)
        return np.tanh(lin(b2, x + xa))
    lat_pi = branch(pol.agent_branch_1, pol.agent_scaling, pol.agent_branch_2)
    lat_vf = branch(pol.value_branch_1, pol.value_scaling, pol.value_branch_2)
    got_pi, got_vf = pol._latents(th.as_tensor(obs))
    assert np.abs(got_pi.detach().numpy() - lat_pi).max() < 2e-6 and np.abs(got_vf.detach().numpy() - lat_vf).max() < 2e-6
    z = lat_pi @ pol.action_net.weight.detach().numpy().astype(np.float64).T + pol.action_net.bias.detach().numpy()
    assert np.abs(pol.logits(th.as_tensor(obs)).detach().numpy() - z).max() < 2e-6
    v = lat_vf @ pol.value_net.weight.detach().numpy().astype(np.float64).T + pol.value_net.bias.detach().numpy()
    assert np.abs(pol.predict_values(th.as_tensor(obs)).detach().numpy() - v).max() < 2e-6


def test_adap_mult_known_answers():
    F, C, L = 7, 3, 4
    pol = _mult(F, C, L)
    obs = th.as_tensor(np.random.default_rng(1).standard_normal((6, F + C)).astype(np.float32))
    # (1) scaling layers at zero: x_a = tanh(0) = 0 and the network is the plain 64-64 MLP on the observation WITHOUT its context
    plain = orc.MlpPolicyOracle(orc.SpaceSpec("box", dim=F), orc.SpaceSpec("discrete", nvec=(L,)))
    with th.no_grad():
        for sc in (pol.agent_scaling, pol.value_scaling):
            sc[0].weight.zero_()
            sc[0].bias.zero_()
        for dst, src in ((plain.policy_net[0], pol.agent_branch_1[0]), (plain.policy_net[2], pol.agent_branch_2[0]),
                         (plain.value_net_mlp[0], pol.value_branch_1[0]), (plain.value_net_mlp[2], pol.value_branch_2[0]),
                         (plain.action_net, pol.action_net), (plain.value_net, pol.value_net)):
            dst.weight.copy_(src.weight)
            dst.bias.copy_(src.bias)
    assert th.equal(pol.logits(obs), plain.logits(obs[:, :F])) and th.equal(pol.predict_values(obs), plain.predict_values(obs[:, :F]))
    # (2) zero context: the scaling output is multiplied away whatever its weights are
    pol2 = _mult(F, C, L, seed=5)
    obs0 = obs.clone()
    obs0[:, F:] = 0
    x = pol2.agent_branch_1(obs0[:, :F])
    assert th.allclose(pol2._latents(obs0)[0], pol2.agent_branch_2(x), atol=0, rtol=0)
    # (3) the pre-activation of branch_2 is LINEAR in the context: latent(c1 + c2) - latent(c1) - latent(c2) + latent(0) = 0 there
    def pre(c):
        o = obs[:, :F]
        xx = pol2.agent_branch_1(o)
        xa = pol2.agent_scaling(xx).view(o.shape[0], 64, C)
        return pol2.agent_branch_2[0](xx + th.matmul(xa, c.unsqueeze(-1)).squeeze(-1))
    c1, c2 = th.randn(6, C), th.randn(6, C)
    assert th.allclose(pre(c1 + c2) - pre(c1) - pre(c2) + pre(th.zeros(6, C)), th.zeros(6, 64), atol=2e-5)
    # (4) one hot context e_c picks column c of the (64, C) view: output rows j * C + c of the scaling layer
    e1 = th.zeros(6, C)
    e1[:, 1] = 1
    o = obs[:, :F]
    xx = pol2.agent_branch_1(o)
    want = pol2.agent_branch_2(xx + pol2.agent_scaling(xx)[:, 1::C])
    assert th.allclose(pol2._latents(th.cat([o, e1], dim=1))[0], want, atol=1e-6)
    # (5) orthogonal init reaches the scaling layers (gain sqrt 2: rows of the 192 x 64 weight are not orthonormal, columns are)
    w = _mult(F, C, L, seed=7).agent_scaling[0].weight.detach()
    assert th.allclose(w.t() @ w, 2.0 * th.eye(64), atol=1e-4)
    # (6) the PPO minibatch loss runs and reaches every parameter of both scaling layers
    pol3 = _mult(F, C, L, seed=9)
    mb = dict(observations=obs, actions=th.randint(0, L, (6, 1)).float(), old_values=th.randn(6), old_log_prob=-th.rand(6),
              advantages=th.randn(6), returns=th.randn(6))
    loss, stats = orc.ppo_minibatch_loss(pol3, mb, orc.PPOHyper())
    loss.backward()
    for sc in (pol3.agent_scaling, pol3.value_scaling):
        assert sc[0].weight.grad is not None and float(sc[0].weight.grad.abs().sum()) > 0
    assert pol3.flat_params().size == sum(prm.numel() for prm in pol3.parameters())     # the device layout holds every parameter
    # (7) ADAP's context term (util.py:97-131) on this network: with the scaling layers at zero the policy ignores its context, every
    # pairwise KL is 0 and the term is exactly 1; with them live it is below 1 and differentiable through the scaling weights
    ctxs = np.random.default_rng(3).standard_normal((4, C)).astype(np.float32)
    assert float(orc.adap_context_loss(pol, obs, C, np.arange(6), ctxs).detach()) == 1.0
    term = orc.adap_context_loss(pol3, obs, C, np.arange(6), ctxs)
    assert 0.0 < float(term.detach()) < 1.0
    pol3.optimizer.zero_grad()
    term.backward()
    assert float(pol3.agent_scaling[0].weight.grad.abs().sum()) > 0 and pol3.value_scaling[0].weight.grad is None


def test_adap_mult_flat_layout_round_trip_and_the_launch_chain_of_its_gradient():
    """(1) the flat vector of ph_adapmult_layout (per net W1 b1 Ws bs W2 b2, heads last; weights input-major) round-trips through the
    oracle; (2) the backward chain csrc/ph_adapmult.hip runs launch by launch -- dh = dz Wo^T, dzh = dh (1 - h^2), dW2 = y^T dzh,
    dy = dzh W2^T, dza = dy ctx (1 - xa^2), dWs = x^T dza, dx = dy + dza Ws^T, dz1 = dx (1 - x^2), dW1 = o^T dz1 -- restated in
    float64 numpy on the flat vector, is autograd's gradient of an arbitrary function of the logits and the value"""
    F, C, L = 7, 3, 4
    pol = _mult(F, C, L, seed=4)
    with th.no_grad():
        for prm in pol.parameters():
            if prm.ndim == 1:
                prm.add_(0.2 * th.randn_like(prm))
    flat = pol.flat_params()
    H_ = 64
    per_net = F * H_ + H_ + H_ * H_ * C + H_ * C + H_ * H_ + H_
    assert flat.size == 2 * per_net + H_ * L + L + H_ + 1
    other = _mult(F, C, L, seed=9)
    other.load_flat_params(flat)
    assert np.array_equal(other.flat_params(), flat)
    rng = np.random.default_rng(0)
    obs = rng.standard_normal((11, F + C)).astype(np.float32)
    assert th.equal(other.logits(th.as_tensor(obs)), pol.logits(th.as_tensor(obs)))
    # an arbitrary scalar of the outputs: autograd ...
    wz, wv = rng.standard_normal((11, L)), rng.standard_normal((11, 1))
    pol.optimizer.zero_grad()
    z = pol.logits(th.as_tensor(obs))
    v = pol.predict_values(th.as_tensor(obs))
    ((z * th.as_tensor(wz, dtype=th.float32)).sum() + (v * th.as_tensor(wv, dtype=th.float32)).sum()).backward()
    g_ref = pol.flat_grads().astype(np.float64)
    # ... and the chain, net by net, on the flat vector
    p = flat.astype(np.float64)
    o, ctx = obs[:, :F].astype(np.float64), obs[:, F:].astype(np.float64)
    g = np.zeros_like(p)
    off = 0
    offs = {}
    for net in ("pi", "vf"):
        for nm, n in (("W1", F * H_), ("b1", H_), ("Ws", H_ * H_ * C), ("bs", H_ * C), ("W2", H_ * H_), ("b2", H_)):
            offs[net + nm] = (off, n)
            off += n
    for nm, n in (("act_W", H_ * L), ("act_b", L), ("val_W", H_), ("val_b", 1)):
        offs[nm] = (off, n)
        off += n
    take = lambda k: p[offs[k][0]:offs[k][0] + offs[k][1]]      # noqa: E731

    def put(k, val):
        g[offs[k][0]:offs[k][0] + offs[k][1]] = np.asarray(val).reshape(-1)
    for net, Wo, bo, dout in (("pi", "act_W", "act_b", wz), ("vf", "val_W", "val_b", wv)):
        W1, Ws, W2 = take(net + "W1").reshape(F, H_), take(net + "Ws").reshape(H_, H_ * C), take(net + "W2").reshape(H_, H_)
        x = np.tanh(o @ W1 + take(net + "b1"))
        xa = np.tanh(x @ Ws + take(net + "bs"))                                     # column j * C + c
        y = x + (xa.reshape(-1, H_, C) * ctx[:, None, :]).sum(-1)
        h = np.tanh(y @ W2 + take(net + "b2"))
        No = dout.shape[1]
        Wout = take(Wo).reshape(H_, No)
        put(Wo, h.T @ dout)
        put(bo, dout.sum(0))
        dzh = (dout @ Wout.T) * (1 - h * h)
        put(net + "W2", y.T @ dzh)
        put(net + "b2", dzh.sum(0))
        dy = dzh @ W2.T
        dza = (dy[:, :, None] * ctx[:, None, :]).reshape(-1, H_ * C) * (1 - xa * xa)
        put(net + "Ws", x.T @ dza)
        put(net + "bs", dza.sum(0))
        dz1 = (dy + dza @ Ws.T) * (1 - x * x)
        put(net + "W1", o.T @ dz1)
        put(net + "b1", dz1.sum(0))
    assert np.abs(g - g_ref).max() <= 2e-5 * max(1.0, np.abs(g_ref).max()), np.abs(g - g_ref).max()
