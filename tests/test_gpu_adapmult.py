"""AdapPolicyMult on the device (csrc/ph_adapmult.hip; pantheonrl/algos/adap/policies.py:136-283) against the oracle's restatement
(oracle/sb3_oracle.py: AdapMultPolicyOracle, pinned by the known-answer tests of tests/test_oracle.py): forward, evaluate_actions,
the fused rollout-buffer row, the PPO minibatch gradient with and without ADAP's context term, and train() chains.  Tolerances are
those of the MlpPolicy / AdapPolicy tests: logits 2e-5, gradients 1e-6 + 2e-4 * max|g|, parameters after a chain 2e-6 per step."""
import ctypes as C

import numpy as np
import pytest
import torch as th

from oracle import sb3_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu

CTX = {"adap_oc": 3, "adap_small": 3}


def _oracle(name, seed=0, perturb=0.3):
    th.manual_seed(seed)
    obs_s, act_s = H.CONFIGS[name]
    pol = orc.AdapMultPolicyOracle(obs_s, act_s, context_size=CTX[name])
    g = th.Generator().manual_seed(seed + 1)
    with th.no_grad():
        for p in pol.parameters():
            if p.ndim == 1:
                p.add_(perturb * th.randn(p.shape, generator=g))
        pol.action_net.weight.add_(perturb * th.randn(pol.action_net.weight.shape, generator=g))
    return pol


def _device(name, orac):
    from pantheonrl_amd import spaces as sp
    from pantheonrl_amd.adap import AdapPolicyMult
    obs_s, act_s = H.CONFIGS[name]
    cs = CTX[name]
    pol = AdapPolicyMult(sp.Box(-np.inf, np.inf, (obs_s.dim - cs,)), H.to_space(act_s), context_size=cs, device="cuda", seed=0)
    assert pol.mlayout.P == orac.flat_params().size
    pol.set_flat_params(orac.flat_params())
    return pol


@pytest.mark.parametrize("name", ["adap_small", "adap_oc"])
@pytest.mark.parametrize("n", [1, 7, 64, 300])
def test_forward_and_evaluate_match_the_oracle(name, n):
    orac = _oracle(name, seed=3)
    pol = _device(name, orac)
    obs_s, _ = H.CONFIGS[name]
    rng = np.random.default_rng(n)
    obs = rng.standard_normal((n, obs_s.dim)).astype(np.float32)
    with th.no_grad():
        z_ref = orac.logits(th.as_tensor(obs)).numpy()
        v_ref = orac.predict_values(th.as_tensor(obs)).numpy()
    z = pol.get_logits(obs).cpu().numpy()
    v = pol.predict_values(obs).cpu().numpy()
    assert np.abs(z - z_ref).max() < 2e-5 and np.abs(v - v_ref).max() < 2e-5, (np.abs(z - z_ref).max(), np.abs(v - v_ref).max())
    # evaluate_actions: log-prob and entropy of given actions
    acts = rng.integers(0, z.shape[1], size=(n, 1))
    with th.no_grad():
        v2, lp_ref, ent_ref = orac.evaluate_actions(th.as_tensor(obs), th.as_tensor(acts))
    values, lp, ent = pol.evaluate_actions(obs, acts.astype(np.float32))
    assert np.abs(lp.cpu().numpy() - lp_ref.numpy()).max() < 2e-5
    assert np.abs(ent.cpu().numpy() - ent_ref.numpy()).max() < 2e-5
    assert np.abs(values.cpu().numpy() - v2.numpy()).max() < 2e-5
    # teacher-forced sampling: the inverse CDF of the oracle's probabilities at the same uniforms
    u = rng.random((n, 1)).astype(np.float32)
    a_dev, _, lp_s = pol.forward(obs, uniforms=u)
    with th.no_grad():
        a_ref, _, lp_sr = orac.forward(th.as_tensor(obs), uniforms=th.as_tensor(u))
    same = a_dev.cpu().numpy().reshape(-1) == a_ref.numpy().reshape(-1)
    assert same.mean() > 0.99                     # a uniform within rounding of a CDF step may fall on the other side
    assert np.abs(lp_s.cpu().numpy()[same] - lp_sr.numpy()[same]).max() < 2e-5


def test_fused_rollout_row_is_what_add_would_write():
    name, T, E = "adap_small", 4, 6
    orac = _oracle(name, seed=5)
    pol = _device(name, orac)
    buf = H.make_device_buffer(name, pol, T, E)
    rng = np.random.default_rng(0)
    obs_s, _ = H.CONFIGS[name]
    starts = np.ones(E, np.float32)
    for t in range(T):
        obs = rng.standard_normal((E, obs_s.dim)).astype(np.float32)
        u = rng.random((E, 1)).astype(np.float32)
        acts, values, logp = pol.forward_and_store(obs, buf, starts, uniforms=u)
        th.cuda.synchronize()
        h = buf.host()
        assert np.array_equal(h["observations"][t].reshape(E, -1), obs)
        assert np.array_equal(h["actions"][t].reshape(-1), acts.cpu().numpy().reshape(-1).astype(np.float32))
        assert np.array_equal(h["values"][t], values.cpu().numpy().reshape(-1))
        assert np.array_equal(h["log_probs"][t], logp.cpu().numpy())
        assert np.array_equal(h["episode_starts"][t], starts) and np.all(h["rewards"][t] == 0)
        starts = (rng.random(E) < 0.3).astype(np.float32)
    assert buf.pos == T and buf.full


def _hyper(nat, hp):
    from pantheonrl_amd.ppo import PPO
    model = PPO.__new__(PPO)
    for k in ("learning_rate", "clip_range", "clip_range_vf", "ent_coef", "vf_coef", "max_grad_norm", "target_kl",
              "normalize_advantage"):
        setattr(model, k, getattr(hp, k))
    return PPO.hyper(model)


def _grad_pair(name, T, E, nb, hp, coef=None, n_ctx=5, n_states=32, seed=5):
    from pantheonrl_amd import _native as nat
    from tests.test_gpu_adap import _adap_struct
    cs = CTX[name]
    orac = _oracle(name, seed=seed)
    ob = H.filled_oracle_buffer(name, orac, T, E, seed=seed)
    pol = _device(name, orac)
    buf = H.make_device_buffer(name, pol, T, E)
    H.upload_buffer(buf, ob)
    idx = np.random.default_rng(nb).permutation(T * E)[:nb]
    keep, ad, loss_t, sidx, ctxs = [], None, None, None, None
    if coef is not None:
        rng = np.random.default_rng(seed + nb)
        ns = min(n_states, nb)
        sidx = np.full(n_states, -1, np.int32)
        sidx[:ns] = rng.permutation(nb)[:ns]
        ctxs = orc.adap_sample_contexts("l2", cs, n_ctx, rng.random((n_ctx, cs)))
        ad, loss_t, _ = _adap_struct(nat, cs, n_ctx, n_states, coef, keep, sidx[None], ctxs[None])
    h = _hyper(nat, hp)
    idx_t = th.as_tensor(np.asarray(idx, np.int32)).cuda()
    g = th.zeros(pol.mlayout.P, device="cuda")
    st = th.zeros(nat.PH_NSTAT, device="cuda")
    pol._bind()
    nat.check(pol.ctx.lib.ph_adapmult_minibatch_grad(pol.ctx.handle, C.byref(pol.spec), cs, pol.params.data_ptr(),
                                                     C.byref(buf.c_struct()), C.byref(h), idx_t.data_ptr(), nb, g.data_ptr(),
                                                     st.data_ptr(), C.byref(ad) if ad is not None else None))
    th.cuda.synchronize()
    flat = ob.flat()
    mb = {k: th.as_tensor(v[idx]) for k, v in flat.items()}
    orac.optimizer.zero_grad()
    loss, stats_ref = orc.ppo_minibatch_loss(orac, mb, hp)
    if coef is not None:
        ns = min(n_states, nb)
        cl = orc.adap_context_loss(orac, mb["observations"], cs, sidx[:ns], ctxs)
        loss = loss + coef * cl
        stats_ref["context_loss"] = cl.item()
    stats_ref["loss"] = loss.item()
    loss.backward()
    return g.cpu().numpy(), orac.flat_grads(), st.cpu().numpy(), stats_ref, None if loss_t is None else float(loss_t.item())


def _assert_grads(g, g_ref):
    scale = np.abs(g_ref).max()
    err = np.abs(g - g_ref)
    assert err.max() <= 1e-6 + 2e-4 * scale, (err.max(), scale, int(err.argmax()))


@pytest.mark.parametrize("name,T,E,nb", [("adap_small", 16, 8, 64), ("adap_small", 8, 4, 20), ("adap_oc", 16, 8, 100),
                                         ("adap_small", 16, 8, 1), ("adap_oc", 64, 64, 4096)])
def test_ppo_minibatch_gradient_matches_autograd(name, T, E, nb):
    hp = orc.PPOHyper(ent_coef=0.01, clip_range_vf=0.3) if nb == 100 else orc.PPOHyper()
    g, g_ref, st, st_ref, _ = _grad_pair(name, T, E, nb, hp)
    _assert_grads(g, g_ref)
    for j, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss")):
        assert abs(st[j] - st_ref[k]) <= 2e-5 + 2e-4 * abs(st_ref[k]), (k, st[j], st_ref[k])
    assert np.abs(g_ref).max() > 1e-4                         # the comparison is not 0 against 0


@pytest.mark.parametrize("name,T,E,nb,n_ctx,n_states", [("adap_small", 16, 8, 64, 5, 32), ("adap_small", 8, 4, 20, 5, 32),
                                                        ("adap_oc", 16, 8, 128, 2, 7), ("adap_small", 16, 8, 64, 16, 5)])
def test_gradient_with_the_context_term_matches_autograd(name, T, E, nb, n_ctx, n_states):
    g, g_ref, st, st_ref, cl = _grad_pair(name, T, E, nb, orc.PPOHyper(), coef=5.0, n_ctx=n_ctx, n_states=n_states)
    _assert_grads(g, g_ref)
    assert abs(cl - st_ref["context_loss"]) <= 1e-5, (cl, st_ref["context_loss"])
    assert abs(st[5] - st_ref["loss"]) <= 1e-5 + 1e-4 * abs(st_ref["loss"]), (st[5], st_ref["loss"])
    # the term is a visible share: without it the gradient is another one
    g0 = _grad_pair(name, T, E, nb, orc.PPOHyper())[0]
    assert np.abs(g - g0).max() > 1e-5


def _model(name, T, E, hp, coef, n_ctx=5, n_states=32, seed=0):
    from pantheonrl_amd import spaces as sp
    from pantheonrl_amd.adap import ADAP
    obs_s, act_s = H.CONFIGS[name]
    cs = CTX[name]
    env = type("E", (), dict(observation_space=sp.Box(-np.inf, np.inf, (obs_s.dim - cs,)), action_space=H.to_space(act_s),
                             _is_dummy_space_env=True))()
    return ADAP("AdapPolicyMult", env, n_steps=T, n_envs=E, batch_size=hp.batch_size, n_epochs=hp.n_epochs,
                learning_rate=hp.learning_rate, clip_range=hp.clip_range, clip_range_vf=hp.clip_range_vf,
                normalize_advantage=hp.normalize_advantage, ent_coef=hp.ent_coef, vf_coef=hp.vf_coef,
                max_grad_norm=hp.max_grad_norm, target_kl=hp.target_kl, seed=seed, context_loss_coeff=coef, context_size=cs,
                num_context_samples=n_ctx, num_state_samples=n_states)


@pytest.mark.parametrize("name,T,E,batch,epochs,coef", [("adap_small", 32, 8, 64, 3, 0.1), ("adap_oc", 25, 5, 64, 2, 1.0)])
def test_train_matches_the_oracle_chain(name, T, E, batch, epochs, coef, tmp_path):
    hp = orc.PPOHyper(batch_size=batch, n_epochs=epochs)
    cs, n_ctx, n_states, seed = CTX[name], 5, 32, 31
    orac = _oracle(name, seed=seed)
    ob = H.filled_oracle_buffer(name, orac, T, E, seed=seed)
    model = _model(name, T, E, hp, coef)
    model.policy.set_flat_params(orac.flat_params())
    H.upload_buffer(model.rollout_buffer, ob)
    N = T * E
    n_mb = (N + batch - 1) // batch
    rng = np.random.default_rng(seed)
    perms = np.stack([rng.permutation(N) for _ in range(epochs)])
    sidx = np.zeros((epochs * n_mb, n_states), np.int32)
    ctxs = np.zeros((epochs * n_mb, n_ctx, cs), np.float32)
    sidx_l = []
    for m in range(epochs * n_mb):
        nb = min(batch, N - (m % n_mb) * batch)
        ns = min(n_states, nb)
        sidx[m, :ns] = rng.permutation(nb)[:ns]
        sidx_l.append(sidx[m, :ns])
        ctxs[m] = orc.adap_sample_contexts("l2", cs, n_ctx, rng.random((n_ctx, cs)))
    model.train(perms=perms, state_idx=sidx, contexts=ctxs)
    stats_ref = orc.ppo_train(orac, ob, hp, perms, adap=orc.AdapTerm(cs, coef, sidx_l, ctxs))
    st, steps = model.last_train_stats, len(stats_ref)
    assert st.shape[0] == steps
    p, p_ref = model.policy.get_flat_params(), orac.flat_params()
    assert np.abs(p - p_ref).max() <= 2e-6 * steps + 1e-6, np.abs(p - p_ref).max()
    for i, s in enumerate(stats_ref):
        for j, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss", "grad_norm")):
            assert abs(st[i, j] - s[k]) <= 2e-4 + 2e-3 * abs(s[k]), (i, k, st[i, j], s[k])
        assert abs(model.last_context_losses[i] - s["context_loss"]) <= 1e-5, (i, model.last_context_losses[i])
    # the checkpoint names the policy class and carries the reference's module names
    path = str(tmp_path / "mult")
    model.save(path)
    from pantheonrl_amd.adap import ADAP, AdapPolicyMult
    again = ADAP.load(path)
    assert isinstance(again.policy, AdapPolicyMult)
    assert np.array_equal(again.policy.get_flat_params(), p)
    sd = model.policy.state_dict()
    assert sd["mlp_extractor.agent_scaling.0.weight"].shape == (64 * cs, 64) and "mlp_extractor.value_branch_2.0.bias" in sd


def test_kl_early_stop_and_samples_drawn_in_the_kernel():
    """default path: nothing teacher-forced (Feistel order, in-kernel state / context samples keyed by the seed); a tight target_kl
    stops the update before its optimizer step, as for MlpPolicy"""
    name, T, E = "adap_small", 16, 8
    orac = _oracle(name, seed=9)
    ob = H.filled_oracle_buffer(name, orac, T, E, seed=9)
    outs = []
    for rep in range(2):
        hp = orc.PPOHyper(batch_size=32, n_epochs=4, learning_rate=3e-2, target_kl=0.01)
        model = _model(name, T, E, hp, coef=0.1, seed=4)
        model.policy.set_flat_params(orac.flat_params())
        H.upload_buffer(model.rollout_buffer, ob)
        model.device_permutations = True
        model.train()
        th.cuda.synchronize()
        outs.append((model.policy.get_flat_params(), model.last_train_stats.copy(), int(model.policy.opt_step.item())))
    (p0, st0, n0), (p1, st1, n1) = outs
    assert 0 < n0 < 4 * 4, n0                                  # the stop happened, after at least one step
    assert n0 == n1 and np.array_equal(p0, p1) and np.array_equal(st0, st1)      # keyed and deterministic
    assert np.isfinite(p0).all() and not np.array_equal(p0, orac.flat_params())


def test_trainer_adap_mult_object_graph(tmp_path, monkeypatch):
    """`trainer.py RPS-v0 ADAP_MULT ADAP_MULT --share-latent` (trainer.py:32-34,129-130,207-208): both learners are ADAP on
    AdapPolicyMult, the partner acts under the ego's context, both update, the saved ego loads as a FIXED partner"""
    from pantheonrl_amd import trainer
    from pantheonrl_amd.adap import ADAP, AdapPolicyMult
    monkeypatch.chdir(tmp_path)
    cfg = '{"n_steps": 32, "batch_size": 16, "n_epochs": 2}'
    ego, partners, _ = trainer.run(["RPS-v0", "ADAP_MULT", "ADAP_MULT", "--share-latent", "--seed", "3", "-t", "96", "--ego-config", cfg,
                                    "--alt-config", cfg, "--ego-save", "m/ego", "--alt-save", "m/alt"])
    assert isinstance(ego, ADAP) and isinstance(ego.policy, AdapPolicyMult) and ego._n_updates == 6
    alt = partners[0]
    assert isinstance(alt.model.policy, AdapPolicyMult) and alt.latent_syncer is ego.policy and alt.iteration >= 2
    assert ego.last_context_losses is not None and (ego.last_context_losses > 0).all() and (ego.last_context_losses <= 1).all()
    assert np.isfinite(ego.policy.get_flat_params()).all()
    loaded = ADAP.load("m/ego")
    assert isinstance(loaded.policy, AdapPolicyMult) and np.array_equal(loaded.policy.get_flat_params(), ego.policy.get_flat_params())
    trainer.run(["RPS-v0", "PPO", "FIXED", "--seed", "3", "-t", "64", "--ego-config", cfg, "--alt-config",
                 '{"type": "ADAP_MULT", "location": "m/ego", "latent_val": [0.0, 1.0, 0.0]}'])
