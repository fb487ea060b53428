"""ADAP's context term on the device (ph_adap_train / ph_adap_minibatch_grad) against the oracle's restatement of
pantheonrl/algos/adap/util.py:97-131 + adap_learn.py:313-320.  Tolerances: gradients 1e-6 + 2e-4 * max|g| (f32 sums in a
different order, v_exp / v_rcp tanh), parameters after a chain 2e-6 per optimizer step."""
import ctypes as C

import numpy as np
import pytest
import torch as th

from oracle import sb3_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu

CTX = {"adap_oc": 3, "adap_small": 3, "adap_multi": 4}


def _adap_struct(nat, cs, n_ctx, n_states, coef, keep, state_idx=None, contexts=None, n_mb=1, sampler="l2", seed=0,
                 want_used=False):
    ad = nat.PhAdapLoss()
    ad.context_size, ad.num_context_samples, ad.num_state_samples = cs, n_ctx, n_states
    ad.sampler, ad.context_loss_coeff, ad.seed = nat.CONTEXT_SAMPLERS[sampler], coef, seed
    if state_idx is not None:
        t = th.as_tensor(np.ascontiguousarray(state_idx, dtype=np.int32)).cuda()
        keep.append(t)
        ad.state_idx = t.data_ptr()
    if contexts is not None:
        t = th.as_tensor(np.ascontiguousarray(contexts, dtype=np.float32)).cuda()
        keep.append(t)
        ad.contexts = t.data_ptr()
    loss = th.zeros(n_mb, device="cuda")
    keep.append(loss)
    ad.context_loss = loss.data_ptr()
    used = None
    if want_used:
        used = (th.full((n_mb, n_states), -1, dtype=th.int32, device="cuda"), th.zeros((n_mb, n_ctx, cs), device="cuda"))
        keep.extend(used)
        ad.used_state_idx, ad.used_contexts = used[0].data_ptr(), used[1].data_ptr()
    return ad, loss, used


def _grad_pair(name, T, E, idx, hp, n_ctx, n_states, coef, seed=5, sampler=None, gemm_mode=0):
    """device and oracle gradient of one minibatch with the context term; samples teacher-forced unless `sampler`"""
    from pantheonrl_amd import _native as nat
    from pantheonrl_amd.ppo import PPO
    cs = CTX[name]
    orac = H.oracle_policy(name, seed=seed)
    ob = H.filled_oracle_buffer(name, orac, T, E, seed=seed)
    pol = H.device_policy(name, orac)
    pol.gemm_mode = gemm_mode
    buf = H.make_device_buffer(name, pol, T, E)
    H.upload_buffer(buf, ob)
    nb = len(idx)
    rng = np.random.default_rng(seed + nb)
    ns = min(n_states, nb)
    keep = []
    if sampler is None:
        sidx = np.full(n_states, -1, np.int32)
        sidx[:ns] = rng.permutation(nb)[:ns]
        ctxs = orc.adap_sample_contexts("l2", cs, n_ctx, rng.random((n_ctx, cs)))
        ad, loss_t, used = _adap_struct(nat, cs, n_ctx, n_states, coef, keep, sidx[None], ctxs[None])
    else:
        ad, loss_t, used = _adap_struct(nat, cs, n_ctx, n_states, coef, keep, sampler=sampler, seed=seed, want_used=True)
    model = PPO.__new__(PPO)
    for k in ("learning_rate", "clip_range", "clip_range_vf", "ent_coef", "vf_coef", "max_grad_norm", "target_kl",
              "normalize_advantage"):
        setattr(model, k, getattr(hp, k))
    h = PPO.hyper(model)
    idx_t = th.as_tensor(np.asarray(idx, np.int32)).cuda()
    g = th.zeros(pol.layout.P, device="cuda")
    st = th.zeros(nat.PH_NSTAT, device="cuda")
    pol._bind()
    nat.check(pol.ctx.lib.ph_adap_minibatch_grad(pol.ctx.handle, C.byref(pol.spec), pol.params.data_ptr(),
                                                 C.byref(buf.c_struct()), C.byref(h), idx_t.data_ptr(), nb, g.data_ptr(),
                                                 st.data_ptr(), gemm_mode, C.byref(ad)))
    th.cuda.synchronize()
    if sampler is not None:
        sidx, ctxs = used[0][0].cpu().numpy(), used[1][0].cpu().numpy()
    flat = ob.flat()
    mb = {k: th.as_tensor(v[idx]) for k, v in flat.items()}
    orac.optimizer.zero_grad()
    loss, stats_ref = orc.ppo_minibatch_loss(orac, mb, hp)
    cl = orc.adap_context_loss(orac, mb["observations"], cs, sidx[:ns], ctxs)
    (loss + coef * cl).backward()
    stats_ref["context_loss"], stats_ref["loss"] = cl.item(), (loss + coef * cl).item()
    return g.cpu().numpy(), orac.flat_grads(), st.cpu().numpy(), stats_ref, float(loss_t.item()), sidx[:ns], ctxs, pol.layout


def _assert_grads(g, g_ref):
    scale = np.abs(g_ref).max()
    err = np.abs(g - g_ref)
    assert err.max() <= 1e-6 + 2e-4 * scale, (err.max(), scale, int(err.argmax()))


@pytest.mark.parametrize("name,T,E,nb,n_ctx,n_states", [
    ("adap_oc", 16, 8, 64, 5, 32),       # the reference's defaults (adap_learn.py:111-116)
    ("adap_small", 16, 8, 100, 5, 32),
    ("adap_small", 8, 4, 20, 5, 32),     # minibatch shorter than num_state_samples: every row is a sampled state
    ("adap_multi", 16, 6, 77, 4, 10),    # three action components: the KL is the components' sum (util.py:30-35)
    ("adap_oc", 16, 8, 128, 2, 7),       # one pair
    ("adap_small", 16, 8, 64, 16, 5),    # one state per workgroup, 120 pairs
    ("adap_small", 16, 8, 64, 8, 33),
    ("adap_oc", 128, 256, 32768, 5, 32),  # a whole bench-size minibatch (two-chunk general gradient kernel + the term)
    ("adap_small", 128, 256, 32768, 5, 32),   # the same on the 64-row fast gradient kernel
])
def test_context_term_gradient_matches_autograd(name, T, E, nb, n_ctx, n_states):
    idx = np.random.default_rng(nb).permutation(T * E)[:nb]
    # a large coefficient so that the context term is a visible share of the gradient
    g, g_ref, st, st_ref, cl, _, _, _ = _grad_pair(name, T, E, idx, orc.PPOHyper(), n_ctx, n_states, coef=5.0)
    _assert_grads(g, g_ref)
    assert abs(cl - st_ref["context_loss"]) <= 1e-5, (cl, st_ref["context_loss"])
    assert abs(st[5] - st_ref["loss"]) <= 1e-5 + 1e-4 * abs(st_ref["loss"]), (st[5], st_ref["loss"])


def test_context_term_alone_and_its_share():
    """coef = 0 is the PPO gradient bit for bit; the difference of two coefficients is linear in the coefficient"""
    from tests.test_gpu_parity import _grad_pair as ppo_pair
    idx = np.random.default_rng(3).permutation(128)[:64]
    hp = orc.PPOHyper()
    g0 = _grad_pair("adap_small", 16, 8, idx, hp, 5, 32, coef=0.0, seed=11)[0]
    gp = ppo_pair("adap_small", 16, 8, idx, hp, seed=11)[0]
    assert np.array_equal(g0, gp)
    g1 = _grad_pair("adap_small", 16, 8, idx, hp, 5, 32, coef=1.0, seed=11)[0]
    g2 = _grad_pair("adap_small", 16, 8, idx, hp, 5, 32, coef=2.0, seed=11)[0]
    d1, d2 = g1 - g0, g2 - g0
    assert np.abs(d1).max() > 1e-4
    assert np.abs(d2 - 2 * d1).max() <= 1e-6 + 1e-5 * np.abs(d2).max()
    lay = H.device_policy("adap_small", H.oracle_policy("adap_small")).layout
    vf = np.r_[lay.vf_W1:lay.act_W, lay.val_W:lay.P]
    assert np.array_equal(d1[vf], np.zeros(len(vf), np.float32))   # the value network takes no part (util.py:118-121)


@pytest.mark.parametrize("sampler", ["l2", "unit_square", "positive_square", "categorical"])
def test_samples_drawn_in_the_kernel(sampler):
    """NULL samples: the kernel draws them; what it reports having used reproduces the gradient through the oracle and
    has the sampler's shape (util.py:42-77) and randperm's (distinct positions inside the minibatch)"""
    idx = np.random.default_rng(1).permutation(128)[:90]
    out = _grad_pair("adap_small", 16, 8, idx, orc.PPOHyper(), 5, 32, coef=3.0, sampler=sampler, seed=7)
    g, g_ref, sidx, ctxs = out[0], out[1], out[5], out[6]
    _assert_grads(g, g_ref)
    assert len(set(sidx.tolist())) == 32 and sidx.min() >= 0 and sidx.max() < 90
    assert ctxs.shape == (5, 3) and (sampler == "categorical" or len({tuple(c) for c in ctxs.tolist()}) == 5)
    if sampler == "l2":
        assert np.abs(np.linalg.norm(ctxs, axis=1) - 1).max() < 1e-6
    elif sampler == "unit_square":
        assert ctxs.min() >= -1 and ctxs.max() < 1 and ctxs.min() < 0
    elif sampler == "positive_square":
        assert ctxs.min() >= 0 and ctxs.max() < 1
    else:
        assert np.array_equal(np.sort(ctxs, axis=1)[:, -1], np.ones(5)) and np.array_equal(ctxs.sum(1), np.ones(5))
    again = _grad_pair("adap_small", 16, 8, idx, orc.PPOHyper(), 5, 32, coef=3.0, sampler=sampler, seed=7)
    assert np.array_equal(again[0], g) and np.array_equal(again[5], sidx) and np.array_equal(again[6], ctxs)
    other = _grad_pair("adap_small", 16, 8, idx, orc.PPOHyper(), 5, 32, coef=3.0, sampler=sampler, seed=8)
    assert not np.array_equal(other[5], sidx) and (sampler == "categorical" or not np.array_equal(other[6], ctxs))


# ----------------------------------------------------------------------------------------------------------------
# ADAP.train(): the whole update chain with the term, and the host surface
# ----------------------------------------------------------------------------------------------------------------
def _adap_model(name, T, E, hp, coef=0.1, n_ctx=5, n_states=32, sampler="l2", seed=0):
    from pantheonrl_amd.adap import ADAP
    from pantheonrl_amd import spaces as sp
    obs_s, act_s = H.CONFIGS[name]
    cs = CTX[name]
    env = type("E", (), dict(observation_space=sp.Box(-np.inf, np.inf, (obs_s.dim - cs,)), action_space=H.to_space(act_s),
                             _is_dummy_space_env=True))()
    return ADAP("AdapPolicy", env, n_steps=T, n_envs=E, batch_size=hp.batch_size, n_epochs=hp.n_epochs,
                learning_rate=hp.learning_rate, clip_range=hp.clip_range, clip_range_vf=hp.clip_range_vf,
                normalize_advantage=hp.normalize_advantage, ent_coef=hp.ent_coef, vf_coef=hp.vf_coef,
                max_grad_norm=hp.max_grad_norm, target_kl=hp.target_kl, seed=seed, context_loss_coeff=coef,
                context_size=cs, num_context_samples=n_ctx, num_state_samples=n_states, context_sampler=sampler)


@pytest.mark.parametrize("name,T,E,batch,epochs,coef", [("adap_small", 32, 8, 64, 3, 0.1), ("adap_oc", 25, 5, 64, 2, 0.1),
                                                        ("adap_multi", 16, 6, 40, 2, 1.0), ("adap_small", 16, 4, 24, 2, 2.0)])
def test_adap_train_matches_oracle(name, T, E, batch, epochs, coef):
    hp = orc.PPOHyper(batch_size=batch, n_epochs=epochs)
    cs, n_ctx, n_states, seed = CTX[name], 5, 32, 31
    orac = H.oracle_policy(name, seed=seed)
    ob = H.filled_oracle_buffer(name, orac, T, E, seed=seed)
    model = _adap_model(name, T, E, hp, coef=coef)
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
        for j, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss",
                               "grad_norm")):
            assert abs(st[i, j] - s[k]) <= 2e-4 + 2e-3 * abs(s[k]), (i, k, st[i, j], s[k])
        assert abs(model.last_context_losses[i] - s["context_loss"]) <= 1e-5, (i, model.last_context_losses[i])
    # the term moved the parameters: the same chain without it ends somewhere else
    plain = H.oracle_policy(name, seed=seed)
    orc.ppo_train(plain, ob, hp, perms)
    assert np.abs(plain.flat_params() - p_ref).max() > 1e-5


def test_adap_train_with_samples_drawn_in_the_kernel_is_keyed_by_the_seed():
    """default path: nothing teacher-forced, no host random numbers at all (Feistel order, in-kernel samples)"""
    hp = orc.PPOHyper(batch_size=64, n_epochs=2)
    orac = H.oracle_policy("adap_small", seed=3)
    ob = H.filled_oracle_buffer("adap_small", orac, 16, 8, seed=3)

    def run(seed):
        model = _adap_model("adap_small", 16, 8, hp, coef=1.0, seed=seed)
        model.device_permutations = True
        model.policy.set_flat_params(orac.flat_params())
        H.upload_buffer(model.rollout_buffer, ob)
        model.train()
        return model.policy.get_flat_params(), model.last_context_losses.copy()
    p0, c0 = run(0)
    p0b, c0b = run(0)
    p1, c1 = run(1)
    assert np.array_equal(p0, p0b) and np.array_equal(c0, c0b)          # deterministic in the seed
    assert not np.array_equal(c0, c1) and not np.array_equal(p0, p1)    # and a function of it
    assert (c0 > 0.5).all() and (c0 <= 1.0).all() and len(set(c0.tolist())) == len(c0)


def test_adap_rollout_rows_carry_the_context_and_contexts_change_at_episode_ends():
    """ADAP.collect_rollouts (adap_learn.py:375-473) vectorised: every stored row = observation ++ the context its column
    had; a column's context is re-drawn exactly when its episode ends; V(terminal) bootstraps use the column's context"""
    from pantheonrl_amd.adap import ADAP
    from pantheonrl_amd import spaces as sp

    class Env:
        num_envs = 4
        observation_space, action_space = sp.Box(-1, 1, (6,)), sp.Discrete(3)

        def __init__(self):
            self.t, self.rng = 0, np.random.default_rng(0)

        def reset(self):
            return self.rng.standard_normal((4, 6)).astype(np.float32)

        def step(self, actions):
            self.t += 1
            dones = np.array([(self.t + e) % (3 + e) == 0 for e in range(4)])
            obs = self.rng.standard_normal((4, 6)).astype(np.float32)
            return obs, np.ones(4, np.float32), dones, [{} for _ in range(4)]

    model = ADAP("AdapPolicy", Env(), n_steps=12, n_envs=4, batch_size=16, n_epochs=1, seed=0, context_size=2)
    model._last_obs, model._last_episode_starts = model.env.reset(), np.ones(4, np.float32)
    seen = []
    orig = model._after_step

    def spy(dones):
        before = model.policy.get_context().copy()
        orig(dones)
        seen.append((np.asarray(dones, bool).copy(), before, model.policy.get_context().copy()))
    model._after_step = spy
    first = model.policy.get_context().copy()
    assert first.shape == (4, 2) and np.abs(np.linalg.norm(first, axis=1) - 1).max() < 1e-6
    model.collect_rollouts()
    rows = model.rollout_buffer.observations.cpu().numpy()          # (T, E, 6 + 2)
    active = first
    for t, (dones, before, after) in enumerate(seen):
        assert np.array_equal(before, active)
        assert np.array_equal(rows[t, :, 6:], active)               # the context the step acted under
        assert np.array_equal(after[~dones], active[~dones])
        assert dones.sum() == 0 or not np.any(np.all(after[dones] == active[dones], axis=1))
        active = after
    assert sum(int(d.sum()) for d, _, _ in seen) > 3
    model.train()                                                   # and the update runs on those rows
    assert model.last_context_losses.shape == (3,) and (model.last_context_losses > 0).all()


def test_adap_agent_partner_side():
    """AdapAgent (adap/agent.py:20-151): buffer rows carry the context, a latent syncer overrides it, done re-draws it"""
    from pantheonrl_amd.adap import ADAP, AdapAgent
    from pantheonrl_amd.common import Observation
    from pantheonrl_amd import spaces as sp
    env = type("E", (), dict(observation_space=sp.Box(-1, 1, (5,)), action_space=sp.Discrete(4), _is_dummy_space_env=True))()
    ego = ADAP("AdapPolicy", env, n_steps=8, batch_size=8, n_epochs=1, seed=0)
    alt = ADAP("AdapPolicy", env, n_steps=8, batch_size=8, n_epochs=1, seed=1)
    solo, synced = AdapAgent(ADAP("AdapPolicy", env, n_steps=8, batch_size=8, n_epochs=1, seed=2)), AdapAgent(
        alt, latent_syncer=ego.policy)
    rng = np.random.default_rng(0)
    ctx_solo = solo.model.policy.get_context().copy()
    for t in range(10):                                  # the 9th get_action trains on the full buffer first
        o = rng.standard_normal(5).astype(np.float32)
        a = solo.get_action(Observation(o))
        assert a.shape == () or a.shape == (1,) or np.isscalar(a)
        row = solo.model.rollout_buffer.observations[solo.model.rollout_buffer.pos - 1, 0].cpu().numpy()
        assert np.array_equal(row[:5], o) and np.array_equal(row[5:], ctx_solo[0])
        solo.update(1.0, t == 4)
        if t == 4:
            assert not np.array_equal(solo.model.policy.get_context(), ctx_solo)
            ctx_solo = solo.model.policy.get_context().copy()
    assert solo.iteration == 1
    o = rng.standard_normal(5).astype(np.float32)
    synced.get_action(Observation(o))
    row = alt.rollout_buffer.observations[0, 0].cpu().numpy()
    assert np.array_equal(row[5:], ego.policy.get_context()[0])
    before = alt.policy.get_context().copy()
    synced.update(0.0, True)
    assert np.array_equal(alt.policy.get_context(), before)      # a synced partner never draws its own (agent.py:148)


def test_adap_refuses_what_it_does_not_implement():
    from pantheonrl_amd.adap import ADAP
    from pantheonrl_amd import spaces as sp
    box = type("E", (), dict(observation_space=sp.Box(-1, 1, (5,)), action_space=sp.Discrete(4), _is_dummy_space_env=True))()
    disc = type("E", (), dict(observation_space=sp.MultiBinary(5), action_space=sp.Discrete(4), _is_dummy_space_env=True))()
    with pytest.raises(ValueError):
        ADAP("AdapPolicyTimes", box)                                        # (AdapPolicyMult: tests/test_gpu_adapmult.py)
    with pytest.raises(Exception):
        ADAP("AdapPolicyMult", box, context_size=5, n_steps=8, batch_size=8, n_epochs=1)   # the device path takes 1 .. 4 components
    with pytest.raises(ValueError):
        ADAP("AdapPolicy", box, context_sampler="natural_numbers")          # (num, 1) contexts need context_size = 1 ...
    nn_model = ADAP("AdapPolicy", box, context_sampler="natural_numbers", context_size=1, n_steps=8, batch_size=8, n_epochs=1)
    assert np.array_equal(nn_model.sample_context(4), np.zeros((4, 1), np.float32))   # ... and the only integer in [0, 1) is 0
    with pytest.raises(ValueError):
        ADAP("AdapPolicy", disc)
    m = ADAP("AdapPolicy", box, n_steps=8, batch_size=8, n_epochs=1, num_context_samples=1)
    with pytest.raises(Exception, match="num_context_samples"):
        m.train()


def test_trainer_adap_object_graph_with_shared_latent(tmp_path, monkeypatch):
    """`trainer.py RPS-v0 ADAP ADAP --share-latent` (trainer.py:65-89,127-128,205-213): one-hot features ++ context, the
    partner acts under the ego's context, both learners update, and a saved ADAP model loads as a FIXED partner"""
    from pantheonrl_amd import trainer
    from pantheonrl_amd.adap import ADAP
    monkeypatch.chdir(tmp_path)
    cfg = '{"n_steps": 32, "batch_size": 16, "n_epochs": 2}'
    ego, partners, _ = trainer.run(["RPS-v0", "ADAP", "ADAP", "--share-latent", "--seed", "3", "-t", "96", "--ego-config", cfg,
                                 "--alt-config", cfg, "--ego-save", "m/ego", "--alt-save", "m/alt"])
    assert isinstance(ego, ADAP) and ego.policy.layout.D == 1 + 3 and ego._n_updates == 6
    alt = partners[0]
    assert alt.latent_syncer is ego.policy and alt.iteration >= 2
    rows = alt.model.rollout_buffer.observations[: alt.model.rollout_buffer.pos, 0].cpu().numpy()
    assert np.array_equal(rows[:, 0], np.ones(len(rows))) and np.abs(np.linalg.norm(rows[:, 1:], axis=1) - 1).max() < 1e-6
    assert ego.last_context_losses is not None and (ego.last_context_losses > 0).all()
    loaded = ADAP.load("m/ego")
    assert loaded.context_size == 3 and np.array_equal(loaded.policy.get_flat_params(), ego.policy.get_flat_params())
    trainer.run(["RPS-v0", "PPO", "FIXED", "--seed", "3", "-t", "64", "--ego-config", cfg, "--alt-config",
                 '{"type": "ADAP", "location": "m/ego", "latent_val": [0.0, 1.0, 0.0]}'])


def test_context_streams_are_per_learner_and_survive_save_load(tmp_path):
    """trainer.py gives the ego and every ADAP partner the same --seed; the reference draws every context from one global
    torch stream (adap/util.py:42-77), so two learners never act under each other's draws: here the generator is keyed by
    (seed, sampling_stream), and a loaded model continues its stream instead of replaying it from the seed."""
    from pantheonrl_amd.adap import ADAP
    hp = orc.PPOHyper(batch_size=8, n_epochs=1)
    ego = _adap_model("adap_small", 8, 2, hp, seed=7)
    from pantheonrl_amd import spaces as sp
    obs_s, act_s = H.CONFIGS["adap_small"]
    env = type("E", (), dict(observation_space=sp.Box(-np.inf, np.inf, (obs_s.dim - 3,)), action_space=H.to_space(act_s),
                             _is_dummy_space_env=True))()
    alt = ADAP("AdapPolicy", env, n_steps=8, n_envs=2, batch_size=8, n_epochs=1, seed=7, sampling_stream=1)
    twin = ADAP("AdapPolicy", env, n_steps=8, n_envs=2, batch_size=8, n_epochs=1, seed=7, sampling_stream=1)
    assert not np.array_equal(ego.policy.get_context(), alt.policy.get_context())     # same seed, different learners
    assert np.array_equal(alt.policy.get_context(), twin.policy.get_context())        # ... yet reproducible
    a = [ego.sample_context(1) for _ in range(4)]
    b = [alt.sample_context(1) for _ in range(4)]
    assert all(not np.array_equal(x, y) for x, y in zip(a, b))
    alt.save(str(tmp_path / "alt"))
    nxt = alt.sample_context(3)
    back = ADAP.load(str(tmp_path / "alt"))
    assert np.array_equal(back.policy.get_context(), alt.policy.get_context())
    assert np.array_equal(back.sample_context(3), nxt)                                # continues, does not restart
