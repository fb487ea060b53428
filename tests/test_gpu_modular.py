"""-m gpu: ModularPolicy / ModularAlgorithm on the device (ph_modular_forward, ph_modular_minibatch_grad, ph_modular_train)
against the oracle's restatement of pantheonrl/algos/modular/policies.py:243-395 and learn.py:221-351.

Tolerances: composed logits / values 2e-5 (as for the plain MlpPolicy); gradients 1e-6 + 2e-4 * max|g| (f32 sums over the
minibatch rows in another order, v_exp / v_rcp tanh); parameters after a chain 2e-6 per optimizer step."""
import ctypes as C

import numpy as np
import pytest
import torch as th

from oracle import sb3_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu

SHAPES = {"mod_oc": (orc.SpaceSpec("box", dim=62), orc.SpaceSpec("discrete", nvec=(6,))),
          "mod_small": (orc.SpaceSpec("box", dim=6), orc.SpaceSpec("discrete", nvec=(5,))),
          "mod_rps": (orc.SpaceSpec("discrete", nvec=(1,)), orc.SpaceSpec("discrete", nvec=(3,))),
          "mod_md": (orc.SpaceSpec("multidiscrete", nvec=(3, 4, 5)), orc.SpaceSpec("discrete", nvec=(8,)))}


def _oracle(name, K, seed=3, perturb=0.3, **kw):
    th.manual_seed(seed)
    obs_s, act_s = SHAPES[name]
    pol = orc.ModularPolicyOracle(obs_s, act_s, num_partners=K, **kw)
    g = th.Generator().manual_seed(seed + 1)
    with th.no_grad():       # biases and the 0.01-gain heads perturbed so that logits and values are not ~0
        seen = set()
        for p in pol.parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            if p.ndim == 1:
                p.add_(perturb * th.randn(p.shape, generator=g))
        for head in [pol.action_net] + [pm["act"] for pm in _unique(pol)]:
            head.weight.add_(perturb * th.randn(head.weight.shape, generator=g))
    return pol


def _unique(pol):
    out, seen = [], set()
    for pm in pol.partners:
        if id(pm) not in seen:
            seen.add(id(pm))
            out.append(pm)
    return out


def _flat(pol, grads=False):
    """the device's parameter order: main network, then every DISTINCT module (baseline shares one)"""
    def vec(t, transpose):
        x = (t.grad if grads else t.detach())
        if x is None:
            x = th.zeros_like(t)
        return (x.t() if transpose else x).contiguous().reshape(-1)
    out = []
    for seq in (pol.policy_net, pol.value_net_mlp):
        for i in (0, 2):
            out += [vec(seq[i].weight, True), vec(seq[i].bias, False)]
    out += [vec(pol.action_net.weight, True), vec(pol.action_net.bias, False), vec(pol.value_net.weight, False),
            vec(pol.value_net.bias, False)]
    for pm in _unique(pol):
        for seq in (pm["pi"], pm["vf"]):
            for i in (0, 2):
                out += [vec(seq[i].weight, True), vec(seq[i].bias, False)]
        out += [vec(pm["act"].weight, True), vec(pm["act"].bias, False), vec(pm["val"].weight, False), vec(pm["val"].bias, False)]
    return th.cat(out).numpy().astype(np.float32).copy()


def _device(name, orac, K, **kw):
    from pantheonrl_amd.modular import ModularPolicy
    obs_s, act_s = SHAPES[name]
    pol = ModularPolicy(H.to_space(obs_s), H.to_space(act_s), device="cuda", seed=0, num_partners=K, **kw)
    flat = _flat(orac)
    assert flat.size == pol.P_total, (flat.size, pol.P_total)
    pol.set_flat_params(flat)
    return pol


def _filled(orac, name, partner, T, E, seed=0):
    rng = np.random.default_rng(seed)
    obs_s, act_s = SHAPES[name]
    buf = orc.RolloutBufferOracle(T, E, obs_s.stored_len, 1)
    es = np.ones(E, np.float32)
    for _ in range(T):
        obs = H.sample_obs(obs_s, E, rng)
        with th.no_grad():
            a, v, lp = orac.forward(th.as_tensor(obs), partner_idx=partner, uniforms=th.as_tensor(rng.random((E, 1)).astype(np.float32)))
        buf.add(obs, a.numpy().astype(np.float32), rng.standard_normal(E).astype(np.float32), es, v.flatten(), lp)
        es = (rng.random(E) < 0.1).astype(np.float32)
    buf.compute_returns_and_advantage(rng.standard_normal(E).astype(np.float32), es)
    return buf


def _device_buffer(pol, name, ob):
    from pantheonrl_amd.ppo import RolloutBuffer
    obs_s, act_s = SHAPES[name]
    buf = RolloutBuffer(ob.T, H.to_space(obs_s), H.to_space(act_s), pol.device, pol.ctx, pol.spec, n_envs=ob.E)
    H.upload_buffer(buf, ob)
    return buf


# ----------------------------------------------------------------------------------------------------------------
# forward family: composed logits and values (policies.py:271-288,325-334,364-395)
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,K,kw", [("mod_oc", 1, {}), ("mod_oc", 2, {}), ("mod_oc", 3, {}), ("mod_small", 2, {"nomain": True}),
                                       ("mod_oc", 3, {"baseline": True}), ("mod_rps", 2, {}), ("mod_md", 2, {})])
@pytest.mark.parametrize("n", [1, 70, 1024])
def test_composed_logits_values_and_distribution_match_the_oracle(name, K, kw, n):
    orac = _oracle(name, K, **kw)
    pol = _device(name, orac, K, **kw)
    rng = np.random.default_rng(n)
    obs = H.sample_obs(SHAPES[name][0], n, rng)
    L = SHAPES[name][1].nvec[0]
    acts = rng.integers(0, L, size=(n, 1))
    mask = (rng.random((n, L)) < 0.7).astype(np.uint8)
    mask[mask.sum(1) == 0, 0] = 1
    for k in range(K):
        with th.no_grad():
            zm_ref, zp_ref = orac.action_logits(th.as_tensor(obs), k)
            v_ref, lp_ref, ent_ref = orac.evaluate_actions(th.as_tensor(obs), th.as_tensor(acts), partner_idx=k)
            vm_ref, lpm_ref, _ = orac.evaluate_actions(th.as_tensor(obs), th.as_tensor(acts), partner_idx=k,
                                                       action_mask=th.as_tensor(mask))
        zm, zp = pol.get_action_logits_from_obs(obs, partner_idx=k)
        assert np.abs(zm.cpu().numpy() - zm_ref.numpy()).max() < 2e-5 and np.abs(zp.cpu().numpy() - zp_ref.numpy()).max() < 2e-5
        v, lp, ent = pol.evaluate_actions(obs, acts, partner_idx=k)
        assert np.abs(v.cpu().numpy() - v_ref.numpy()).max() < 3e-5
        assert np.abs(lp.cpu().numpy() - lp_ref.numpy()).max() < 3e-5 and np.abs(ent.cpu().numpy() - ent_ref.numpy()).max() < 3e-5
        _, lpm, _ = pol.evaluate_actions(obs, acts, partner_idx=k, action_mask=mask)      # policies.py:330-333: -30 * (~mask)
        assert np.abs(lpm.cpu().numpy() - lpm_ref.numpy()).max() < 1e-4
        # sampling with teacher-forced uniforms picks the oracle's inverse-CDF action wherever the uniform is not on a boundary
        u = rng.random((n, 1)).astype(np.float32)
        with th.no_grad():
            a_ref, _, _ = orac.forward(th.as_tensor(obs), partner_idx=k, uniforms=th.as_tensor(u))
        a, _, _ = pol.forward(obs, partner_idx=k, uniforms=u)
        assert (a.cpu().numpy().reshape(-1) == a_ref.numpy().reshape(-1)).mean() > 0.999


def test_forward_and_store_writes_the_partner_rollout_row():
    """collect_rollouts (learn.py:182-207): forward(obs, partner_idx) fused with RolloutBuffer.add into THAT partner's buffer"""
    orac = _oracle("mod_oc", 2)
    pol = _device("mod_oc", orac, 2)
    from pantheonrl_amd.ppo import RolloutBuffer
    obs_s, act_s = SHAPES["mod_oc"]
    E = 37
    rb = RolloutBuffer(4, H.to_space(obs_s), H.to_space(act_s), pol.device, pol.ctx, pol.spec, n_envs=E)
    rng = np.random.default_rng(0)
    obs = H.sample_obs(obs_s, E, rng)
    es = (rng.random(E) < 0.5).astype(np.float32)
    a, v, lp = pol.forward_and_store(obs, rb, es, partner_idx=1)
    h = rb.host()
    assert rb.pos == 1 and np.array_equal(h["observations"][0], obs) and np.array_equal(h["episode_starts"][0], es)
    assert np.array_equal(h["actions"][0, :, 0], a.cpu().numpy().reshape(-1).astype(np.float32))
    assert np.array_equal(h["values"][0], v.cpu().numpy().reshape(-1)) and np.array_equal(h["log_probs"][0], lp.cpu().numpy())
    v_ref, lp_ref, _ = orac.evaluate_actions(th.as_tensor(obs), a.cpu(), partner_idx=1)
    assert np.abs(h["values"][0] - v_ref.detach().numpy().reshape(-1)).max() < 3e-5
    assert np.abs(h["log_probs"][0] - lp_ref.detach().numpy()).max() < 3e-5


# ----------------------------------------------------------------------------------------------------------------
# one minibatch: the gradient of the whole loss (learn.py:244-318) w.r.t. every parameter
# ----------------------------------------------------------------------------------------------------------------
def _grad_pair(name, K, partner, T, E, nb, coef, hp, kw=None, seed=5, gemm_mode=0):
    from pantheonrl_amd import _native as nat
    from pantheonrl_amd.modular import ModularAlgorithm  # noqa: F401  (import check)
    kw = kw or {}
    orac = _oracle(name, K, seed=seed, **kw)
    ob = _filled(orac, name, partner, T, E, seed=seed)
    pol = _device(name, orac, K, **kw)
    pol.gemm_mode = gemm_mode
    buf = _device_buffer(pol, name, ob)
    idx = np.random.default_rng(nb).permutation(T * E)[:nb]
    mb = ob.minibatch(idx) if hasattr(ob, "minibatch") else next(iter(ob.get(nb, idx)))
    for p in orac.parameters():
        p.grad = None
    loss, st_ref = orc.modular_minibatch_loss(orac, mb, hp, partner, coef)
    loss.backward()
    g_ref = _flat(orac, grads=True)
    hpc = nat.PhPpoHyper()
    hpc.learning_rate, hpc.clip_range = hp.learning_rate, hp.clip_range
    hpc.clip_range_vf = -1.0 if hp.clip_range_vf is None else hp.clip_range_vf
    hpc.ent_coef, hpc.vf_coef, hpc.max_grad_norm, hpc.target_kl = hp.ent_coef, hp.vf_coef, hp.max_grad_norm, -1.0
    hpc.normalize_advantage, hpc.adam_beta1, hpc.adam_beta2, hpc.adam_eps = 1, 0.9, 0.999, 1e-5
    grad = th.zeros(pol.P_total, device="cuda")
    stats = th.zeros(nat.PH_NSTAT, device="cuda")
    idx_t = th.as_tensor(idx.astype(np.int32)).cuda()
    pol._bind()
    nat.check(pol.ctx.lib.ph_modular_minibatch_grad(pol.ctx.handle, C.byref(pol.spec), C.byref(pol.mod), pol.params.data_ptr(),
                                                    partner, C.byref(buf.c_struct()), C.byref(hpc), idx_t.data_ptr(), nb,
                                                    float(coef), grad.data_ptr(), stats.data_ptr(), gemm_mode))
    th.cuda.synchronize()
    return grad.cpu().numpy(), g_ref, stats.cpu().numpy(), st_ref, pol


def _assert_grads(g, g_ref, pol, what=""):
    scale = np.abs(g_ref).max()
    err = np.abs(g - g_ref)
    assert err.max() <= 1e-6 + 2e-4 * scale, (what, err.max(), scale, int(err.argmax()), pol.layout.P, pol.module_layout.P)


@pytest.mark.parametrize("name,K,partner,nb,coef,kw", [
    ("mod_small", 1, 0, 64, 0.0, {}), ("mod_small", 2, 0, 64, 0.5, {}), ("mod_small", 2, 1, 64, 0.5, {}),
    ("mod_oc", 3, 1, 200, 0.3, {}), ("mod_oc", 2, 0, 64, 1.0, {"nomain": True}), ("mod_oc", 3, 2, 130, 0.7, {"baseline": True}),
    ("mod_rps", 2, 1, 96, 0.4, {}), ("mod_md", 2, 0, 64, 0.4, {}), ("mod_oc", 2, 1, 1, 0.2, {})])
def test_minibatch_gradient_of_the_full_loss_matches_autograd(name, K, partner, nb, coef, kw):
    hp = orc.PPOHyper(ent_coef=0.01, clip_range_vf=0.3 if nb == 130 else None)
    if nb == 1:
        pytest.skip("a one-row minibatch normalises its advantage with std = nan in the reference (learn.py:261)")
    g, g_ref, st, st_ref, pol = _grad_pair(name, K, partner, 16, 16, nb, coef, hp, kw)
    _assert_grads(g, g_ref, pol, (name, K, partner))
    for i, key in enumerate(("policy_loss", "value_loss", "entropy_loss")):
        assert abs(st[i] - st_ref[key]) <= 1e-5 + 1e-4 * abs(st_ref[key]), (key, st[i], st_ref[key])
    assert abs(st[4] - st_ref["approx_kl"]) <= 1e-5 and abs(st[7] - st_ref["marginal_reg"]) <= 1e-5
    assert abs(st[5] - st_ref["loss"]) <= 2e-5 + 1e-4 * abs(st_ref["loss"])
    # parameters the loss does not reach have exactly zero gradient: the value side of the modules that are not trained
    ml, off = pol.module_layout, pol.layout.P
    for m in range(pol.n_modules):
        if m == int(pol.mod.module_of[partner]):
            continue
        base = off + m * ml.P
        assert not g[base + ml.vf_W1:base + ml.act_W].any() and not g[base + ml.val_W:base + ml.P].any()
        if coef > 0:
            assert np.abs(g[base:base + ml.vf_W1]).max() > 0        # ... while the regulariser reaches their policy side


def test_minibatch_gradient_at_the_bench_size_and_valu_cross_check():
    """32 768 rows of Overcooked shapes, two partners, both gradient products paths (MFMA tiles; gemm_mode 1 = the k-ordered
    fmaf restatement of the same tiles: bitwise the same slabs)"""
    hp = orc.PPOHyper()
    g, g_ref, st, st_ref, pol = _grad_pair("mod_oc", 2, 1, 128, 1024, 32768, 0.5, hp)
    _assert_grads(g, g_ref, pol, "32768 rows")
    g_small, g_ref_small, _, _, pol = _grad_pair("mod_oc", 2, 1, 16, 16, 200, 0.5, hp)
    g_valu, _, _, _, _ = _grad_pair("mod_oc", 2, 1, 16, 16, 200, 0.5, hp, gemm_mode=1)
    assert np.array_equal(g_small, g_valu)


def test_marginal_regulariser_known_answer_on_the_device():
    """learn.py:309-316 by hand (tests/test_oracle.py's case): main logits 0 -> uniform over 4 actions; two partners whose
    logits are the constants d1, d2 -> every row's term is sum_a | 1/4 - (softmax(d1) + softmax(d2)) / 2 |"""
    from pantheonrl_amd import _native as nat
    from pantheonrl_amd.modular import ModularPolicy
    from pantheonrl_amd.ppo import RolloutBuffer
    from pantheonrl_amd import spaces as sp
    obs_space, act_space = sp.Box(-np.inf, np.inf, (3,)), sp.Discrete(4)
    pol = ModularPolicy(obs_space, act_space, device="cuda", seed=0, num_partners=2)
    flat = pol.get_flat_params()
    lay, ml = pol.layout, pol.module_layout
    d = [np.array([0.5, -0.2, 0.1, 0.0], np.float32), np.array([-1.0, 0.3, 0.0, 0.7], np.float32)]
    flat[lay.act_W:lay.act_W + 64 * 4] = 0
    flat[lay.act_b:lay.act_b + 4] = 0
    for m in range(2):
        base = pol.module_offset(m)
        flat[base + ml.act_W:base + ml.act_W + 64 * 4] = 0
        flat[base + ml.act_b:base + ml.act_b + 4] = d[m]
    pol.set_flat_params(flat)
    T, E = 4, 8
    rb = RolloutBuffer(T, obs_space, act_space, pol.device, pol.ctx, pol.spec, n_envs=E)
    rng = np.random.default_rng(0)
    rb.observations.copy_(th.as_tensor(rng.standard_normal((T, E, 3)).astype(np.float32)))
    rb.advantages.copy_(th.as_tensor(rng.standard_normal((T, E)).astype(np.float32)))
    rb.log_probs.fill_(-1.3)
    hpc = nat.PhPpoHyper()
    hpc.learning_rate, hpc.clip_range, hpc.clip_range_vf, hpc.vf_coef, hpc.max_grad_norm = 3e-4, 0.2, -1.0, 0.5, 0.5
    hpc.target_kl, hpc.normalize_advantage, hpc.adam_beta1, hpc.adam_beta2, hpc.adam_eps = -1.0, 1, 0.9, 0.999, 1e-5
    grad, stats = th.zeros(pol.P_total, device="cuda"), th.zeros(nat.PH_NSTAT, device="cuda")
    idx = th.arange(T * E, dtype=th.int32, device="cuda")
    pol._bind()
    nat.check(pol.ctx.lib.ph_modular_minibatch_grad(pol.ctx.handle, C.byref(pol.spec), C.byref(pol.mod), pol.params.data_ptr(), 0,
                                                    C.byref(rb.c_struct()), C.byref(hpc), idx.data_ptr(), T * E, 1.0,
                                                    grad.data_ptr(), stats.data_ptr(), 0))
    sm = [np.exp(x.astype(np.float64)) / np.exp(x.astype(np.float64)).sum() for x in d]
    want = np.abs(0.25 - (sm[0] + sm[1]) / 2).sum()
    assert abs(float(stats[7].item()) - want) < 1e-6


# ----------------------------------------------------------------------------------------------------------------
# ModularAlgorithm.train (learn.py:221-351): partner by partner, one buffer each, one clip + Adam over everything
# ----------------------------------------------------------------------------------------------------------------
def _algo(name, K, T, E, hp, coef, kw=None, seed=0):
    from pantheonrl_amd.modular import ModularAlgorithm
    obs_s, act_s = SHAPES[name]
    env = type("E", (), dict(observation_space=H.to_space(obs_s), action_space=H.to_space(act_s), _is_dummy_space_env=True))()
    return ModularAlgorithm("ModularPolicy", env, n_steps=T, n_envs=E, batch_size=hp.batch_size, n_epochs=hp.n_epochs,
                            learning_rate=hp.learning_rate, clip_range=hp.clip_range, clip_range_vf=hp.clip_range_vf,
                            ent_coef=hp.ent_coef, vf_coef=hp.vf_coef, max_grad_norm=hp.max_grad_norm, target_kl=hp.target_kl,
                            seed=seed, marginal_reg_coef=coef, policy_kwargs=dict(num_partners=K, **(kw or {})))


@pytest.mark.parametrize("name,K,T,E,batch,epochs,coef,kw", [
    ("mod_small", 2, 8, 8, 16, 2, 0.3, {}), ("mod_oc", 2, 16, 8, 64, 3, 0.5, {}), ("mod_oc", 3, 8, 8, 40, 2, 0.0, {}),
    ("mod_small", 3, 8, 8, 32, 2, 0.4, {"baseline": True}), ("mod_small", 2, 8, 8, 32, 2, 0.4, {"nomain": True})])
def test_modular_train_chain_matches_the_oracle(name, K, T, E, batch, epochs, coef, kw):
    """the whole update chain, twice (a second train() continues every parameter's own Adam step count: a partner's value side
    joined the optimizer when that partner was first trained, torch 1.13 zero_grad leaves zero gradients behind)"""
    hp = orc.PPOHyper(batch_size=batch, n_epochs=epochs, ent_coef=0.01)
    orac = _oracle(name, K, seed=7, **kw)
    model = _algo(name, K, T, E, hp, coef, kw)
    model.policy.set_flat_params(_flat(orac))
    n_steps = 0
    for round_ in range(2):
        bufs = [_filled(orac, name, k, T, E, seed=20 + 3 * round_ + k) for k in range(K)]
        for rb, ob in zip(model.rollout_buffer, bufs):
            H.upload_buffer(rb, ob)
        perms = [[np.random.default_rng(100 * round_ + 10 * k + ep).permutation(T * E) for ep in range(epochs)] for k in range(K)]
        st_ref = orc.modular_train(orac, bufs, hp, coef, perms=perms)
        model.train(perms=np.asarray(perms))
        n_steps += len(st_ref)
        d = np.abs(model.policy.get_flat_params() - _flat(orac)).max()
        assert d <= 2e-6 * n_steps, (round_, d, n_steps)
        st = model.last_train_stats.reshape(-1, 8)
        assert len(st) == len(st_ref)
        for row, ref in zip(st, st_ref):
            assert abs(row[0] - ref["policy_loss"]) <= 2e-5 + 2e-4 * abs(ref["policy_loss"])
            assert abs(row[1] - ref["value_loss"]) <= 2e-5 + 2e-4 * abs(ref["value_loss"])
            assert abs(row[7] - ref["marginal_reg"]) <= 2e-5 and abs(row[6] - ref["grad_norm"]) <= 1e-5 + 2e-4 * ref["grad_norm"]
    assert int(model.policy.opt_step.item()) == n_steps
    first = model.policy.mod_first.cpu().numpy()
    per_partner = epochs * (-(-T * E // batch))
    assert list(first) == ([0] if kw.get("baseline") else [k * per_partner for k in range(K)])


def test_target_kl_ends_a_partner_epochs_after_a_whole_epoch():
    """learn.py:320-334: every minibatch of an epoch steps; the mean of that epoch's KLs is tested afterwards and ends THIS
    partner's epochs only -- the next partner starts afresh"""
    hp = orc.PPOHyper(batch_size=16, n_epochs=4, learning_rate=3e-3, target_kl=1e-7)
    orac = _oracle("mod_small", 2, seed=9)
    model = _algo("mod_small", 2, 8, 8, hp, 0.2)
    model.policy.set_flat_params(_flat(orac))
    bufs = [_filled(orac, "mod_small", k, 8, 8, seed=40 + k) for k in range(2)]
    for rb, ob in zip(model.rollout_buffer, bufs):
        H.upload_buffer(rb, ob)
    perms = [[np.random.default_rng(10 * k + ep).permutation(64) for ep in range(4)] for k in range(2)]
    st_ref = orc.modular_train(orac, bufs, hp, 0.2, perms=perms)
    model.train(perms=np.asarray(perms))
    ran = (np.abs(model.last_train_stats).sum(-1) > 0)            # (2 partners, 16 minibatch slots)
    per_partner = [sum(1 for s in st_ref if s["partner"] == k) for k in range(2)]
    assert [int(r.sum()) for r in ran] == per_partner and all(p % 4 == 0 and p < 16 for p in per_partner)
    assert np.abs(model.policy.get_flat_params() - _flat(orac)).max() <= 2e-5
    assert int(model.policy.opt_step.item()) == len(st_ref)


def test_modular_save_load_and_trainer_object_graph(tmp_path, monkeypatch):
    """`trainer.py RPS-v0 ModularAlgorithm PPO PPO` (trainer.py:131-135): one module per partner, one rollout buffer per
    partner filled with that partner in the seat (set_partnerid), both partners learn; save / load / LOAD with a new partner set"""
    from pantheonrl_amd import trainer
    from pantheonrl_amd.modular import ModularAlgorithm
    monkeypatch.chdir(tmp_path)
    cfg = '{"n_steps": 16, "batch_size": 16, "n_epochs": 2, "marginal_reg_coef": 0.5}'
    alt = '{"n_steps": 16, "batch_size": 16, "n_epochs": 1}'
    ego, partners, env = trainer.run(["RPS-v0", "ModularAlgorithm", "PPO", "PPO", "--seed", "3", "-t", "64", "--ego-config", cfg,
                                      "--alt-config", alt, "--alt-config", alt, "--ego-save", "m/ego"])
    assert isinstance(ego, ModularAlgorithm) and ego.policy.num_partners == 2 and len(ego.rollout_buffer) == 2
    assert ego.num_timesteps == 64 and ego._n_updates == 4 and int(ego.policy.opt_step.item()) == 2 * 2 * 2
    assert all(p.n_steps > 0 or p.model.rollout_buffer.pos > 0 or p.model._n_updates > 0 for p in partners)   # each sat in the seat
    assert list(ego.policy.mod_first.cpu().numpy()) == [0, 2]
    loaded = ModularAlgorithm.load("m/ego")
    assert np.array_equal(loaded.policy.get_flat_params(), ego.policy.get_flat_params())
    assert list(loaded.policy.mod_first.cpu().numpy()) == [0, 2] and loaded.marginal_reg_coef == 0.5
    sd = ego.policy.state_dict()
    assert "partner_mlp_extractor.1.policy_net.0.weight" in sd and sd["partner_action_net.0.weight"].shape == (3, 64)
    # LOAD as the ego of a run with three partners: the main network is kept, the modules are drawn afresh (trainer.py:121-123)
    ego2, _, _ = trainer.run(["RPS-v0", "LOAD", "PPO", "PPO", "PPO", "--seed", "4", "-t", "48", "--ego-config",
                              '{"type": "ModularAlgorithm", "location": "m/ego"}', "--alt-config", alt, "--alt-config", alt,
                              "--alt-config", alt])
    assert ego2.policy.num_partners == 3 and len(ego2.rollout_buffer) == 3 and ego2.num_timesteps >= 48
