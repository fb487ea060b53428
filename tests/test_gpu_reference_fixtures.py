"""gpu: the reference-generated fixtures (tests/golden/ref_*.json, ref_adap_context.npz -- produced by the REFERENCE's own Python text,
tests/golden/make_reference_fixtures.py) replayed through the DEVICE path: `pantheonrl_amd.common` classes around gfx950-backed PPO
learners, and `ph_adap_minibatch_grad` for ADAP's context term.

What the policy samples differs from what the reference's recording model scripted, so the comparison is on everything that does not
depend on the sampled action: the stream the ego sees, the callback sequence every partner receives, the rows (observations,
episode starts, late additive rewards -- f32 adds in the reference's order, bit-exact) each device buffer holds when GAE is asked
for, what GAE is handed (the cached values of the previous call, the last update's done), and the train / reset order.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch as th

from tests import refdrive as rd

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def _device_ppo(D, n_act, n_steps, seed):
    from pantheonrl_amd import PPO
    from pantheonrl_amd import spaces as sp
    env = type("E", (), dict(observation_space=sp.Box(-np.inf, np.inf, (D,)), action_space=sp.Discrete(n_act),
                             _is_dummy_space_env=True))()
    return PPO("MlpPolicy", env, n_steps=n_steps, batch_size=n_steps, n_epochs=1, seed=seed)


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, th.Tensor) else np.asarray(x)


class _Tap:
    """a partner seat that logs in RecordingAgent's format and forwards to the real device learner"""

    def __init__(self, agent):
        self.agent, self.log, self.acts = agent, [], []

    def get_action(self, obs, record=True):
        self.log.append(["act", rd.plain(obs.obs), rd.plain(obs.state), rd.plain(obs.action_mask), bool(record)])
        a = self.agent.get_action(obs, record)
        self.acts.append(int(a))
        return a

    def update(self, reward, done):
        self.log.append(["upd", rd.plain(reward), bool(done)])
        self.agent.update(reward, done)


def _rows_from_log(log):
    """the buffer rows the reference's OnPolicyAgent would hold after this callback sequence (agents.py:172-179,197-198): per recorded
    action its observation, the done of the update before it (first row: True, :98), and the f32 running sum of its rewards"""
    obs, starts, rew, last = [], [], [], True
    for e in log:
        if e[0] == "act":
            obs.append(e[1])
            starts.append(float(last))
            rew.append(np.float32(0))
        else:
            rew[-1] = np.float32(rew[-1] + np.float32(e[1]))
            last = e[2]
    return np.asarray(obs, np.float32), np.asarray(starts, np.float32), np.asarray(rew, np.float32)


def test_round_robin_partners_on_the_device_receive_the_reference_callbacks():
    """BASELINE config 4's host logic (multiagentenv.py:118-125,149-243) with three gfx950-backed OnPolicyAgent partners: the ego's
    stream and every partner's callback log equal the reference-generated ones; each device buffer then holds exactly the rows that
    callback sequence implies"""
    from pantheonrl_amd.common import OnPolicyAgent, SimultaneousEnv
    ref = _json("ref_multiagent.json")["simultaneous"]
    T, K = 120, 3

    class Env(rd._SimGame, SimultaneousEnv):
        def __init__(self):
            SimultaneousEnv.__init__(self)
            rd._SimGame.__init__(self, 0, T)

    env = Env()
    taps = [_Tap(OnPolicyAgent(_device_ppo(3, 4, 64, seed=k))) for k in range(K)]
    for t in taps:
        env.add_partner_agent(t)
    ego = json.loads(json.dumps(rd._ego_loop(env, T)))
    assert ego == ref["ego"]
    for k, t in enumerate(taps):
        assert json.loads(json.dumps(t.log)) == ref["partners"][k], f"partner {k}"
        obs, starts, rew = _rows_from_log(ref["partners"][k])
        n = len(obs)
        assert 10 < n <= 64 and t.agent.model.rollout_buffer.pos == n and t.agent.iteration == 0
        got = t.agent.model.rollout_buffer.host()
        assert np.array_equal(got["observations"][:n, 0], obs)
        assert np.array_equal(got["episode_starts"][:n, 0], starts)
        assert np.array_equal(got["rewards"][:n, 0], rew), (got["rewards"][:n, 0] - rew)
        assert np.array_equal(got["actions"][:n, 0, 0].astype(int), np.asarray(t.acts))
        assert 0 <= min(t.acts) and max(t.acts) < 4
    # the joint action the game received: the ego's 0 and whatever the seated partner played
    flat = [a for t in taps for a in t.acts]
    assert sorted(j[1] for j in env.joint) == sorted(flat)


def test_onpolicy_agent_on_the_device_follows_the_reference_schedule():
    """agents.py:111-203 with a gfx950-backed PPO under the fixture's stream of (observation, updates): at every buffer fill GAE is
    handed the PREVIOUS call's values and the last update's done, the buffer holds the reference's rows, train and reset follow in
    that order, and the agent's counters / ep_info_buffer equal the reference's after every call"""
    from pantheonrl_amd.common import Observation, OnPolicyAgent
    kw = rd.RECORDED_ONLY
    ref = _json("ref_onpolicy_agent.json")["recorded_only"]
    model = _device_ppo(kw["D"], 3, kw["n_steps"], seed=0)
    agent = OnPolicyAgent(model, tb_log_name="fixture_agent")
    buf, events = model.rollout_buffer, []
    real_gae, real_train, real_reset = buf.compute_returns_and_advantage, model.train, buf.reset

    def gae(last_values, dones):
        th.cuda.synchronize()
        h = buf.host()
        events.append(["gae", _np(last_values).copy(), dones, h["rewards"][:, 0].copy(), h["observations"][:, 0].copy(),
                       h["episode_starts"][:, 0].copy()])
        real_gae(last_values=last_values, dones=dones)

    def train(*a, **k):
        events.append(["train", buf.host()["advantages"][:, 0].copy()])
        real_train(*a, **k)

    def reset():
        events.append(["reset"])
        real_reset()
    buf.compute_returns_and_advantage, model.train, buf.reset = gae, train, reset

    ref_ev = ref["events"]
    ref_states = [e for e in ref_ev if e[0] == "state"]
    ref_updated = [e for e in ref_ev if e[0] == "updated"]
    ref_adds = [e for e in ref_ev if e[0] == "add"]
    values_seen, n_upd = [], 0
    for i, (obs, record, upd) in enumerate(rd.onpolicy_script(kw["seed"], kw["n_calls"], kw["D"], kw["p_skip"])):
        events.append(["get_action", record])
        act = agent.get_action(Observation(obs), record=record)
        assert 0 <= int(act) < 3
        values_seen.append(_np(agent.values).reshape(-1).copy())
        st = ref_states[i]
        assert [agent.n_steps, agent.num_timesteps, agent.iteration] == st[1:4] and rd.plain(agent._last_episode_starts) == st[4]
        assert rd.plain(list(model.ep_info_buffer)) == st[6]
        for r, d in upd:
            agent.update(r, d)
            u = ref_updated[n_upd]
            n_upd += 1
            th.cuda.synchronize()
            assert np.array_equal(buf.rewards[:, 0].cpu().numpy(), np.asarray(u[3], np.float32)), (i, u)
            assert rd.plain(agent._last_episode_starts) == u[4] and rd.plain(list(model.ep_info_buffer)) == u[5]
    skeleton = [e[0] for e in events]
    assert skeleton == [e[0] for e in ref_ev if e[0] in ("get_action", "gae", "train", "reset")]
    assert skeleton.count("train") == 5
    gaes, ref_gaes, call = [e for e in events if e[0] == "gae"], [e for e in ref_ev if e[0] == "gae"], -1
    fill = 0
    for e in events:
        if e[0] == "get_action":
            call += 1
        if e[0] != "gae":
            continue
        r = ref_gaes[fill]
        assert np.array_equal(e[1].reshape(-1), values_seen[call - 1])                 # D-1: V of the PREVIOUS observation
        assert e[2] == r[2] and isinstance(e[2], bool)                                # the last update's done
        assert np.array_equal(e[3], np.asarray(r[3], np.float32))                     # rewards, bit for bit
        rows = ref_adds[fill * kw["n_steps"]:(fill + 1) * kw["n_steps"]]
        assert np.array_equal(e[4], np.asarray([a[2][0] for a in rows], np.float32))
        assert np.array_equal(e[5], np.asarray([float(a[5][0]) for a in rows], np.float32))
        fill += 1
    assert fill == len(gaes) == len(ref_gaes)
    adv = [e[1] for e in events if e[0] == "train"]
    assert all(np.isfinite(a).all() and np.abs(a).max() > 0 for a in adv)             # GAE ran before train


ADAP_SPEC = {"discrete6": ("discrete", (6,)), "multi_7_12": ("multidiscrete", (7, 12)), "two_contexts": ("discrete", (5,))}


@pytest.mark.parametrize("name", list(ADAP_SPEC))
def test_device_context_term_matches_the_reference_function(name):
    """`ph_adap_minibatch_grad` against the REFERENCE's get_context_kl_loss (adap/util.py:97-131) on the states and contexts the
    reference drew: the reported loss within 1e-5, and coef * d(loss)/d(params) -- isolated as g(coef) - g(0) -- within
    2e-6 + 2e-4 * coef * max|g_ref| (f32 sums in another order; v_exp / v_rcp tanh and softmax)"""
    from pantheonrl_amd import _native as nat
    from pantheonrl_amd import spaces as sp
    from pantheonrl_amd.ppo import PPO, ActorCriticPolicy, RolloutBuffer
    from oracle import sb3_oracle as orc
    from tests.golden.make_reference_fixtures import ADAP_CASES
    c = ADAP_CASES[name]
    z = np.load(os.path.join(GOLDEN, "ref_adap_context.npz"))
    obs, params = z[f"{name}.observations"], z[f"{name}.params"]
    sidx, ctxs = z[f"{name}.state_idx"], z[f"{name}.contexts"]
    B, cs = c["B"], c["ctx"]
    obs_space = sp.Box(-np.inf, np.inf, (c["F"] + cs,))
    act_space = sp.Discrete(c["nvec"][0]) if len(c["nvec"]) == 1 else sp.MultiDiscrete(list(c["nvec"]))
    pol = ActorCriticPolicy(obs_space, act_space, device="cuda", seed=0)
    pol.set_flat_params(params)
    buf = RolloutBuffer(B, obs_space, act_space, pol.device, pol.ctx, pol.spec, n_envs=1)      # env-major flat row = t
    rng = np.random.default_rng(0)
    buf.observations.copy_(th.as_tensor(obs).reshape(B, 1, -1))
    buf.actions.copy_(th.as_tensor(np.stack([rng.integers(0, n, B) for n in c["nvec"]], 1).astype(np.float32)).reshape(B, 1, -1))
    for k in ("advantages", "returns", "values", "log_probs"):
        getattr(buf, k).copy_(th.as_tensor(rng.standard_normal((B, 1)).astype(np.float32) * (0.1 if k == "log_probs" else 1.0)))
    buf.pos, buf.full = B, True
    model = PPO.__new__(PPO)
    for k, v in vars(orc.PPOHyper()).items():
        setattr(model, k, v)
    hp = PPO.hyper(model)
    idx = th.arange(B, dtype=th.int32, device="cuda")
    coef = 5.0

    def grad(coefficient):
        ad = nat.PhAdapLoss()
        ad.context_size, ad.num_context_samples, ad.num_state_samples = cs, c["n_ctx"], c["n_states"]
        ad.sampler, ad.context_loss_coeff, ad.seed = nat.CONTEXT_SAMPLERS[c["sampler"]], coefficient, 0
        s = np.full(c["n_states"], -1, np.int32)
        s[:len(sidx)] = sidx
        keep = [th.as_tensor(s[None]).cuda(), th.as_tensor(np.ascontiguousarray(ctxs[None], np.float32)).cuda(), th.zeros(1, device="cuda")]
        ad.state_idx, ad.contexts, ad.context_loss = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
        g = th.zeros(pol.layout.P, device="cuda")
        st = th.zeros(nat.PH_NSTAT, device="cuda")
        pol._bind()
        nat.check(pol.ctx.lib.ph_adap_minibatch_grad(pol.ctx.handle, C.byref(pol.spec), pol.params.data_ptr(), C.byref(buf.c_struct()),
                                                     C.byref(hp), idx.data_ptr(), B, g.data_ptr(), st.data_ptr(), 0, C.byref(ad)))
        th.cuda.synchronize()
        return g.cpu().numpy(), float(keep[2].item())

    g0, _ = grad(0.0)
    g1, loss = grad(coef)
    assert abs(loss - float(z[f"{name}.loss"])) <= 1e-5, (loss, float(z[f"{name}.loss"]))
    g_ref = coef * z[f"{name}.grad"]
    err = np.abs((g1 - g0) - g_ref)
    assert err.max() <= 2e-6 + 2e-4 * np.abs(g_ref).max(), (err.max(), np.abs(g_ref).max())
    assert np.abs(g_ref).max() > 1e-3


# ---- the reference's PPO update loop on the device (tests/golden/ref_ppo_train.npz) --------------------------------------------------------
TRAIN_NAMES = ["ppo_discrete6", "ppo_clipvf_klstop", "adap_discrete6", "adap_multi_7_12"]


@pytest.mark.parametrize("gemm_mode", [0, 2])
@pytest.mark.parametrize("name", TRAIN_NAMES)
def test_device_train_matches_the_reference_train_loop(name, gemm_mode):
    """`ph_ppo_train` / `ph_adap_train` (through pantheonrl_amd.PPO.train / ADAP.train, both product matrix-product modes) against what the
    REFERENCE's own ADAP.train text (pantheonrl/algos/adap/adap_learn.py:229-371 -- the in-tree copy of SB3's PPO.train loop -- over
    AdapPolicy.evaluate_actions, adap/policies.py:97-135) did to the same parameters on the same buffer, index orders and, for the
    context term, the states and contexts the reference drew.  context_loss_coeff = 0 cases run as PLAIN PPO on the device.
    Tolerances: parameters 2e-6 per optimizer step + 1e-6 (f32 sums in another order, v_exp / v_rcp tanh and softmax; Adam
    normalises, so one step moves a parameter by about lr whatever the gradient's scale); logged means 2e-4 + 2e-3 relative;
    the number of optimizer steps (KL early stop) exact."""
    from pantheonrl_amd import PPO
    from pantheonrl_amd import spaces as sp
    from pantheonrl_amd.adap import ADAP
    from tests.golden.make_reference_fixtures import TRAIN_CASES
    c = TRAIN_CASES[name]
    z = np.load(os.path.join(GOLDEN, "ref_ppo_train.npz"))
    g = {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + ".")}
    cs, T, E = c["ctx"], c["T"], c["E"]
    act_space = sp.Discrete(c["nvec"][0]) if len(c["nvec"]) == 1 else sp.MultiDiscrete(list(c["nvec"]))
    kw = dict(n_steps=T, n_envs=E, batch_size=c["batch"], n_epochs=c["epochs"], learning_rate=c["lr"], clip_range=c["clip"],
              clip_range_vf=c["clip_vf"], ent_coef=c["ent"], vf_coef=c["vf"], max_grad_norm=c["max_norm"], target_kl=c["target_kl"], seed=0)
    adap = c["coef"] != 0
    if adap:    # the environment's observation is the features; the stored rows carry the context behind them (adap_learn.py:448-452)
        env = type("E", (), dict(observation_space=sp.Box(-np.inf, np.inf, (c["F"],)), action_space=act_space, _is_dummy_space_env=True))()
        model = ADAP("AdapPolicy", env, context_loss_coeff=c["coef"], context_size=cs, num_context_samples=c["n_ctx"],
                     num_state_samples=c["n_states"], context_sampler=c["sampler"], **kw)
    else:
        env = type("E", (), dict(observation_space=sp.Box(-np.inf, np.inf, (c["F"] + cs,)), action_space=act_space,
                                 _is_dummy_space_env=True))()
        model = PPO("MlpPolicy", env, **kw)
    model.policy.gemm_mode = gemm_mode
    model.policy.set_flat_params(g["params0"])
    rb = model.rollout_buffer
    for k in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns"):
        getattr(rb, k).copy_(th.as_tensor(g[k]))
    rb.pos, rb.full = rb.buffer_size, True
    n_mb = -(-T * E // c["batch"])
    n_seen, n_steps = int(g["n_minibatches_seen"]), int(g["n_steps"])
    if adap:
        sidx = np.zeros((c["epochs"] * n_mb, c["n_states"]), np.int32)
        sidx[:n_seen] = np.where(g["state_idx"] >= 0, g["state_idx"], 0)
        ctxs = np.zeros((c["epochs"] * n_mb, c["n_ctx"], cs), np.float32)
        ctxs[:n_seen] = g["contexts"]
        model.train(perms=g["perms"], state_idx=sidx, contexts=ctxs)
    else:
        model.train(perms=g["perms"])
    st = model.last_train_stats
    assert int((st[:, 7] > 0).sum()) == n_steps, (st[:, 7], n_steps)                      # optimizer steps applied: exact
    p = model.policy.get_flat_params()
    err = np.abs(p - g["params_final"]).max()
    assert err <= 2e-6 * n_steps + 1e-6, err
    assert np.abs(g["params_final"] - g["params0"]).max() > 1e-3
    logged = model.logger.name_to_value
    keys = ["train/entropy_loss", "train/policy_gradient_loss", "train/value_loss", "train/clip_fraction", "train/approx_kl", "train/loss"]
    if adap:
        keys.append("train/context_kl_loss")
    for k in keys:
        want = float(g["log." + k])
        tol = 2e-4 + 2e-3 * abs(want)
        if k == "train/clip_fraction":                                                    # a count of rows over a threshold: one row may flip
            tol = 1.0 / c["batch"] / max(n_seen, 1) + 1e-6
        assert abs(float(logged[k]) - want) <= tol, (k, float(logged[k]), want)
    assert logged["train/n_updates"] == int(g["log.train/n_updates"])


# ---- the reference's ModularPolicy / ModularAlgorithm.train text on the device (tests/golden/ref_modular.npz) ---------------------------------
MODULAR_NAMES = ["two_partners", "three_partners_klstop", "nomain"]


@pytest.mark.parametrize("name", MODULAR_NAMES)
def test_device_modular_train_matches_the_reference_train_loop(name):
    """`ph_modular_train` (through pantheonrl_amd.ModularAlgorithm.train) against what the REFERENCE's ModularAlgorithm.train text
    (pantheonrl/algos/modular/learn.py:221-351) did over its own ModularPolicy text (modular/policies.py:57-395) to the same parameters,
    per-partner buffers and index orders: how many optimizer steps every partner got (the per-epoch KL rule, exact), the parameters
    after the last one within 2e-6 per step, the three logged means within 2e-4 + 2e-3 relative."""
    from pantheonrl_amd import spaces as sp
    from pantheonrl_amd.modular import ModularAlgorithm
    from tests.golden.make_reference_fixtures import MODULAR_CASES
    c = MODULAR_CASES[name]
    z = np.load(os.path.join(GOLDEN, "ref_modular.npz"))
    g = {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + ".")}
    K, T, E = c["K"], c["T"], c["E"]
    env = type("E", (), dict(observation_space=sp.Box(-np.inf, np.inf, (c["D"],)), action_space=sp.Discrete(c["n_act"]),
                             _is_dummy_space_env=True))()
    model = ModularAlgorithm("ModularPolicy", env, n_steps=T, n_envs=E, batch_size=c["batch"], n_epochs=c["epochs"], learning_rate=c["lr"],
                             clip_range=c["clip"], clip_range_vf=c["clip_vf"], ent_coef=c["ent"], vf_coef=c["vf"],
                             max_grad_norm=c["max_norm"], target_kl=c["target_kl"], seed=0, marginal_reg_coef=c["coef"],
                             policy_kwargs=dict(num_partners=K, **c["kw"]))
    assert g["params0"].size == model.policy.P_total
    model.policy.set_flat_params(g["params0"])
    for k, rb in enumerate(model.rollout_buffer):
        for f in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns"):
            getattr(rb, f).copy_(th.as_tensor(g[f"buf{k}.{f}"]))
        rb.pos, rb.full = rb.buffer_size, True
    model.train(perms=g["perms"])
    n_steps, n_mb = int(g["n_steps"]), -(-T * E // c["batch"])
    st = model.last_train_stats                                                            # (K, n_epochs * n_mb, 8); rows that ran are non-zero
    ran = [int((np.abs(st[k]).sum(1) > 0).sum()) for k in range(K)]
    assert ran == [int((g["step_partner"] == k).sum()) for k in range(K)], (ran, g["step_partner"])
    assert [r // n_mb for r in ran] == g["epochs_run"].tolist()
    assert int(model.policy.opt_step.item()) == n_steps
    err = np.abs(model.policy.get_flat_params() - g["params_final"]).max()
    assert err <= 2e-6 * n_steps + 1e-6, err
    logged = model.logger.name_to_value
    for key in ("train/entropy_loss", "train/policy_gradient_loss", "train/value_loss"):
        want = float(g["log." + key])
        assert abs(float(logged[key]) - want) <= 2e-4 + 2e-3 * abs(want), (key, float(logged[key]), want)


@pytest.mark.parametrize("name", MODULAR_NAMES)
def test_device_modular_forward_matches_the_reference_policy_text(name):
    """`ph_modular_forward` (through ModularPolicy.evaluate_actions / get_action_logits_from_obs on the device) against the reference's
    ModularPolicy.evaluate_actions -- with and without the action mask of policies.py:330-333 -- and get_action_logits_from_obs on the
    same parameters and rows: values, log-probabilities, entropies and both logit sets within 2e-5 (the tolerance of the forward
    tests against the oracle: v_exp / v_rcp tanh and softmax)"""
    from pantheonrl_amd import spaces as sp
    from pantheonrl_amd.modular import ModularPolicy
    from tests.golden.make_reference_fixtures import MODULAR_CASES
    c = MODULAR_CASES[name]
    z = np.load(os.path.join(GOLDEN, "ref_modular.npz"))
    g = {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + ".")}
    pol = ModularPolicy(sp.Box(-np.inf, np.inf, (c["D"],)), sp.Discrete(c["n_act"]), device="cuda", seed=0, num_partners=c["K"], **c["kw"])
    assert g["params0"].size == pol.P_total
    pol.set_flat_params(g["params0"])
    obs, acts, mask = g["fwd.obs"], g["fwd.actions"].astype(np.float32).reshape(-1, 1), g["fwd.mask"]
    for k in range(c["K"]):
        v, lp, ent = pol.evaluate_actions(obs, acts, partner_idx=k)
        vm, lpm, entm = pol.evaluate_actions(obs, acts, partner_idx=k, action_mask=mask)
        zm, zp = pol.get_action_logits_from_obs(obs, partner_idx=k)
        for key, val in (("values", v), ("log_prob", lp), ("entropy", ent), ("masked_log_prob", lpm), ("masked_entropy", entm),
                         ("main_logits", zm), ("partner_logits", zp)):
            want = g[f"fwd.{key}.{k}"]
            got = _np(val).reshape(want.shape)
            assert np.abs(got - want).max() <= 2e-5, (k, key, np.abs(got - want).max())
        assert np.array_equal(_np(v), _np(vm))                                             # the mask does not touch the value


# ---- the reference's BC text on the device (tests/golden/ref_bc.npz) -------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["discrete6", "l2_ragged"])
def test_device_bc_train_matches_the_reference_bc_text(name):
    """`ph_bc_train` (through pantheonrl_amd.bc.BC.train) against what the REFERENCE's BC.train / _calculate_loss text (pantheonrl/algos/
    bc.py:270-353) did over torch's DataLoader in the batch order it drew: every batch's seven statistics within 2e-5 + 2e-4 relative.
    Parameters: Adam with torch's defaults (lr 1e-3, eps 1e-8 -- what the reference's constructor builds) moves an entry by ~lr per
    step whatever its gradient, so f32 noise in a near-zero gradient entry can flip that entry's step: median |d| <= 2e-6, at most 1 %
    of the entries beyond 1e-4, none beyond 2.5 steps (the bound of tests/test_gpu_bc.py against the oracle)."""
    from pantheonrl_amd import spaces as sp
    from pantheonrl_amd.bc import BC
    from pantheonrl_amd.common import TransitionsMinimal
    from tests.golden.make_reference_fixtures import BC_CASES
    c = BC_CASES[name]
    z = np.load(os.path.join(GOLDEN, "ref_bc.npz"))
    g = {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + ".")}
    clone = BC(sp.Box(-np.inf, np.inf, (c["D"],)), sp.Discrete(c["nvec"][0]), expert_data=TransitionsMinimal(g["obs"], g["acts"]),
               ent_weight=c["ent"], l2_weight=c["l2"])
    clone.policy.set_flat_params(g["params0"])
    st = clone.train(n_epochs=c["epochs"], orders=g["orders"])
    assert st.shape[0] == len(g["stats"]) and int(clone.opt_step.item()) == len(g["stats"])
    for i in range(len(g["stats"])):
        for j, k in enumerate(("neglogp", "entropy", "ent_loss", "prob_true_act", "l2_norm", "l2_loss", "loss")):
            want = g["stats"][i, j]
            assert abs(st[i, j] - want) <= 2e-5 + 2e-4 * abs(want), (i, k, st[i, j], want)
    d = np.abs(clone.policy.get_flat_params() - g["params_final"])
    assert np.median(d) <= 2e-6 and d.max() <= 2.5e-3 and (d > 1e-4).mean() <= 0.01, (np.median(d), d.max(), (d > 1e-4).mean())


# ---- the integer action-mask rule on the device (tests/golden/ref_action_mask.npz: the reference's PettingZooAECWrapper text) ---------------
@pytest.mark.parametrize("case", ["mpe8_L5", "wide_L20"])
def test_device_illegal_action_repair_is_the_reference_wrappers(case):
    """`ph_fix_illegal_actions` against what the REFERENCE's PettingZooAECWrapper.n_step (pantheonrl/envs/pettingzoo.py:78-84) stepped its
    base environment with for the same (mask, sample) pairs: bit-exact (north_star: integer action masks)."""
    from pantheonrl_amd import _native as nat
    from pantheonrl_amd import spaces as sp
    from pantheonrl_amd.ppo import ActorCriticPolicy
    z = np.load(os.path.join(GOLDEN, "ref_action_mask.npz"))
    masks, sampled, want = z[case + ".masks"], z[case + ".sampled"], z[case + ".stepped_with"]
    n, L = masks.shape
    pol = ActorCriticPolicy(sp.Box(-np.inf, np.inf, (4,)), sp.Discrete(L), device="cuda", seed=0)
    acts = th.as_tensor(sampled.astype(np.int32)).cuda()
    m_dev = th.as_tensor(np.ascontiguousarray(masks, np.uint8)).cuda()
    pol._bind()
    nat.check(pol.ctx.lib.ph_fix_illegal_actions(pol.ctx.handle, acts.data_ptr(), m_dev.data_ptr(), n, L))
    th.cuda.synchronize()
    assert np.array_equal(acts.cpu().numpy(), want)
