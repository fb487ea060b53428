"""-m gpu: the owning-handle layer of the C ABI driven through pure ctypes + numpy -- no torch tensor, stream or allocation in
the call path (callee-owned device memory, caller-owned host arrays).  create -> set_params -> act (record) ->
add_reward -> gae -> train -> export / get_params, every stage against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest
import torch as th

from oracle import sb3_oracle as orc
from pantheonrl_amd import _native as nat
from pantheonrl_amd import spaces as sp
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _np(a, dtype=np.float32):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_handle_abi_end_to_end_without_torch_in_the_call_path():
    lib = nat.load()
    name, T, E = "overcooked", 12, 5
    obs_s, act_s = H.CONFIGS[name]
    spec = sp.make_spec(H.to_space(obs_s), H.to_space(act_s))
    h = C.c_void_p()
    assert lib.ph_agent_create(0, C.byref(spec), E, T, 0.99, 0.95, 123, C.byref(h)) == 0, lib.ph_agent_last_error()
    lay = nat.PhLayout()
    assert lib.ph_agent_layout(h, C.byref(lay)) == 0 and (lay.D, lay.L, lay.P) == (62, 6, 16839)
    orac = H.oracle_policy(name, seed=4)
    assert lib.ph_agent_set_params(h, _p(_np(orac.flat_params()))) == 0
    back = np.zeros(lay.P, np.float32)
    assert lib.ph_agent_get_params(h, _p(back)) == 0 and np.array_equal(back, orac.flat_params())

    # rollout: teacher-forced uniforms on the device, the sampled actions teacher-forced into the oracle
    rng = np.random.default_rng(0)
    ob = orc.RolloutBufferOracle(T, E, 62, 1)
    starts = np.ones(E, np.float32)
    values = None
    for t in range(T):
        obs = rng.standard_normal((E, 62)).astype(np.float32)
        u = rng.random((E, 1)).astype(np.float32)
        acts, vals, logp = np.zeros((E, 1), np.int32), np.zeros(E, np.float32), np.zeros(E, np.float32)
        assert lib.ph_agent_act(h, _p(obs), None, _p(u), 0, 1, _p(starts), _p(acts), _p(vals), _p(logp)) == 0, \
            lib.ph_agent_last_error()
        with th.no_grad():
            v_ref, lp_ref, _ = orac.evaluate_actions(th.as_tensor(obs), th.as_tensor(acts[:, 0].astype(np.int64)))
        assert np.abs(vals - v_ref.numpy().reshape(-1)).max() < 2e-5 and np.abs(logp - lp_ref.numpy()).max() < 2e-5
        rew = rng.standard_normal(E).astype(np.float32)
        assert lib.ph_agent_add_reward(h, _p(rew), None) == 0
        ob.add(obs, acts.astype(np.float32), rew, starts, th.as_tensor(vals), th.as_tensor(logp))
        starts = (rng.random(E) < 0.2).astype(np.float32)
        values = vals
    pos = C.c_int(-1)
    assert lib.ph_agent_pos(h, C.byref(pos)) == 0 and pos.value == T
    full = lib.ph_agent_act(h, _p(_np(np.zeros((E, 62)))), None, None, 0, 1, _p(starts), None, None, None)
    assert full != 0 and b"full buffer" in lib.ph_agent_last_error()

    # GAE (bit-exact serial mode) and export
    assert lib.ph_agent_gae(h, _p(_np(values)), _p(starts), 1) == 0
    ob.compute_returns_and_advantage(th.as_tensor(values), starts)
    out = {k: np.zeros((T, E) + ((62,) if k == "observations" else (1,) if k == "actions" else ()), np.float32)
           for k in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns")}
    assert lib.ph_agent_export_buffer(h, *[_p(out[k]) for k in out]) == 0
    for k in out:
        assert np.array_equal(out[k].reshape(getattr(ob, k).shape), getattr(ob, k)), k

    # PPO.train(), teacher-forced permutations, against the oracle
    hp, hpo = nat.PhPpoHyper(), orc.PPOHyper(batch_size=20, n_epochs=2)
    hp.learning_rate, hp.clip_range, hp.clip_range_vf, hp.ent_coef, hp.vf_coef = 3e-4, 0.2, -1.0, 0.0, 0.5
    hp.max_grad_norm, hp.target_kl, hp.normalize_advantage = 0.5, -1.0, 1
    hp.adam_beta1, hp.adam_beta2, hp.adam_eps = 0.9, 0.999, 1e-5
    perms = np.stack([np.random.default_rng(ep).permutation(T * E) for ep in range(2)]).astype(np.int32)
    stats = np.zeros((2 * 3, nat.PH_NSTAT), np.float32)
    assert lib.ph_agent_train(h, C.byref(hp), 2, 20, _p(perms), 0, _p(stats)) == 0, lib.ph_agent_last_error()
    ref = orc.ppo_train(orac, ob, hpo, perms)
    assert lib.ph_agent_get_params(h, _p(back)) == 0
    assert np.abs(back - orac.flat_params()).max() <= 2e-6 * len(ref) + 1e-6
    assert (stats[:, 7] == 1).all() and abs(stats[0, 5] - ref[0]["loss"]) < 1e-4
    m, v, step = np.zeros(lay.P, np.float32), np.zeros(lay.P, np.float32), C.c_int(0)
    assert lib.ph_agent_get_optimizer(h, _p(m), _p(v), C.byref(step)) == 0 and step.value == len(ref) and np.abs(m).max() > 0

    # buffer reset, import / export round trip, misuse is reported
    assert lib.ph_agent_buffer_reset(h) == 0 and lib.ph_agent_pos(h, C.byref(pos)) == 0 and pos.value == 0
    assert lib.ph_agent_export_buffer(h, None, None, _p(out["rewards"]), None, None, None, None, None) == 0
    assert not out["rewards"].any()
    assert lib.ph_agent_import_buffer(h, None, None, _p(_np(ob.rewards)), None, None, None, None, None, T) == 0
    assert lib.ph_agent_export_buffer(h, None, None, _p(out["rewards"]), None, None, None, None, None) == 0
    assert np.array_equal(out["rewards"], ob.rewards)
    assert lib.ph_agent_add_reward(None, None, None) != 0 and lib.ph_agent_train(h, None, 1, 1, None, 0, None) != 0
    assert lib.ph_agent_destroy(h) == 0
