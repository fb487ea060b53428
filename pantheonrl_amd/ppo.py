"""The model surface PantheonRL's OnPolicyAgent / trainer.py expect from `stable_baselines3.PPO`, backed by the
gfx950 engine (libpantheon_hip.so).

Reference call sites this module answers (paths relative to the reference root):
  * PPO(policy='MlpPolicy', env=, device=, seed=, verbose=, tensorboard_log=, **cfg)   trainer.py:108-126,196-203
  * model.rollout_buffer.{add, reset, compute_returns_and_advantage, rewards, pos}      pantheonrl/common/agents.py:123-130,157,172-179,196-198
  * model.policy.forward(obs_tensor) / .observation_space / .action_space / .device     pantheonrl/common/util.py:75-79, agents.py:170-171
  * model.train(), model.n_steps, model.logger, model.set_logger, model.ep_info_buffer  agents.py:102-109,126,134-155
  * model.learn(total_timesteps=, tb_log_name=), model.save(path), PPO.load(path)        trainer.py:140-149,410-432
Semantics follow SB3 1.7.0 as restated in SURVEY.md Appendix A.

torch is used for device memory, streams and host<->device copies only; all arithmetic of the hot path (buffer
writes, GAE, forward, PPO update) runs in the hand-written HIP kernels.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import io
import json
import time
import zipfile
from collections import deque
from typing import Sequence, Any, Dict, Optional, Tuple

import numpy as np
import torch as th

from . import _native as nat
from . import spaces as sp
from .logger import Logger, configure_logger

HID = nat.PH_HIDDEN


def _require_cuda(device) -> th.device:
    dev = th.device("cuda" if device in (None, "auto") else device)
    if dev.type != "cuda":
        raise nat.NativeError(
            f"device={device!r}: the PantheonRL MI355X engine runs on a gfx950 GPU only (no CPU fallback)")
    if not th.cuda.is_available():
        raise nat.NativeError("no HIP device visible to torch: the PantheonRL MI355X engine has no CPU fallback")
    if dev.index is None:
        dev = th.device("cuda", th.cuda.current_device())
    return dev


def _f32_dev(x, dev: th.device, shape: Tuple[int, ...]) -> th.Tensor:
    """numpy / list / scalar / tensor -> contiguous float32 device tensor of `shape` (host data are copied)."""
    if isinstance(x, th.Tensor):
        t = x.detach()
    else:
        arr = np.asarray(x)
        t = th.as_tensor(arr if arr.flags.writeable else arr.copy())
    return t.to(device=dev, dtype=th.float32).reshape(shape).contiguous()


_RAW_STREAM = getattr(th._C, "_cuda_getCurrentRawStream", None)


def _raw_stream(device) -> int:
    """the caller's current stream on `device` as the raw handle (what ph_ctx_set_stream takes); the private getter costs a third
    of torch.cuda.current_stream(...).cuda_stream, and this runs in front of every engine call"""
    if _RAW_STREAM is not None and device.index is not None:
        return _RAW_STREAM(device.index)
    return th.cuda.current_stream(device).cuda_stream


class RolloutBuffer:
    """Device-resident SB3 RolloutBuffer.  Arrays are torch views of HBM, time-major (T, E, ...)."""

    def __init__(self, buffer_size: int, observation_space, action_space, device, ctx: nat.Context,
                 spec: nat.PhSpec, gae_lambda: float = 0.95, gamma: float = 0.99, n_envs: int = 1):
        self.buffer_size, self.n_envs = int(buffer_size), int(n_envs)
        self.observation_space, self.action_space = observation_space, action_space
        self.obs_shape = sp.obs_stored_shape(observation_space)
        self.action_dim = sp.action_dim(action_space)
        self.device, self.ctx, self.spec = device, ctx, spec
        self.gamma, self.gae_lambda = float(gamma), float(gae_lambda)
        self.gae_mode = 0
        T, E, D, A = self.buffer_size, self.n_envs, int(np.prod(self.obs_shape)), self.action_dim
        z = lambda *s: th.zeros(*s, dtype=th.float32, device=device)  # noqa: E731
        self.observations = z(T, E, D)
        self.actions = z(T, E, A)
        self.rewards, self.returns, self.episode_starts = z(T, E), z(T, E), z(T, E)
        self.values, self.log_probs, self.advantages = z(T, E), z(T, E), z(T, E)
        self._c = nat.PhRollout()
        self._c.T, self._c.E = T, E
        for name in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages",
                     "returns"):
            setattr(self._c, name, getattr(self, name).data_ptr())
        self.pos, self.full = 0, False

    # -- SB3 surface --------------------------------------------------------------------------------------
    def reset(self) -> None:
        self._bind()
        nat.check(self.ctx.lib.ph_buffer_reset(self.ctx.handle, C.byref(self.spec), C.byref(self._c)))
        self.pos, self.full = 0, False

    def add(self, obs, action, reward, episode_start, value, log_prob) -> None:
        """RolloutBuffer.add <- agents.py:172-179.  Inputs (numpy or tensors, host or device) are copied."""
        if self.pos >= self.buffer_size:
            raise nat.NativeError("RolloutBuffer.add on a full buffer")
        E, D, A, dev = self.n_envs, int(np.prod(self.obs_shape)), self.action_dim, self.device
        o, a = _f32_dev(obs, dev, (E, D)), _f32_dev(action, dev, (E, A))
        s, v, lp = _f32_dev(episode_start, dev, (E,)), _f32_dev(value, dev, (E,)), _f32_dev(log_prob, dev, (E,))
        self._bind()
        nat.check(self.ctx.lib.ph_buffer_add(self.ctx.handle, C.byref(self.spec), C.byref(self._c), self.pos,
                                             o.data_ptr(), a.data_ptr(), s.data_ptr(), v.data_ptr(), lp.data_ptr()))
        self.pos += 1
        r = np.asarray(reward.detach().cpu() if isinstance(reward, th.Tensor) else reward, dtype=np.float32)
        if np.any(r != 0):  # the row is written with reward 0 (agents.py:175); a non-zero reward is the ego path
            self.add_reward(r)
        self.full = self.pos == self.buffer_size

    def add_reward(self, reward, env_mask=None, pos: Optional[int] = None) -> None:
        """buf.rewards[pos-1][e] += reward[e] <- Agent.update, agents.py:198 (vectorised over envs)."""
        row = (self.pos - 1) % self.buffer_size if pos is None else pos   # rewards[pos - 1]: numpy's wrap at pos == 0 (agents.py:198)
        if (env_mask is None and self.n_envs == 1 and isinstance(reward, np.ndarray) and reward.size == 1):
            reward = float(reward.reshape(-1)[0])
        if env_mask is None and isinstance(reward, (int, float, np.floating, np.integer)):
            # a scalar for every environment of the row (Agent.update's own signature): it travels as a kernel argument
            self._bind()
            nat.check(self.ctx.lib.ph_buffer_add_reward_const(self.ctx.handle, C.byref(self._c), row, float(reward)))
            return
        r = _f32_dev(np.broadcast_to(np.asarray(reward, np.float32), (self.n_envs,))
                     if not isinstance(reward, th.Tensor) else reward, self.device, (self.n_envs,))
        m = None
        if env_mask is not None:
            m = th.as_tensor(env_mask).to(device=self.device, dtype=th.uint8).contiguous()
        self._bind()
        nat.check(self.ctx.lib.ph_buffer_add_reward(self.ctx.handle, C.byref(self._c), row, r.data_ptr(), nat.ptr(m)))

    def compute_returns_and_advantage(self, last_values, dones) -> None:
        """GAE <- agents.py:127-130."""
        E, dev = self.n_envs, self.device
        lv = _f32_dev(last_values, dev, (E,))
        dn = _f32_dev(np.broadcast_to(np.asarray(dones, np.float32), (E,)) if not isinstance(dones, th.Tensor)
                      else dones, dev, (E,))
        self._bind()
        nat.check(self.ctx.lib.ph_gae(self.ctx.handle, C.byref(self._c), lv.data_ptr(), dn.data_ptr(), self.gamma,
                                      self.gae_lambda, int(self.gae_mode)))

    # -- helpers ---------------------------------------------------------------------------------------------
    def _bind(self) -> None:
        self.ctx.set_stream(_raw_stream(self.device))

    def c_struct(self) -> nat.PhRollout:
        return self._c

    def c_ref(self):
        """byref(c_struct()), made once (the per-step host path passes it every environment step)"""
        ref = self.__dict__.get("_c_byref")
        if ref is None:
            ref = self._c_byref = C.byref(self._c)
        return ref

    def add_reward_scalar(self, reward: float) -> None:
        """add_reward for Agent.update's own signature -- one float for the last row written -- without the dispatch around it"""
        self._bind()
        rc = self.ctx.lib.ph_buffer_add_reward_const(self.ctx.handle, self.c_ref(), (self.pos - 1) % self.buffer_size, reward)
        if rc:
            nat.check(rc)

    def host(self) -> Dict[str, np.ndarray]:
        """copy of every array on the host (tests, save/export)."""
        return {k: getattr(self, k).detach().cpu().numpy() for k in (
            "observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns")}


# SB3 state_dict names of the MlpPolicy tensors, in the order of the flat parameter vector
_SD = (("mlp_extractor.policy_net.0", "pi_W1", "pi_b1"), ("mlp_extractor.policy_net.2", "pi_W2", "pi_b2"),
       ("mlp_extractor.value_net.0", "vf_W1", "vf_b1"), ("mlp_extractor.value_net.2", "vf_W2", "vf_b2"),
       ("action_net", "act_W", "act_b"), ("value_net", "val_W", "val_b"))


class _HostStep:
    """forward_and_store_host's staging for one (row count, observation shape): arrays, their addresses, argument references"""

    def __init__(self, pol, obs_shape):
        lay = pol.layout
        n = int(np.prod(obs_shape)) // lay.D
        if n * lay.D != int(np.prod(obs_shape)) or n <= 0:
            raise ValueError(f"observation rows of shape {tuple(obs_shape)} do not hold whole rows of {lay.D} values")
        self.n = n
        self.rows = np.empty((n, lay.D), np.float32)
        self.rows_nd = self.rows.reshape(obs_shape)          # assignment converts dtype and layout in one copy
        self.acts = np.empty((n, lay.A), np.int32)
        self.acts_shaped = self.acts.reshape((-1,) + tuple(pol.action_space.shape))
        self.vals, self.logp, self.es = np.empty((n, 1), np.float32), np.empty((n,), np.float32), np.empty((n,), np.float32)
        self.rows_ptr, self.acts_ptr = self.rows.ctypes.data, self.acts.ctypes.data
        self.vals_ptr, self.logp_ptr, self.es_ptr = self.vals.ctypes.data, self.logp.ctypes.data, self.es.ctypes.data
        self.spec_ref = C.byref(pol.spec)
        self.params, self.params_ptr = pol.params, pol.params.data_ptr()


class ActorCriticPolicy:
    """SB3 MlpPolicy (FlattenExtractor + MlpExtractor pi=[64,64], vf=[64,64], tanh) with parameters resident in HBM
    as one flat input-major vector (layout in include/pantheon_hip.h)."""

    # OnPolicyAgent.get_action may use forward_and_store_host (one native call per environment step) for host observations;
    # subclasses that build their rows on the device (AdapPolicy appends the context there) switch it off
    host_step_path = True
    # the parameter vector is the plain 64-64 MLP in ph_layout order, i.e. what the fused MLP kernels read (ph_ppo_train_multi, the
    # 16-row rollout forwards, the exchange / round-robin engines).  Policies with another network (AdapPolicyMult, ModularPolicy)
    # carry the same `spec` for their rollout buffer but a different vector: they say False and those entry points refuse them
    fused_mlp_kernels = True

    def __init__(self, observation_space, action_space, lr: float = 3e-4, device="cuda",
                 ortho_init: bool = True, seed: Optional[int] = None, sampling_stream: int = 0):
        if type(action_space).__name__ == "Box" and not getattr(self, "gaussian_head", False):
            # (UnsupportedPolicyConfig is defined below; this is the same refusal, raised before the device is touched)
            raise UnsupportedPolicyConfig(f"{type(self).__name__} is written for the categorical heads (Discrete / MultiDiscrete "
                                          "actions); Box action spaces run on PPO's GaussianActorCriticPolicy (DiagGaussian head)")
        self.device = _require_cuda(device)
        self.observation_space, self.action_space = observation_space, action_space
        self.spec = sp.make_spec(observation_space, action_space)
        self.layout = nat.layout_of(self.spec)
        self.ctx = nat.Context(self.device.index)
        # how the 64x64 products of the gradient kernels are computed (include/pantheon_hip.h, gemm_mode): 2 = three bf16 planes per
        # float32 operand, six matrix-pipe terms per product (float32 accuracy, the fast path; specs the split kernel does not
        # take run mode 0), 0 = exact float32 MFMA, 1 = VALU fmaf chain (debug cross-check of mode 0).  PH_GEMM_MODE overrides.
        self.gemm_mode = int(os.environ.get("PH_GEMM_MODE", "2"))
        lay = self.layout
        self.params = th.zeros(lay.P, dtype=th.float32, device=self.device)
        self.adam_m = th.zeros_like(self.params)
        self.adam_v = th.zeros_like(self.params)
        self.opt_step = th.zeros(1, dtype=th.int32, device=self.device)
        self.lr = float(lr(1.0)) if callable(lr) else float(lr)   # a schedule: its value at the start (informational)
        self._seed = int(seed) if seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        if sampling_stream:
            # Philox key of this model's action sampling.  The reference draws every agent's actions from ONE global torch
            # stream, so two agents built with the same --seed still sample independently; here each model owns a counter-based
            # stream, and two models keyed by the bare seed would draw identical uniforms step after step (an RPS PPO-vs-PPO
            # run would start with nothing but ties).  Stream k of seed s gets the key splitmix64(s, k).
            z = (self._seed * 0x9E3779B97F4A7C15 + int(sampling_stream) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
            self._seed = int((z ^ (z >> 31)) & 0x7FFFFFFFFFFFFFFF)
        self._counter = 0
        self._host_out: Dict[tuple, "_HostStep"] = {}   # forward_and_store_host's staging, per observation-rows shape
        self._init_weights(ortho_init)

    # -- parameters -------------------------------------------------------------------------------------------
    def _shapes(self):
        lay = self.layout
        return {"pi_W1": (lay.F, HID), "pi_W2": (HID, HID), "vf_W1": (lay.F, HID), "vf_W2": (HID, HID),
                "act_W": (HID, lay.L), "val_W": (HID, 1)}

    def _init_weights(self, ortho_init: bool) -> None:
        """orthogonal init with SB3's gains (modular/policies.py:229-241); host-side, once."""
        gains = {"pi_W1": np.sqrt(2), "pi_W2": np.sqrt(2), "vf_W1": np.sqrt(2), "vf_W2": np.sqrt(2), "act_W": 0.01,
                 "val_W": 1.0}
        flat = th.zeros(self.layout.P, dtype=th.float32)
        for name, (fin, fout) in self._shapes().items():
            w = th.empty(fout, fin)  # torch Linear layout [out][in]
            if ortho_init:
                th.nn.init.orthogonal_(w, gain=gains[name])
            else:
                th.nn.init.kaiming_uniform_(w, a=np.sqrt(5))
            off = getattr(self.layout, name)
            flat[off:off + fin * fout] = w.t().contiguous().reshape(-1)
        self.params.copy_(flat)

    def state_dict(self) -> Dict[str, th.Tensor]:
        """torch.nn.Linear-shaped CPU tensors under SB3's module names."""
        flat, lay, shapes, out = self.params.detach().cpu(), self.layout, self._shapes(), {}
        for mod, wname, bname in _SD:
            fin, fout = shapes[wname]
            woff, boff = getattr(lay, wname), getattr(lay, bname)
            out[mod + ".weight"] = flat[woff:woff + fin * fout].reshape(fin, fout).t().contiguous()
            out[mod + ".bias"] = flat[boff:boff + fout].clone()
        return out

    def load_state_dict(self, sd: Dict[str, th.Tensor]) -> None:
        flat, lay, shapes = th.zeros(self.layout.P), self.layout, self._shapes()
        for mod, wname, bname in _SD:
            fin, fout = shapes[wname]
            woff, boff = getattr(lay, wname), getattr(lay, bname)
            flat[woff:woff + fin * fout] = th.as_tensor(sd[mod + ".weight"]).float().reshape(fout, fin).t().reshape(-1)
            flat[boff:boff + fout] = th.as_tensor(sd[mod + ".bias"]).float().reshape(-1)
        self.params.copy_(flat)

    def get_flat_params(self) -> np.ndarray:
        return self.params.detach().cpu().numpy().copy()

    def set_flat_params(self, flat) -> None:
        self.params.copy_(th.as_tensor(np.asarray(flat, np.float32)))

    # -- forward family -----------------------------------------------------------------------------------------
    def _obs(self, obs) -> th.Tensor:
        D = self.layout.D
        if isinstance(obs, th.Tensor):
            t = obs.detach()
        else:
            t = th.as_tensor(np.asarray(obs))
        return t.to(device=self.device, dtype=th.float32).reshape(-1, D).contiguous()

    def _bind(self) -> None:
        self.ctx.set_stream(_raw_stream(self.device))

    def _launch(self, obs_t, *, mask=None, uniforms=None, given=None, deterministic=False, want_logits=False,
                want_entropy=False, rb: Optional[RolloutBuffer] = None, pos: int = 0, episode_start=None):
        n, lay, dev = obs_t.shape[0], self.layout, self.device
        acts = th.empty((n, lay.A), dtype=th.int32, device=dev)
        values = th.empty((n, 1), dtype=th.float32, device=dev)
        logp = th.empty((n,), dtype=th.float32, device=dev)
        logits = th.empty((n, lay.L), dtype=th.float32, device=dev) if want_logits else None
        ent = th.empty((n,), dtype=th.float32, device=dev) if want_entropy else None
        m = None if mask is None else th.as_tensor(mask).to(device=dev, dtype=th.uint8).reshape(n, lay.L).contiguous()
        u = None if uniforms is None else _f32_dev(uniforms, dev, (n, lay.A))
        g = None if given is None else _f32_dev(given, dev, (n, lay.A))
        es = None if episode_start is None else _f32_dev(episode_start, dev, (n,))
        self._bind()
        self._counter += 1
        nat.check(self.ctx.lib.ph_policy_forward(
            self.ctx.handle, C.byref(self.spec), self.params.data_ptr(), obs_t.data_ptr(), n, nat.ptr(m), nat.ptr(u),
            nat.ptr(g), self._seed, self._counter, int(bool(deterministic)), acts.data_ptr(), None, values.data_ptr(),
            logp.data_ptr(), nat.ptr(ent), nat.ptr(logits), C.byref(rb.c_struct()) if rb is not None else None,
            int(pos), nat.ptr(es), None, int(self.gemm_mode)))
        return acts, values, logp, ent, logits

    def _shape_actions(self, acts: th.Tensor) -> th.Tensor:
        # SB3: actions.reshape((-1,) + action_space.shape); Discrete has shape ()
        return acts.long().reshape((-1,) + tuple(self.action_space.shape))

    def forward(self, obs, deterministic: bool = False, action_mask=None, uniforms=None):
        """-> (actions, values (n,1), log_prob (n,)) like ActorCriticPolicy.forward (util.py:79)."""
        acts, values, logp, _, _ = self._launch(self._obs(obs), mask=action_mask, uniforms=uniforms,
                                                deterministic=deterministic)
        return self._shape_actions(acts), values, logp

    __call__ = forward

    def forward_and_store(self, obs, rb: RolloutBuffer, episode_start, deterministic: bool = False, action_mask=None,
                          uniforms=None):
        """forward fused with RolloutBuffer.add(reward = 0) at rb.pos (agents.py:162 + 172-179 in one launch)."""
        if rb.pos >= rb.buffer_size:
            raise nat.NativeError("RolloutBuffer.add on a full buffer")
        acts, values, logp, _, _ = self._launch(self._obs(obs), mask=action_mask, uniforms=uniforms,
                                                deterministic=deterministic, rb=rb, pos=rb.pos,
                                                episode_start=episode_start)
        rb.pos += 1
        rb.full = rb.pos == rb.buffer_size
        return self._shape_actions(acts), values, logp

    def forward_and_store_host(self, obs_rows: np.ndarray, rb: RolloutBuffer, episode_start, deterministic: bool = False,
                               as_numpy: bool = False):
        """forward_and_store for an environment that lives on the host: host rows in, host results out, ONE native call and one
        wait (ph_policy_act_host) instead of the tensor operations of the general path.  Same kernel, same (seed, counter):
        bitwise the same samples.  -> (actions np.int64 shaped like SB3's, values (n,1), log_prob (n,)); the last two as CPU
        tensors, or -- as_numpy, what OnPolicyAgent asks for: it only keeps the values for the next GAE -- as numpy copies.

        This runs once per environment step of an n_envs = 1 loop, where the Python around the launch is most of the step
        (scripts/host_step_overhead.py): the staging arrays, their addresses and the argument references are made once per
        (row count, observation shape) and reused."""
        pos = rb.pos
        if pos >= rb.buffer_size:
            raise nat.NativeError("RolloutBuffer.add on a full buffer")
        obs_rows = np.asarray(obs_rows)
        hs = self._host_out.get(obs_rows.shape)
        if hs is None:
            hs = self._host_out[obs_rows.shape] = _HostStep(self, obs_rows.shape)
        hs.rows_nd[...] = obs_rows
        hs.es[:] = episode_start
        if self.params is not hs.params:          # (load_state_dict / set_flat_params may have replaced the tensor)
            hs.params, hs.params_ptr = self.params, self.params.data_ptr()
        self._bind()
        self._counter += 1
        rc = self.ctx.lib.ph_policy_act_host(
            self.ctx.handle, hs.spec_ref, hs.params_ptr, hs.rows_ptr, hs.n, hs.es_ptr, self._seed, self._counter,
            1 if deterministic else 0, hs.acts_ptr, hs.vals_ptr, hs.logp_ptr, rb.c_ref(), pos, int(self.gemm_mode))
        if rc:
            nat.check(rc)
        rb.pos = pos + 1
        rb.full = rb.pos == rb.buffer_size
        shaped = hs.acts_shaped.astype(np.int64)
        if as_numpy:
            return shaped, hs.vals.copy(), hs.logp.copy()
        return shaped, th.from_numpy(hs.vals.copy()), th.from_numpy(hs.logp.copy())

    def evaluate_actions(self, obs, actions, action_mask=None):
        """-> (values (n,1), log_prob (n,), entropy (n,))  (modular/policies.py:364-383)."""
        obs_t = self._obs(obs)
        _, values, logp, ent, _ = self._launch(obs_t, mask=action_mask, given=actions, want_entropy=True)
        return values, logp, ent

    def predict_values(self, obs) -> th.Tensor:
        return self._launch(self._obs(obs), deterministic=True)[1]

    def get_logits(self, obs, action_mask=None) -> th.Tensor:
        return self._launch(self._obs(obs), mask=action_mask, deterministic=True, want_logits=True)[4]

    def predict(self, obs, deterministic: bool = False):
        acts, _, _ = self.forward(obs, deterministic=deterministic)
        return acts.cpu().numpy(), None

    def reset_noise(self, n_envs: int = 1) -> None:  # gSDE hook (util.py:109-111); MlpPolicy default has none
        return None

    def set_training_mode(self, mode: bool) -> None:
        return None


class GaussianActorCriticPolicy(ActorCriticPolicy):
    """SB3's MlpPolicy over a Box (continuous) action space: the same two 64-64 tanh towers, `action_net` gives the A means and
    `log_std` (A entries, initialised to 0: SB3's log_std_init) is a free parameter -- DiagGaussianDistribution.  Actions are float32
    rows, stored UNclipped in the rollout buffer; `clip_actions` (reference pantheonrl/common/util.py:84-99) clips what the
    environment gets.  Runs on the general forward / gradient kernels (ph_policy_forward, ph_ppo_train); the 16-row rollout
    forwards, the one-launch rollouts and the split-bf16 gradient kernels are written for the categorical heads and refuse the
    spec in the library.  `uniforms` of forward() teacher-forces the STANDARD-NORMAL draws here."""

    host_step_path = False   # ph_policy_act_host returns int32 actions
    gaussian_head = True
    # the vectorised agents, the one-launch rollouts and the joint update hand int32 action rows around: they refuse this policy by
    # require_mlp_kernels; PPO.train()'s own entry point (ph_ppo_train: the general gradient kernel) knows the head
    fused_mlp_kernels = False

    def _init_weights(self, ortho_init: bool) -> None:
        super()._init_weights(ortho_init)          # log_std stays 0 behind val_b
        lay = self.layout
        assert lay.P == lay.val_b + 1 + lay.A

    def log_std(self) -> th.Tensor:
        lay = self.layout
        return self.params[lay.val_b + 1:lay.val_b + 1 + lay.A]

    def state_dict(self) -> Dict[str, th.Tensor]:
        out = super().state_dict()
        out["log_std"] = self.log_std().detach().cpu().clone()
        return out

    def load_state_dict(self, sd: Dict[str, th.Tensor]) -> None:
        super().load_state_dict(sd)
        lay = self.layout
        self.params[lay.val_b + 1:lay.val_b + 1 + lay.A] = th.as_tensor(sd["log_std"]).float().reshape(-1).to(self.device)

    def _launch(self, obs_t, *, mask=None, uniforms=None, given=None, deterministic=False, want_logits=False,
                want_entropy=False, rb: Optional[RolloutBuffer] = None, pos: int = 0, episode_start=None):
        if mask is not None:
            raise nat.NativeError("action masks belong to the categorical heads")
        n, lay, dev = obs_t.shape[0], self.layout, self.device
        acts = th.empty((n, lay.A), dtype=th.float32, device=dev)
        values = th.empty((n, 1), dtype=th.float32, device=dev)
        logp = th.empty((n,), dtype=th.float32, device=dev)
        means = th.empty((n, lay.L), dtype=th.float32, device=dev) if want_logits else None
        ent = th.empty((n,), dtype=th.float32, device=dev) if want_entropy else None
        u = None if uniforms is None else _f32_dev(uniforms, dev, (n, lay.A))
        g = None if given is None else _f32_dev(given, dev, (n, lay.A))
        es = None if episode_start is None else _f32_dev(episode_start, dev, (n,))
        self._bind()
        self._counter += 1
        nat.check(self.ctx.lib.ph_policy_forward(
            self.ctx.handle, C.byref(self.spec), self.params.data_ptr(), obs_t.data_ptr(), n, None, nat.ptr(u),
            nat.ptr(g), self._seed, self._counter, int(bool(deterministic)), None, acts.data_ptr(), values.data_ptr(),
            logp.data_ptr(), nat.ptr(ent), nat.ptr(means), C.byref(rb.c_struct()) if rb is not None else None,
            int(pos), nat.ptr(es), None, int(self.gemm_mode)))
        return acts, values, logp, ent, means

    def _shape_actions(self, acts: th.Tensor) -> th.Tensor:
        return acts.reshape((-1,) + tuple(self.action_space.shape))

    def predict(self, obs, deterministic: bool = False):
        """SB3 BasePolicy.predict: Box actions are clipped to the space (the policy does not squash)"""
        acts, _, _ = self.forward(obs, deterministic=deterministic)
        return np.clip(acts.cpu().numpy(), self.action_space.low, self.action_space.high), None

    def forward_and_store_host(self, *a, **k):
        raise nat.NativeError("GaussianActorCriticPolicy: the one-call host step returns integer actions; use forward_and_store")


def require_mlp_kernels(policy, who: str) -> None:
    """refuse a policy whose parameter vector is not the plain MLP's before an engine path would read it as one"""
    if not getattr(policy, "fused_mlp_kernels", True):
        raise nat.NativeError(f"{who}: {type(policy).__name__} does not run on the fused MLP kernels (its parameter vector has "
                              "another layout); use its own algorithm class")


class _SingleEnvVec:
    """What SB3 wraps a lone env into: DummyVecEnv([Monitor(env)]) (trainer.py:119) -- batch of one, auto-reset on
    done, episode return/length reported through info['episode']."""

    num_envs = 1

    def __init__(self, env):
        self.env = env
        self.observation_space, self.action_space = env.observation_space, env.action_space
        self._ret, self._len = 0.0, 0

    def reset(self):
        self._ret, self._len = 0.0, 0
        return np.asarray(self.env.reset())[None]

    def step(self, actions):
        obs, rew, done, info = self.env.step(actions[0])
        self._ret += float(rew)
        self._len += 1
        info = dict(info)
        if done:
            info["episode"] = {"r": self._ret, "l": self._len}
            info["terminal_observation"] = obs
            self._ret, self._len = 0.0, 0
            obs = self.env.reset()
        return (np.asarray(obs)[None], np.asarray([rew], np.float32), np.asarray([done]), [info])


class UnsupportedPolicyConfig(ValueError):
    """A `policy_kwargs` entry (trainer.py:108-126,196-203 splat `--ego-config` / `--alt-config` JSON into the constructor) or a space
    asks for a network the gfx950 kernels do not implement.  Raised at construction, never at the first forward."""


def check_policy_kwargs(policy_kwargs: Optional[Dict[str, Any]]) -> Dict[str, Any]:
    """What the kernels implement is SB3 1.7.0's MlpPolicy default and nothing else: FlattenExtractor, separate towers
    `net_arch=[dict(pi=[64, 64], vf=[64, 64])]`, Tanh, no gSDE (modular/policies.py:112-114,214-218).  `policy_kwargs` may restate that
    default (and pick `ortho_init`); any other entry is refused by name.  -> the keyword arguments for ActorCriticPolicy"""
    out: Dict[str, Any] = {}
    for key, val in dict(policy_kwargs or {}).items():
        if key == "net_arch":
            arch = val[0] if isinstance(val, (list, tuple)) and len(val) == 1 else val
            ok = isinstance(arch, dict) and sorted(arch) == ["pi", "vf"] and all(list(arch[k]) == [64, 64] for k in ("pi", "vf"))
            if not ok:
                raise UnsupportedPolicyConfig(
                    f"policy_kwargs['net_arch'] = {val!r}: the MI355X engine implements net_arch=[dict(pi=[64, 64], vf=[64, 64])] "
                    "only (SB3 1.7.0's MlpPolicy default, reference pantheonrl/algos/modular/policies.py:112-114)")
        elif key == "activation_fn":
            name = val if isinstance(val, str) else getattr(val, "__name__", repr(val))
            if name.lower() != "tanh":
                raise UnsupportedPolicyConfig(f"policy_kwargs['activation_fn'] = {name}: the kernels implement Tanh only "
                                              "(reference pantheonrl/algos/modular/policies.py:54)")
        elif key == "ortho_init":
            out["ortho_init"] = bool(val)
        elif key in ("use_sde", "squash_output", "use_expln", "full_std") and not val:
            continue
        elif key in ("optimizer_class", "optimizer_kwargs", "features_extractor_class", "features_extractor_kwargs",
                     "normalize_images", "log_std_init", "sde_net_arch", "use_sde", "squash_output", "use_expln", "full_std"):
            raise UnsupportedPolicyConfig(f"policy_kwargs[{key!r}]: the engine's policy is fixed to FlattenExtractor + Adam(eps=1e-5) + "
                                          "categorical heads (reference pantheonrl/algos/modular/policies.py:84-88,112-114)")
        else:
            raise UnsupportedPolicyConfig(f"policy_kwargs[{key!r}] is not an ActorCriticPolicy argument this engine knows")
    return out


def check_action_space(action_space) -> None:
    """the engine's heads: Categorical / MultiCategorical (every kernel) and DiagGaussian for one-dimensional Box spaces
    (GaussianActorCriticPolicy: general kernels; `clip_actions`' np.clip is reference pantheonrl/common/util.py:84-99)"""
    kind = type(action_space).__name__
    if kind == "Box":
        shape = tuple(getattr(action_space, "shape", ()))
        if len(shape) != 1 or shape[0] > nat.PH_MAX_BOX_ACT:
            raise UnsupportedPolicyConfig(f"action space {action_space!r}: Box action spaces of one dimension and at most "
                                          f"{nat.PH_MAX_BOX_ACT} components (DiagGaussian head, general kernels)")
        return
    if kind not in ("Discrete", "MultiDiscrete"):
        raise UnsupportedPolicyConfig(
            f"action space {action_space!r}: the MI355X engine implements Discrete / MultiDiscrete (categorical) and Box (DiagGaussian, "
            f"clipped by reference pantheonrl/common/util.py:84-99) action heads; a {kind} action space would need SB3's Bernoulli head")


class PPO:
    """Drop-in for `stable_baselines3.PPO` on the surface PantheonRL uses (see module docstring)."""

    def __init__(self, policy="MlpPolicy", env=None, learning_rate: float = 3e-4, n_steps: int = 2048,
                 batch_size: int = 64, n_epochs: int = 10, gamma: float = 0.99, gae_lambda: float = 0.95,
                 clip_range: float = 0.2, clip_range_vf: Optional[float] = None, normalize_advantage: bool = True,
                 ent_coef: float = 0.0, vf_coef: float = 0.5, max_grad_norm: float = 0.5,
                 target_kl: Optional[float] = None, tensorboard_log: Optional[str] = None, verbose: int = 0,
                 seed: Optional[int] = None, device="cuda", n_envs: Optional[int] = None, use_sde: bool = False,
                 sde_sample_freq: int = -1, _init_setup_model: bool = True, sampling_stream: int = 0,
                 policy_kwargs: Optional[Dict[str, Any]] = None):
        if policy not in ("MlpPolicy", ActorCriticPolicy):
            raise UnsupportedPolicyConfig("the MI355X engine implements SB3's MlpPolicy")
        if use_sde:
            raise UnsupportedPolicyConfig("gSDE is not on the categorical PPO path")
        self._policy_args = check_policy_kwargs(policy_kwargs)
        if env is not None and hasattr(env, "action_space"):
            check_action_space(env.action_space)
        self.device = _require_cuda(device)
        self.learning_rate, self.n_steps, self.batch_size, self.n_epochs = learning_rate, n_steps, batch_size, n_epochs
        self.gamma, self.gae_lambda, self.clip_range, self.clip_range_vf = gamma, gae_lambda, clip_range, clip_range_vf
        self.normalize_advantage, self.ent_coef, self.vf_coef = normalize_advantage, ent_coef, vf_coef
        self.max_grad_norm, self.target_kl = max_grad_norm, target_kl
        self.tensorboard_log, self.verbose, self.seed = tensorboard_log, verbose, seed
        self.use_sde, self.sde_sample_freq = False, sde_sample_freq
        self.sampling_stream = int(sampling_stream)
        self.num_timesteps, self._n_updates, self._custom_logger = 0, 0, False
        self.ep_info_buffer: deque = deque(maxlen=100)
        self._logger: Optional[Logger] = None
        self.start_time = time.time()
        self.env = None
        self.n_envs = int(n_envs or 1)
        self._last_obs, self._last_episode_starts = None, None
        self.permutation_seed = 0 if seed is None else int(seed)
        self.device_permutations = False  # True: Feistel permutations generated in-kernel (no host RNG, no H2D)
        self.last_train_stats: Optional[np.ndarray] = None
        if seed is not None:
            self.set_random_seed(seed)
        if env is not None:
            self._attach_env(env, n_envs)
        if _init_setup_model and env is not None:
            self._setup_model()

    # -- construction -------------------------------------------------------------------------------------------
    def set_random_seed(self, seed: int) -> None:
        """SB3 set_random_seed: random, numpy and torch (drives init and np.random.permutation)."""
        import random
        random.seed(seed)
        np.random.seed(seed)
        th.manual_seed(seed)

    def _attach_env(self, env, n_envs: Optional[int] = None) -> None:
        self.observation_space, self.action_space = env.observation_space, env.action_space
        if hasattr(env, "num_envs"):
            self.env = env
        elif hasattr(env, "step") and hasattr(env, "reset") and not getattr(env, "_is_dummy_space_env", False):
            self.env = _SingleEnvVec(env)
        else:
            self.env = env  # a DummyEnv that only carries spaces (partner models, trainer.py:95)
        self.n_envs = int(n_envs or getattr(self.env, "num_envs", 1))

    def _setup_model(self) -> None:
        check_action_space(self.action_space)
        policy_cls = GaussianActorCriticPolicy if type(self.action_space).__name__ == "Box" else ActorCriticPolicy
        self.policy = policy_cls(self.observation_space, self.action_space, lr=self.learning_rate,
                                 device=self.device, seed=self.seed, sampling_stream=self.sampling_stream,
                                 **getattr(self, "_policy_args", {}))
        self.rollout_buffer = RolloutBuffer(self.n_steps, self.observation_space, self.action_space, self.device,
                                            self.policy.ctx, self.policy.spec, gae_lambda=self.gae_lambda,
                                            gamma=self.gamma, n_envs=self.n_envs)

    def set_env(self, env) -> None:
        self._attach_env(env)

    # -- logger (agents.py:102-107) ---------------------------------------------------------------------------------
    @property
    def logger(self) -> Logger:
        if self._logger is None:
            self._logger = configure_logger(self.verbose, self.tensorboard_log, "PPO")
        return self._logger

    def set_logger(self, logger: Logger) -> None:
        self._logger = logger
        self._custom_logger = True

    # -- PPO.train() (agents.py:155) -----------------------------------------------------------------------------------
    @staticmethod
    def _schedule(value, progress_remaining: float) -> float:
        """SB3 get_schedule_fn: a constant, or a callable of progress_remaining (1 at the start of learn(), 0 at the end)"""
        return float(value(progress_remaining)) if callable(value) else float(value)

    def hyper(self) -> nat.PhPpoHyper:
        """this train() call's hyper-parameters; learning_rate / clip_range / clip_range_vf may be schedules evaluated at
        _current_progress_remaining (adap_learn.py:233-244: _update_learning_rate, clip_range(progress))"""
        h = nat.PhPpoHyper()
        prog = getattr(self, "_current_progress_remaining", 1.0)
        h.learning_rate, h.clip_range = self._schedule(self.learning_rate, prog), self._schedule(self.clip_range, prog)
        h.clip_range_vf = -1.0 if self.clip_range_vf is None else self._schedule(self.clip_range_vf, prog)
        h.ent_coef, h.vf_coef, h.max_grad_norm = float(self.ent_coef), float(self.vf_coef), float(self.max_grad_norm)
        h.target_kl = -1.0 if self.target_kl is None else float(self.target_kl)
        h.normalize_advantage = int(bool(self.normalize_advantage))
        h.adam_beta1, h.adam_beta2, h.adam_eps = 0.9, 0.999, 1e-5
        return h

    def train(self, perms: Optional[np.ndarray] = None, sync_stats: bool = True) -> None:
        """One PPO update over the (full) rollout buffer.  `perms` (n_epochs, T*E) teacher-forces the index order;
        default is np.random.permutation per epoch exactly like SB3's RolloutBuffer.get."""
        rb, pol = self.rollout_buffer, self.policy
        N = rb.buffer_size * rb.n_envs
        n_mb = (N + self.batch_size - 1) // self.batch_size
        perm_t = None
        if perms is None and not self.device_permutations:
            perms = np.stack([np.random.permutation(N) for _ in range(self.n_epochs)])
        if perms is not None:
            perm_t = th.as_tensor(np.ascontiguousarray(perms, dtype=np.int32)).to(self.device)
            assert perm_t.shape == (self.n_epochs, N)
        stats = getattr(self, "_stats_dev", None)   # reused across calls: no allocation on the update path
        if stats is None or stats.shape[0] != self.n_epochs * n_mb:
            stats = th.zeros((self.n_epochs * n_mb, nat.PH_NSTAT), dtype=th.float32, device=self.device)
        opt = nat.PhOptState()
        opt.params, opt.adam_m, opt.adam_v = pol.params.data_ptr(), pol.adam_m.data_ptr(), pol.adam_v.data_ptr()
        opt.step = pol.opt_step.data_ptr()
        hp = self.hyper()
        pol._bind()
        self.permutation_seed += 1
        self._train_native(pol, opt, rb, hp, perm_t, stats)
        self._n_updates += self.n_epochs
        self._stats_dev = stats
        if sync_stats:
            st = stats.cpu().numpy()
            # ph_ctx_step_errors counts over the context's life: only expiries since the previous train() condemn THIS update
            expired = pol.ctx.step_errors() if pol.ctx.exclusive_hint else 0
            seen, self._step_errors_seen = getattr(self, "_step_errors_seen", 0), expired
            if (st[:, 7] < 0).any() or expired > seen:
                raise nat.NativeError("PPO.train: a wait of the one-launch optimizer step expired (is the device really this "
                                      "learner's alone? ph_set_exclusive_device): the update was applied only in part")
            self.last_train_stats = st
            applied = st[:, 7] > 0
            # SB3 appends a minibatch's losses BEFORE the KL check (adap_learn.py:282-327), so the minibatch that triggers the
            # early stop is in the logged means although its optimizer step is skipped; later minibatches never run
            n_used = int(applied.sum()) + (1 if not applied.all() else 0)
            used = st[:max(n_used, 1)]
            lg = self.logger
            ret = rb.returns.flatten()
            var_y = float(ret.var(unbiased=False))     # explained_variance(values, returns), adap_learn.py:350-352
            ev = float("nan") if var_y == 0 else 1.0 - float((ret - rb.values.flatten()).var(unbiased=False)) / var_y
            lg.record("train/explained_variance", ev)
            lg.record("train/entropy_loss", float(used[:, 2].mean()))
            lg.record("train/policy_gradient_loss", float(used[:, 0].mean()))
            lg.record("train/value_loss", float(used[:, 1].mean()))
            # ... except approx_kl: `approx_kl_divs` is emptied at the top of every epoch (adap_learn.py:248-250), so the logged mean is
            # over the minibatches of the LAST epoch that ran (reference-generated fixture tests/golden/ref_ppo_train.npz)
            lg.record("train/approx_kl", float(used[((len(used) - 1) // n_mb) * n_mb:, 4].mean()))
            lg.record("train/clip_fraction", float(used[:, 3].mean()))
            lg.record("train/loss", float(used[-1, 5]))
            lg.record("train/n_updates", self._n_updates, exclude="tensorboard")
            lg.record("train/clip_range", float(hp.clip_range))
            lg.record("train/learning_rate", float(hp.learning_rate))

    def _train_native(self, pol, opt, rb, hp, perm_t, stats) -> None:
        """the update itself; subclasses with an additional loss term (ADAP) issue their own entry point here"""
        if not getattr(pol, "gaussian_head", False):
            require_mlp_kernels(pol, "PPO.train")
        nat.check(pol.ctx.lib.ph_ppo_train(pol.ctx.handle, C.byref(pol.spec), C.byref(opt), C.byref(rb.c_struct()),
                                           C.byref(hp), int(self.n_epochs), int(self.batch_size), nat.ptr(perm_t),
                                           int(self.permutation_seed), stats.data_ptr(), int(pol.gemm_mode)))

    def _train_call(self, keep: list) -> "nat.PhTrainCall":
        """this learner's train() arguments as a ph_train_call (device permutations; statistics stay on the device)"""
        rb, pol = self.rollout_buffer, self.policy
        require_mlp_kernels(pol, "PPO.train_joint")
        N = rb.buffer_size * rb.n_envs
        n_mb = (N + self.batch_size - 1) // self.batch_size
        stats = getattr(self, "_stats_dev", None)
        if stats is None or stats.shape[0] != self.n_epochs * n_mb:
            stats = th.zeros((self.n_epochs * n_mb, nat.PH_NSTAT), dtype=th.float32, device=self.device)
        self._stats_dev = stats
        opt = nat.PhOptState()
        opt.params, opt.adam_m, opt.adam_v = pol.params.data_ptr(), pol.adam_m.data_ptr(), pol.adam_v.data_ptr()
        opt.step = pol.opt_step.data_ptr()
        hp, rbc = self.hyper(), rb.c_struct()
        keep += [opt, hp, rbc]
        self.permutation_seed += 1
        call = nat.PhTrainCall()
        call.ctx, call.spec, call.opt, call.rb, call.hyper = pol.ctx.handle, C.pointer(pol.spec), C.pointer(opt), \
            C.pointer(rbc), C.pointer(hp)
        call.n_epochs, call.batch_size, call.perms = int(self.n_epochs), int(self.batch_size), None
        call.perm_seed, call.stats, call.gemm_mode = int(self.permutation_seed), stats.data_ptr(), int(pol.gemm_mode)
        return call

    @staticmethod
    def train_joint(models: Sequence["PPO"]) -> None:
        """train() of several independent learners in one call (ph_ppo_train_multi): same results as calling train() on
        each, with the learners' gradient launches chained so that the small launches of one overlap the large launch of
        the next.  Every learner's context must already be bound to the stream its launches should go to."""
        keep: list = []
        calls = (nat.PhTrainCall * len(models))(*[m._train_call(keep) for m in models])
        lib = models[0].policy.ctx.lib
        nat.check(lib.ph_ppo_train_multi(calls, len(models)))
        for m in models:
            m._n_updates += m.n_epochs

    # -- OnPolicyAlgorithm.learn() for the ego (trainer.py:413) -----------------------------------------
    def collect_rollouts(self, forced_uniforms=None, callback=None) -> bool:
        """SB3 1.7.0 OnPolicyAlgorithm.collect_rollouts for the ego (trainer.py:413).  `forced_uniforms[t]` (E, A) teacher-forces
        the sampling uniforms of step t (tests).  `callback`: an SB3-style BaseCallback -- on_rollout_start() before the first
        step, update_locals(locals()) + on_step() after every vectorised environment step (a False return ends the rollout and
        learn(), as in SB3: the witness is modular/learn.py:173,195-196), on_rollout_end() is learn()'s."""
        env, rb, pol = self.env, self.rollout_buffer, self.policy
        rb.reset()
        dones = np.zeros(self.n_envs, dtype=bool)
        new_obs = self._last_obs
        if callback is not None and hasattr(callback, "on_rollout_start"):
            callback.on_rollout_start()
        for t in range(self.n_steps):
            if forced_uniforms is None and getattr(pol, "host_step_path", False) and isinstance(self._last_obs, np.ndarray):
                # host environment: one native call per step (stage in, forward + row write, results out)
                act_np, _, _ = pol.forward_and_store_host(self._last_obs, rb, self._last_episode_starts, as_numpy=True)
                actions = act_np
            else:
                actions, _, _ = pol.forward_and_store(self._last_obs, rb, self._last_episode_starts,
                                                      uniforms=None if forced_uniforms is None else forced_uniforms[t])
                act_np = actions.cpu().numpy()
            if type(self.action_space).__name__ == "Box":   # SB3 collect_rollouts: the environment gets the clipped action,
                act_np = np.clip(act_np, self.action_space.low, self.action_space.high)   # the buffer row keeps the raw sample
            new_obs, rewards, dones, infos = env.step(act_np)
            self.num_timesteps += self.n_envs
            if callback is not None and hasattr(callback, "on_step"):
                if hasattr(callback, "update_locals"):
                    callback.update_locals(locals())
                if callback.on_step() is False:
                    return False
            rewards = np.asarray(rewards, np.float32).copy()
            for idx, info in enumerate(infos):
                ep = info.get("episode") if isinstance(info, dict) else None
                if ep is not None:
                    self.ep_info_buffer.append(ep)
                # an episode cut by a time limit is not a terminal state: bootstrap the cut-off return with the value of
                # the terminal observation [SB3 1.7.0: rewards[idx] += gamma * V(terminal_observation)], in float32
                if (isinstance(info, dict) and dones[idx] and info.get("terminal_observation") is not None
                        and info.get("TimeLimit.truncated", False)):
                    v_term = self._terminal_value(np.asarray(info["terminal_observation"], np.float32).reshape(1, -1), idx)
                    rewards[idx] += np.float32(self.gamma) * np.float32(v_term.reshape(-1)[0].item())
            rb.add_reward(rewards)
            self._last_obs = new_obs
            self._last_episode_starts = np.asarray(dones, np.float32)
            self._after_step(dones)
        values = pol.predict_values(new_obs)  # ego bootstraps with V(o_T)
        rb.compute_returns_and_advantage(last_values=values, dones=np.asarray(dones, np.float32))
        return True

    def _terminal_value(self, terminal_obs: np.ndarray, env_index: int) -> th.Tensor:
        """V(terminal observation) of environment column `env_index` for the time-limit bootstrap"""
        return self.policy.predict_values(terminal_obs)

    def _after_step(self, dones) -> None:
        """hook after every vectorised environment step of collect_rollouts (ADAP re-draws contexts here)"""

    def _extra_state(self) -> dict:
        """JSON-serialisable learner state beyond the hyper-parameters (ADAP: context generator and current contexts)"""
        return {}

    def _load_extra_state(self, extra: dict) -> None:
        pass

    def learn(self, total_timesteps: int, log_interval: int = 1, tb_log_name: str = "PPO",
              reset_num_timesteps: bool = True, callback=None, **_ignored) -> "PPO":
        if self.env is None or not hasattr(self.env, "step"):
            raise ValueError("learn() needs a steppable environment")
        if not self._custom_logger:
            self._logger = configure_logger(self.verbose, self.tensorboard_log, tb_log_name)
        if not reset_num_timesteps:
            total_timesteps += self.num_timesteps   # SB3 _setup_learn: continue for `total_timesteps` MORE steps
        if reset_num_timesteps or self._last_obs is None:
            if reset_num_timesteps:
                self.num_timesteps = 0
            self._last_obs = self.env.reset()
            self._last_episode_starts = np.ones(self.n_envs, np.float32)
        self.start_time = time.time()
        self._total_timesteps, start_steps = total_timesteps, self.num_timesteps
        iteration = 0
        # callback: SB3's BaseCallback protocol where available (init_callback / on_training_start / on_rollout_start /
        # on_step per vectorised step / on_rollout_end / on_training_end), or a plain callable(locals, globals) -> bool called
        # once per rollout; False stops training
        cb_obj = callback if hasattr(callback, "on_rollout_end") else None
        if cb_obj is not None and hasattr(cb_obj, "init_callback"):
            cb_obj.init_callback(self)
        if cb_obj is not None and hasattr(cb_obj, "on_training_start"):
            cb_obj.on_training_start(locals(), globals())
        while self.num_timesteps < total_timesteps:
            if self.collect_rollouts(callback=cb_obj) is False:
                break
            iteration += 1
            # SB3 _update_current_progress_remaining(num_timesteps, total_timesteps)
            self._current_progress_remaining = 1.0 - float(self.num_timesteps - start_steps) / float(
                max(total_timesteps - start_steps, 1))
            if cb_obj is not None:
                cb_obj.on_rollout_end()
                if getattr(cb_obj, "stop_training", False):
                    break
            elif callable(callback) and callback(locals(), globals()) is False:
                break
            if log_interval is not None and iteration % log_interval == 0:
                lg = self.logger
                fps = int(self.num_timesteps / max(time.time() - self.start_time, 1e-9))
                lg.record("time/iterations", iteration, exclude="tensorboard")
                if len(self.ep_info_buffer) > 0:
                    lg.record("rollout/ep_rew_mean", float(np.mean([e["r"] for e in self.ep_info_buffer])))
                    lg.record("rollout/ep_len_mean", float(np.mean([e["l"] for e in self.ep_info_buffer])))
                lg.record("time/fps", fps)
                lg.record("time/time_elapsed", int(time.time() - self.start_time), exclude="tensorboard")   # (SB3's learn; witness modular/learn.py:396)
                lg.record("time/total_timesteps", self.num_timesteps, exclude="tensorboard")
                lg.dump(step=self.num_timesteps)
            self.train()
        if cb_obj is not None and hasattr(cb_obj, "on_training_end"):
            cb_obj.on_training_end()
        return self

    def predict(self, obs, deterministic: bool = False):
        return self.policy.predict(obs, deterministic)

    # -- save / load (trainer.py:419-432, 140-157).  Own container; see DESIGN.md for the SB3 mapping. --------------
    _HP = ("learning_rate", "n_steps", "batch_size", "n_epochs", "gamma", "gae_lambda", "clip_range", "clip_range_vf",
           "normalize_advantage", "ent_coef", "vf_coef", "max_grad_norm", "target_kl", "seed", "n_envs", "tensorboard_log",
           "verbose", "sampling_stream")

    @staticmethod
    def _space_to_json(space) -> Dict[str, Any]:
        k = type(space).__name__
        if k == "Box":
            return {"type": k, "shape": list(space.shape), "low": np.asarray(space.low).tolist(),
                    "high": np.asarray(space.high).tolist()}
        if k == "MultiDiscrete":
            return {"type": k, "nvec": [int(v) for v in space.nvec]}
        return {"type": k, "n": int(space.n)}

    @staticmethod
    def _space_from_json(d: Dict[str, Any]):
        if d["type"] == "Box":
            return sp.Box(np.asarray(d["low"], np.float32), np.asarray(d["high"], np.float32), d["shape"])
        if d["type"] == "MultiDiscrete":
            return sp.MultiDiscrete(d["nvec"])
        if d["type"] == "Discrete":
            return sp.Discrete(d["n"])
        return sp.MultiBinary(d["n"])

    def save(self, path: str) -> None:
        path = path if str(path).endswith(".zip") else str(path) + ".zip"
        import os
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        data = {k: getattr(self, k) for k in self._HP}
        prog = getattr(self, "_current_progress_remaining", 1.0)
        for k in ("learning_rate", "clip_range", "clip_range_vf"):      # a schedule is stored as its current value
            if callable(data[k]):
                data[k] = self._schedule(data[k], prog)
        data.update(observation_space=self._space_to_json(self.observation_space),
                    action_space=self._space_to_json(self.action_space), num_timesteps=self.num_timesteps,
                    _n_updates=self._n_updates, format="pantheonrl_amd-1", extra=self._extra_state())
        pol = self.policy
        with zipfile.ZipFile(path, "w") as zf:
            zf.writestr("data", json.dumps(data))
            b = io.BytesIO()
            th.save(pol.state_dict(), b)
            zf.writestr("policy.pth", b.getvalue())
            b = io.BytesIO()
            th.save({"exp_avg": pol.adam_m.cpu(), "exp_avg_sq": pol.adam_v.cpu(), "step": int(pol.opt_step.item())}, b)
            zf.writestr("policy.optimizer.pth", b.getvalue())

    @classmethod
    def load(cls, path: str, env=None, device="cuda", **kwargs) -> "PPO":
        path = path if str(path).endswith(".zip") else str(path) + ".zip"
        with zipfile.ZipFile(path) as zf:
            try:
                data = json.loads(zf.read("data"))
            except Exception as exc:  # noqa: BLE001
                raise ValueError(f"{path}: unreadable 'data' entry ({exc})") from exc
            if data.get("format") != "pantheonrl_amd-1":
                # a stable-baselines3 zip keeps cloudpickled objects (schedules, gym spaces) under "data": they cannot be
                # rebuilt without SB3 / gym.  Its policy.pth uses the same module names as ours -- convert with
                # PPO(...).policy.load_state_dict(torch.load("policy.pth")) after constructing the model from its spaces.
                raise ValueError(f"{path} is not a pantheonrl_amd checkpoint (format tag {data.get('format')!r}); "
                                 "stable-baselines3 zips are not readable here -- see PPO.load's comment for the "
                                 "state_dict route")
            sd = th.load(io.BytesIO(zf.read("policy.pth")), map_location="cpu")
            opt = th.load(io.BytesIO(zf.read("policy.optimizer.pth")), map_location="cpu")
        hp = {k: data[k] for k in cls._HP if k in data}
        hp.update(kwargs)
        model = cls(env=None, device=device, **hp)
        model.observation_space = cls._space_from_json(data["observation_space"])
        model.action_space = cls._space_from_json(data["action_space"])
        model.n_envs = int(hp.get("n_envs") or 1)
        if env is not None:
            model._attach_env(env)
        model._setup_model()
        model.policy.load_state_dict(sd)
        model.policy.adam_m.copy_(opt["exp_avg"])
        model.policy.adam_v.copy_(opt["exp_avg_sq"])
        model.policy.opt_step.fill_(int(opt["step"]))
        model.num_timesteps, model._n_updates = data["num_timesteps"], data["_n_updates"]
        model._load_extra_state(data.get("extra") or {})
        return model
