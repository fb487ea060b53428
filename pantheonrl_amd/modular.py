"""ModularAlgorithm / ModularPolicy on the engine (reference pantheonrl/algos/modular/{learn,policies}.py).

ModularPolicy (policies.py:40-395) is the ordinary MlpPolicy ("main") plus one module per partner: a 64-64 policy tower and a
64-64 value tower that both read the MAIN POLICY LATENT (policies.py:254,281), an action head and a value head; logits and
values are sums (policies.py:286,325-328; `nomain` keeps the partner's logits alone, `baseline` shares one module between all
partners, policies.py:255-257).  ModularAlgorithm (learn.py:22-403) is PPO over that policy with

  * one rollout buffer per partner (learn.py:134-144), filled partner by partner with `set_partnerid` switching the
    environment's partner and `forward(obs, partner_idx)` composing the heads (learn.py:155-218,376-379);
  * `train()` walking the partners' buffers one after the other (learn.py:237-334): always-normalised advantages, the
    marginal regulariser over EVERY partner's composed logits on every minibatch (learn.py:298-318), the KL test after a whole
    epoch on the mean of its KLs.

Everything arithmetic runs in ph_modular.hip (tower kernels, loss kernel, one clip + Adam over the main network and every
module) behind `ph_modular_forward` / `ph_modular_train`; there is no CPU fallback.  Shapes: observations of at most 64
features (Box, Discrete or MultiDiscrete), one Discrete action head of at most 8 logits.
"""
from __future__ import annotations

import ctypes as C
import io
import json
import zipfile
from typing import Any, Dict, List, Optional

import numpy as np
import torch as th

from . import _native as nat
from . import spaces as sp
from .ppo import HID, PPO, ActorCriticPolicy, RolloutBuffer, _SD, _f32_dev

# module tensors in the order of a module's parameter block (torch.nn.Linear names of policies.py:214-219,252-260)
_MOD_SD = (("partner_mlp_extractor.{k}.policy_net.0", "pi_W1", "pi_b1"), ("partner_mlp_extractor.{k}.policy_net.2", "pi_W2", "pi_b2"),
           ("partner_mlp_extractor.{k}.value_net.0", "vf_W1", "vf_b1"), ("partner_mlp_extractor.{k}.value_net.2", "vf_W2", "vf_b2"),
           ("partner_action_net.{k}", "act_W", "act_b"), ("partner_value_net.{k}", "val_W", "val_b"))


class ModularPolicy(ActorCriticPolicy):
    """pantheonrl.algos.modular.policies.ModularPolicy with the FlattenExtractor defaults (net_arch = partner_net_arch =
    [dict(pi=[64, 64], vf=[64, 64])], tanh).  Parameters: the main network in ph_layout order, then every module's block in the
    ph_layout order of a (Box(64), same action space) network; `state_dict()` uses the reference's module names."""

    host_step_path = False   # the composed forward (main + partner towers) has its own launches
    fused_mlp_kernels = False   # main network + per-partner modules in one vector: ph_modular_* only

    def __init__(self, observation_space, action_space, lr: float = 3e-4, device="cuda", ortho_init: bool = True,
                 seed: Optional[int] = None, sampling_stream: int = 0, num_partners: int = 1, baseline: bool = False,
                 nomain: bool = False):
        if not 1 <= int(num_partners) <= nat.PH_MOD_MAX:
            raise ValueError(f"num_partners must be in [1, {nat.PH_MOD_MAX}]")
        self.num_partners, self.baseline, self.nomain = int(num_partners), bool(baseline), bool(nomain)
        self._ortho = ortho_init
        super().__init__(observation_space, action_space, lr=lr, device=device, ortho_init=ortho_init, seed=seed,
                         sampling_stream=sampling_stream)
        self._build_modules()

    # -- parameters -------------------------------------------------------------------------------------------------
    def _mod_struct(self) -> nat.PhModular:
        m = nat.PhModular()
        m.num_partners = self.num_partners
        m.n_modules = 1 if self.baseline else self.num_partners
        for k in range(self.num_partners):
            m.module_of[k] = 0 if self.baseline else k
        m.nomain = int(self.nomain)
        return m

    def _build_modules(self) -> None:
        """policies.py:243-265: the main network is what ActorCriticPolicy.__init__ built; the modules are appended"""
        self.mod = self._mod_struct()
        main, module, total = nat.PhLayout(), nat.PhLayout(), C.c_int(0)
        nat.check(self.ctx.lib.ph_modular_layout(C.byref(self.spec), C.byref(self.mod), C.byref(main), C.byref(module),
                                                 C.byref(total)))
        if main.F > 64 or main.A != 1 or main.L > 8:
            raise ValueError("ModularPolicy on the engine: observations of at most 64 features and one Discrete action head of at "
                             "most 8 logits")
        self.module_layout, self.P_total, self.n_modules = module, int(total.value), int(self.mod.n_modules)
        main_params = self.params
        self.params = th.zeros(self.P_total, dtype=th.float32, device=self.device)
        self.params[:main.P].copy_(main_params)
        self.adam_m, self.adam_v = th.zeros_like(self.params), th.zeros_like(self.params)
        # optimizer step count before module m's value side first received a gradient, -1 = never (torch's per-parameter step)
        self.mod_first = th.full((self.n_modules,), -1, dtype=th.int32, device=self.device)
        self.do_init_weights(init_partner=True)

    def _module_shapes(self):
        return {"pi_W1": (HID, HID), "pi_W2": (HID, HID), "vf_W1": (HID, HID), "vf_W2": (HID, HID),
                "act_W": (HID, self.layout.L), "val_W": (HID, 1)}

    def module_offset(self, m: int) -> int:
        return self.layout.P + m * self.module_layout.P

    def do_init_weights(self, init_main: bool = False, init_partner: bool = False) -> None:
        """policies.py:221-241: orthogonal init, gains sqrt(2) / 0.01 / 1, biases 0 (trainer.py:122 re-draws the partner
        modules of a loaded model with init_partner=True)"""
        gains = {"pi_W1": np.sqrt(2), "pi_W2": np.sqrt(2), "vf_W1": np.sqrt(2), "vf_W2": np.sqrt(2), "act_W": 0.01, "val_W": 1.0}
        flat = self.params.detach().cpu().clone()
        if init_main:
            for name, (fin, fout) in self._shapes().items():
                w = th.empty(fout, fin)
                th.nn.init.orthogonal_(w, gain=gains[name])
                off = getattr(self.layout, name)
                flat[off:off + fin * fout] = w.t().contiguous().reshape(-1)
            for b in ("pi_b1", "pi_b2", "vf_b1", "vf_b2", "act_b", "val_b"):
                off = getattr(self.layout, b)
                flat[off:off + {"act_b": self.layout.L, "val_b": 1}.get(b, HID)] = 0
        if init_partner:
            ml = self.module_layout
            for m in range(self.n_modules):
                base = self.module_offset(m)
                flat[base:base + ml.P] = 0
                for name, (fin, fout) in self._module_shapes().items():
                    w = th.empty(fout, fin)
                    if self._ortho:
                        th.nn.init.orthogonal_(w, gain=gains[name])
                    else:
                        th.nn.init.kaiming_uniform_(w, a=np.sqrt(5))
                    off = base + getattr(ml, name)
                    flat[off:off + fin * fout] = w.t().contiguous().reshape(-1)
        self.params.copy_(flat)

    def state_dict(self) -> Dict[str, th.Tensor]:
        flat, out = self.params.detach().cpu(), {}
        shapes = self._shapes()
        for mod, wname, bname in _SD:
            fin, fout = shapes[wname]
            woff, boff = getattr(self.layout, wname), getattr(self.layout, bname)
            out[mod + ".weight"] = flat[woff:woff + fin * fout].reshape(fin, fout).t().contiguous()
            out[mod + ".bias"] = flat[boff:boff + fout].clone()
        ms, ml = self._module_shapes(), self.module_layout
        for k in range(self.num_partners):       # `baseline` partners alias module 0 (policies.py:255-257): written per partner
            base = self.module_offset(int(self.mod.module_of[k]))
            for mod, wname, bname in _MOD_SD:
                fin, fout = ms[wname]
                woff, boff = base + getattr(ml, wname), base + getattr(ml, bname)
                out[mod.format(k=k) + ".weight"] = flat[woff:woff + fin * fout].reshape(fin, fout).t().contiguous()
                out[mod.format(k=k) + ".bias"] = flat[boff:boff + fout].clone()
        return out

    def load_state_dict(self, sd: Dict[str, th.Tensor]) -> None:
        flat, shapes = self.params.detach().cpu().clone(), self._shapes()
        for mod, wname, bname in _SD:
            fin, fout = shapes[wname]
            woff, boff = getattr(self.layout, wname), getattr(self.layout, bname)
            flat[woff:woff + fin * fout] = th.as_tensor(sd[mod + ".weight"]).float().reshape(fout, fin).t().reshape(-1)
            flat[boff:boff + fout] = th.as_tensor(sd[mod + ".bias"]).float().reshape(-1)
        ms, ml = self._module_shapes(), self.module_layout
        for k in range(self.num_partners):
            if (_MOD_SD[0][0].format(k=k) + ".weight") not in sd:
                continue                       # a checkpoint with fewer partners: the others keep their initialisation
            base = self.module_offset(int(self.mod.module_of[k]))
            for mod, wname, bname in _MOD_SD:
                fin, fout = ms[wname]
                woff, boff = base + getattr(ml, wname), base + getattr(ml, bname)
                flat[woff:woff + fin * fout] = th.as_tensor(sd[mod.format(k=k) + ".weight"]).float().reshape(fout, fin).t().reshape(-1)
                flat[boff:boff + fout] = th.as_tensor(sd[mod.format(k=k) + ".bias"]).float().reshape(-1)
        self.params.copy_(flat)

    def overwrite_main(self, other: ActorCriticPolicy) -> None:
        """policies.py:267-269: take another policy's main network (and a fresh optimizer)"""
        self.params[:self.layout.P].copy_(other.params[:self.layout.P])
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.opt_step.zero_()
        self.mod_first.fill_(-1)

    # -- forward family (every call names the partner whose module composes the heads) --------------------------------
    def _launch(self, obs_t, *, partner_idx: int = 0, mask=None, uniforms=None, given=None, deterministic=False,
                want_logits=False, want_entropy=False, rb: Optional[RolloutBuffer] = None, pos: int = 0, episode_start=None):
        n, lay, dev = obs_t.shape[0], self.layout, self.device
        acts = th.empty((n, 1), dtype=th.int32, device=dev)
        values = th.empty((n, 1), dtype=th.float32, device=dev)
        logp = th.empty((n,), dtype=th.float32, device=dev)
        zm = th.empty((n, lay.L), dtype=th.float32, device=dev) if want_logits else None
        zp = th.empty((n, lay.L), dtype=th.float32, device=dev) if want_logits else None
        ent = th.empty((n,), dtype=th.float32, device=dev) if want_entropy else None
        m = None if mask is None else th.as_tensor(mask).to(device=dev, dtype=th.uint8).reshape(n, lay.L).contiguous()
        u = None if uniforms is None else _f32_dev(uniforms, dev, (n, 1))
        g = None if given is None else _f32_dev(given, dev, (n, 1))
        es = None if episode_start is None else _f32_dev(episode_start, dev, (n,))
        self._bind()
        self._counter += 1
        nat.check(self.ctx.lib.ph_modular_forward(
            self.ctx.handle, C.byref(self.spec), C.byref(self.mod), self.params.data_ptr(), int(partner_idx), obs_t.data_ptr(), n,
            nat.ptr(m), nat.ptr(u), nat.ptr(g), self._seed, self._counter, int(bool(deterministic)), acts.data_ptr(), None,
            values.data_ptr(), logp.data_ptr(), nat.ptr(ent), nat.ptr(zm), nat.ptr(zp),
            C.byref(rb.c_struct()) if rb is not None else None, int(pos), nat.ptr(es), None, int(self.gemm_mode)))
        return acts, values, logp, ent, (zm, zp)

    def forward(self, obs, partner_idx: int = 0, deterministic: bool = False, action_mask=None, uniforms=None):
        """policies.py:271-288 -> (actions, values (n, 1), log_prob (n,))"""
        acts, values, logp, _, _ = self._launch(self._obs(obs), partner_idx=partner_idx, mask=action_mask, uniforms=uniforms,
                                                deterministic=deterministic)
        return self._shape_actions(acts), values, logp

    __call__ = forward

    def forward_and_store(self, obs, rb: RolloutBuffer, episode_start, partner_idx: int = 0, deterministic: bool = False,
                          action_mask=None, uniforms=None):
        if rb.pos >= rb.buffer_size:
            raise nat.NativeError("RolloutBuffer.add on a full buffer")
        acts, values, logp, _, _ = self._launch(self._obs(obs), partner_idx=partner_idx, mask=action_mask, uniforms=uniforms,
                                                deterministic=deterministic, rb=rb, pos=rb.pos, episode_start=episode_start)
        rb.pos += 1
        rb.full = rb.pos == rb.buffer_size
        return self._shape_actions(acts), values, logp

    def evaluate_actions(self, obs, actions, partner_idx: int = 0, action_mask=None):
        """policies.py:364-383 -> (values (n, 1), log_prob (n,), entropy (n,))"""
        _, values, logp, ent, _ = self._launch(self._obs(obs), partner_idx=partner_idx, mask=action_mask, given=actions,
                                               want_entropy=True)
        return values, logp, ent

    def get_action_logits_from_obs(self, obs, partner_idx: int = 0, action_mask=None):
        """policies.py:385-395 -> (main_logits, partner_logits); a mask multiplies both (sets masked options to 0)"""
        _, _, _, _, (zm, zp) = self._launch(self._obs(obs), partner_idx=partner_idx, deterministic=True, want_logits=True)
        if action_mask is not None:
            mk = th.as_tensor(action_mask).to(device=self.device, dtype=th.float32).reshape(zm.shape)
            zm, zp = zm * mk, zp * mk
        return zm, zp

    def predict_values(self, obs, partner_idx: int = 0) -> th.Tensor:
        return self._launch(self._obs(obs), partner_idx=partner_idx, deterministic=True)[1]

    def get_logits(self, obs, action_mask=None, partner_idx: int = 0) -> th.Tensor:
        zm, zp = self.get_action_logits_from_obs(obs, partner_idx)
        z = zp if self.nomain else zm + zp
        if action_mask is not None:
            z = z - 30.0 * (1.0 - th.as_tensor(action_mask).to(device=self.device, dtype=th.float32).reshape(z.shape))
        return z

    def predict(self, obs, deterministic: bool = False, partner_idx: int = 0):
        acts, _, _ = self.forward(obs, partner_idx=partner_idx, deterministic=deterministic)
        return acts.cpu().numpy(), None


class ModularAlgorithm(PPO):
    """pantheonrl.algos.modular.learn.ModularAlgorithm: the constructor surface of learn.py:27-54 on top of `PPO`
    (`policy_kwargs` carries num_partners / baseline / nomain, trainer.py:131-135)."""

    def __init__(self, policy=ModularPolicy, env=None, *args, policy_kwargs: Optional[Dict[str, Any]] = None,
                 marginal_reg_coef: float = 0.0, **kwargs):
        if policy not in ("ModularPolicy", ModularPolicy):
            raise ValueError("ModularAlgorithm runs ModularPolicy")
        self.policy_kwargs = dict(policy_kwargs or {})
        self.marginal_reg_coef = float(marginal_reg_coef)
        if kwargs.get("batch_size", 64) <= 1:
            raise ValueError("`batch_size` must be greater than 1 (learn.py:88-90)")
        super().__init__("MlpPolicy", env, *args, **kwargs)

    _HP = PPO._HP + ("marginal_reg_coef", "policy_kwargs")

    def _setup_model(self) -> None:                     # learn.py:117-155
        self.policy = ModularPolicy(self.observation_space, self.action_space, lr=self.learning_rate, device=self.device,
                                    seed=self.seed, sampling_stream=self.sampling_stream, **self.policy_kwargs)
        self.rollout_buffer = [RolloutBuffer(self.n_steps, self.observation_space, self.action_space, self.device,
                                             self.policy.ctx, self.policy.spec, gae_lambda=self.gae_lambda, gamma=self.gamma,
                                             n_envs=self.n_envs) for _ in range(self.policy.num_partners)]

    # -- learn.py:155-218 -------------------------------------------------------------------------------------------
    def _set_partnerid(self, partner_idx: int) -> None:
        """`env.envs[0].set_partnerid(partner_idx)` (learn.py:191,376-379): every environment behind the vector wrapper that
        has partners switches to `partner_idx`; an environment without set_partnerid is left alone ("unable to switch")"""
        env = self.env
        for e in (getattr(env, "envs", None) or [getattr(env, "env", env)]):
            fn = getattr(e, "set_partnerid", None)
            if fn is not None:
                try:
                    fn(partner_idx)
                except Exception:  # noqa: BLE001 -- learn.py:377-379
                    pass

    def collect_rollouts(self, partner_idx: int = 0, forced_uniforms=None, callback=None) -> bool:
        env, rb, pol = self.env, self.rollout_buffer[partner_idx], self.policy
        rb.reset()
        if callback is not None and hasattr(callback, "on_rollout_start"):
            callback.on_rollout_start()
        last_dones = None          # learn.py:176: _last_dones = None at the start of every rollout
        new_obs, dones = self._last_obs, np.zeros(self.n_envs, dtype=bool)
        values = None
        for t in range(self.n_steps):
            # The reference hands the buffer `self._last_dones`, None on the first step of a rollout (learn.py:176,207): numpy
            # stores that as nan in row 0 of episode_starts, a row GAE never reads (it reads rows 1 .. T-1 and `dones`).  Row 0
            # is written as 0 here.
            es = np.zeros(self.n_envs, np.float32) if last_dones is None else np.asarray(last_dones, np.float32)
            actions, values, _ = pol.forward_and_store(self._last_obs, rb, es, partner_idx=partner_idx,
                                                       uniforms=None if forced_uniforms is None else forced_uniforms[t])
            self._set_partnerid(partner_idx)
            new_obs, rewards, dones, infos = env.step(actions.cpu().numpy())
            self.num_timesteps += self.n_envs
            if callback is not None and hasattr(callback, "on_step"):
                if hasattr(callback, "update_locals"):
                    callback.update_locals(locals())
                if callback.on_step() is False:
                    return False
            for info in infos:
                ep = info.get("episode") if isinstance(info, dict) else None
                if ep is not None:
                    self.ep_info_buffer.append(ep)
            rb.add_reward(np.asarray(rewards, np.float32))
            self._last_obs, last_dones = new_obs, dones
        # learn.py:214: bootstrap with the LAST STEP's values (not V(new_obs)) and that step's dones
        rb.compute_returns_and_advantage(last_values=values, dones=np.asarray(dones, np.float32))
        return True

    # -- learn.py:221-351 -------------------------------------------------------------------------------------------
    def train(self, perms=None, sync_stats: bool = True) -> None:
        """`perms[k][epoch]` (num_partners, n_epochs, T*E) teacher-forces the buffers' permutations; default
        np.random.permutation per partner and epoch (SB3's RolloutBuffer.get)."""
        pol, K = self.policy, self.policy.num_partners
        rb0 = self.rollout_buffer[0]
        N = rb0.buffer_size * rb0.n_envs
        n_mb = (N + self.batch_size - 1) // self.batch_size
        perm_t = None
        if perms is None and not self.device_permutations:
            perms = np.stack([np.stack([np.random.permutation(N) for _ in range(self.n_epochs)]) for _ in range(K)])
        if perms is not None:
            perm_t = th.as_tensor(np.ascontiguousarray(perms, dtype=np.int32)).to(self.device)
            assert perm_t.shape == (K, self.n_epochs, N)
        stats = th.zeros((K * self.n_epochs * n_mb, nat.PH_NSTAT), dtype=th.float32, device=self.device)
        opt = nat.PhOptState()
        opt.params, opt.adam_m, opt.adam_v = pol.params.data_ptr(), pol.adam_m.data_ptr(), pol.adam_v.data_ptr()
        opt.step = pol.opt_step.data_ptr()
        hp = self.hyper()
        rbs = (nat.PhRollout * K)(*[rb.c_struct() for rb in self.rollout_buffer])
        pol._bind()
        self.permutation_seed += 1
        nat.check(pol.ctx.lib.ph_modular_train(pol.ctx.handle, C.byref(pol.spec), C.byref(pol.mod), C.byref(opt),
                                               pol.mod_first.data_ptr(), rbs, C.byref(hp), int(self.n_epochs),
                                               int(self.batch_size), nat.ptr(perm_t), int(self.permutation_seed),
                                               stats.data_ptr(), float(self.marginal_reg_coef), int(pol.gemm_mode)))
        self._n_updates += self.n_epochs
        if sync_stats:
            st = stats.cpu().numpy()
            self.last_train_stats = st.reshape(K, self.n_epochs * n_mb, nat.PH_NSTAT)
            ran = st[np.abs(st).sum(1) > 0]
            if len(ran):
                lg = self.logger                  # learn.py:339-342 (the other keys are commented out in the reference)
                lg.record("train/entropy_loss", float(ran[:, 2].mean()))
                lg.record("train/policy_gradient_loss", float(ran[:, 0].mean()))
                lg.record("train/value_loss", float(ran[:, 1].mean()))
                lg.record("train/marginal_reg_loss", float(ran[:, 7].mean()))

    def learn(self, total_timesteps: int, log_interval: int = 1, tb_log_name: str = "OnPolicyAlgorithm",
              reset_num_timesteps: bool = True, callback=None, **_ignored) -> "ModularAlgorithm":
        """learn.py:353-403: every iteration collects one rollout PER PARTNER (set_partnerid before each), then trains"""
        import time
        from .logger import configure_logger
        if self.env is None or not hasattr(self.env, "step"):
            raise ValueError("learn() needs a steppable environment")
        if not self._custom_logger:
            self._logger = configure_logger(self.verbose, self.tensorboard_log, tb_log_name)
        if not reset_num_timesteps:
            total_timesteps += self.num_timesteps
        if reset_num_timesteps or self._last_obs is None:
            if reset_num_timesteps:
                self.num_timesteps = 0
            self._last_obs = self.env.reset()
        self.start_time = time.time()
        start_steps, iteration = self.num_timesteps, 0
        cb_obj = callback if hasattr(callback, "on_rollout_end") else None
        if cb_obj is not None and hasattr(cb_obj, "init_callback"):
            cb_obj.init_callback(self)
        if cb_obj is not None and hasattr(cb_obj, "on_training_start"):
            cb_obj.on_training_start(locals(), globals())
        while self.num_timesteps < total_timesteps:
            go_on = True
            for k in range(self.policy.num_partners):
                self._set_partnerid(k)
                go_on = self.collect_rollouts(partner_idx=k, callback=cb_obj)
                if cb_obj is not None and go_on:
                    cb_obj.on_rollout_end()
            if go_on is False:
                break
            iteration += 1
            self._current_progress_remaining = 1.0 - float(self.num_timesteps - start_steps) / float(
                max(total_timesteps - start_steps, 1))
            if callable(callback) and cb_obj is None and callback(locals(), globals()) is False:
                break
            if log_interval is not None and iteration % log_interval == 0:
                lg = self.logger
                lg.record("time/iterations", iteration, exclude="tensorboard")
                if len(self.ep_info_buffer) > 0:
                    lg.record("rollout/ep_rew_mean", float(np.mean([e["r"] for e in self.ep_info_buffer])))
                    lg.record("rollout/ep_len_mean", float(np.mean([e["l"] for e in self.ep_info_buffer])))
                lg.record("time/fps", int(self.num_timesteps / max(time.time() - self.start_time, 1e-9)))
                lg.record("time/time_elapsed", int(time.time() - self.start_time), exclude="tensorboard")      # learn.py:396
                lg.record("time/total_timesteps", self.num_timesteps, exclude="tensorboard")
                lg.dump(step=self.num_timesteps)
            self.train()
        if cb_obj is not None and hasattr(cb_obj, "on_training_end"):
            cb_obj.on_training_end()
        return self

    def predict(self, obs, deterministic: bool = False, partner_idx: int = 0):
        return self.policy.predict(obs, deterministic, partner_idx=partner_idx)

    def set_num_partners(self, num_partners: int) -> None:
        """trainer.py:121-123 for a LOADed ego: `policy.do_init_weights(init_partner=True); policy.num_partners = len(args.alt)`
        -- the main network is kept, the partner modules are drawn afresh for this run's partners.  (The reference only
        re-initialises the modules it already has and then changes the count, which breaks on a different number of partners;
        here the module list is rebuilt for the new count.)"""
        pol = self.policy
        main = pol.params[:pol.layout.P].clone()
        pol.num_partners = int(num_partners)
        self.policy_kwargs["num_partners"] = int(num_partners)
        pol.params = main
        pol._build_modules()
        pol.opt_step.zero_()
        self.rollout_buffer = [RolloutBuffer(self.n_steps, self.observation_space, self.action_space, self.device,
                                             pol.ctx, pol.spec, gae_lambda=self.gae_lambda, gamma=self.gamma,
                                             n_envs=self.n_envs) for _ in range(pol.num_partners)]

    # -- save / load: PPO's container + the per-module optimizer bookkeeping -------------------------------------------
    def _extra_state(self) -> dict:
        return {"mod_first": [int(v) for v in self.policy.mod_first.cpu().numpy()]}

    def _load_extra_state(self, extra: dict) -> None:
        mf = extra.get("mod_first")
        if mf is not None and len(mf) == self.policy.n_modules:
            self.policy.mod_first.copy_(th.as_tensor(np.asarray(mf, np.int32)))
