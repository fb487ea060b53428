"""ADAP on the engine: PantheonRL's ADAP learner (pantheonrl/algos/adap) behind the same surface.

ADAP (adap_learn.py:30-227) is PPO over a policy that also reads a latent context vector, plus one extra loss term:

* ``AdapPolicy`` (adap/policies.py:21-146) feeds ``features ++ context`` to the ordinary MlpExtractor, so on the engine it IS the
  MlpPolicy over a Box of ``observation + context_size`` components: the forward, GAE and PPO gradient kernels run unchanged.
* the rollout stores every observation with the context that was active (adap_learn.py:448-452, agent.py:117-121); the context is
  re-drawn when an episode ends (adap_learn.py:457-461, agent.py:147-151).
* ``train()`` adds ``context_loss_coeff * get_context_kl_loss`` to every minibatch loss (adap_learn.py:313-320, util.py:97-131):
  ``ph_adap_train`` -- one small launch per minibatch next to the PPO gradient launch, folded into the same gradient reduction,
  clip and Adam step (include/pantheon_hip.h).

``AdapPolicyMult`` (policies.py:136-283: the multiplicative-latent extractor, a different network) has its own parameter layout and
launch chain (csrc/ph_adapmult.hip) behind the same classes: ``ADAP("AdapPolicyMult", ...)``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch as th

from . import _native as nat
from . import spaces as sp
from .common.agents import OnPolicyAgent
from .ppo import PPO, ActorCriticPolicy, RolloutBuffer


def _l2(ctx_size, num, rng):
    c = rng.random((num, ctx_size)).astype(np.float32) * 2 - 1
    return c / np.sqrt(np.sum(c ** 2, axis=-1, dtype=np.float32)).reshape(num, 1)


def _categorical(ctx_size, num, rng):
    c = np.zeros((num, ctx_size), np.float32)
    c[np.arange(num), rng.integers(0, ctx_size, size=num)] = 1
    return c


# adap/util.py:42-95.  "natural_numbers" (util.py:80-89) yields a (num, 1) vector of integers in [0, ctx_size) whatever
# context_size is: AdapPolicy.set_context only takes it when context_size == 1 (and the only integer is then 0), which is the
# one configuration the engine accepts it in.
SAMPLERS = {"l2": _l2,
            "unit_square": lambda n, num, rng: rng.random((num, n)).astype(np.float32) * 2 - 1,
            "positive_square": lambda n, num, rng: rng.random((num, n)).astype(np.float32),
            "categorical": _categorical,
            "natural_numbers": lambda n, num, rng: rng.integers(0, n, size=(num, 1)).astype(np.float32)}


class AdapPolicy(ActorCriticPolicy):
    """adap/policies.py:21-146.  ``observation_space`` is the ENVIRONMENT's space, like the reference's; the network input is
    ``features ++ context`` (policies.py:104-119), so the engine sees a Box of (features + context_size) components.  For a
    Discrete / MultiDiscrete environment the features are the one-hot encoding SB3's preprocess_obs builds: the rows this
    policy hands to the engine (and that the rollout buffer therefore stores) are the features, not the raw integers."""

    host_step_path = False   # the rows are built on the device (features ++ context, _obs)

    def __init__(self, observation_space, action_space, context_size: int = 3, **kw):
        kind = sp._kind(observation_space)
        if kind == "Box":
            self._nvec, feat = None, int(np.prod(sp.obs_stored_shape(observation_space)))
        elif kind in ("Discrete", "MultiDiscrete"):
            self._nvec = [int(observation_space.n)] if kind == "Discrete" else [int(v) for v in observation_space.nvec]
            feat = int(sum(self._nvec))
        else:
            raise ValueError(f"ADAP on the engine: unsupported observation space {observation_space!r}")
        self.context_size = int(context_size)
        self.full_observation_space = sp.Box(-np.inf, np.inf, (feat + self.context_size,))
        super().__init__(self.full_observation_space, action_space, **kw)
        self.observation_space = observation_space
        self.env_obs_len = int(np.prod(sp.obs_stored_shape(observation_space)))   # stored length of a raw observation
        self.context: Optional[np.ndarray] = None   # (1, ctx) shared by every row, or (n, ctx) one per environment column
        self._context_dev: Optional[th.Tensor] = None

    def set_context(self, ctxt) -> None:            # policies.py:96-97
        self.context = np.asarray(ctxt.detach().cpu() if isinstance(ctxt, th.Tensor) else ctxt, np.float32).reshape(
            -1, self.context_size)
        self._context_dev = th.as_tensor(self.context).to(self.device)   # one upload per change, not one per forward

    def get_context(self) -> np.ndarray:            # policies.py:99-100
        return self.context

    def features(self, raw: th.Tensor) -> th.Tensor:
        """SB3 preprocess_obs + FlattenExtractor of bare environment observations (n, env_obs_len)"""
        if self._nvec is None:
            return raw
        idx = raw.long()
        parts = [th.nn.functional.one_hot(idx[:, i].clamp(0, n - 1), n).float() for i, n in enumerate(self._nvec)]
        return th.cat(parts, dim=1)

    def _obs(self, obs) -> th.Tensor:
        """rows that already carry a context pass (evaluate_actions, policies.py:121-134); bare environment observations
        get the current one appended (_get_latent, policies.py:104-119).  Full rows are 2-D and always longer than raw ones."""
        t = obs.detach() if isinstance(obs, th.Tensor) else th.as_tensor(np.asarray(obs))
        t = t.to(device=self.device, dtype=th.float32)
        if t.ndim >= 2 and t.shape[-1] == self.layout.D:
            return t.reshape(-1, self.layout.D).contiguous()
        t = self.features(t.reshape(-1, self.env_obs_len))
        if self.context is None:
            raise nat.NativeError("AdapPolicy: no context set")
        c = self._context_dev
        if c.shape[0] != t.shape[0]:
            if c.shape[0] != 1:
                raise nat.NativeError(f"AdapPolicy: {c.shape[0]} contexts for {t.shape[0]} observation rows")
            c = c.expand(t.shape[0], -1)
        return th.cat((t, c), dim=1).contiguous()


class AdapPolicyMult(AdapPolicy):
    """adap/policies.py:136-283: AdapPolicy whose extractor is MultModel -- per net  x = tanh(W1 o + b1),
    x_a = tanh(Ws x + bs) viewed as (64, C),  latent = tanh(W2 (x + x_a @ ctx) + b2)  on rows features ++ context.  Same rows, same
    rollout buffer and same surface as AdapPolicy; the parameters follow ph_adapmult_layout (include/pantheon_hip.h) and the
    network runs as a chain of small launches (csrc/ph_adapmult.hip)."""

    # reference module names (policies.py:207-236) -> (weight offset, bias offset, fan in, fan out); gains: SB3's init_weights loop
    # reaches every Linear of mlp_extractor with sqrt(2) (modular/policies.py:229-241 is the same loop)
    fused_mlp_kernels = False   # parameters follow ph_adapmult_layout: only ADAP's own launches (ph_adapmult_*) may read them

    _MODS = (("mlp_extractor.agent_branch_1.0", "pi_W1", "pi_b1"), ("mlp_extractor.agent_scaling.0", "pi_Ws", "pi_bs"),
             ("mlp_extractor.agent_branch_2.0", "pi_W2", "pi_b2"), ("mlp_extractor.value_branch_1.0", "vf_W1", "vf_b1"),
             ("mlp_extractor.value_scaling.0", "vf_Ws", "vf_bs"), ("mlp_extractor.value_branch_2.0", "vf_W2", "vf_b2"),
             ("action_net", "act_W", "act_b"), ("value_net", "val_W", "val_b"))

    def __init__(self, observation_space, action_space, context_size: int = 3, **kw):
        ortho = kw.get("ortho_init", True)
        super().__init__(observation_space, action_space, context_size=context_size, **kw)
        self.mlayout = nat.adapmult_layout_of(self.spec, self.context_size)      # raises for shapes the device path does not take
        P = self.mlayout.P
        self.params = th.zeros(P, dtype=th.float32, device=self.device)
        self.adam_m = th.zeros_like(self.params)
        self.adam_v = th.zeros_like(self.params)
        self._init_weights(ortho)

    def _mshapes(self):
        m, H, C = self.mlayout, 64, self.context_size
        return {"pi_W1": (m.Fo, H), "pi_Ws": (H, H * C), "pi_W2": (H, H), "vf_W1": (m.Fo, H), "vf_Ws": (H, H * C), "vf_W2": (H, H),
                "act_W": (H, m.L), "val_W": (H, 1)}

    def _init_weights(self, ortho_init: bool) -> None:
        if not hasattr(self, "mlayout"):      # the base constructor's call: its MlpPolicy-sized vector is replaced right after
            return super()._init_weights(ortho_init)
        flat = th.zeros(self.mlayout.P, dtype=th.float32)
        for name, (fin, fout) in self._mshapes().items():
            w = th.empty(fout, fin)
            gain = 0.01 if name == "act_W" else (1.0 if name == "val_W" else float(np.sqrt(2)))
            if ortho_init:
                th.nn.init.orthogonal_(w, gain=gain)
            else:
                th.nn.init.kaiming_uniform_(w, a=np.sqrt(5))
            off = getattr(self.mlayout, name)
            flat[off:off + fin * fout] = w.t().contiguous().reshape(-1)
        self.params.copy_(flat)

    def state_dict(self):
        flat, m, shapes, out = self.params.detach().cpu(), self.mlayout, self._mshapes(), {}
        for mod, wname, bname in self._MODS:
            fin, fout = shapes[wname]
            woff, boff = getattr(m, wname), getattr(m, bname)
            out[mod + ".weight"] = flat[woff:woff + fin * fout].reshape(fin, fout).t().contiguous()
            out[mod + ".bias"] = flat[boff:boff + fout].clone()
        return out

    def load_state_dict(self, sd) -> None:
        flat, m, shapes = th.zeros(self.mlayout.P), self.mlayout, self._mshapes()
        for mod, wname, bname in self._MODS:
            fin, fout = shapes[wname]
            woff, boff = getattr(m, wname), getattr(m, bname)
            flat[woff:woff + fin * fout] = th.as_tensor(sd[mod + ".weight"]).float().reshape(fout, fin).t().reshape(-1)
            flat[boff:boff + fout] = th.as_tensor(sd[mod + ".bias"]).float().reshape(-1)
        self.params.copy_(flat)

    def _launch(self, obs_t, *, mask=None, uniforms=None, given=None, deterministic=False, want_logits=False,
                want_entropy=False, rb: Optional[RolloutBuffer] = None, pos: int = 0, episode_start=None):
        from .ppo import _f32_dev
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
        nat.check(self.ctx.lib.ph_adapmult_forward(
            self.ctx.handle, C.byref(self.spec), int(self.context_size), self.params.data_ptr(), obs_t.data_ptr(), n, nat.ptr(m),
            nat.ptr(u), nat.ptr(g), self._seed, self._counter, int(bool(deterministic)), acts.data_ptr(), None, values.data_ptr(),
            logp.data_ptr(), nat.ptr(ent), nat.ptr(logits), C.byref(rb.c_struct()) if rb is not None else None, int(pos),
            nat.ptr(es)))
        return acts, values, logp, ent, logits


class ADAP(PPO):
    """pantheonrl.algos.adap.adap_learn.ADAP: same constructor surface (adap_learn.py:86-117) on top of `PPO`."""

    def __init__(self, policy=AdapPolicy, env=None, *args, context_loss_coeff: float = 0.1, context_size: int = 3,
                 num_context_samples: int = 5, context_sampler: str = "l2", num_state_samples: int = 32,
                 policy_kind: Optional[str] = None, **kwargs):
        policy = policy_kind or policy          # (checkpoints name the policy class: ADAP.load passes policy_kind)
        if policy in ("AdapPolicy", AdapPolicy):
            self._policy_cls = AdapPolicy
        elif policy in ("AdapPolicyMult", AdapPolicyMult):
            self._policy_cls = AdapPolicyMult
        else:
            raise ValueError(f"ADAP policies: 'AdapPolicy' (concatenated context) or 'AdapPolicyMult' (multiplicative), not {policy!r}")
        self.policy_kind = self._policy_cls.__name__
        if context_sampler not in SAMPLERS:
            raise ValueError(f"unknown context sampler {context_sampler!r} (one of {sorted(SAMPLERS)})")
        if context_sampler == "natural_numbers" and int(context_size) != 1:
            raise ValueError("context_sampler='natural_numbers' draws (num, 1) contexts (adap/util.py:80-89): it needs context_size=1")
        self.context_loss_coeff, self.context_size = float(context_loss_coeff), int(context_size)
        self.num_context_samples, self.num_state_samples = int(num_context_samples), int(num_state_samples)
        self.context_sampler = context_sampler
        # One generator per learner: trainer.py hands the ego and every ADAP partner the same --seed, and the reference draws all
        # contexts from ONE global torch stream (adap/util.py:42-77), so agents never see each other's draws.  The seed is
        # therefore salted with the learner's sampling stream (as the action-sampling key is); its state travels with save/load.
        seed, stream = kwargs.get("seed"), int(kwargs.get("sampling_stream", 0) or 0)
        self.context_rng = np.random.default_rng() if seed is None else np.random.default_rng([int(seed), stream])
        self.last_context_losses: Optional[np.ndarray] = None
        super().__init__("MlpPolicy", env, *args, **kwargs)

    _HP = PPO._HP + ("context_loss_coeff", "context_size", "num_context_samples", "context_sampler", "num_state_samples",
                     "policy_kind")

    def sample_context(self, num: int = 1) -> np.ndarray:
        return SAMPLERS[self.context_sampler](self.context_size, num, self.context_rng)

    def _extra_state(self) -> dict:                 # saved under "extra" by PPO.save
        ctx = self.policy.get_context()
        return {"context_rng": self.context_rng.bit_generator.state,
                "context": None if ctx is None else np.asarray(ctx, np.float32).tolist()}

    def _load_extra_state(self, extra: dict) -> None:
        if extra.get("context_rng") is not None:    # continue the stream instead of replaying it from the seed
            self.context_rng.bit_generator.state = extra["context_rng"]
        if extra.get("context") is not None:
            self.policy.set_context(np.asarray(extra["context"], np.float32))

    def _setup_model(self) -> None:                 # adap_learn.py:208-215
        self.policy = self._policy_cls(self.observation_space, self.action_space, context_size=self.context_size,
                                       lr=self.learning_rate, device=self.device, seed=self.seed,
                                       sampling_stream=self.sampling_stream)
        self.rollout_buffer = RolloutBuffer(self.n_steps, self.policy.full_observation_space, self.action_space, self.device,
                                            self.policy.ctx, self.policy.spec, gae_lambda=self.gae_lambda, gamma=self.gamma,
                                            n_envs=self.n_envs)
        self.full_obs_shape = self.rollout_buffer.obs_shape
        # one context per environment column (the reference runs one environment: one context)
        self.policy.set_context(self.sample_context(self.n_envs))

    # -- ADAP.train(): PPO.train() with the context term ---------------------------------------------------------------
    def adap_struct(self, n_mb_total: int, state_idx=None, contexts=None, keep: Optional[list] = None) -> nat.PhAdapLoss:
        ad = nat.PhAdapLoss()
        ad.context_size, ad.num_context_samples = self.context_size, self.num_context_samples
        ad.num_state_samples, ad.sampler = self.num_state_samples, nat.CONTEXT_SAMPLERS[self.context_sampler]
        ad.context_loss_coeff, ad.seed = self.context_loss_coeff, int(self.permutation_seed)
        keep = keep if keep is not None else []
        if state_idx is not None:
            t = th.as_tensor(np.ascontiguousarray(state_idx, dtype=np.int32)).to(self.device)
            assert t.shape == (n_mb_total, self.num_state_samples)
            keep.append(t)
            ad.state_idx = t.data_ptr()
        if contexts is not None:
            t = th.as_tensor(np.ascontiguousarray(contexts, dtype=np.float32)).to(self.device)
            assert t.shape == (n_mb_total, self.num_context_samples, self.context_size)
            keep.append(t)
            ad.contexts = t.data_ptr()
        cl = getattr(self, "_ctx_loss_dev", None)
        if cl is None or cl.shape[0] != n_mb_total:
            cl = th.zeros(n_mb_total, dtype=th.float32, device=self.device)
        self._ctx_loss_dev = cl
        ad.context_loss = cl.data_ptr()
        self._adap_keep = keep
        return ad

    def _train_native(self, pol, opt, rb, hp, perm_t, stats) -> None:
        N = rb.buffer_size * rb.n_envs
        n_mb = (N + self.batch_size - 1) // self.batch_size
        forced = getattr(self, "_forced_samples", None) or (None, None)
        ad = self.adap_struct(self.n_epochs * n_mb, forced[0], forced[1])
        if isinstance(pol, AdapPolicyMult):
            nat.check(pol.ctx.lib.ph_adapmult_train(pol.ctx.handle, C.byref(pol.spec), int(pol.context_size), C.byref(opt),
                                                    C.byref(rb.c_struct()), C.byref(hp), int(self.n_epochs), int(self.batch_size),
                                                    nat.ptr(perm_t), int(self.permutation_seed), stats.data_ptr(), C.byref(ad)))
            return
        nat.check(pol.ctx.lib.ph_adap_train(pol.ctx.handle, C.byref(pol.spec), C.byref(opt), C.byref(rb.c_struct()),
                                            C.byref(hp), int(self.n_epochs), int(self.batch_size), nat.ptr(perm_t),
                                            int(self.permutation_seed), stats.data_ptr(), int(pol.gemm_mode), C.byref(ad)))

    def train(self, perms: Optional[np.ndarray] = None, sync_stats: bool = True, state_idx=None, contexts=None) -> None:
        """`state_idx` (n_minibatches, num_state_samples) / `contexts` (n_minibatches, num_context_samples, context_size)
        teacher-force th.randperm and the sampler of util.py:106,113-114 (tests); default: drawn in the kernel"""
        self._forced_samples = (state_idx, contexts)
        try:
            super().train(perms=perms, sync_stats=sync_stats)
        finally:
            self._forced_samples = None
        if sync_stats:
            st, cl = self.last_train_stats, self._ctx_loss_dev.cpu().numpy()
            applied = st[:, 7] > 0
            n_used = max(int(applied.sum()) + (1 if not applied.all() else 0), 1)
            self.last_context_losses = cl
            # adap_learn.py:359 -- of the LAST epoch that ran: `context_kl_divs` is emptied at the top of every epoch (:250)
            rb = self.rollout_buffer
            n_mb = -(-(rb.buffer_size * rb.n_envs) // self.batch_size)
            self.logger.record("train/context_kl_loss", float(cl[((n_used - 1) // n_mb) * n_mb:n_used].mean()))

    # -- ADAP.collect_rollouts (adap_learn.py:375-473): a fresh context for every column whose episode ended ----------
    def _after_step(self, dones) -> None:
        done = np.asarray(dones, bool).reshape(-1)
        if done.any():
            ctx = self.policy.get_context()
            if ctx.shape[0] == 1 and self.n_envs > 1:
                ctx = np.repeat(ctx, self.n_envs, axis=0)
            ctx = ctx.copy()
            ctx[done] = self.sample_context(int(done.sum()))
            self.policy.set_context(ctx)


    def _terminal_value(self, terminal_obs: np.ndarray, env_index: int) -> th.Tensor:
        ctx = self.policy.get_context()
        row = ctx[env_index if ctx.shape[0] > 1 else 0].reshape(1, -1)
        pol = self.policy
        feats = pol.features(th.as_tensor(np.asarray(terminal_obs, np.float32)).to(pol.device).reshape(1, pol.env_obs_len))
        return pol.predict_values(th.cat((feats, th.as_tensor(row).to(pol.device)), dim=1))


class AdapAgent(OnPolicyAgent):
    """pantheonrl.algos.adap.agent.AdapAgent: an OnPolicyAgent whose model is an ADAP learner.  `latent_syncer` (the ego's
    policy) makes the partner act under the ego's current context (agent.py:72-73, trainer.py:213); without it the agent
    re-draws its own context when an episode ends (agent.py:147-151)."""

    def __init__(self, model: ADAP, log_interval=None, tensorboard_log=None, tb_log_name="AdapAgent",
                 latent_syncer: Optional[AdapPolicy] = None):
        super().__init__(model, log_interval=log_interval, tensorboard_log=tensorboard_log, tb_log_name=tb_log_name)
        self.latent_syncer = latent_syncer

    def get_action(self, obs, record: bool = True) -> np.ndarray:
        if self.latent_syncer is not None:
            self.model.policy.set_context(self.latent_syncer.get_context())
        return super().get_action(obs, record)

    def _shape_obs(self, raw_obs, buf) -> np.ndarray:
        return np.asarray(raw_obs, np.float32).reshape(-1, self.model.policy.env_obs_len)

    def update(self, reward: float, done: bool) -> None:
        super().update(reward, done)
        if done and self.latent_syncer is None:
            self.model.policy.set_context(self.model.sample_context(1))
