"""ADAP on the engine: PantheonRL's ADAP learner (pantheonrl/algos/adap) behind the same surface.

ADAP (adap_learn.py:30-227) is PPO over a policy that also reads a latent context vector, plus one extra loss term:

* ``AdapPolicy`` (adap/policies.py:21-146) feeds ``features ++ context`` to the ordinary MlpExtractor, so on the engine it IS the
  MlpPolicy over a Box of ``observation + context_size`` components: the forward, GAE and PPO gradient kernels run unchanged.
* the rollout stores every observation with the context that was active (adap_learn.py:448-452, agent.py:117-121); the context is
  re-drawn when an episode ends (adap_learn.py:457-461, agent.py:147-151).
* ``train()`` adds ``context_loss_coeff * get_context_kl_loss`` to every minibatch loss (adap_learn.py:313-320, util.py:97-131):
  ``ph_adap_train`` -- one small launch per minibatch next to the PPO gradient launch, folded into the same gradient reduction,
  clip and Adam step (include/pantheon_hip.h).

Not built: ``AdapPolicyMult`` (policies.py:149-283, the multiplicative-latent extractor is a different network).
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


class ADAP(PPO):
    """pantheonrl.algos.adap.adap_learn.ADAP: same constructor surface (adap_learn.py:86-117) on top of `PPO`."""

    def __init__(self, policy=AdapPolicy, env=None, *args, context_loss_coeff: float = 0.1, context_size: int = 3,
                 num_context_samples: int = 5, context_sampler: str = "l2", num_state_samples: int = 32, **kwargs):
        if policy not in ("AdapPolicy", AdapPolicy):
            raise ValueError("the engine implements AdapPolicy (concatenated context); AdapPolicyMult is not built")
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

    _HP = PPO._HP + ("context_loss_coeff", "context_size", "num_context_samples", "context_sampler", "num_state_samples")

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
        self.policy = AdapPolicy(self.observation_space, self.action_space, context_size=self.context_size,
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
            self.logger.record("train/context_kl_loss", float(cl[:n_used].mean()))   # adap_learn.py:359

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
