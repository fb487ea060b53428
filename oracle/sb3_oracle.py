"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  ``pantheonrl_amd`` never does.

What it is
----------
A from-scratch CPU restatement (numpy + torch-CPU, float32) of the arithmetic
that PantheonRL's ``OnPolicyAgent`` delegates to ``stable-baselines3==1.7.0``
(reference ``setup.py:17``): the rollout buffer, the GAE pass, the MlpPolicy
forward / evaluate_actions and ``PPO.train()``, plus PantheonRL's own ADAP context
term (``pantheonrl/algos/adap/util.py:97-131``, whose module imports gym / SB3) and -- ahead of any device path --
``ModularPolicy`` / ``ModularAlgorithm.train`` (``pantheonrl/algos/modular``).

PARITY UNPINNED FOR THE SB3 CLASSES (RolloutBuffer incl. GAE, the plain
ActorCriticPolicy assembly, the distribution wrappers, collect_rollouts):
stable-baselines3 is not vendored under /root/reference and is not installed in
this image, and the reference has no tests or golden vectors (SURVEY.md section
4, 8c).  PINNED to the reference's own executable text, through fixtures that
tests/golden/make_reference_fixtures.py generates by RUNNING that text from
/root/reference (tests/test_reference_fixtures.py compares this module with them
to 1e-6): ``ppo_train`` / ``ppo_minibatch_loss`` / ``AdapTerm`` against ADAP.train
(adap_learn.py:229-371, the in-tree copy of SB3's PPO.train loop) over
AdapPolicy.evaluate_actions; ``adap_context_loss`` and the samplers against
adap/util.py; ``ModularPolicyOracle`` / ``modular_train`` against ModularPolicy /
ModularAlgorithm.train (incl. init gains, Adam eps, the mask offset);
``bc_loss`` / ``bc_train`` against pantheonrl/algos/bc.py.  The restatement follows

* the in-tree copies of the SB3 code that PantheonRL carries:
  ``pantheonrl/algos/adap/adap_learn.py:229-371`` (PPO.train),
  ``adap_learn.py:400-473`` (collect_rollouts / GAE call),
  ``pantheonrl/algos/modular/policies.py:84-88,112-114,214-241,273-290,364-383``
  (Adam eps, net_arch, ortho gains, forward / evaluate_actions order),
* SURVEY.md Appendix A (SB3 1.7.0 semantics restated from memory),
* the closed-form known-answer tests of SURVEY.md Appendix C,

and it is built from the *same torch primitives SB3 itself calls*
(``torch.distributions.Categorical``, ``nn.init.orthogonal_``,
``torch.optim.Adam``, ``clip_grad_norm_``, ``F.mse_loss``, autograd), so the
only unpinned part is the call sequence, not the primitives.

Everything here is executed "the way SB3 executes it": per-step ``add`` with
host copies, a Python loop over T for GAE, an eager-autograd minibatch loop.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch as th
from torch import nn
from torch.nn import functional as F

HIDDEN = 64  # SB3 MlpPolicy default net_arch=[dict(pi=[64,64], vf=[64,64])]  (modular/policies.py:112-114)


# --------------------------------------------------------------------------------------
# spaces (gym is absent): the minimal description the arithmetic needs
# --------------------------------------------------------------------------------------
@dataclass
class SpaceSpec:
    """kind in {"box","discrete","multidiscrete"}; nvec only for the discrete kinds."""
    kind: str
    dim: int = 0                      # Box: length
    nvec: Tuple[int, ...] = ()        # Discrete(n) -> (n,), MultiDiscrete(nvec) -> nvec

    @property
    def stored_len(self) -> int:      # SB3 obs_shape / action_dim
        return self.dim if self.kind == "box" else len(self.nvec)

    @property
    def flat_len(self) -> int:        # features after preprocess_obs / number of logits
        return self.dim if self.kind == "box" else int(sum(self.nvec))


def preprocess_obs(obs: th.Tensor, space: SpaceSpec) -> th.Tensor:
    """SB3 ``preprocess_obs``: Box -> float; Discrete -> one-hot; MultiDiscrete -> concat one-hots."""
    if space.kind == "box":
        return obs.float()
    obs = obs.long().reshape(obs.shape[0], len(space.nvec))
    parts = [F.one_hot(obs[:, i], num_classes=int(n)).float() for i, n in enumerate(space.nvec)]
    return th.cat(parts, dim=-1)


# --------------------------------------------------------------------------------------
# A.1 RolloutBuffer
# --------------------------------------------------------------------------------------
class RolloutBufferOracle:
    """SB3 1.7.0 ``RolloutBuffer``; call sites agents.py:123-130,157,172-179,196-198."""

    def __init__(self, n_steps: int, n_envs: int, obs_len: int, act_len: int,
                 gamma: float = 0.99, gae_lambda: float = 0.95):
        self.T, self.E, self.D, self.A = n_steps, n_envs, obs_len, act_len
        self.gamma, self.gae_lambda = gamma, gae_lambda
        self.reset()

    def reset(self) -> None:
        T, E = self.T, self.E
        self.observations = np.zeros((T, E, self.D), np.float32)
        self.actions = np.zeros((T, E, self.A), np.float32)
        for name in ("rewards", "returns", "episode_starts", "values", "log_probs", "advantages"):
            setattr(self, name, np.zeros((T, E), np.float32))
        self.pos, self.full, self.generator_ready = 0, False, False

    def add(self, obs, action, reward, episode_start, value, log_prob) -> None:
        E = self.E
        self.observations[self.pos] = np.array(obs, dtype=np.float32).reshape(E, self.D).copy()
        self.actions[self.pos] = np.array(action, dtype=np.float32).reshape(E, self.A).copy()
        self.rewards[self.pos] = np.array(reward, dtype=np.float32).copy()
        self.episode_starts[self.pos] = np.array(episode_start, dtype=np.float32).copy()
        self.values[self.pos] = th.as_tensor(value).clone().cpu().numpy().flatten()
        self.log_probs[self.pos] = th.as_tensor(log_prob).clone().cpu().numpy().reshape(-1)
        self.pos += 1
        if self.pos == self.T:
            self.full = True

    def compute_returns_and_advantage(self, last_values, dones) -> None:
        """SURVEY A.2 verbatim structure: Python loop over T, ~6 numpy vector ops per iteration."""
        last_values = th.as_tensor(last_values).clone().cpu().numpy().flatten().astype(np.float32)
        last_gae_lam = 0
        for step in reversed(range(self.T)):
            if step == self.T - 1:
                next_non_terminal = 1.0 - dones
                next_values = last_values
            else:
                next_non_terminal = 1.0 - self.episode_starts[step + 1]
                next_values = self.values[step + 1]
            delta = self.rewards[step] + self.gamma * next_values * next_non_terminal - self.values[step]
            last_gae_lam = delta + self.gamma * self.gae_lambda * next_non_terminal * last_gae_lam
            self.advantages[step] = last_gae_lam
        self.returns = self.advantages + self.values

    @staticmethod
    def swap_and_flatten(arr: np.ndarray) -> np.ndarray:
        shape = arr.shape
        if len(shape) < 3:
            shape = shape + (1,)
        return arr.swapaxes(0, 1).reshape(shape[0] * shape[1], *shape[2:])

    def flat(self):
        """env-major flattening (row e*T + t) of everything ``get`` yields."""
        f = self.swap_and_flatten
        return dict(observations=f(self.observations), actions=f(self.actions),
                    old_values=f(self.values).reshape(-1), old_log_prob=f(self.log_probs).reshape(-1),
                    advantages=f(self.advantages).reshape(-1), returns=f(self.returns).reshape(-1))

    def get(self, batch_size: Optional[int], indices: Optional[np.ndarray] = None):
        assert self.full, "rollout buffer must be full"
        n = self.T * self.E
        if indices is None:
            indices = np.random.permutation(n)
        data = self.flat()
        if batch_size is None:
            batch_size = n
        start = 0
        while start < n:
            sl = indices[start:start + batch_size]
            yield {k: th.as_tensor(v[sl]) for k, v in data.items()}
            start += batch_size


def gae_reference(rewards, values, episode_starts, last_values, dones, gamma=0.99, gae_lambda=0.95):
    """Functional form of A.2 on (T,E) float32 arrays.  Returns (advantages, returns)."""
    T, E = rewards.shape
    buf = RolloutBufferOracle(T, E, 1, 1, gamma, gae_lambda)
    buf.rewards[:] = rewards
    buf.values[:] = values
    buf.episode_starts[:] = episode_starts
    buf.compute_returns_and_advantage(np.asarray(last_values, np.float32), np.asarray(dones, np.float32))
    return buf.advantages.copy(), buf.returns.copy()


def gae_float64(rewards, values, episode_starts, last_values, dones, gamma=0.99, gae_lambda=0.95):
    """Same recurrence in float64 -- used only to state fp32 tolerances."""
    r, v, s = (np.asarray(a, np.float64) for a in (rewards, values, episode_starts))
    T, E = r.shape
    adv = np.zeros((T, E))
    last = np.zeros(E)
    for t in reversed(range(T)):
        if t == T - 1:
            nnt, nv = 1.0 - np.asarray(dones, np.float64), np.asarray(last_values, np.float64)
        else:
            nnt, nv = 1.0 - s[t + 1], v[t + 1]
        delta = r[t] + gamma * nv * nnt - v[t]
        last = delta + gamma * gae_lambda * nnt * last
        adv[t] = last
    return adv, adv + v


# --------------------------------------------------------------------------------------
# A.4 MlpPolicy
# --------------------------------------------------------------------------------------
class MlpPolicyOracle(nn.Module):
    """SB3 ``ActorCriticPolicy`` with FlattenExtractor and the default MlpExtractor."""

    def __init__(self, obs_space: SpaceSpec, act_space: SpaceSpec, lr: float = 3e-4, ortho_init: bool = True):
        super().__init__()
        assert act_space.kind in ("discrete", "multidiscrete") or isinstance(self, GaussianMlpPolicyOracle), \
            "categorical family here; Box action spaces: GaussianMlpPolicyOracle"
        self.obs_space, self.act_space = obs_space, act_space
        Fdim, L = obs_space.flat_len, act_space.flat_len
        self.policy_net = nn.Sequential(nn.Linear(Fdim, HIDDEN), nn.Tanh(), nn.Linear(HIDDEN, HIDDEN), nn.Tanh())
        self.value_net_mlp = nn.Sequential(nn.Linear(Fdim, HIDDEN), nn.Tanh(), nn.Linear(HIDDEN, HIDDEN), nn.Tanh())
        self.action_net = nn.Linear(HIDDEN, L)
        self.value_net = nn.Linear(HIDDEN, 1)
        if ortho_init:  # gains: modular/policies.py:229-241
            for mod, gain in ((self.policy_net, np.sqrt(2)), (self.value_net_mlp, np.sqrt(2)),
                              (self.action_net, 0.01), (self.value_net, 1.0)):
                for m in mod.modules():
                    if isinstance(m, nn.Linear):
                        nn.init.orthogonal_(m.weight, gain=gain)
                        m.bias.data.fill_(0.0)
        # Adam eps=1e-5: modular/policies.py:84-88
        self.optimizer = th.optim.Adam(self.parameters(), lr=lr, eps=1e-5)

    # -- distribution helpers -------------------------------------------------------
    def _split(self, logits: th.Tensor) -> List[th.Tensor]:
        return list(th.split(logits, list(self.act_space.nvec), dim=1))

    def _latents(self, obs: th.Tensor):
        feats = preprocess_obs(obs, self.obs_space)
        return self.policy_net(feats), self.value_net_mlp(feats)

    def logits(self, obs: th.Tensor, action_mask: Optional[th.Tensor] = None) -> th.Tensor:
        z = self.action_net(self._latents(obs)[0])
        if action_mask is not None:  # modular/policies.py:330-333: logits -= 30 * (~mask)
            z = z - 30.0 * (1.0 - action_mask.float())
        return z

    def forward(self, obs: th.Tensor, deterministic: bool = False, uniforms: Optional[th.Tensor] = None,
                action_mask: Optional[th.Tensor] = None):
        """-> (actions (n, A) int64, values (n,1), log_prob (n,)).  Order as modular/policies.py:273-290.

        ``uniforms`` (n, n_components) teacher-forces sampling by inverse CDF so a device RNG can be
        compared; without it ``Categorical.sample()`` is used like SB3.
        """
        latent_pi, latent_vf = self._latents(obs)
        values = self.value_net(latent_vf)
        z = self.action_net(latent_pi)
        if action_mask is not None:
            z = z - 30.0 * (1.0 - action_mask.float())
        acts, logp = [], 0.0
        for c, zc in enumerate(self._split(z)):
            dist = th.distributions.Categorical(logits=zc)
            if deterministic:
                a = th.argmax(dist.probs, dim=1)
            elif uniforms is not None:
                a = inverse_cdf_sample(dist.probs, uniforms[:, c])
            else:
                a = dist.sample()
            acts.append(a)
            logp = logp + dist.log_prob(a)
        return th.stack(acts, dim=1), values, logp

    def evaluate_actions(self, obs: th.Tensor, actions: th.Tensor, action_mask: Optional[th.Tensor] = None):
        """-> (values (n,1), log_prob (n,), entropy (n,)).  modular/policies.py:364-383."""
        latent_pi, latent_vf = self._latents(obs)
        z = self.action_net(latent_pi)
        if action_mask is not None:
            z = z - 30.0 * (1.0 - action_mask.float())
        actions = actions.long().reshape(obs.shape[0], -1)
        logp, ent = 0.0, 0.0
        for c, zc in enumerate(self._split(z)):
            dist = th.distributions.Categorical(logits=zc)
            logp = logp + dist.log_prob(actions[:, c])
            ent = ent + dist.entropy()
        return self.value_net(latent_vf), logp, ent

    def predict_values(self, obs: th.Tensor) -> th.Tensor:
        return self.value_net(self._latents(obs)[1])

    # -- flat parameter vector in the layout include/pantheon_hip.h documents ------------
    def flat_params(self) -> np.ndarray:
        """[pi_W1(F,H) pi_b1 pi_W2(H,H) pi_b2 vf_W1 vf_b1 vf_W2 vf_b2 act_W(H,L) act_b val_W(H) val_b];
        weights stored input-major (the transpose of torch's [out][in])."""
        out = []
        for seq in (self.policy_net, self.value_net_mlp):
            for idx in (0, 2):
                out += [seq[idx].weight.detach().t().contiguous().reshape(-1), seq[idx].bias.detach()]
        out += [self.action_net.weight.detach().t().contiguous().reshape(-1), self.action_net.bias.detach(),
                self.value_net.weight.detach().reshape(-1), self.value_net.bias.detach()]
        return th.cat(out).numpy().astype(np.float32).copy()

    def load_flat_params(self, flat: np.ndarray) -> None:
        flat = th.as_tensor(np.asarray(flat, np.float32))
        o = 0

        def take(n):
            nonlocal o
            v = flat[o:o + n]
            o += n
            return v
        with th.no_grad():
            for seq in (self.policy_net, self.value_net_mlp):
                for idx in (0, 2):
                    lin = seq[idx]
                    lin.weight.copy_(take(lin.weight.numel()).reshape(lin.in_features, lin.out_features).t())
                    lin.bias.copy_(take(lin.bias.numel()))
            lin = self.action_net
            lin.weight.copy_(take(lin.weight.numel()).reshape(lin.in_features, lin.out_features).t())
            lin.bias.copy_(take(lin.bias.numel()))
            self.value_net.weight.copy_(take(HIDDEN).reshape(1, HIDDEN))
            self.value_net.bias.copy_(take(1))
        assert o == flat.numel()

    def flat_grads(self) -> np.ndarray:
        out = []
        for seq in (self.policy_net, self.value_net_mlp):
            for idx in (0, 2):
                out += [seq[idx].weight.grad.t().contiguous().reshape(-1), seq[idx].bias.grad]
        out += [self.action_net.weight.grad.t().contiguous().reshape(-1), self.action_net.bias.grad,
                self.value_net.weight.grad.reshape(-1), self.value_net.bias.grad]
        return th.cat(out).numpy().astype(np.float32).copy()


class GaussianMlpPolicyOracle(MlpPolicyOracle):
    """SB3 1.7.0 ``ActorCriticPolicy`` over a Box action space **[SB3-mem]**: ``DiagGaussianDistribution`` --
    ``proba_distribution_net`` = (``Linear(latent_pi, A)`` for the means, ``log_std = nn.Parameter(ones(A) * log_std_init)``,
    ``log_std_init = 0``); ``action_net`` keeps the 0.01 orthogonal gain.  ``log_prob`` / ``entropy`` are ``Normal``'s summed over
    the action dimensions (``sum_independent_dims``); actions are NOT squashed or clipped inside the policy -- the caller clips what
    the environment gets (reference ``pantheonrl/common/util.py:84-99``), the rollout buffer keeps the raw sample.
    Flat parameter vector: MlpPolicyOracle's, then ``log_std`` (include/pantheon_hip.h)."""

    def __init__(self, obs_space: SpaceSpec, act_space: SpaceSpec, lr: float = 3e-4, ortho_init: bool = True):
        assert act_space.kind == "box"
        super().__init__(obs_space, act_space, lr=lr, ortho_init=ortho_init)
        self.log_std = nn.Parameter(th.zeros(act_space.dim))
        self.optimizer = th.optim.Adam(self.parameters(), lr=lr, eps=1e-5)

    def _dist(self, latent_pi: th.Tensor):
        mean = self.action_net(latent_pi)
        return th.distributions.Normal(mean, th.ones_like(mean) * self.log_std.exp())

    def forward(self, obs: th.Tensor, deterministic: bool = False, uniforms: Optional[th.Tensor] = None,
                action_mask: Optional[th.Tensor] = None):
        """-> (actions (n, A) float32, values (n,1), log_prob (n,)).  ``uniforms`` (n, A) teacher-forces the STANDARD-NORMAL
        draws (actions = mean + std * eps); without it ``Normal.rsample()`` like SB3."""
        assert action_mask is None
        latent_pi, latent_vf = self._latents(obs)
        values = self.value_net(latent_vf)
        dist = self._dist(latent_pi)
        if deterministic:
            a = dist.mean
        elif uniforms is not None:
            a = dist.mean + dist.stddev * uniforms.float()
        else:
            a = dist.rsample()
        return a, values, dist.log_prob(a).sum(dim=1)

    def evaluate_actions(self, obs: th.Tensor, actions: th.Tensor, action_mask: Optional[th.Tensor] = None):
        assert action_mask is None
        latent_pi, latent_vf = self._latents(obs)
        dist = self._dist(latent_pi)
        actions = actions.float().reshape(obs.shape[0], -1)
        return self.value_net(latent_vf), dist.log_prob(actions).sum(dim=1), dist.entropy().sum(dim=1)

    def flat_params(self) -> np.ndarray:
        return np.concatenate([super().flat_params(), self.log_std.detach().numpy().astype(np.float32)])

    def load_flat_params(self, flat: np.ndarray) -> None:
        flat = np.asarray(flat, np.float32)
        A = self.act_space.dim
        super().load_flat_params(flat[:-A])
        with th.no_grad():
            self.log_std.copy_(th.as_tensor(flat[-A:]))

    def flat_grads(self) -> np.ndarray:
        return np.concatenate([super().flat_grads(), self.log_std.grad.numpy().astype(np.float32)])


class AdapMultPolicyOracle(MlpPolicyOracle):
    """``AdapPolicyMult`` = ``AdapPolicy`` with ``MultModel`` as its extractor (adap/policies.py:136-283), SB3's default
    ``net_arch=[dict(pi=[64, 64], vf=[64, 64])]``, tanh: the yardstick of the device path (csrc/ph_adapmult.hip).

    The stored observation is ``features ++ context`` (adap_learn.py:448-452) as for ``AdapPolicy``; ``MultModel.forward``
    splits it again (policies.py:268-272) and, per net (policies.py:239-264):
        x      = tanh(W1 o + b1)                      branch_1 on the observation WITHOUT the context
        x_a    = tanh(Ws x + bs)    (64 -> 64 * C)     scaling, viewed as (64, C): element [j][c] is output row j * C + c
        latent = tanh(W2 (x + x_a @ ctx) + b2)        branch_2
    Orthogonal init with gain sqrt(2) reaches every Linear of the extractor, the scaling layers included (SB3 applies
    ``init_weights`` to ``mlp_extractor`` as a whole: modular/policies.py:229-241 is the same loop)."""

    def __init__(self, obs_space: SpaceSpec, act_space: SpaceSpec, context_size: int = 3, lr: float = 3e-4,
                 ortho_init: bool = True):
        nn.Module.__init__(self)
        assert obs_space.kind == "box", "features ++ context rows are Box rows (adap.py turns Discrete features into one-hots)"
        assert act_space.kind in ("discrete", "multidiscrete")
        self.obs_space, self.act_space, self.context_size = obs_space, act_space, int(context_size)
        Fdim, L, C = obs_space.flat_len - self.context_size, act_space.flat_len, self.context_size
        assert Fdim > 0

        def net():
            return (nn.Sequential(nn.Linear(Fdim, HIDDEN), nn.Tanh()), nn.Sequential(nn.Linear(HIDDEN, HIDDEN * C), nn.Tanh()),
                    nn.Sequential(nn.Linear(HIDDEN, HIDDEN), nn.Tanh()))
        self.agent_branch_1, self.agent_scaling, self.agent_branch_2 = net()
        self.value_branch_1, self.value_scaling, self.value_branch_2 = net()
        self.action_net = nn.Linear(HIDDEN, L)
        self.value_net = nn.Linear(HIDDEN, 1)
        if ortho_init:
            extractor = (self.agent_branch_1, self.agent_scaling, self.agent_branch_2, self.value_branch_1, self.value_scaling,
                         self.value_branch_2)
            for mod, gain in [(m, np.sqrt(2)) for m in extractor] + [(self.action_net, 0.01), (self.value_net, 1.0)]:
                for m in mod.modules():
                    if isinstance(m, nn.Linear):
                        nn.init.orthogonal_(m.weight, gain=gain)
                        m.bias.data.fill_(0.0)
        self.optimizer = th.optim.Adam(self.parameters(), lr=lr, eps=1e-5)

    def _branch(self, b1, scaling, b2, o: th.Tensor, ctx: th.Tensor) -> th.Tensor:
        x = b1(o)
        x_a = scaling(x).view(o.shape[0], HIDDEN, self.context_size)          # policies.py:245 / 258
        return b2(x + th.matmul(x_a, ctx.unsqueeze(-1)).squeeze(-1))          # policies.py:246-247 / 259-260

    def _latents(self, obs: th.Tensor):
        feats = preprocess_obs(obs, self.obs_space)
        o, ctx = feats[:, :-self.context_size], feats[:, -self.context_size:]  # policies.py:270-271
        return (self._branch(self.agent_branch_1, self.agent_scaling, self.agent_branch_2, o, ctx),
                self._branch(self.value_branch_1, self.value_scaling, self.value_branch_2, o, ctx))

    # -- flat vector in ph_adapmult_layout's order (include/pantheon_hip.h): per net W1 b1 Ws bs W2 b2 (pi, then vf), then the heads;
    #    weights input-major, i.e. the scaling layer as [64][64 C] with column j C + c = torch's output row j C + c
    def _linears(self):
        return [self.agent_branch_1[0], self.agent_scaling[0], self.agent_branch_2[0], self.value_branch_1[0],
                self.value_scaling[0], self.value_branch_2[0], self.action_net]

    def flat_params(self) -> np.ndarray:
        out = []
        for lin in self._linears():
            out += [lin.weight.detach().t().contiguous().reshape(-1), lin.bias.detach()]
        out += [self.value_net.weight.detach().reshape(-1), self.value_net.bias.detach()]
        return th.cat(out).numpy().astype(np.float32).copy()

    def load_flat_params(self, flat: np.ndarray) -> None:
        flat = th.as_tensor(np.asarray(flat, np.float32))
        o = 0
        with th.no_grad():
            for lin in self._linears():
                n = lin.weight.numel()
                lin.weight.copy_(flat[o:o + n].reshape(lin.in_features, lin.out_features).t())
                o += n
                lin.bias.copy_(flat[o:o + lin.bias.numel()])
                o += lin.bias.numel()
            self.value_net.weight.copy_(flat[o:o + HIDDEN].reshape(1, HIDDEN))
            self.value_net.bias.copy_(flat[o + HIDDEN:o + HIDDEN + 1])
            o += HIDDEN + 1
        assert o == flat.numel()

    def flat_grads(self) -> np.ndarray:
        zero = lambda t: th.zeros_like(t) if t.grad is None else t.grad   # noqa: E731
        out = []
        for lin in self._linears():
            out += [zero(lin.weight).t().contiguous().reshape(-1), zero(lin.bias)]
        out += [zero(self.value_net.weight).reshape(-1), zero(self.value_net.bias)]
        return th.cat(out).numpy().astype(np.float32).copy()


def inverse_cdf_sample(probs: th.Tensor, u: th.Tensor) -> th.Tensor:
    """action = number of prefix sums (float32, left to right) that are <= u, clamped to n-1."""
    n = probs.shape[1]
    acc = th.zeros(probs.shape[0], dtype=th.float32)
    a = th.zeros(probs.shape[0], dtype=th.long)
    for k in range(n - 1):
        acc = acc + probs[:, k].float()
        a = a + (u.float() >= acc).long()
    return a


def fix_illegal_actions(actions: np.ndarray, masks: np.ndarray) -> np.ndarray:
    """env-side fix-up, pettingzoo.py:81-82: illegal action -> first legal index.  Integer, bit-exact."""
    actions = np.asarray(actions).astype(np.int64).copy()
    masks = np.asarray(masks)
    for e in range(actions.shape[0]):
        if not masks[e][actions[e]]:
            actions[e] = masks[e].tolist().index(1)
    return actions


# --------------------------------------------------------------------------------------
# A.3 PPO.train()
# --------------------------------------------------------------------------------------
@dataclass
class PPOHyper:
    """defaults: adap_learn.py:90-103 (mirror of SB3's)."""
    learning_rate: float = 3e-4
    n_epochs: int = 10
    batch_size: int = 64
    clip_range: float = 0.2
    clip_range_vf: Optional[float] = None
    normalize_advantage: bool = True
    ent_coef: float = 0.0
    vf_coef: float = 0.5
    max_grad_norm: float = 0.5
    target_kl: Optional[float] = None


def ppo_minibatch_loss(policy: MlpPolicyOracle, mb: dict, hp: PPOHyper):
    """One minibatch of adap_learn.py:253-327 (without the ADAP context term).  Returns (loss, stats)."""
    actions = mb["actions"]
    if policy.act_space.kind == "discrete":
        actions = actions.long().flatten()
    values, log_prob, entropy = policy.evaluate_actions(mb["observations"], actions)
    values = values.flatten()
    advantages = mb["advantages"]
    if hp.normalize_advantage and len(advantages) > 1:
        advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    ratio = th.exp(log_prob - mb["old_log_prob"])
    policy_loss_1 = advantages * ratio
    policy_loss_2 = advantages * th.clamp(ratio, 1 - hp.clip_range, 1 + hp.clip_range)
    policy_loss = -th.min(policy_loss_1, policy_loss_2).mean()
    clip_fraction = th.mean((th.abs(ratio - 1) > hp.clip_range).float()).item()
    if hp.clip_range_vf is None:
        values_pred = values
    else:
        values_pred = mb["old_values"] + th.clamp(values - mb["old_values"], -hp.clip_range_vf, hp.clip_range_vf)
    value_loss = F.mse_loss(mb["returns"], values_pred)
    entropy_loss = -th.mean(entropy)
    loss = policy_loss + hp.ent_coef * entropy_loss + hp.vf_coef * value_loss
    with th.no_grad():
        log_ratio = log_prob - mb["old_log_prob"]
        approx_kl = th.mean((th.exp(log_ratio) - 1) - log_ratio).item()
    stats = dict(policy_loss=policy_loss.item(), value_loss=value_loss.item(), entropy_loss=entropy_loss.item(),
                 clip_fraction=clip_fraction, approx_kl=approx_kl, loss=loss.item())
    return loss, stats


# --------------------------------------------------------------------------------------
# ADAP's context term (pantheonrl/algos/adap): PPO loss + context_loss_coeff * context loss
# --------------------------------------------------------------------------------------
def adap_sample_contexts(sampler: str, ctx_size: int, num: int, uniforms: np.ndarray) -> np.ndarray:
    """The SAMPLERS of adap/util.py:42-77 with th.rand teacher-forced by ``uniforms`` (num, ctx_size) in [0,1):
    "l2": 2u-1 scaled to unit length (util.py:42-51), "unit_square": 2u-1 (54-59), "positive_square": u (62-67),
    "categorical": one-hot at floor(u[:,0] * ctx_size) (70-77: th.randint teacher-forced by the first uniform)."""
    u = th.as_tensor(np.asarray(uniforms, np.float32)).reshape(num, ctx_size)
    if sampler == "l2":
        c = u * 2 - 1
        c = c / (th.sum(c ** 2, dim=-1).reshape(num, 1)) ** (1 / 2)
    elif sampler == "unit_square":
        c = u * 2 - 1
    elif sampler == "positive_square":
        c = u
    elif sampler == "categorical":
        c = th.zeros(num, ctx_size)
        c[th.arange(num), th.clamp((u[:, 0] * ctx_size).long(), max=ctx_size - 1)] = 1
    elif sampler == "natural_numbers":   # util.py:80-89: (num, 1) integers in [0, ctx_size), th.randint teacher-forced by u[:, 0]
        c = th.clamp((u[:, :1] * ctx_size).long(), max=ctx_size - 1).float()
    else:
        raise ValueError(sampler)
    return c.numpy()


@dataclass
class AdapTerm:
    """ADAP.__init__ defaults adap_learn.py:111-116; per-minibatch samples teacher-force th.randperm and the sampler."""
    context_size: int = 3
    context_loss_coeff: float = 0.1
    state_idx: Optional[Sequence[np.ndarray]] = None    # [minibatch] -> (<= num_state_samples,) positions in the minibatch
    contexts: Optional[Sequence[np.ndarray]] = None     # [minibatch] -> (num_context_samples, context_size)


def adap_context_loss(policy: MlpPolicyOracle, observations: th.Tensor, context_size: int, state_idx, contexts) -> th.Tensor:
    """``get_context_kl_loss`` (adap/util.py:97-131): the observations carry the rollout's context in their last
    ``context_size`` components (adap_learn.py:448-452); the sampled states are re-evaluated under every sampled context
    (AdapPolicy._get_latent: features ++ context, adap/policies.py:104-119) and the loss is the mean over context pairs
    (a, b), a before b, of mean_s exp(-KL(pi(.|s,a) || pi(.|s,b)))."""
    from itertools import combinations
    original = observations[:, :-context_size]
    states = original[th.as_tensor(np.asarray(state_idx, np.int64))]
    n = states.shape[0]
    dists = []
    for c in np.asarray(contexts, np.float32):
        feats = th.cat((states, th.as_tensor(c).reshape(1, -1).repeat(n, 1)), dim=1)
        dists.append([th.distributions.Categorical(logits=z) for z in policy._split(policy.logits(feats))])
    cls = []
    for a, b in combinations(dists, 2):
        # util.py:16-39: MultiCategorical -> sum of the per-component KLs; Categorical -> torch's kl_divergence
        kl = sum(th.distributions.kl.kl_divergence(p, q) for p, q in zip(a, b))
        cls.append(th.mean(th.exp(-kl)))
    return sum(cls) / len(cls)


def ppo_train(policy: MlpPolicyOracle, buf: RolloutBufferOracle, hp: PPOHyper,
              perms: Optional[Sequence[np.ndarray]] = None, adap: Optional[AdapTerm] = None) -> List[dict]:
    """SB3 ``PPO.train()``.  ``perms[epoch]`` teacher-forces ``np.random.permutation``.  With ``adap`` the
    loss of every minibatch gains ``context_loss_coeff * context_loss`` (ADAP.train, adap_learn.py:313-320)."""
    for g in policy.optimizer.param_groups:
        g["lr"] = hp.learning_rate
    all_stats: List[dict] = []
    continue_training = True
    mbi = -1
    for epoch in range(hp.n_epochs):
        idx = None if perms is None else np.asarray(perms[epoch])
        for mb in buf.get(hp.batch_size, idx):
            mbi += 1
            loss, stats = ppo_minibatch_loss(policy, mb, hp)
            if adap is not None:
                cl = adap_context_loss(policy, mb["observations"], adap.context_size, adap.state_idx[mbi], adap.contexts[mbi])
                loss = loss + adap.context_loss_coeff * cl
                stats["context_loss"] = cl.item()
                stats["loss"] = loss.item()
            if hp.target_kl is not None and stats["approx_kl"] > 1.5 * hp.target_kl:
                continue_training = False
                stats["stopped"] = True
                all_stats.append(stats)
                break
            policy.optimizer.zero_grad()
            loss.backward()
            gn = th.nn.utils.clip_grad_norm_(policy.parameters(), hp.max_grad_norm)
            stats["grad_norm"] = float(gn)
            policy.optimizer.step()
            all_stats.append(stats)
        if not continue_training:
            break
    return all_stats


# --------------------------------------------------------------------------------------
# SB3-style rollout+update iteration on synthetic inputs: the cpu_baseline leg of bench.py
# --------------------------------------------------------------------------------------
def synthetic_iteration(policy: MlpPolicyOracle, buf: RolloutBufferOracle, hp: PPOHyper,
                        obs_seq: np.ndarray, rew_seq: np.ndarray, done_seq: np.ndarray) -> None:
    """One whole PPO iteration executed the way OnPolicyAgent + SB3 execute it (agents.py:111-203):
    per step: no_grad forward, buffer.add with host copies, reward +=; then GAE loop; then train()."""
    T, E = buf.T, buf.E
    buf.reset()
    last_starts = np.ones(E, np.float32)
    values = None
    for t in range(T):
        with th.no_grad():
            actions, values, logp = policy.forward(th.as_tensor(obs_seq[t]))
        buf.add(obs_seq[t], actions.numpy(), np.zeros(E, np.float32), last_starts, values, logp)
        buf.rewards[buf.pos - 1] += rew_seq[t]          # Agent.update: agents.py:198
        last_starts = done_seq[t].astype(np.float32)    # agents.py:197
    buf.compute_returns_and_advantage(values, last_starts)  # quirk D-1: V(o_{T-1})
    ppo_train(policy, buf, hp)


# --------------------------------------------------------------------------------------
# the ego's rollout loop: OnPolicyAlgorithm.collect_rollouts of SB3 1.7.0
# --------------------------------------------------------------------------------------
def collect_rollouts(policy: MlpPolicyOracle, buf: RolloutBufferOracle, env, last_obs: np.ndarray,
                     last_episode_starts: np.ndarray, gamma: float = 0.99, forced_actions=None):
    """SB3 1.7.0 ``collect_rollouts`` for the ego (reference call site trainer.py:413 -> ``ego.learn``; in-tree witness of
    the loop's shape: adap_learn.py:400-473, an older SB3 copy).  ``env`` is a VecEnv-like object:
    ``step(actions) -> (obs, rewards, dones, infos)`` with auto-reset and ``infos[i]["terminal_observation"]``.

    The part the in-tree copy predates [SB3-mem, v1.5+]: when an episode ends by TIME LIMIT
    (``infos[i].get("TimeLimit.truncated", False)``) the transition's reward is bootstrapped with the value of the terminal
    observation, ``rewards[i] += gamma * V(terminal_observation)``, before the row is added.
    ``forced_actions[t]`` (E, A) teacher-forces the sampled actions.  Returns (new_obs, dones)."""
    T, E = buf.T, buf.E
    buf.reset()
    new_obs, dones = last_obs, np.zeros(E, bool)
    for t in range(T):
        with th.no_grad():
            actions, values, logp = policy.forward(th.as_tensor(np.asarray(last_obs, np.float32)))
            if forced_actions is not None:
                actions = th.as_tensor(np.asarray(forced_actions[t])).reshape(actions.shape)
                values, logp, _ = policy.evaluate_actions(th.as_tensor(np.asarray(last_obs, np.float32)), actions)
        act_np = actions.numpy()
        new_obs, rewards, dones, infos = env.step(act_np)
        rewards = np.asarray(rewards, np.float32).copy()
        for i, done in enumerate(dones):
            if done and infos[i].get("terminal_observation") is not None and infos[i].get("TimeLimit.truncated", False):
                with th.no_grad():
                    term = th.as_tensor(np.asarray(infos[i]["terminal_observation"], np.float32)).reshape(1, -1)
                    terminal_value = policy.predict_values(term)[0]
                rewards[i] += (gamma * terminal_value).numpy().reshape(())
        buf.add(np.asarray(last_obs, np.float32), act_np.reshape(E, -1), rewards, last_episode_starts, values, logp)
        last_obs, last_episode_starts = new_obs, np.asarray(dones, np.float32)
    with th.no_grad():
        values = policy.predict_values(th.as_tensor(np.asarray(new_obs, np.float32)))
    buf.compute_returns_and_advantage(values, np.asarray(dones, np.float32))
    return new_obs, dones


# --------------------------------------------------------------------------------------
# ModularAlgorithm / ModularPolicy (pantheonrl/algos/modular): what ph_modular_forward / ph_modular_train (ph_modular.hip)
# are tested against (tests/test_gpu_modular.py).
# --------------------------------------------------------------------------------------
class ModularPolicyOracle(MlpPolicyOracle):
    """``ModularPolicy`` (modular/policies.py:40-395) with the FlattenExtractor defaults: the main network is the ordinary
    MlpPolicy; every partner owns a second MlpExtractor whose INPUT is the main policy latent (``:254``: input_dim =
    latent_dim_pi; forward ``:281``), with pi / vf towers 64-64 (``:101-105``), an action head and a value head
    (``:214-219``).  Logits and values are sums (``:325-328``: main_logits + partner_logits, or partner_logits alone with
    ``nomain``; ``:286``: value_net(latent_vf) + partner_value_net(partner_latent_vf)).  Orthogonal gains as for the main
    network (``:229-241``).  ``baseline`` shares one partner module between all partners (``:255-257``)."""

    def __init__(self, obs_space: SpaceSpec, act_space: SpaceSpec, num_partners: int = 1, lr: float = 3e-4,
                 ortho_init: bool = True, nomain: bool = False, baseline: bool = False):
        super().__init__(obs_space, act_space, lr=lr, ortho_init=ortho_init)
        self.num_partners, self.nomain = int(num_partners), bool(nomain)
        L = act_space.flat_len

        def module():
            m = nn.ModuleDict(dict(
                pi=nn.Sequential(nn.Linear(HIDDEN, HIDDEN), nn.Tanh(), nn.Linear(HIDDEN, HIDDEN), nn.Tanh()),
                vf=nn.Sequential(nn.Linear(HIDDEN, HIDDEN), nn.Tanh(), nn.Linear(HIDDEN, HIDDEN), nn.Tanh()),
                act=nn.Linear(HIDDEN, L), val=nn.Linear(HIDDEN, 1)))
            if ortho_init:
                for mod, gain in ((m["pi"], np.sqrt(2)), (m["vf"], np.sqrt(2)), (m["act"], 0.01), (m["val"], 1.0)):
                    for lin in mod.modules():
                        if isinstance(lin, nn.Linear):
                            nn.init.orthogonal_(lin.weight, gain=gain)
                            lin.bias.data.fill_(0.0)
            return m
        mods = [module() for _ in range(self.num_partners)]
        if baseline:
            mods = [mods[0]] * self.num_partners
        self.partners = nn.ModuleList(mods)
        self.optimizer = th.optim.Adam(self.parameters(), lr=lr, eps=1e-5)   # policies.py:263 over ALL parameters

    def _towers(self, obs: th.Tensor, partner_idx: int):
        latent_pi, latent_vf = self._latents(obs)
        pm = self.partners[partner_idx]
        return latent_pi, latent_vf, pm["pi"](latent_pi), pm["vf"](latent_pi), pm   # BOTH partner towers read latent_pi

    def action_logits(self, obs: th.Tensor, partner_idx: int):
        """``get_action_logits_from_obs`` (policies.py:385-395) without a mask -> (main_logits, partner_logits)."""
        latent_pi, _, p_pi, _, pm = self._towers(obs, partner_idx)
        return self.action_net(latent_pi), pm["act"](p_pi)

    def _mean_actions(self, latent_pi, p_pi, pm, action_mask):
        z = pm["act"](p_pi) if self.nomain else self.action_net(latent_pi) + pm["act"](p_pi)
        if action_mask is not None:     # policies.py:330-333; the clamp at :334 discards its result
            z = z - 30.0 * (1.0 - action_mask.float())
        return z

    def forward(self, obs: th.Tensor, partner_idx: int = 0, deterministic: bool = False,
                uniforms: Optional[th.Tensor] = None, action_mask: Optional[th.Tensor] = None):
        """policies.py:271-288 -> (actions (n, A), values (n, 1), log_prob (n,))"""
        latent_pi, latent_vf, p_pi, p_vf, pm = self._towers(obs, partner_idx)
        z = self._mean_actions(latent_pi, p_pi, pm, action_mask)
        acts, logp = [], 0.0
        for c, zc in enumerate(self._split(z)):
            dist = th.distributions.Categorical(logits=zc)
            if deterministic:
                a = th.argmax(dist.probs, dim=1)
            elif uniforms is not None:
                a = inverse_cdf_sample(dist.probs, uniforms[:, c])
            else:
                a = dist.sample()
            acts.append(a)
            logp = logp + dist.log_prob(a)
        return th.stack(acts, dim=1), self.value_net(latent_vf) + pm["val"](p_vf), logp

    def evaluate_actions(self, obs: th.Tensor, actions: th.Tensor, partner_idx: int = 0,
                         action_mask: Optional[th.Tensor] = None):
        """policies.py:364-383 -> (values (n, 1), log_prob (n,), entropy (n,))"""
        latent_pi, latent_vf, p_pi, p_vf, pm = self._towers(obs, partner_idx)
        z = self._mean_actions(latent_pi, p_pi, pm, action_mask)
        actions = actions.long().reshape(obs.shape[0], -1)
        logp, ent = 0.0, 0.0
        for c, zc in enumerate(self._split(z)):
            dist = th.distributions.Categorical(logits=zc)
            logp = logp + dist.log_prob(actions[:, c])
            ent = ent + dist.entropy()
        return self.value_net(latent_vf) + pm["val"](p_vf), logp, ent

    def flat_params(self) -> np.ndarray:
        """the main network in MlpPolicyOracle's layout, then per partner [pi_W1 pi_b1 pi_W2 pi_b2 vf_W1 vf_b1 vf_W2 vf_b2
        act_W(H,L) act_b val_W(H) val_b], weights input-major"""
        out = [th.as_tensor(super().flat_params())]
        for pm in self.partners:
            for seq in (pm["pi"], pm["vf"]):
                for idx in (0, 2):
                    out += [seq[idx].weight.detach().t().contiguous().reshape(-1), seq[idx].bias.detach()]
            out += [pm["act"].weight.detach().t().contiguous().reshape(-1), pm["act"].bias.detach(),
                    pm["val"].weight.detach().reshape(-1), pm["val"].bias.detach()]
        return th.cat(out).numpy().astype(np.float32).copy()


def modular_marginal_regularization(policy: ModularPolicyOracle, observations: th.Tensor) -> th.Tensor:
    """modular/learn.py:298-318 for a single Categorical head: every partner's (main_logits, partner_logits) on the minibatch;
    main_probs = mean over partners of softmax(main_logits) (the same tensor num_partners times), composed_probs = mean over
    partners of softmax(main_logits + partner_logits); loss = mean over rows of sum_a |main_probs - composed_probs|."""
    def probs(z):                      # exp(z - logsumexp z), the reference's way of writing the softmax (:313-314)
        return (z - th.logsumexp(z, dim=-1, keepdim=True)).exp()
    z_main, z_sum = [], []
    for k in range(policy.num_partners):
        zm, zp = policy.action_logits(observations, k)
        z_main.append(zm)
        z_sum.append(zm + zp)
    p_alone = probs(th.stack(z_main)).mean(dim=0)          # (rows, actions)
    p_with_partner = probs(th.stack(z_sum)).mean(dim=0)
    return (p_alone - p_with_partner).abs().sum(dim=1).mean()


def modular_minibatch_loss(policy: ModularPolicyOracle, mb: dict, hp: PPOHyper, partner_idx: int, marginal_reg_coef: float):
    """One minibatch of ModularAlgorithm.train (modular/learn.py:244-318): the PPO terms with the partner's module in the
    network (advantages are ALWAYS normalised: no normalize_advantage switch, no len > 1 guard, ``:260-261``), plus
    ``marginal_reg_coef * marginal_regularization_loss`` (``:318``).  approx_kl is the plain mean(old_log_prob - log_prob)
    of ``:327``, not SB3 1.7's estimator."""
    actions = mb["actions"]
    if policy.act_space.kind == "discrete":
        actions = actions.long().flatten()
    values, log_prob, entropy = policy.evaluate_actions(mb["observations"], actions, partner_idx=partner_idx)
    values = values.flatten()
    advantages = mb["advantages"]
    advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    ratio = th.exp(log_prob - mb["old_log_prob"])
    policy_loss = -th.min(advantages * ratio, advantages * th.clamp(ratio, 1 - hp.clip_range, 1 + hp.clip_range)).mean()
    if hp.clip_range_vf is None:
        values_pred = values
    else:
        values_pred = mb["old_values"] + th.clamp(values - mb["old_values"], -hp.clip_range_vf, hp.clip_range_vf)
    value_loss = F.mse_loss(mb["returns"], values_pred)
    entropy_loss = -th.mean(entropy)
    reg = modular_marginal_regularization(policy, mb["observations"])
    loss = policy_loss + hp.ent_coef * entropy_loss + hp.vf_coef * value_loss + marginal_reg_coef * reg
    stats = dict(policy_loss=policy_loss.item(), value_loss=value_loss.item(), entropy_loss=entropy_loss.item(),
                 marginal_reg=reg.item(), loss=loss.item(),
                 approx_kl=th.mean(mb["old_log_prob"] - log_prob).item())
    return loss, stats


def modular_train(policy: ModularPolicyOracle, bufs: Sequence[RolloutBufferOracle], hp: PPOHyper, marginal_reg_coef: float = 0.0,
                  perms: Optional[Sequence[Sequence[np.ndarray]]] = None) -> List[dict]:
    """``ModularAlgorithm.train`` (modular/learn.py:221-351): partner by partner (one rollout buffer each, ``:134-144``),
    ``n_epochs`` passes over that partner's buffer; the optimiser step comes BEFORE the KL bookkeeping and the target-KL test
    ends the partner's epoch loop only after a whole epoch (``:320-334``: mean of the epoch's KLs).  ``perms[partner][epoch]``
    teacher-forces the buffer's permutation."""
    for g in policy.optimizer.param_groups:
        g["lr"] = hp.learning_rate
    all_stats: List[dict] = []
    for k, buf in enumerate(bufs):
        for epoch in range(hp.n_epochs):
            kls = []
            idx = None if perms is None else np.asarray(perms[k][epoch])
            for mb in buf.get(hp.batch_size, idx):
                loss, stats = modular_minibatch_loss(policy, mb, hp, k, marginal_reg_coef)
                # The reference pins torch==1.13.1 (setup.py:15), whose Optimizer.zero_grad defaults to set_to_none=False:
                # gradients are zeroed IN PLACE, so a parameter that has received a gradient once keeps taking part in every
                # later step (g = 0: its moments decay, its own step count advances), while one the loss has never reached
                # (another partner's value tower before that partner's first turn) has grad None and is skipped.  torch >= 2.0
                # flipped the default; the pinned behaviour is requested explicitly.
                policy.optimizer.zero_grad(set_to_none=False)
                loss.backward()
                stats["grad_norm"] = float(th.nn.utils.clip_grad_norm_(policy.parameters(), hp.max_grad_norm))
                policy.optimizer.step()
                stats["partner"] = k
                kls.append(stats["approx_kl"])
                all_stats.append(stats)
            if hp.target_kl is not None and np.mean(kls) > 1.5 * hp.target_kl:
                break
    return all_stats


# --------------------------------------------------------------------------------------
# Behavioural cloning: FeedForward32Policy + BC._calculate_loss + BC.train  (reference pantheonrl/algos/bc.py)
# --------------------------------------------------------------------------------------
BC_HIDDEN = 32


class FeedForward32Oracle(nn.Module):
    """``FeedForward32Policy`` (pantheonrl/common/util.py:114-123): SB3 ActorCriticPolicy with ``net_arch=[32, 32]`` -- in SB3
    1.7.0 a plain list of ints is a SHARED trunk, so policy and value heads sit on the same 32-32 tanh network.  Orthogonal
    init with SB3's gains (sqrt(2) trunk, 0.01 action_net, 1 value_net; modular/policies.py:229-241)."""

    def __init__(self, obs_space: SpaceSpec, act_space: SpaceSpec, ortho_init: bool = True):
        super().__init__()
        self.obs_space, self.act_space = obs_space, act_space
        Fdim, L = obs_space.flat_len, act_space.flat_len
        self.shared_net = nn.Sequential(nn.Linear(Fdim, BC_HIDDEN), nn.Tanh(), nn.Linear(BC_HIDDEN, BC_HIDDEN), nn.Tanh())
        self.action_net = nn.Linear(BC_HIDDEN, L)
        self.value_net = nn.Linear(BC_HIDDEN, 1)
        if ortho_init:
            for mod, gain in ((self.shared_net, np.sqrt(2)), (self.action_net, 0.01), (self.value_net, 1.0)):
                for m in mod.modules():
                    if isinstance(m, nn.Linear):
                        nn.init.orthogonal_(m.weight, gain=gain)
                        m.bias.data.fill_(0.0)

    def _latent(self, obs: th.Tensor) -> th.Tensor:
        return self.shared_net(preprocess_obs(obs, self.obs_space))

    def logits(self, obs: th.Tensor) -> th.Tensor:
        return self.action_net(self._latent(obs))

    def evaluate_actions(self, obs: th.Tensor, actions: th.Tensor):
        """-> (values (n,1), log_prob (n,), entropy (n,))"""
        latent = self._latent(obs)
        z = self.action_net(latent)
        actions = actions.long().reshape(obs.shape[0], -1)
        logp, ent = 0.0, 0.0
        for c, zc in enumerate(th.split(z, list(self.act_space.nvec), dim=1)):
            dist = th.distributions.Categorical(logits=zc)
            logp = logp + dist.log_prob(actions[:, c])
            ent = ent + dist.entropy()
        return self.value_net(latent), logp, ent

    def _linears(self):
        return (self.shared_net[0], self.shared_net[2], self.action_net)

    def flat_params(self) -> np.ndarray:
        """[W1(F,32) b1 W2(32,32) b2 act_W(32,L) act_b val_W(32) val_b], weights input-major (ph_bc_layout)"""
        out = []
        for lin in self._linears():
            out += [lin.weight.detach().t().contiguous().reshape(-1), lin.bias.detach()]
        out += [self.value_net.weight.detach().reshape(-1), self.value_net.bias.detach()]
        return th.cat(out).numpy().astype(np.float32).copy()

    def load_flat_params(self, flat: np.ndarray) -> None:
        flat, o = th.as_tensor(np.asarray(flat, np.float32)), 0
        with th.no_grad():
            for lin in self._linears():
                n = lin.weight.numel()
                lin.weight.copy_(flat[o:o + n].reshape(lin.in_features, lin.out_features).t())
                o += n
                lin.bias.copy_(flat[o:o + lin.bias.numel()])
                o += lin.bias.numel()
            self.value_net.weight.copy_(flat[o:o + BC_HIDDEN].reshape(1, BC_HIDDEN))
            self.value_net.bias.copy_(flat[o + BC_HIDDEN:o + BC_HIDDEN + 1])
        assert o + BC_HIDDEN + 1 == flat.numel()


def bc_loss(policy: FeedForward32Oracle, obs: th.Tensor, acts: th.Tensor, ent_weight: float = 1e-3, l2_weight: float = 0.0):
    """``BC._calculate_loss`` (bc.py:270-315): returns (loss, stats_dict)."""
    _, log_prob, entropy = policy.evaluate_actions(obs, acts)
    prob_true_act = th.exp(log_prob).mean()
    log_prob, entropy = log_prob.mean(), entropy.mean()
    l2_norm = sum(th.sum(th.square(w)) for w in policy.parameters()) / 2     # divide by 2 to cancel the square's gradient
    ent_loss, neglogp, l2_loss = -ent_weight * entropy, -log_prob, l2_weight * l2_norm
    loss = neglogp + ent_loss + l2_loss
    return loss, dict(neglogp=neglogp.item(), loss=loss.item(), entropy=entropy.item(), ent_loss=ent_loss.item(),
                      prob_true_act=prob_true_act.item(), l2_norm=l2_norm.item(), l2_loss=l2_loss.item())


def bc_train(policy: FeedForward32Oracle, obs_data: np.ndarray, acts_data: np.ndarray, orders: Sequence[np.ndarray],
             batch_size: int = 32, ent_weight: float = 1e-3, l2_weight: float = 0.0, optimizer=None,
             max_batches: int = 0) -> List[dict]:
    """``BC.train`` (bc.py:316-353): for every epoch walk the DataLoader -- consecutive slices of ``batch_size`` rows of that
    epoch's shuffled order (``orders[epoch]`` teacher-forces ``shuffle=True``; the last slice may be short) -- and take one
    optimizer step per batch.  Optimizer: ``torch.optim.Adam(policy.parameters())`` with torch's defaults (bc.py:186-237)."""
    opt = optimizer or th.optim.Adam(policy.parameters())
    out = []
    for order in orders:
        for start in range(0, len(order), batch_size):
            idx = np.asarray(order[start:start + batch_size])
            loss, stats = bc_loss(policy, th.as_tensor(obs_data[idx]), th.as_tensor(acts_data[idx]), ent_weight, l2_weight)
            opt.zero_grad()
            loss.backward()
            opt.step()
            out.append(stats)
            if max_batches and len(out) >= max_batches:
                return out
    return out
