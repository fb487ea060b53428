"""Behavioural cloning on the MI355X engine -- the surface of the reference's `pantheonrl.algos.bc` (bc.py:113-366):

    clone = BC(observation_space=, action_space=, expert_data=TransitionsMinimal, l2_weight=, device=)   # bctrainer.py:96-100
    clone.train(n_epochs=...)            # or n_batches=...
    clone.save_policy(path);  policy = reconstruct_policy(path)

The policy is the reference's default `FeedForward32Policy` (pantheonrl/common/util.py:114-123: one shared 32-32 tanh trunk
under action_net and value_net).  Training runs as ONE launch of a persistent workgroup (`ph_bc_train`, csrc/ph_bc.hip) that
walks every minibatch of every epoch with the parameters resident in LDS; the expert data are uploaded once.  torch is used
for device memory and the shuffles' host RNG only.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch as th

from . import _native as nat
from . import spaces as sp
from .common.trajsaver import TransitionsMinimal
from .ppo import _require_cuda


class FeedForward32Policy:
    """SB3 ActorCriticPolicy(net_arch=[32, 32]) with the parameters in HBM as one flat vector (ph_bc_layout)."""

    def __init__(self, observation_space, action_space, device="cuda", ortho_init: bool = True, seed: Optional[int] = None):
        self.device = _require_cuda(device)
        self.observation_space, self.action_space = observation_space, action_space
        if type(action_space).__name__ == "Box":
            raise ValueError("BC's FeedForward32Policy clones categorical (Discrete / MultiDiscrete) actions (reference bc.py:291-303 "
                             "on the engine's heads); Box action spaces are PPO's GaussianActorCriticPolicy only")
        self.spec = sp.make_spec(observation_space, action_space)
        self.ctx = nat.Context(self.device.index)
        self.layout = nat.PhBcLayout()
        nat.check(self.ctx.lib.ph_bc_layout_of(C.byref(self.spec), C.byref(self.layout)))
        lay = self.layout
        self.params = th.zeros(lay.P, dtype=th.float32, device=self.device)
        self._seed = int(seed) if seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        self._counter = 0
        H = nat.PH_BC_HIDDEN
        flat = th.zeros(lay.P)
        for off, fin, fout, gain in ((lay.W1, lay.F, H, np.sqrt(2)), (lay.W2, H, H, np.sqrt(2)), (lay.act_W, H, lay.L, 0.01),
                                     (lay.val_W, H, 1, 1.0)):
            w = th.empty(fout, fin)
            if ortho_init:
                th.nn.init.orthogonal_(w, gain=gain)          # gains: modular/policies.py:229-241
            else:
                th.nn.init.kaiming_uniform_(w, a=np.sqrt(5))
            flat[off:off + fin * fout] = w.t().contiguous().reshape(-1)
        self.params.copy_(flat)

    _SD = (("mlp_extractor.shared_net.0", "W1", "b1"), ("mlp_extractor.shared_net.2", "W2", "b2"),
           ("action_net", "act_W", "act_b"), ("value_net", "val_W", "val_b"))

    def _shapes(self):
        lay, H = self.layout, nat.PH_BC_HIDDEN
        return {"W1": (lay.F, H), "W2": (H, H), "act_W": (H, lay.L), "val_W": (H, 1)}

    def state_dict(self) -> Dict[str, th.Tensor]:
        flat, lay, out = self.params.detach().cpu(), self.layout, {}
        for mod, wname, bname in self._SD:
            fin, fout = self._shapes()[wname]
            woff, boff = getattr(lay, wname), getattr(lay, bname)
            out[mod + ".weight"] = flat[woff:woff + fin * fout].reshape(fin, fout).t().contiguous()
            out[mod + ".bias"] = flat[boff:boff + fout].clone()
        return out

    def load_state_dict(self, sd: Dict[str, th.Tensor]) -> None:
        flat, lay = th.zeros(self.layout.P), self.layout
        for mod, wname, bname in self._SD:
            fin, fout = self._shapes()[wname]
            woff, boff = getattr(lay, wname), getattr(lay, bname)
            flat[woff:woff + fin * fout] = th.as_tensor(sd[mod + ".weight"]).float().reshape(fout, fin).t().reshape(-1)
            flat[boff:boff + fout] = th.as_tensor(sd[mod + ".bias"]).float().reshape(-1)
        self.params.copy_(flat)

    def get_flat_params(self) -> np.ndarray:
        return self.params.detach().cpu().numpy().copy()

    def set_flat_params(self, flat) -> None:
        self.params.copy_(th.as_tensor(np.asarray(flat, np.float32)))

    def _launch(self, obs, *, given=None, uniforms=None, deterministic=False, want_logits=False, mask=None):
        lay, dev = self.layout, self.device
        t = obs.detach() if isinstance(obs, th.Tensor) else th.as_tensor(np.asarray(obs))
        obs_t = t.to(device=dev, dtype=th.float32).reshape(-1, lay.D).contiguous()
        n = obs_t.shape[0]
        acts = th.empty((n, lay.A), dtype=th.int32, device=dev)
        values, logp, ent = (th.empty(n, dtype=th.float32, device=dev) for _ in range(3))
        logits = th.empty((n, lay.L), dtype=th.float32, device=dev) if want_logits else None
        f32 = lambda x: None if x is None else th.as_tensor(np.asarray(x, np.float32)).to(dev).reshape(n, lay.A).contiguous()  # noqa: E731
        g, u = f32(given), f32(uniforms)
        m = None if mask is None else th.as_tensor(mask).to(device=dev, dtype=th.uint8).reshape(n, lay.L).contiguous()
        self.ctx.set_stream(th.cuda.current_stream(dev).cuda_stream)
        self._counter += 1
        nat.check(self.ctx.lib.ph_bc_forward(self.ctx.handle, C.byref(self.spec), self.params.data_ptr(), obs_t.data_ptr(), n,
                                             nat.ptr(m), nat.ptr(u), nat.ptr(g), self._seed, self._counter,
                                             int(bool(deterministic)), acts.data_ptr(), values.data_ptr(), logp.data_ptr(),
                                             ent.data_ptr(), nat.ptr(logits)))
        return acts, values.reshape(n, 1), logp, ent, logits

    def forward(self, obs, deterministic: bool = False, action_mask=None, uniforms=None):
        """-> (actions, values (n,1), log_prob (n,)) like ActorCriticPolicy.forward (util.py:79): usable by StaticPolicyAgent"""
        acts, values, logp, _, _ = self._launch(obs, deterministic=deterministic, uniforms=uniforms, mask=action_mask)
        return acts.long().reshape((-1,) + tuple(self.action_space.shape)), values, logp

    __call__ = forward

    def evaluate_actions(self, obs, actions):
        _, values, logp, ent, _ = self._launch(obs, given=actions)
        return values, logp, ent

    def get_logits(self, obs) -> th.Tensor:
        return self._launch(obs, deterministic=True, want_logits=True)[4]

    def predict(self, obs, deterministic: bool = False):
        return self.forward(obs, deterministic=deterministic)[0].cpu().numpy(), None

    def reset_noise(self, n_envs: int = 1) -> None:
        return None


class BC:
    """Behavioural cloning (bc.py:113-366).  `expert_data`: a TransitionsMinimal (bctrainer.py:88-94)."""

    DEFAULT_BATCH_SIZE: int = 32

    def __init__(self, observation_space, action_space, *, policy_class=FeedForward32Policy, policy_kwargs=None,
                 expert_data: Optional[TransitionsMinimal] = None, optimizer_kwargs: Optional[dict] = None,
                 ent_weight: float = 1e-3, l2_weight: float = 0.0, device="cuda", batch_size: Optional[int] = None):
        if optimizer_kwargs and "weight_decay" in optimizer_kwargs:
            raise ValueError("Use the parameter l2_weight instead of weight_decay.")     # bc.py:213-216
        if policy_class is not FeedForward32Policy:
            raise ValueError("the engine's BC trains the reference's default FeedForward32Policy")
        self.observation_space, self.action_space = observation_space, action_space
        self.policy = FeedForward32Policy(observation_space, action_space, device=device, **(policy_kwargs or {}))
        self.device = self.policy.device
        ok = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8)          # torch.optim.Adam defaults (bc.py:186,235-237)
        ok.update(optimizer_kwargs or {})
        self.optimizer_kwargs = ok
        self.ent_weight, self.l2_weight = float(ent_weight), float(l2_weight)
        self.batch_size = int(batch_size or self.DEFAULT_BATCH_SIZE)
        P = self.policy.layout.P
        self.adam_m = th.zeros(P, dtype=th.float32, device=self.device)
        self.adam_v = th.zeros(P, dtype=th.float32, device=self.device)
        self.opt_step = th.zeros(1, dtype=th.int32, device=self.device)
        self._obs = self._acts = None
        self.last_stats: Optional[np.ndarray] = None
        if expert_data is not None:
            self.set_expert_data_loader(expert_data)

    def set_expert_data_loader(self, expert_data: TransitionsMinimal) -> None:
        """upload the (obs, acts) table once; batches are drawn on the device from a per-epoch shuffled order (bc.py:239-267)"""
        lay = self.policy.layout
        obs = np.asarray(expert_data.obs, np.float32).reshape(len(expert_data), lay.D)
        acts = np.asarray(expert_data.acts, np.float32).reshape(len(expert_data), lay.A)
        self._obs = th.as_tensor(obs).to(self.device).contiguous()
        self._acts = th.as_tensor(acts).to(self.device).contiguous()

    def hyper(self) -> nat.PhBcHyper:
        h = nat.PhBcHyper()
        h.learning_rate = float(self.optimizer_kwargs["lr"])
        h.adam_beta1, h.adam_beta2 = (float(b) for b in self.optimizer_kwargs["betas"])
        h.adam_eps, h.ent_weight, h.l2_weight = float(self.optimizer_kwargs["eps"]), self.ent_weight, self.l2_weight
        return h

    def train(self, *, n_epochs: Optional[int] = None, n_batches: Optional[int] = None, orders: Optional[np.ndarray] = None,
              log_interval: int = 100, on_epoch_end=None, on_batch_end=None) -> np.ndarray:
        """exactly one of n_epochs / n_batches (bc.py:316-332).  `orders` (n_epochs, N) teacher-forces the shuffles (default:
        np.random.permutation per epoch).  Returns the per-minibatch statistics (minibatches, 8): _native.BC_STAT_NAMES."""
        if (n_epochs is None) == (n_batches is None):
            raise ValueError("Must provide exactly one of `n_epochs` and `n_batches` arguments.")
        if self._obs is None:
            raise ValueError("no expert data: call set_expert_data_loader first")
        N = int(self._obs.shape[0])
        per_epoch = -(-N // self.batch_size)
        epochs = int(n_epochs) if n_epochs is not None else -(-int(n_batches) // per_epoch)
        if orders is None:
            orders = np.stack([np.random.permutation(N) for _ in range(epochs)])
        orders = np.ascontiguousarray(orders, dtype=np.int32).reshape(epochs, N)
        total = epochs * per_epoch if n_batches is None else int(n_batches)
        order_t = th.as_tensor(orders).to(self.device)
        stats = th.zeros((total, nat.PH_BC_NSTAT), dtype=th.float32, device=self.device)
        opt = nat.PhOptState()
        pol = self.policy
        opt.params, opt.adam_m, opt.adam_v = pol.params.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr()
        opt.step = self.opt_step.data_ptr()
        hp = self.hyper()
        pol.ctx.set_stream(th.cuda.current_stream(self.device).cuda_stream)

        def launch(order_ptr, n_rows, n_ep, max_b, stats_row):
            nat.check(pol.ctx.lib.ph_bc_train(pol.ctx.handle, C.byref(pol.spec), C.byref(opt), self._obs.data_ptr(),
                                              self._acts.data_ptr(), order_ptr, n_rows, self.batch_size, n_ep, max_b, C.byref(hp),
                                              stats.data_ptr() + 4 * nat.PH_BC_NSTAT * stats_row))
        if on_batch_end is None and on_epoch_end is None:
            # the whole run as ONE launch of the persistent workgroup (parameters never leave LDS)
            launch(order_t.data_ptr(), N, epochs, 0 if n_batches is None else int(n_batches), 0)
        else:
            # A callback must see the state the reference's loop shows it (bc.py:333-353: on_batch_end after every optimizer
            # step, on_epoch_end after every pass): the run is cut at the callback's granularity -- one launch per epoch, or
            # one per batch -- with the optimizer state carried on the device: same kernel, same order of steps (the results
            # differ from the single launch only through the rounding of Adam's bias corrections, which one launch carries as
            # running fp64 products and a fresh launch restarts from pow(beta, step)).
            done = 0
            for ep in range(epochs):
                left = total - done
                if left <= 0:
                    break
                row = order_t.data_ptr() + 4 * N * ep
                if on_batch_end is None:
                    nb = min(per_epoch, left)
                    launch(row, N, 1, 0 if nb == per_epoch else nb, done)
                    done += nb
                else:
                    for b in range(min(per_epoch, left)):
                        lo = b * self.batch_size
                        launch(row + 4 * lo, min(self.batch_size, N - lo), 1, 0, done)
                        done += 1
                        on_batch_end()
                if on_epoch_end is not None and (n_batches is None or done < total):   # n_batches: the loop returns mid-epoch (bc.py:141-143)
                    on_epoch_end()
        self.last_stats = stats.cpu().numpy()
        if log_interval and getattr(self, "logger", None) is not None:
            for b in range(0, total, int(log_interval)):      # bc.py:343-347: every log_interval batches
                for k, name in enumerate(nat.BC_STAT_NAMES):
                    self.logger.record(f"bc/{name}", float(self.last_stats[b, k]))
                self.logger.dump(step=b)
        return self.last_stats

    def save_policy(self, policy_path: str) -> None:
        """bc.py:355-360 saves the torch module; here the state_dict under SB3's module names plus the spaces"""
        from .ppo import PPO
        th.save({"format": "pantheonrl_amd-bc-1", "state_dict": self.policy.state_dict(),
                 "observation_space": PPO._space_to_json(self.observation_space),
                 "action_space": PPO._space_to_json(self.action_space)}, policy_path)


def reconstruct_policy(policy_path: str, device="cuda") -> FeedForward32Policy:
    """bc.py:33-47"""
    from .ppo import PPO
    blob = th.load(policy_path, map_location="cpu")
    if not isinstance(blob, dict) or blob.get("format") != "pantheonrl_amd-bc-1":
        raise ValueError(f"{policy_path} is not a policy saved by pantheonrl_amd.bc.BC.save_policy")
    pol = FeedForward32Policy(PPO._space_from_json(blob["observation_space"]), PPO._space_from_json(blob["action_space"]),
                              device=device)
    pol.load_state_dict(blob["state_dict"])
    return pol
