"""BASELINE config 4 on K + 1 ranks (here sharing one GPU over gloo): ego on rank 0, partner k on rank 1 + k, per-environment
round-robin partner ids, partner observations routed from the ego's rank.  After the run every rank's buffers travel to
rank 0, which replays the environments (all of them, or RR_SAMPLE of them spread over the vector) through the reference's control
flow (pantheonrl/common/multiagentenv.py:149-243) with replay agents and compares buffers row by row.  RR_ENV_IMPL picks whose
statement of that control flow does the replay: "product" = pantheonrl_amd.common.SimultaneousEnv (itself tested against
hand-derived expectations), "oracle" = oracle/multiagent_oracle.py (an independent restatement that shares no code with the
product).  RR_E / RR_T / RR_ITER / RR_SAMPLE size the run: BASELINE config 4 as written is RR_E=1024, four ranks."""
import os
import sys

import numpy as np
import torch as th
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pantheonrl_amd import PPO, roundrobin as rr, spaces as sp  # noqa: E402
from pantheonrl_amd.common import Agent, SimultaneousEnv  # noqa: E402
from pantheonrl_amd.vec import SyntheticRollouts  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
K = world - 1
dev_index = int(os.environ.get("LOCAL_RANK", "0")) % th.cuda.device_count()
th.cuda.set_device(dev_index)
dist.init_process_group(os.environ.get("RR_BACKEND", "gloo"))
device = th.device("cuda", dev_index)
E, T, D, ITER, BONUS = int(os.environ.get("RR_E", "24")), int(os.environ.get("RR_T", "8")), 62, int(os.environ.get("RR_ITER", "3")), 0.25
SAMPLE = int(os.environ.get("RR_SAMPLE", "0")) or E
ENV_IMPL = os.environ.get("RR_ENV_IMPL", "product")
HORIZON = int(os.environ.get("RR_HORIZON", "5"))
T_PARTNER = int(os.environ.get("RR_T_PARTNER", "64"))      # long enough that no partner trains during the replayed run
obs_space, act_space = sp.Box(-np.inf, np.inf, (D,)), sp.Discrete(6)
env = type("S", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()
data_ego = SyntheticRollouts(obs_space, E, T, HORIZON, 0, device)
data_alt = SyntheticRollouts(obs_space, E, T, HORIZON, 1, device)
model = PPO("MlpPolicy", env, n_steps=T if rank == 0 else T_PARTNER, n_envs=E, batch_size=E * 4, n_epochs=1,
            seed=100 + rank)
side = rr.make_rank(model, K, T, data_ego=data_ego, obs_alt=data_alt.obs, bonus=BONUS)
# default: the engine-side carrier (one native call per iteration and rank, IPC-mapped receive areas); PH_RR_NATIVE=0: the
# host-driven one (torch.distributed collectives and torch book-keeping per step)
assert side.native == (os.environ.get("PH_RR_NATIVE", "1") != "0"), side.native
trace = []
for it in range(ITER):
    dist.barrier()
    side.run_iteration()
    th.cuda.synchronize()
    if rank == 0:
        trace.append(dict(ego_actions=model.rollout_buffer.actions[..., 0].cpu().numpy().astype(np.int64).copy(),
                          alt_actions=side.alt_actions.cpu().numpy().copy(), partner=side.partner_trace.cpu().numpy().copy(),
                          rewards=model.rollout_buffer.rewards.cpu().numpy().copy(),
                          starts=model.rollout_buffer.episode_starts.cpu().numpy().copy(),
                          obs=model.rollout_buffer.observations.cpu().numpy().copy()))
mine = None
if rank > 0:
    rb, ag = model.rollout_buffer, side.agent
    mine = dict(pos=ag.pos.cpu().numpy(), obs=rb.observations.cpu().numpy(), actions=rb.actions[..., 0].cpu().numpy(),
                rewards=rb.rewards.cpu().numpy(), starts=rb.episode_starts.cpu().numpy(), updates=side.updates)
everyone = [None] * world
dist.all_gather_object(everyone, mine)

if rank == 0 and T_PARTNER >= ITER * T:
    obs0, obs1 = data_ego.obs.cpu().numpy(), data_alt.obs.cpu().numpy()
    base, done = data_ego.rewards.cpu().numpy(), data_ego.dones.cpu().numpy()

    CLOCK = [0]          # global step of the environment being replayed

    class Replay(Agent):
        """plays the actions the device partner produced; logs what OnPolicyAgent would store"""

        def __init__(self):
            self.rows, self.last_start = [], True

        def get_action(self, obs, record=True):
            self.rows.append(dict(obs=np.asarray(obs.obs, np.float32), start=self.last_start, reward=np.float32(0), action=None,
                                  g=CLOCK[0]))
            return None     # the scripted game looks the action up itself

        def update(self, reward, done):
            self.rows[-1]["reward"] = np.float32(self.rows[-1]["reward"] + np.float32(reward))
            self.last_start = bool(done)

    class ScriptedGame:
        """the scripted 2-player game itself (multi_reset / multi_step); `wrapper` = whoever drives it"""

        def __init__(self, e):
            self.e, self.g, self.wrapper = e, 0, None

        def _active_partner(self):
            w = self.wrapper
            return w.partners[0][w.partnerids[0]] if ENV_IMPL == "product" else w.partners[w.partnerid]

        def multi_reset(self):
            t = self.g % T
            return obs0[t, self.e], obs1[t, self.e]

        def multi_step(self, ego_action, alt_action):
            it, t = divmod(self.g, T)
            a1 = int(trace[it]["alt_actions"][t, self.e])        # what the active partner's device forward sampled
            self._active_partner().rows[-1]["action"] = a1
            r = np.float32(base[t, self.e] + (np.float32(BONUS) if int(ego_action) == a1 else np.float32(0)))
            d = bool(done[t, self.e])
            self.g += 1
            tn = self.g % T
            return (obs0[tn, self.e], obs1[tn, self.e]), (r, r), d, {}

    if ENV_IMPL == "product":
        class Scripted(ScriptedGame, SimultaneousEnv):
            observation_space, action_space = obs_space, act_space

            def __init__(self, e):
                SimultaneousEnv.__init__(self)
                ScriptedGame.__init__(self, e)
                self.wrapper = self
    else:
        from oracle.multiagent_oracle import RoundRobinSimultaneousOracle

    sample = sorted({int(round(i * (E - 1) / max(SAMPLE - 1, 1))) for i in range(SAMPLE)})
    for e in sample:
        partners = [Replay() for _ in range(K)]
        if ENV_IMPL == "product":
            game = Scripted(e)
            for p in partners:
                game.add_partner_agent(p)
        else:
            inner = ScriptedGame(e)
            game = RoundRobinSimultaneousOracle(inner, partners)
            inner.wrapper = game
        ego_rows = []
        ob = game.reset()
        start = True
        for g in range(ITER * T):
            it, t = divmod(g, T)
            CLOCK[0] = g
            assert np.array_equal(np.asarray(ob, np.float32), obs0[t, e])
            ob_next, r, d, info = game.step(int(trace[it]["ego_actions"][t, e]))
            assert info["_partnerid"][0] == trace[it]["partner"][t, e], (e, g, info["_partnerid"], trace[it]["partner"][t, e])
            ego_rows.append((t, start, np.float32(r)))
            start = d
            ob = game.reset() if d else ob_next              # DummyVecEnv auto-reset (SB3): reset() resamples the partner
        # ego buffer of the last iteration
        for (t, st, r) in ego_rows[-T:]:
            assert trace[-1]["rewards"][t, e] == r, (e, t, trace[-1]["rewards"][t, e], r)
            assert bool(trace[-1]["starts"][t, e]) == st
            assert np.array_equal(trace[-1]["obs"][t, e], obs0[t, e])
        # partner columns: every recorded row of every partner, in order
        for k, p in enumerate(partners):
            dev = everyone[1 + k]
            n = int(dev["pos"][e])
            assert n == len(p.rows), (e, k, n, len(p.rows))
            for i, row in enumerate(p.rows):
                assert np.array_equal(dev["obs"][i, e], row["obs"]), (e, k, i)
                assert int(dev["actions"][i, e]) == row["action"], (e, k, i)
                # the reward of the run's very last step reaches a partner with the NEXT step's routing block
                want = np.float32(0) if row["g"] == ITER * T - 1 else row["reward"]
                assert dev["rewards"][i, e] == want, (e, k, i, dev["rewards"][i, e], want)
                assert bool(dev["starts"][i, e]) == row["start"], (e, k, i)
    used = sorted({int(v) for tr in trace for v in np.unique(tr["partner"])})
    assert used == list(range(K)), used                       # every partner was played against
    print("RR_REPLAY_OK", flush=True)
if rank > 0 and T_PARTNER < ITER * T:
    assert side.updates >= 1, "a partner whose columns filled must have trained"
if side.native:
    assert side.link.timeouts() == 0, side.link.timeouts()
dist.barrier()
print(f"RR_OK rank {rank}/{world} native={side.native}", flush=True)
dist.destroy_process_group()
