"""TEST INFRASTRUCTURE: scripted inputs and drivers that run the SAME scenario through either implementation of PantheonRL's
host-side surface -- the reference's own classes (loaded from /root/reference by tests/golden/make_reference_fixtures.py, build
container only) or the product's (`pantheonrl_amd.common`) -- and return a JSON-able log of everything observable at the surface.

The reference run is committed as tests/golden/ref_*.json / ref_transitions/*.npy; the product must reproduce those logs exactly.
Nothing in this module knows a rule of either implementation: games are scripts (seeded tables of observations, rewards and end
flags), agents and models only record what they are handed.  A "framework" is any namespace with the reference's class names
(SimultaneousEnv, TurnBasedEnv, MultiAgentEnv, Observation, OnPolicyAgent, HistoryQueue, ...).
"""
from __future__ import annotations

import io
import json
from collections import deque
from types import SimpleNamespace
from typing import Any, List

import numpy as np
import torch as th


# ---------------------------------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------------------------------
def plain(x: Any):
    """numpy / torch / tuples -> nested python lists and scalars (exact: f32 -> python float is lossless)"""
    if isinstance(x, th.Tensor):
        x = x.detach().cpu().numpy()
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, np.integer, np.bool_)):
        return x.item()
    if isinstance(x, (list, tuple)):
        return [plain(v) for v in x]
    if isinstance(x, dict):
        return {str(k): plain(v) for k, v in x.items()}
    return x


class RecordingAgent:
    """a partner that plays a fixed action and logs every callback it receives (duck-typed: neither implementation checks the
    class of a partner)"""

    def __init__(self, action=1):
        self.action, self.log = action, []

    def get_action(self, obs, record=True):
        self.log.append(["act", plain(obs.obs), plain(obs.state), plain(obs.action_mask), bool(record)])
        return self.action

    def update(self, reward, done):
        self.log.append(["upd", plain(reward), bool(done)])


def _ego_loop(env, T: int, ego_action=0) -> list:
    """what an SB3-style learner sees: reset, then T steps with a reset after every done"""
    out = [["reset", plain(env.reset())]]
    for _ in range(T):
        obs, rew, done, info = env.step(ego_action)
        out.append(["step", plain(obs), plain(rew), bool(done), plain(info["_partnerid"])])
        if done:
            out.append(["reset", plain(env.reset())])
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# (i) MultiAgentEnv.step / reset / _get_actions / _update_players under scripted games
# ---------------------------------------------------------------------------------------------------------------------------
class _SimGame:
    """scripted 2-player simultaneous game: observation / reward / done tables indexed by a global move counter"""

    def __init__(self, seed: int, T: int, D: int = 3, p_done: float = 0.2):
        rng = np.random.default_rng(seed)
        self.obs = rng.standard_normal((T + 1, 2, D)).astype(np.float32)
        self.rew = rng.standard_normal((T, 2)).astype(np.float32)
        self.done = rng.random(T) < p_done
        self.g = 0
        self.joint = []

    def multi_reset(self):
        return self.obs[self.g, 0], self.obs[self.g, 1]

    def multi_step(self, a0, a1):
        t = self.g
        self.g += 1
        self.joint.append([plain(a0), plain(a1)])
        return (self.obs[self.g, 0], self.obs[self.g, 1]), (float(self.rew[t, 0]), float(self.rew[t, 1])), bool(self.done[t]), {}


def drive_simultaneous(fw, seed: int = 0, T: int = 120, K: int = 3) -> dict:
    """ego vs K round-robin partners (BASELINE config 4's host logic) -- reference multiagentenv.py:118-125,149-243,395-409"""
    class Env(_SimGame, fw.SimultaneousEnv):
        def __init__(self):
            fw.SimultaneousEnv.__init__(self)
            _SimGame.__init__(self, seed, T)

    env = Env()
    partners = [RecordingAgent(action=k + 1) for k in range(K)]
    for p in partners:
        env.add_partner_agent(p)
    ego = _ego_loop(env, T)
    return {"ego": ego, "partners": [p.log for p in partners], "joint_actions": env.joint}


class _TurnGame:
    """scripted 2-player alternating game; a game never ends on its first move (the reference raises if the partner, moving
    first, ends it before the ego has moved: multiagentenv.py:234-235)"""

    def __init__(self, seed: int, T: int, D: int = 2, p_done: float = 0.25):
        rng = np.random.default_rng(seed)
        n = 4 * T + 8
        self.obs = rng.integers(0, 5, size=(n, D)).astype(np.int64)
        self.rew = rng.integers(-2, 3, size=(n, 2)).astype(np.float64)
        self.done = rng.random(n) < p_done
        self.g = 0
        self.moves_this_game = 0
        self.moves = []

    def _move(self, who: str, action):
        t = self.g
        self.g += 1
        self.moves_this_game += 1
        self.moves.append([who, plain(action)])
        done = bool(self.done[t]) and self.moves_this_game > 2
        return self.obs[self.g], (float(self.rew[t, 0]), float(self.rew[t, 1])), done, {}

    def ego_step(self, action):
        return self._move("ego", action)

    def alt_step(self, action):
        return self._move("alt", action)

    def multi_reset(self, egofirst):
        self.moves_this_game = 0
        self.moves.append(["reset", bool(egofirst)])
        return self.obs[self.g]


def drive_turnbased(fw, seed: int = 1, T: int = 100, K: int = 2) -> dict:
    """turn-based game, who starts drawn from numpy's global stream (multiagentenv.py:307-327), round-robin partners"""
    class Env(_TurnGame, fw.TurnBasedEnv):
        def __init__(self):
            fw.TurnBasedEnv.__init__(self, probegostart=0.5)
            _TurnGame.__init__(self, seed, T)

    np.random.seed(1000 + seed)
    env = Env()
    partners = [RecordingAgent(action=k + 1) for k in range(K)]
    for p in partners:
        env.add_partner_agent(p)
    ego = _ego_loop(env, T)
    return {"ego": ego, "partners": [p.log for p in partners], "moves": env.moves}


def drive_three_player(fw, seed: int = 2, T: int = 90) -> dict:
    """3 seats, the ego in the MIDDLE one (seat->partner-list index mapping, multiagentenv.py:84-91), one player moves per
    n_step in rotation (the PettingZoo AEC adapter's shape, pettingzoo.py:54-103), partners resampled at random from numpy's
    global stream (multiagentenv.py:113-116), observations carry a state and an action mask (observation.py:7-25)"""
    rng = np.random.default_rng(seed)
    n = 3 * T + 9
    obs_t = rng.standard_normal((n, 2)).astype(np.float32)
    rew_t = rng.integers(-1, 2, size=(n, 3)).astype(np.float64)
    done_t = rng.random(n) < 0.08
    mask_t = (rng.random((n, 4)) < 0.7).astype(np.int64)

    class Env(fw.MultiAgentEnv):
        def __init__(self, partners):
            super().__init__(ego_ind=1, n_players=3, resample_policy="random", partners=partners)
            self.g, self.turn, self.n_moves, self.log = 0, 0, 0, []

        def _view(self):
            return fw.Observation(obs_t[self.g], state=obs_t[self.g] * 2, action_mask=mask_t[self.g])

        def n_step(self, actions):
            t = self.g
            self.g += 1
            self.n_moves += 1
            self.log.append([self.turn, plain(actions)])
            self.turn = (self.turn + 1) % 3
            done = bool(done_t[t]) and self.n_moves > 3
            return (self.turn,), (self._view(),), tuple(float(v) for v in rew_t[t]), done, {}

        def n_reset(self):
            self.turn, self.n_moves = 0, 0
            return (0,), (self._view(),)

    np.random.seed(2000 + seed)
    seat0 = [RecordingAgent(action=10), RecordingAgent(action=11)]
    seat2 = [RecordingAgent(action=20), RecordingAgent(action=21), RecordingAgent(action=22)]
    env = Env([seat0, seat2])
    ego = _ego_loop(env, T)
    errors = []
    for bad in (dict(ego_ind=0, n_players=3, resample_policy="robin"), dict(ego_ind=0, n_players=2, resample_policy="nope"),
                dict(ego_ind=0, n_players=3, partners=[[RecordingAgent()]]), dict(ego_ind=0, n_players=2, partners=[[]])):
        try:
            type("Bad", (Env,), {"__init__": lambda self, kw=bad: fw.MultiAgentEnv.__init__(self, **kw)})()
            errors.append(None)
        except Exception as e:  # noqa: BLE001 -- the class name is the thing compared
            errors.append(type(e).__name__)
    try:
        env._get_partner_num(1)
        errors.append(None)
    except Exception as e:  # noqa: BLE001
        errors.append(type(e).__name__)
    return {"ego": ego, "seat0": [p.log for p in seat0], "seat2": [p.log for p in seat2], "n_step": env.log, "errors": errors,
            "partner_num": [env._get_partner_num(0), env._get_partner_num(2)]}


# ---------------------------------------------------------------------------------------------------------------------------
# (ii) OnPolicyAgent.get_action / update over a recording model -- reference agents.py:92-203
# ---------------------------------------------------------------------------------------------------------------------------
class RecordingBuffer:
    """the four members of SB3's RolloutBuffer the agent touches (agents.py:124-130,157,172-179,197-198); `add` keeps SB3's row
    write for rewards (the agent later does `rewards[pos - 1][0] += r`) and logs the rest"""

    def __init__(self, n_steps: int, events: list):
        self.n_steps, self.events = n_steps, events
        self.rewards = np.zeros((n_steps, 1), np.float32)
        self.pos = 0
        self.obs_shape = None

    def compute_returns_and_advantage(self, last_values, dones):
        self.events.append(["gae", plain(last_values), plain(dones), plain(self.rewards[:, 0])])

    def reset(self):
        self.events.append(["reset"])
        self.rewards[:] = 0
        self.pos = 0

    def add(self, obs, action, reward, episode_start, value, log_prob):
        self.events.append(["add", self.pos, plain(obs), plain(action), plain(reward), plain(episode_start), plain(value),
                            plain(log_prob)])
        self.rewards[self.pos % self.n_steps] = np.asarray(reward, np.float32)
        self.pos += 1


class RecordingLogger:
    def __init__(self, events):
        self.events = events

    def record(self, key, value, exclude=None):
        self.events.append(["log", key, plain(value), exclude])

    def dump(self, step=0):
        self.events.append(["dump", plain(step)])


class RecordingModel:
    """`OnPolicyAlgorithm`-shaped recorder: scripted policy outputs, every call the agent makes lands in `events`"""

    def __init__(self, obs_space, act_space, n_steps: int, seed: int, verbose: int = 0):
        self.events: list = []
        self.n_steps, self.verbose = n_steps, verbose
        self.use_sde, self.sde_sample_freq = False, -1
        self.action_space, self.observation_space = act_space, obs_space
        self.rollout_buffer = RecordingBuffer(n_steps, self.events)
        self.rollout_buffer.obs_shape = tuple(obs_space.shape)
        self.logger = None
        self.ep_info_buffer = deque(maxlen=100)
        rng = np.random.default_rng(seed)
        model = self

        class Policy:
            observation_space, action_space, device = obs_space, act_space, "cpu"

            def forward(self, obs, deterministic=False, **kw):
                o = obs.detach().cpu().numpy() if isinstance(obs, th.Tensor) else np.asarray(obs)
                a = th.as_tensor(rng.integers(0, 3, size=(o.shape[0],) + tuple(act_space.shape)))
                v = th.as_tensor(rng.standard_normal((o.shape[0], 1)).astype(np.float32))
                lp = th.as_tensor(-rng.random(o.shape[0]).astype(np.float32))
                model.events.append(["forward", plain(o), plain(a), plain(v), plain(lp)])
                return a, v, lp

            def reset_noise(self, n=1):
                model.events.append(["reset_noise", n])
        self.policy = Policy()

    def set_logger(self, logger):
        self.logger = RecordingLogger(self.events)      # whatever configure_logger built is replaced by the recorder

    def train(self):
        self.events.append(["train"])


def onpolicy_script(seed: int, n_calls: int, D: int, p_skip: float = 0.1):
    """the stream the environment side produces: per get_action an observation, whether it is recorded, and the (reward, done)
    updates that follow it (0, 1 or several: rewards add up, the last done wins -- agents.py:44-47)"""
    rng = np.random.default_rng(seed)
    script = []
    for i in range(n_calls):
        obs = rng.standard_normal(D).astype(np.float32)
        record = bool(rng.random() >= p_skip)
        n_upd = int(rng.choice([0, 1, 1, 1, 2, 3]))
        upd = [(float(np.float32(rng.standard_normal())), bool(rng.random() < 0.2)) for _ in range(n_upd)]
        script.append((obs, record, upd))
    return script


# every call recorded (what MultiAgentEnv does: multiagentenv.py:156 never passes record=False) -- the case the device replays
RECORDED_ONLY = dict(seed=9, n_steps=6, n_calls=31, verbose=0, D=4, p_skip=0.0)


def drive_onpolicy_agent(fw, spaces, seed: int = 3, n_steps: int = 5, n_calls: int = 23, verbose: int = 1, D: int = 4,
                         p_skip: float = 0.1) -> dict:
    """3+ buffer fills through OnPolicyAgent on a recording model: train trigger (:126), the cached-value bootstrap and
    `_last_episode_starts[0]` hand-over (:127-130, D-1), logging with the running episode excluded (:132-153, D-5), row
    contents (:172-179), non-recorded calls (:181, D-4), additive rewards (:198, D-2), ep_info_buffer (:109,166-168,199-203)"""
    obs_space = spaces.Box(-np.inf, np.inf, (D,), np.float32)
    act_space = spaces.Discrete(3)
    model = RecordingModel(obs_space, act_space, n_steps, seed, verbose=verbose)
    agent = fw.OnPolicyAgent(model, tb_log_name="fixture_agent")
    ev = model.events
    returned = []
    for obs, record, upd in onpolicy_script(seed, n_calls, D, p_skip):
        ev.append(["get_action", bool(record)])
        act = agent.get_action(fw.Observation(obs), record=record)
        returned.append(plain(act))
        ev.append(["state", agent.n_steps, agent.num_timesteps, agent.iteration, plain(agent._last_episode_starts),
                   plain(agent.values), plain(list(model.ep_info_buffer))])
        for r, d in upd:
            agent.update(r, d)
            ev.append(["updated", plain(r), d, plain(model.rollout_buffer.rewards[:, 0]), plain(agent._last_episode_starts),
                       plain(list(model.ep_info_buffer))])
    return {"events": ev, "returned": returned, "name": agent.name, "log_interval": agent.log_interval}


# ---------------------------------------------------------------------------------------------------------------------------
# (iii) HistoryQueue and the frame-stack wrappers -- reference wrappers.py:37-71,233-349, util.py:32-60
# ---------------------------------------------------------------------------------------------------------------------------
def drive_history_queue(fw, seed: int = 4) -> dict:
    rng = np.random.default_rng(seed)
    out = {}
    for name, default, size, n in (("vec3x4", [0.0, -1.0, 2.5], 4, 11), ("scalar_x3", [0], 3, 7), ("size1", [7, 8], 1, 4)):
        cast = float if isinstance(default[0], float) else int
        pushes = [[cast(v) for v in row] for row in rng.integers(-9, 10, size=(n, len(default)))]
        q = fw.HistoryQueue(default, size)
        views = [plain(q.add(p)) for p in pushes]
        q.reset()
        out[name] = {"default": default, "size": size, "pushes": pushes, "views": views, "after_reset": plain(q.add(pushes[0]))}
    return out


def drive_framestack(fw, spaces, seed: int = 5, T: int = 40) -> dict:
    class Sim(_SimGame, fw.SimultaneousEnv):
        def __init__(self):
            fw.SimultaneousEnv.__init__(self)
            _SimGame.__init__(self, seed, T, D=2)
            self.observation_space = spaces.Box(np.asarray([-5.0, -6.0], np.float32), np.asarray([5.0, 6.0], np.float32),
                                                dtype=np.float32)
            self.action_space = spaces.Discrete(3)

    class Turn(_TurnGame, fw.TurnBasedEnv):
        def __init__(self):
            fw.TurnBasedEnv.__init__(self, probegostart=0.5)
            _TurnGame.__init__(self, seed, T, D=2)
            self.observation_space = spaces.MultiDiscrete([5, 5])
            self.action_space = spaces.Discrete(3)

    out = {}
    for name, base in (("simultaneous", Sim), ("turnbased", Turn)):
        np.random.seed(3000 + seed)
        inner = base()
        inner.add_partner_agent(RecordingAgent(action=2))
        env = fw.frame_wrap(inner, 3)
        ego = _ego_loop(env, T)
        sp = env.observation_space
        out[name] = {"ego": ego, "partner": inner.partners[0][0].log, "wrapper": type(env).__name__,
                     "space": [type(sp).__name__, plain(getattr(sp, "low", None)), plain(getattr(sp, "high", None)),
                               plain(getattr(sp, "nvec", None))]}
    other = {"Discrete": spaces.Discrete(4), "MultiBinary": spaces.MultiBinary(3)}
    out["calculate_space"] = {}
    for k, s in other.items():
        c = fw.calculate_space(s, 2)
        holder = SimpleNamespace(observation_space=s)
        out["calculate_space"][k] = [type(c).__name__, plain(getattr(c, "nvec", None)), plain(getattr(c, "n", None)),
                                     fw.get_space_size(s), plain(fw.get_default_obs(holder))]
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# (iv) recorders and the `.npy` wire format -- reference wrappers.py:82-230, trajsaver.py:130-232
# ---------------------------------------------------------------------------------------------------------------------------
def _npy_bytes(transitions) -> bytes:
    f = io.BytesIO()
    transitions.write_transition(f)
    return f.getvalue()


def drive_recorders(fw, spaces, seed: int = 6, T: int = 37) -> dict:
    """-> {name: bytes of the .npy the recorder's transitions write}, plus the split views read back"""
    class Sim(_SimGame, fw.SimultaneousEnv):
        def __init__(self):
            fw.SimultaneousEnv.__init__(self)
            _SimGame.__init__(self, seed, T, D=3)
            self.observation_space = spaces.Box(-np.inf, np.inf, (3,), np.float32)
            self.action_space = spaces.Discrete(3)

    class Turn(_TurnGame, fw.TurnBasedEnv):
        def __init__(self):
            fw.TurnBasedEnv.__init__(self, probegostart=0.5)
            _TurnGame.__init__(self, seed, T, D=2)
            self.observation_space = spaces.MultiDiscrete([5, 5])
            self.action_space = spaces.Discrete(3)

    files, views = {}, {}
    for name, base in (("simultaneous", Sim), ("turnbased", Turn)):
        np.random.seed(4000 + seed)
        inner = base()
        inner.add_partner_agent(RecordingAgent(action=2))
        env = fw.recorder_wrap(inner)
        _ego_loop(env, T, ego_action=1)
        tr = env.get_transitions()
        files[name] = _npy_bytes(tr)
        ego, alt = tr.get_ego_transitions(), tr.get_alt_transitions()
        # TransitionsMinimal.write_transition concatenates obs and acts as they are (trajsaver.py:130-132): acts must be 2-D
        ego2, alt2 = (type(t)(t.obs, np.reshape(t.acts, (len(t), -1))) for t in (ego, alt))
        files[name + "_ego"], files[name + "_alt"] = _npy_bytes(ego2), _npy_bytes(alt2)
        back = type(tr).read_transition(io.BytesIO(files[name]), inner.observation_space, inner.action_space)
        views[name] = {"wrapper": type(env).__name__, "n_ego": len(ego), "n_alt": len(alt),
                       "fields": {k: plain(np.asarray(v)) for k, v in vars(back).items()},
                       "item0": plain({k: v for k, v in ego[0].items()})}
        back_min = type(ego).read_transition(io.BytesIO(files[name + "_ego"]), inner.observation_space, inner.action_space)
        views[name]["ego_back"] = [plain(back_min.obs), plain(back_min.acts)]
    return {"files": files, "views": views}


def drive_observation(fw) -> dict:
    """Observation defaults and the two extractors -- reference observation.py:7-43"""
    a, s, m = np.arange(3.0), np.arange(5.0), np.asarray([1, 0, 1])
    o1, o2 = fw.Observation(a), fw.Observation(a, state=s, action_mask=m)
    return {"default_state_is_obs": o1.state is o1.obs, "default_mask": o1.action_mask,
            "extract_obs": plain(fw.extract_obs(o2)), "extract_partial": plain(fw.extract_partial_obs(o2)),
            "state": plain(o2.state)}


# ---------------------------------------------------------------------------------------------------------------------------
# (ix) trainer.py: recording doubles for everything the script constructs -- the object graph of a run as an event list
#      (reference trainer.py:92-228,407-432 run as a script; pantheonrl_amd.trainer.run with its names replaced by the same doubles)
# ---------------------------------------------------------------------------------------------------------------------------
class TrainerDoubles:
    """`names()` -> {name the trainer module binds: double}.  Every construction / call that shapes the run lands in `events` with the
    objects replaced by labels: ["env", id, config], ["frame_wrap", env, n], ["model", kind, policy, config], ["agent", class, ...],
    ["add_partner", env, agent], ["learn", model, config], ["save", model, path], ..."""

    def __init__(self):
        self.events = []
        ev = self.events

        def lab(x):
            if isinstance(x, (list, tuple)):
                return [lab(v) for v in x]
            if isinstance(x, dict):
                return {k: lab(v) for k, v in x.items()}
            if isinstance(x, type):
                return getattr(x, "label", x.__name__)
            return getattr(x, "label", plain(x))
        self.lab = lab

        class Env:
            env_id = "?"

            def __init__(self, **config):
                self.label = self.env_id
                ev.append(["env", self.env_id, lab(config)])

            def getDummyEnv(self, player_num):
                d = type(self).__new__(type(self))
                d.label = f"dummy{player_num}({self.label})"
                return d

            def add_partner_agent(self, agent, player_num=1):
                ev.append(["add_partner", self.label, lab(agent), player_num])

            def get_transitions(self):
                ev.append(["get_transitions", self.label])
                return Transitions()

        class Transitions:
            label = "transitions"

            def write_transition(self, path):
                ev.append(["write_transition", path])

        class Wrapped:
            def __init__(self, kind, env, *extra):
                self.env, self.label = env, f"{kind}({env.label})"
                ev.append([kind, env.label] + [plain(x) for x in extra])

            def getDummyEnv(self, player_num):
                d = Wrapped.__new__(Wrapped)
                d.env, d.label = self.env.getDummyEnv(player_num), f"dummy{player_num}({self.label})"
                return d

            def add_partner_agent(self, agent, player_num=1):
                ev.append(["add_partner", self.label, lab(agent), player_num])

            def get_transitions(self):
                ev.append(["get_transitions", self.label])
                return Transitions()

        class RPSEnv(Env):
            env_id = "RPS-v0"

        class LiarEnv(Env):
            env_id = "LiarsDice-v0"
        self.RPSEnv, self.LiarEnv, self.Wrapped = RPSEnv, LiarEnv, Wrapped

        class Policy:
            def __init__(self, owner):
                self.label, self.owner, self.num_partners_set = f"{owner}.policy", owner, None

            def set_context(self, value):
                ev.append(["set_context", self.label, plain(np.asarray(value, np.float32))])

            def do_init_weights(self, init_main=False, init_partner=False):      # reference, LOAD of a ModularAlgorithm ego (:121-123)
                ev.append(["do_init_weights", self.label, bool(init_main), bool(init_partner)])

            def __setattr__(self, k, v):
                if k == "num_partners":
                    ev.append(["num_partners", self.label, int(v)])
                object.__setattr__(self, k, v)

        counter = {}

        def model_class(kind):
            class Model:
                label_kind = kind

                def __init__(self, policy=None, **config):
                    counter[kind] = counter.get(kind, 0) + 1
                    self.label = f"{kind}#{counter[kind]}"
                    self.policy = Policy(self.label)
                    ev.append(["model", kind, lab(policy), lab(config), self.label])

                @classmethod
                def load(cls, location, **kw):
                    m = cls.__new__(cls)
                    counter[kind] = counter.get(kind, 0) + 1
                    m.label = f"{kind}#{counter[kind]}"
                    m.policy = Policy(m.label)
                    ev.append(["load", kind, location, m.label])
                    return m

                def set_env(self, env):
                    ev.append(["set_env", self.label, lab(env)])

                def set_num_partners(self, n):                                    # the product's form of :121-123
                    ev.append(["set_num_partners", self.label, int(n)])

                def learn(self, **config):
                    ev.append(["learn", self.label, lab(config)])

                def save(self, path):
                    ev.append(["save", self.label, path])
            Model.__name__ = kind
            return Model
        self.PPO, self.ADAP, self.ModularAlgorithm = model_class("PPO"), model_class("ADAP"), model_class("ModularAlgorithm")

        def agent_class(kind, has_model=True):
            class AgentDouble:
                def __init__(self, *args, **kw):
                    counter[kind] = counter.get(kind, 0) + 1
                    self.label = f"{kind}#{counter[kind]}"
                    if has_model and args:
                        self.model = args[0]
                    ev.append(["agent", kind, lab(list(args)), lab(kw), self.label])
            AgentDouble.__name__ = kind
            return AgentDouble
        self.OnPolicyAgent, self.AdapAgent = agent_class("OnPolicyAgent"), agent_class("AdapAgent")
        self.StaticPolicyAgent = agent_class("StaticPolicyAgent", has_model=False)
        self.RPSWeightedAgent, self.LiarDefaultAgent = agent_class("RPSWeightedAgent", False), agent_class("LiarDefaultAgent", False)

        def policy_name(n):
            return type(n, (), {"label": n})
        self.AdapPolicy, self.AdapPolicyMult, self.ModularPolicy = policy_name("AdapPolicy"), policy_name("AdapPolicyMult"), policy_name("ModularPolicy")

    def make(self, env_id, **config):
        return {"RPS-v0": self.RPSEnv, "LiarsDice-v0": self.LiarEnv}[env_id](**config)

    def frame_wrap(self, env, numframes):
        return self.Wrapped("frame_wrap", env, numframes)

    def recorder_wrap(self, env, numframes=None):
        return self.Wrapped("recorder_wrap", env)


def normalise_trainer_events(events, product: bool):
    """the deliberate differences between the two scripts folded away, each named here:
    * the product gives every partner model `sampling_stream = index + 1` (an action-sampling stream of its own under the shared seed);
    * a LOADed ego: the reference wraps the environment as DummyVecEnv([Monitor(env)]) for set_env and re-initialises a
      ModularAlgorithm's partner modules through policy.do_init_weights(init_partner=True) + policy.num_partners = n
      (trainer.py:118-123); the product hands set_env the environment and calls set_num_partners(n);
    * frame stacking: the reference wraps the partner-side dummy environment as well (trainer.py:98-99), the product asks the wrapped
      environment for its dummy -- the same observation space either way;
    * FIXED ADAP partners: latent_val goes through torch.tensor there, numpy here (compared as lists)."""
    out = []
    for e in events:
        e = json.loads(json.dumps(e))
        if e[0] == "model" and isinstance(e[3], dict):
            e[3].pop("sampling_stream", None)
            if isinstance(e[3].get("env"), str):
                e[3]["env"] = e[3]["env"].replace("frame_wrap(dummy1(", "dummy1(frame_wrap(")
        if e[0] == "frame_wrap" and e[1].startswith("dummy1("):
            continue
        if e[0] == "set_env" and e[2].startswith("vec(monitor("):
            e[2] = e[2][len("vec(monitor("):-2]
        if e[0] == "do_init_weights":
            continue
        if e[0] == "num_partners":
            e = ["set_num_partners", e[1].replace(".policy", ""), e[2]]
        out.append(e)
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# (ii, continued) StaticPolicyAgent (agents.py:54-79) and RecordingAgentWrapper (agents.py:365-413) over the recording policy
# ---------------------------------------------------------------------------------------------------------------------------
def drive_static_and_wrapper(fw, spaces, seed: int = 12, n_calls: int = 9, D: int = 3) -> dict:
    obs_space = spaces.Box(-np.inf, np.inf, (D,), np.float32)
    act_space = spaces.Discrete(3)
    model = RecordingModel(obs_space, act_space, 4, seed)
    static = fw.StaticPolicyAgent(model.policy)
    wrapped = fw.RecordingAgentWrapper(static)
    rng = np.random.default_rng(seed)
    returned = []
    for i in range(n_calls):
        obs = rng.standard_normal(D).astype(np.float32)
        agent = static if i % 3 == 0 else wrapped                     # both paths; only the wrapper's calls are recorded by it
        returned.append(plain(agent.get_action(fw.Observation(obs), record=bool(i % 2))))
        agent.update(float(i), bool(i % 4 == 3))                      # a no-op for the static agent, forwarded by the wrapper
    tr = wrapped.get_transitions()
    return {"events": model.events, "returned": returned, "transitions_class": type(tr).__name__, "obs": plain(tr.obs), "acts": plain(tr.acts),
            "obs_dtype": str(np.asarray(tr.obs).dtype), "acts_dtype": str(np.asarray(tr.acts).dtype)}


# ---------------------------------------------------------------------------------------------------------------------------
# (xii) the ego's loop: learn() / collect_rollouts() on recording doubles -- reference modular/learn.py:157-219,353-403 and
#       adap/adap_learn.py:377-473 (the in-tree copy of SB3's OnPolicyAlgorithm.collect_rollouts + the context lines)
# ---------------------------------------------------------------------------------------------------------------------------
class LoopDoubles:
    """policy / per-partner buffers / one-environment vector env / callback that script their outputs from a seeded stream and log every
    call.  A buffer row is logged ONCE, when it is complete: the reference completes it in `add(obs, actions, rewards, starts, values,
    log_probs)`, the product in `forward_and_store(...)` + `add_reward(rewards)`."""

    def __init__(self, seed: int, D: int, n_act: int, K: int = 1, ctx: int = 0, p_done: float = 0.2):
        self.events = ev = []
        rng = np.random.default_rng(seed)
        loop = self

        def outputs(n):
            a = rng.integers(0, n_act, size=(n,))
            v = rng.standard_normal((n, 1)).astype(np.float32)
            lp = (-rng.random(n)).astype(np.float32)
            return a, v, lp

        class Policy:
            num_partners = K
            context = np.zeros((1, ctx), np.float32)

            def forward(self, obs, partner_idx=None, deterministic=False):                 # the reference's call
                o = obs.detach().cpu().numpy() if isinstance(obs, th.Tensor) else np.asarray(obs)
                a, v, lp = outputs(o.reshape(1, -1).shape[0] if o.ndim == 1 else o.shape[0])
                ev.append(["forward", plain(o.reshape(-1)), partner_idx])
                return th.as_tensor(a), th.as_tensor(v), th.as_tensor(lp)

            __call__ = forward

            def forward_and_store(self, obs, rb, episode_start, partner_idx=None, uniforms=None, **kw):   # the product's fused call
                o = np.asarray(obs, np.float32)
                a, v, lp = outputs(1)
                ev.append(["forward", plain(o.reshape(-1)), partner_idx])
                rb.pending = [plain(o.reshape(-1)), plain(a), plain(np.asarray(episode_start, np.float32)), plain(v), plain(lp)]
                return th.as_tensor(a), th.as_tensor(v), th.as_tensor(lp)

            def predict_values(self, obs, partner_idx=None):                                # the product's bootstrap of the ego
                o = np.asarray(obs, np.float32)
                a, v, lp = outputs(1)
                ev.append(["forward", plain(o.reshape(-1)), partner_idx])
                return th.as_tensor(v)

            def get_context(self):
                return self.context

            def set_context(self, c):
                self.context = np.asarray(c.detach().cpu().numpy() if isinstance(c, th.Tensor) else c, np.float32).reshape(1, -1)
                ev.append(["set_context", plain(self.context.reshape(-1))])

            def reset_noise(self, n=1):
                ev.append(["reset_noise", n])
        self.policy = Policy()

        class Buffer:
            def __init__(self, k):
                self.k, self.pending, self.obs_shape, self.pos, self.full = k, None, (D,), 0, False

            def reset(self):
                ev.append(["buffer.reset", self.k])
                self.pos, self.full = 0, False

            def add(self, obs, actions, rewards, starts, values, log_probs):               # reference: the whole row at once
                none_to_zero = [0.0] if starts is None else starts                         # (learn.py:176,207: None on a rollout's first step)
                ev.append(["row", self.k, plain(np.asarray(obs, np.float32).reshape(-1)), plain(np.asarray(actions).reshape(-1)),
                           plain(np.asarray(rewards, np.float32).reshape(-1)), plain(np.asarray(none_to_zero, np.float32).reshape(-1)),
                           plain(values), plain(log_probs)])

            def add_reward(self, rewards):                                                  # product: the row's reward arrives after the step
                o, a, es, v, lp = self.pending
                ev.append(["row", self.k, o, a, plain(np.asarray(rewards, np.float32).reshape(-1)), es, v, lp])
                self.pending = None

            def compute_returns_and_advantage(self, last_values, dones):
                ev.append(["gae", self.k, plain(last_values), plain(np.asarray(dones, np.float32).reshape(-1))])
        self.buffers = [Buffer(k) for k in range(K)]

        class Game:
            def set_partnerid(self, k):
                ev.append(["set_partnerid", int(k)])

        class VecEnv:
            num_envs = 1
            envs = [Game()]

            def reset(self):
                ev.append(["env.reset"])
                return rng.standard_normal((1, D)).astype(np.float32)

            def step(self, actions):
                ev.append(["env.step", plain(np.asarray(actions).reshape(-1))])
                done = bool(rng.random() < p_done)
                info = {"episode": {"r": float(np.float32(rng.standard_normal())), "l": int(rng.integers(1, 9))}} if done else {}
                return (rng.standard_normal((1, D)).astype(np.float32), np.asarray([rng.standard_normal()], np.float32),
                        np.asarray([done]), [info])
        self.env = VecEnv()

        class Callback:
            def init_callback(self, model):
                ev.append(["callback.init"])

            def on_training_start(self, l, g):
                ev.append(["callback.training_start"])

            def on_rollout_start(self):
                ev.append(["callback.rollout_start"])

            def update_locals(self, l):
                pass

            def on_step(self):
                ev.append(["callback.step"])
                return True

            def on_rollout_end(self):
                ev.append(["callback.rollout_end"])

            def on_training_end(self):
                ev.append(["callback.training_end"])
        self.callback = Callback()

        class Logger:
            def record(self, key, value, exclude=None):
                if not key.startswith("time/") or key in ("time/iterations", "time/total_timesteps"):
                    ev.append(["log", key, plain(value), exclude])
                else:
                    ev.append(["log", key, "<clock>", exclude])

            def dump(self, step=0):
                ev.append(["dump", plain(step)])
        self.logger = Logger()

    def train(self):
        self.events.append(["train"])

