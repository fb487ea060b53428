"""Pins the integer game fixtures to the REFERENCE's own code (build container only; `/root/reference` never travels).

    python tests/golden/check_against_reference.py              # verify both fixtures against the reference's functions
    python tests/golden/check_against_reference.py --regenerate # rewrite game_traces.npz from the reference's functions

What runs here is the reference's source, loaded from where it lies (nothing is copied into this repository):
`pantheonrl/envs/liargym/liar.py` (LiarEnv.player_step / sanitize_action / eval_bluff / getObs / multi_reset, :53-102) and
`pantheonrl/envs/rpsgym/rps.py` (RPSEnv.multi_step, :41-45).  Those files import `gym` and two PantheonRL modules that
import `gym` / `stable_baselines3` (absent here); the rule functions themselves use nothing from them beyond two module-level
space constants and the base-class names, so the modules are executed with inert stand-ins for exactly those names
(`gym.spaces.*` constructors that keep their arguments; `Agent`, `TurnBasedEnv`, `SimultaneousEnv` as empty base classes).
No game rule comes from a stand-in.

Checked / produced:
  * tests/golden/liar_hand_worked.json -- five games worked by hand from the source text: every step's observation, reward
    and done flag is replayed through the reference's LiarEnv;
  * tests/golden/game_traces.npz -- 192 seeded Liar's Dice tables x 14 steps and 999 RPS rounds: every live step is replayed
    through the reference's LiarEnv / RPSEnv; `--regenerate` writes the file from those replays (same seeds and raw moves as
    make_game_traces.py, so the product's restatement and the reference must agree bit for bit or the not-gpu test fails).
With this the integer rows of the oracle are REFERENCE-generated.  The float path (SB3 1.7.0 arithmetic) stays unpinned:
stable-baselines3 is absent and the reference holds no vectors for it.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("PANTHEON_REFERENCE", "/root/reference")


def _stand_ins() -> dict:
    """inert modules for the names the two game files import but whose behaviour the rule functions never use"""
    class _Space:
        def __init__(self, *a, **k):
            self.args, self.kwargs = a, k

    gym = types.ModuleType("gym")
    gym.spaces = types.ModuleType("gym.spaces")
    for name in ("MultiDiscrete", "Discrete", "Box", "MultiBinary"):
        setattr(gym.spaces, name, type(name, (_Space,), {}))
    gym.Env = object

    class _Base:
        def __init__(self, *a, **k):
            pass
    agents = types.ModuleType("pantheonrl.common.agents")
    agents.Agent = _Base
    mae = types.ModuleType("pantheonrl.common.multiagentenv")
    mae.TurnBasedEnv, mae.SimultaneousEnv, mae.MultiAgentEnv = _Base, _Base, _Base
    pkg, common = types.ModuleType("pantheonrl"), types.ModuleType("pantheonrl.common")
    return {"gym": gym, "gym.spaces": gym.spaces, "pantheonrl": pkg, "pantheonrl.common": common,
            "pantheonrl.common.agents": agents, "pantheonrl.common.multiagentenv": mae}


def load_reference_games(root: str = REFERENCE):
    """-> (liar module namespace, rps module namespace), executed from the reference's files"""
    saved = {k: sys.modules.get(k) for k in _stand_ins()}
    sys.modules.update(_stand_ins())
    try:
        out = []
        for rel in ("pantheonrl/envs/liargym/liar.py", "pantheonrl/envs/rpsgym/rps.py"):
            path = os.path.join(root, rel)
            ns = {"__name__": "reference_" + os.path.basename(rel)[:-3], "__file__": path}
            with open(path) as fh:
                exec(compile(fh.read(), path, "exec"), ns)   # the reference's text, run in place
            out.append(ns)
        return out[0], out[1]
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _table(liar_ns, egohand, althand):
    t = liar_ns["LiarEnv"]()
    t.history, t.egohand, t.althand = [], [int(v) for v in egohand], [int(v) for v in althand]
    return t


def check_hand_worked(liar_ns) -> int:
    games = json.load(open(os.path.join(HERE, "liar_hand_worked.json")))["games"]
    n = 0
    for g in games:
        t = _table(liar_ns, g["egohand"], g["althand"])
        for st in g["steps"]:
            obs, rew, done, _ = t.player_step(np.asarray(st["raw"]), bool(st["is_ego"]))
            assert [int(v) for v in obs] == st["obs"], (g["note"], st, obs)
            assert [int(v) for v in rew] == st["rew"] and bool(done) == st["done"], (g["note"], st, rew, done)
            n += 1
    return n


def replay_traces(liar_ns, rps_ns, z) -> dict:
    """every step of the committed inputs (hands, who moves first, raw moves) through the reference -> outputs"""
    env = rps_ns["RPSEnv"]()
    rps = np.asarray([env.multi_step(int(a), int(b))[1][0] for a, b in zip(z["rps_ego"], z["rps_alt"])], np.float32)
    hands, acts = z["liar_hands"], z["liar_acts"]
    S, E = acts.shape[:2]
    obs = np.zeros((S, E, 30), np.float32)
    rew = np.zeros((S, E, 2), np.float32)
    done = np.zeros((S, E), np.uint8)
    alive = np.zeros((S, E), np.uint8)
    tables = [_table(liar_ns, hands[e, :6], hands[e, 6:]) for e in range(E)]
    live = np.ones(E, bool)
    turn = z["liar_ego_first"].astype(bool).copy()
    for s in range(S):
        alive[s] = live
        for e in np.nonzero(live)[0]:
            o, r, d, _ = tables[e].player_step(acts[s, e], bool(turn[e]))
            obs[s, e], rew[s, e], done[s, e] = np.asarray(o, np.float32), r, d
            if d:
                live[e] = False
        turn = ~turn
    return dict(rps_ego_reward=rps, liar_obs=obs, liar_rew=rew, liar_done=done, liar_alive=alive)


def check_dice(liar_ns) -> None:
    """multi_reset / randRoll (liar.py:23-27,98-102) under numpy's global generator: the hands the committed traces start from"""
    z = np.load(os.path.join(HERE, "game_traces.npz"))
    np.random.seed(7)
    for e in range(z["liar_hands"].shape[0]):
        t = liar_ns["LiarEnv"]()
        t.multi_reset(True)
        assert t.egohand + t.althand == [int(v) for v in z["liar_hands"][e]], e


def main() -> int:
    if not os.path.isdir(REFERENCE):
        print(f"{REFERENCE} not present: this check runs in the build container only")
        return 0
    liar_ns, rps_ns = load_reference_games()
    n_hand = check_hand_worked(liar_ns)
    path = os.path.join(HERE, "game_traces.npz")
    z = dict(np.load(path))
    out = replay_traces(liar_ns, rps_ns, z)
    if "--regenerate" in sys.argv:
        z.update(out)
        np.savez_compressed(path, **z)
        print("game_traces.npz rewritten from the reference's LiarEnv / RPSEnv")
    for k, v in out.items():
        assert np.array_equal(v, z[k]), f"game_traces.npz[{k}] differs from the reference's output"
    check_dice(liar_ns)
    print(f"reference check ok: {n_hand} hand-worked steps, {int(out['liar_alive'].sum())} trace steps, "
          f"{len(out['rps_ego_reward'])} RPS rounds, {z['liar_hands'].shape[0]} dealt tables")
    return 0


if __name__ == "__main__":
    sys.exit(main())
