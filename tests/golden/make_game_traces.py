"""Regenerates tests/golden/game_traces.npz: seeded dice and raw moves of the two in-tree integer games and the
observations / rewards / done flags they lead to.

    python tests/golden/make_game_traces.py

THESE ARE RESTATEMENT OUTPUT, NOT REFERENCE OUTPUT: the reference's games (pantheonrl/envs/rpsgym/rps.py:41-45,
pantheonrl/envs/liargym/liar.py:53-102) import `gym`, which is absent here, so they cannot be executed; the traces come
from this repository's Python restatement of the same rules (pantheonrl_amd/envs/{rps,liar}.py).  They (i) pin that
restatement against drift (not-gpu test) and (ii) give the device kernels a file-based expectation that does not run the
product's Python games at test time.  The independent anchor is tests/golden/liar_hand_worked.json: games worked by hand
from the reference's source text, which both the restatement and the kernels must reproduce.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from pantheonrl_amd.envs.liar import LiarEnv  # noqa: E402
from pantheonrl_amd.envs.rps import rps_payoff  # noqa: E402


def main():
    rng = np.random.default_rng(2024)
    out = {}
    a0, a1 = rng.integers(0, 3, 999).astype(np.int32), rng.integers(0, 3, 999).astype(np.int32)
    out.update(rps_ego=a0, rps_alt=a1, rps_ego_reward=rps_payoff(a0, a1).astype(np.float32))
    E, S = 192, 14
    np.random.seed(7)                      # LiarEnv.multi_reset rolls with numpy's global generator
    tables = [LiarEnv() for _ in range(E)]
    hands = np.zeros((E, 12), np.int32)
    for e, t in enumerate(tables):
        t.multi_reset(True)
        hands[e, :6], hands[e, 6:] = t.egohand, t.althand
    ego_turn = rng.random(E) < 0.5
    acts = np.zeros((S, E, 2), np.int32)
    obs = np.zeros((S, E, 30), np.float32)
    rew = np.zeros((S, E, 2), np.float32)
    done = np.zeros((S, E), np.uint8)
    alive = np.zeros((S, E), np.uint8)
    live = np.ones(E, bool)
    turn = ego_turn.copy()
    for s in range(S):
        acts[s] = np.stack([rng.integers(0, 7, E), rng.integers(0, 12, E)], 1)
        if s < 3:
            acts[s, :, 1] = np.minimum(acts[s, :, 1], 3 * s + 2)     # keep some games going for a few raises
        alive[s] = live
        for e in np.nonzero(live)[0]:
            o, r, d, _ = tables[e].player_step(acts[s, e], bool(turn[e]))
            obs[s, e], rew[s, e], done[s, e] = np.asarray(o, np.float32), r, d
            if d:
                live[e] = False
        turn = ~turn
    assert not live.any()
    out.update(liar_hands=hands, liar_ego_first=ego_turn.astype(np.uint8), liar_acts=acts, liar_obs=obs, liar_rew=rew,
               liar_done=done, liar_alive=alive)
    np.savez_compressed(os.path.join(HERE, "game_traces.npz"), **out)
    print("wrote game_traces.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
