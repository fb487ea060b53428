"""`bctrainer.py` of the reference (bctrainer.py:1-104) on the engine: behavioural cloning from a recorded trajectory.

    python -m pantheonrl_amd.bctrainer RPS-v0 demo.npy --total-epochs 10 --save clone.pt [--choose-alt] [--l2 0] [-f N]

Same positional arguments and flags.  The trajectory is what `trainer.py --record FILE` / `tester.py --record FILE` wrote
(`common/trajsaver.py`); the whole training run is one persistent kernel launch (`ph_bc_train`)."""
from __future__ import annotations

import argparse
import json

from .bc import BC
from .common import trajsaver
from .common.multiagentenv import SimultaneousEnv
from .trainer import generate_env


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="BC algorithm given a trajectory (flags as in PantheonRL's bctrainer.py)")
    p.add_argument("env", help="The environment the trajectory was recorded in")
    p.add_argument("trajectory", type=str, help="Location of trajectory")
    p.add_argument("--choose-alt", action="store_true", help="Train from the alt trajectory (default is ego)")
    p.add_argument("--total-epochs", "-t", type=int, default=10)
    p.add_argument("--l2", type=float, default=0, help="Value of l2 weight of BC algorithm")
    p.add_argument("--device", "-d", default="cuda")
    p.add_argument("--env-config", type=json.loads, default={})
    p.add_argument("--framestack", "-f", type=int, default=1)
    p.add_argument("--save", help="File to save the agent into")
    return p


def run(argv=None) -> BC:
    args = build_parser().parse_args(argv)
    args.record = None
    print(f"Arguments: {args}")
    env, altenv = generate_env(args)
    print(f"Environment: {env}; Partner env: {altenv}")
    cls = trajsaver.SimultaneousTransitions if isinstance(env, SimultaneousEnv) else trajsaver.TurnBasedTransitions
    side = altenv if args.choose_alt else env                       # bctrainer.py:83-84
    transition = cls.read_transition(args.trajectory, side.observation_space, side.action_space)
    data = transition.get_alt_transitions() if args.choose_alt else transition.get_ego_transitions()
    clone = BC(observation_space=side.observation_space, action_space=side.action_space, expert_data=data,
               l2_weight=args.l2, device=args.device)
    clone.train(n_epochs=args.total_epochs)
    if args.save is not None:
        clone.save_policy(args.save)
    return clone


if __name__ == "__main__":
    run()
