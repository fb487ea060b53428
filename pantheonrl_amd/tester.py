"""`tester.py` of the reference (tester.py:1-198) on the engine: play a loaded ego against a loaded or default partner.

    python -m pantheonrl_amd.tester RPS-v0 PPO DEFAULT --ego-load models/ego --alt-config '{"r": 1}' -t 100 [--record FILE]

Same positional arguments and flags (`--render` is accepted and ignored: the in-tree games draw nothing).  Ego types: PPO, ADAP
(with `--ego-config '{"latent_val": [...]}'`), BC; partner types: the same plus DEFAULT."""
from __future__ import annotations

import argparse
import json
from typing import List

import numpy as np

from .common import StaticPolicyAgent
from .trainer import EnvException, gen_load, gen_partner, generate_env

EGO_LIST = ["PPO", "BC", "ADAP"]
PARTNER_LIST = ["PPO", "DEFAULT", "BC", "ADAP"]


def input_check(args) -> None:
    """tester.py:14-31"""
    if args.ego not in EGO_LIST or args.alt not in PARTNER_LIST:
        raise EnvException(f"ego must be one of {EGO_LIST}, alt one of {PARTNER_LIST}")
    args.ego_config.setdefault("verbose", 1)
    if args.ego_load is None:
        raise EnvException("Need to provide file for ego to load")
    if (args.alt_load is None) != (args.alt == "DEFAULT"):
        raise EnvException("Load policy if and only if alt is not DEFAULT")


def generate_agent(env, policy_type: str, config: dict, location, args):
    """tester.py:34-38: the environment's default agent, or a fixed agent around the loaded policy"""
    if policy_type == "DEFAULT":
        return gen_partner("DEFAULT", config, env, None, args, 0)
    cfg = {k: v for k, v in config.items() if k != "verbose"}
    cfg.setdefault("device", args.device)
    return StaticPolicyAgent(gen_load(cfg, policy_type, location).policy)


def run_test(ego, env, num_episodes: int) -> List[float]:
    """tester.py:41-63"""
    env.set_ego_extractor(lambda obs: obs)
    rewards = []
    for _ in range(num_episodes):
        obs, done, reward = env.reset(), False, 0.0
        while not done:
            action = ego.get_action(obs, False)
            obs, newreward, done, _ = env.step(action)
            reward += newreward
        rewards.append(reward)
    print(f"Average Reward: {sum(rewards) / num_episodes}")
    print(f"Standard Deviation: {np.std(rewards)}")
    return rewards


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Test ego and partner in an environment (flags as in PantheonRL's tester.py)")
    p.add_argument("env")
    p.add_argument("ego")
    p.add_argument("alt")
    p.add_argument("--total-episodes", "-t", type=int, default=100)
    p.add_argument("--device", "-d", default="cuda")
    p.add_argument("--seed", "-s", type=int)
    p.add_argument("--ego-config", type=json.loads, default={})
    p.add_argument("--alt-config", type=json.loads, default={})
    p.add_argument("--env-config", type=json.loads, default={})
    p.add_argument("--framestack", "-f", type=int, default=1)
    p.add_argument("--record", "-r")
    p.add_argument("--render", action="store_true")
    p.add_argument("--ego-load")
    p.add_argument("--alt-load")
    return p


def run(argv=None) -> List[float]:
    args = build_parser().parse_args(argv)
    input_check(args)
    args.tensorboard_log, args.tensorboard_name, args.verbose_partner = None, None, False
    if args.seed is not None:
        np.random.seed(args.seed)
    print(f"Arguments: {args}")
    env, altenv = generate_env(args)
    print(f"Environment: {env}; Partner env: {altenv}")
    ego = generate_agent(env, args.ego, args.ego_config, args.ego_load, args)
    print(f"Ego: {ego}")
    alt = generate_agent(altenv, args.alt, args.alt_config, args.alt_load, args)
    env.add_partner_agent(alt)
    print(f"Alt: {alt}")
    rewards = run_test(ego, env, args.total_episodes)
    if args.record is not None:
        env.get_transitions().write_transition(args.record)
    return rewards


if __name__ == "__main__":
    run()
