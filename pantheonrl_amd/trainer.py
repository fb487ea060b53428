"""The object graph of the reference's training CLI (trainer.py:92-137,182-256,394-432) on the MI355X engine.

    python -m pantheonrl_amd.trainer RPS-v0 PPO PPO --preset 1 --seed 0 -t 10000        (BASELINE config 1)

Same positional arguments and flags as the reference for the part of the surface that sits on the PPO path:
env in {RPS-v0, LiarsDice-v0}; ego in {PPO, ADAP, ADAP_MULT, ModularAlgorithm, LOAD}; each partner in {PPO, ADAP, ADAP_MULT, FIXED, DEFAULT}; JSON configs splatted
into the constructors; `--framestack`, `--preset 1`, `--ego-save/--alt-save`, `--tensorboard-log/-name`, `--seed`, `--device`,
`--total-timesteps`, `--share-latent` (ADAP ego + ADAP partners act under the ego's context).  `--record FILE` writes the episode
transitions in the reference's `.npy` format.  A ModularAlgorithm ego gets one partner module per partner agent
(`policy_kwargs = dict(num_partners=len(args.alt))`, trainer.py:131-135).  ADAP_MULT = ADAP with AdapPolicyMult (trainer.py:129-130,207-208).
"""
from __future__ import annotations

import argparse
import json
from typing import List, Tuple

import numpy as np

from . import envs as _envs
from .common import OnPolicyAgent, StaticPolicyAgent
from .common.wrappers import frame_wrap, recorder_wrap
from .envs.liar import LiarDefaultAgent, LiarEnv
from .envs.rps import RPSEnv, RPSWeightedAgent
from .adap import ADAP, AdapAgent, AdapPolicy, AdapPolicyMult
from .modular import ModularAlgorithm, ModularPolicy
from .ppo import PPO

ADAP_TYPES = ["ADAP", "ADAP_MULT"]            # trainer.py:32
EGO_LIST = ["PPO", "ModularAlgorithm", "LOAD"] + ADAP_TYPES
PARTNER_LIST = ["PPO", "DEFAULT", "FIXED"] + ADAP_TYPES
OUT_OF_SCOPE = {"BC"}


class EnvException(Exception):
    """Raise when parameters do not align with the environment (reference trainer.py:37)."""


def input_check(args) -> None:
    """reject combinations the engine cannot honour, with the reference's error class, and complete the configs the way the
    reference's does (trainer.py:41-63: one `{}` per partner when no --alt-config was given, `verbose = 1` for the ego unless its
    config says otherwise, --tensorboard-log and --tensorboard-name together or not at all).  --share-latent's check is the caller's
    next step, as in the reference's main (trainer.py:399-403); pinned to the reference's text by tests/golden/ref_trainer_cli.json"""
    if args.env not in _envs.REGISTRY:
        raise EnvException(f"unknown or out-of-scope environment {args.env!r}; available: {sorted(_envs.REGISTRY)}")
    for name in [args.ego] + list(args.alt):
        if name in OUT_OF_SCOPE:
            raise EnvException(f"{name} agents are outside the PPO rollout+update path this engine implements")
    if args.ego not in EGO_LIST:
        raise EnvException(f"ego must be one of {EGO_LIST}")
    for name in args.alt:
        if name not in PARTNER_LIST:
            raise EnvException(f"partners must be among {PARTNER_LIST}")
    if args.alt_config is None:                         # trainer.py:50-52
        args.alt_config = [{} for _ in args.alt]
    elif len(args.alt_config) != len(args.alt):
        raise EnvException("Number of partners is different from number of configs")
    if "verbose" not in args.ego_config:                # trainer.py:57-59
        args.ego_config["verbose"] = 1
    if (args.tensorboard_log is not None) != (args.tensorboard_name is not None):       # trainer.py:61-63
        raise EnvException("Must define log and names for tensorboard")
    if args.framestack > 1 and args.env_config.get("framestack_incompatible", False):
        raise EnvException("this environment cannot be frame-stacked")


def latent_check(args) -> None:
    """--share-latent: every agent must be ADAP with the ego's context size and sampler (trainer.py:65-89)"""
    if args.ego not in ADAP_TYPES or not all(v in ADAP_TYPES for v in args.alt):
        raise EnvException("both agents must be ADAP or ADAP_MULT to share latent spaces")
    args.ego_config.setdefault("context_size", 3)
    args.ego_config.setdefault("context_sampler", "l2")
    for conf in args.alt_config:
        for key in ("context_size", "context_sampler"):
            if conf.setdefault(key, args.ego_config[key]) != args.ego_config[key]:
                raise EnvException("both agents must have similar configs to share latent spaces")


def generate_env(args) -> Tuple[object, object]:
    """env + the partner-side dummy env, optionally frame-stacked (trainer.py:92-104)"""
    env = _envs.make(args.env, **args.env_config)
    altenv = env.getDummyEnv(1)
    if args.framestack > 1:
        env = frame_wrap(env, args.framestack)
        altenv = env.getDummyEnv(1)   # the wrapped env carries the stacked observation space for both seats
    if args.record is not None:
        env = recorder_wrap(env)      # trainer.py:101-102
    return env, altenv


def gen_load(config: dict, policy_type: str, location: str):
    if policy_type in ADAP_TYPES:   # trainer.py:140-147: a fixed ADAP policy acts under a given latent value
        if "latent_val" not in config:
            raise EnvException("latent_val needs to be specified for FIXED ADAP policy")
        agent = ADAP.load(location, device=config.get("device", "cuda"))
        agent.policy.set_context(np.asarray(config.pop("latent_val"), np.float32))
        return agent
    if policy_type == "BC":         # trainer.py:152-153: BCShell(reconstruct_policy(location)) -- an object with a .policy
        from .bc import reconstruct_policy
        return _bc_shell(reconstruct_policy(location, device=config.get("device", "cuda")))
    if policy_type == "ModularAlgorithm":     # trainer.py:150-151
        return ModularAlgorithm.load(location, device=config.get("device", "cuda"))
    if policy_type != "PPO":
        raise EnvException("Not a valid FIXED/LOAD policy")
    return PPO.load(location, device=config.get("device", "cuda"))


class _BCShell:
    """pantheonrl.algos.bc.BCShell (bc.py:29-31): just enough of a model for StaticPolicyAgent / tester.py"""

    def __init__(self, policy):
        self.policy = policy


def _bc_shell(policy) -> _BCShell:
    return _BCShell(policy)


def generate_ego(env, args):
    """the ego learner (trainer.py:107-137)"""
    kwargs = dict(args.ego_config)
    kwargs.update(env=env, device=args.device, tensorboard_log=args.tensorboard_log)
    if args.seed is not None:
        kwargs["seed"] = args.seed
    if args.ego == "LOAD":
        model = gen_load(kwargs, kwargs["type"], kwargs["location"])
        model.set_env(env)
        if kwargs["type"] == "ModularAlgorithm":     # trainer.py:121-123: fresh partner modules for the partners of THIS run
            model.set_num_partners(len(args.alt))
        return model
    if args.ego == "ADAP":          # trainer.py:127-128
        return ADAP(policy=AdapPolicy, **kwargs)
    if args.ego == "ADAP_MULT":     # trainer.py:129-130
        return ADAP(policy=AdapPolicyMult, **kwargs)
    if args.ego == "ModularAlgorithm":               # trainer.py:131-135
        return ModularAlgorithm(policy=ModularPolicy, policy_kwargs=dict(num_partners=len(args.alt)), **kwargs)
    return PPO(policy="MlpPolicy", **kwargs)


def gen_partner(kind: str, config: dict, altenv, ego, args, index: int):
    """one partner agent (trainer.py:182-213)"""
    config = dict(config)
    if kind == "FIXED":
        return StaticPolicyAgent(gen_load(config, config["type"], config["location"]).policy)
    if kind == "DEFAULT":
        base = altenv
        while hasattr(base, "env"):      # unwrap frame-stack / recorder wrappers down to the game itself
            base = base.env
        if isinstance(base, RPSEnv):
            return RPSWeightedAgent(**config)
        if config:
            raise EnvException("No config possible for this default agent")
        if isinstance(base, LiarEnv):
            return LiarDefaultAgent()
        raise EnvException("No default policy available")
    agentarg = {}
    if args.tensorboard_log is not None:
        agentarg = {"tensorboard_log": args.tensorboard_log,
                    "tb_log_name": f"{args.tensorboard_name}_alt_{index}"}
    config.update(env=altenv, device=args.device, verbose=args.verbose_partner)
    if args.seed is not None:
        config["seed"] = args.seed
    # same seed as the ego (same initial weights, as in the reference: set_random_seed(seed) runs before every model's
    # init), but an action-sampling stream of its own -- see ActorCriticPolicy.__init__
    config["sampling_stream"] = index + 1
    if kind in ADAP_TYPES:          # trainer.py:205-213
        shared = ego.policy if args.share_latent else None
        return AdapAgent(ADAP(policy=AdapPolicy if kind == "ADAP" else AdapPolicyMult, **config), latent_syncer=shared, **agentarg)
    return OnPolicyAgent(PPO(policy="MlpPolicy", **config), **agentarg)


def generate_partners(altenv, env, ego, args) -> List:
    partners = []
    for i, kind in enumerate(args.alt):
        agent = gen_partner(kind, args.alt_config[i], altenv, ego, args, i)
        print(f"Partner {i}: {agent}")
        env.add_partner_agent(agent)
        partners.append(agent)
    return partners


def preset(args, preset_id: int):
    """default log / save names (trainer.py:231-256)"""
    if preset_id != 1:
        raise Exception("Invalid preset id")
    env_name = args.env
    if "layout_name" in args.env_config:
        env_name = f"{args.env}-{args.env_config['layout_name']}"
    seed = 0 if args.seed is None else args.seed
    args.tensorboard_log = args.tensorboard_log or "logs"
    args.tensorboard_name = args.tensorboard_name or f"{env_name}-{args.ego}{args.alt[0]}-{seed}"
    args.ego_save = args.ego_save or f"models/{env_name}-{args.ego}-ego-{seed}"
    args.alt_save = args.alt_save or f"models/{env_name}-{args.alt[0]}-alt-{seed}"
    return args


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Train an ego agent with partner agents in a multi-agent environment "
                                            "on the MI355X PPO engine (flags as in PantheonRL's trainer.py)")
    p.add_argument("env", help="environment id")
    p.add_argument("ego", help="algorithm of the ego agent")
    p.add_argument("alt", nargs="+", help="algorithm(s) of the partner agent(s)")
    p.add_argument("--total-timesteps", "-t", type=int, default=500000)
    p.add_argument("--device", "-d", default="auto")     # (the reference's default; the engine takes it as "cuda": there is no other device)
    p.add_argument("--seed", "-s", type=int)
    p.add_argument("--ego-config", type=json.loads, default={})
    # the reference's form is ONE flag with a JSON string per partner (nargs='*', trainer.py:356-359); repeating the flag works as well
    p.add_argument("--alt-config", type=json.loads, nargs="*", action="extend")
    p.add_argument("--env-config", type=json.loads, default={})
    p.add_argument("--framestack", "-f", type=int, default=1)
    p.add_argument("--record", "-r")
    p.add_argument("--ego-save")
    p.add_argument("--alt-save")
    p.add_argument("--share-latent", "-l", action="store_true")
    p.add_argument("--tensorboard-log")
    p.add_argument("--tensorboard-name")
    p.add_argument("--verbose-partner", action="store_true")
    p.add_argument("--preset", type=int)
    p.add_argument("--n-envs", type=int, default=1,
                   help="(extension) >1: E copies of the game resident on the device, PPO-vs-PPO self-play without the "
                        "host in the step loop")
    return p


def run_vectorised(args):
    """`<env> PPO PPO --n-envs E`: the reference's object graph with every table on the device.
    total_timesteps counts ego transitions over all tables, as SB3 does for a VecEnv."""
    import torch as th

    from .envs.vec import RaggedVecOnPolicyAgent, VecLiarsDice, VecLiarSelfPlay, VecRPS, selfplay_iteration
    from .vec import VecOnPolicyAgent
    if args.ego != "PPO" or list(args.alt) != ["PPO"]:
        raise EnvException("--n-envs supports the PPO-vs-PPO self-play pairing")
    if args.framestack > 1 or args.record is not None:
        raise EnvException("--n-envs cannot be combined with --framestack / --record")
    game = {"RPS-v0": VecRPS, "LiarsDice-v0": VecLiarsDice}.get(args.env)
    if game is None:
        raise EnvException(f"no device-resident form of {args.env}")
    E = int(args.n_envs)
    spaces = type("Spaces", (), dict(observation_space=game.observation_space, action_space=game.action_space,
                                     _is_dummy_space_env=True))()
    models = []
    for offset, config in enumerate((dict(args.ego_config), dict(args.alt_config[0]))):
        config.setdefault("n_steps", 128)
        config.setdefault("batch_size", max(64, E * config["n_steps"] // 4))
        config.update(env=spaces, device=args.device, n_envs=E)
        if args.seed is not None:
            config["seed"] = args.seed + offset
        model = PPO(policy="MlpPolicy", **config)
        model.device_permutations = True
        models.append(model)
    ego = VecOnPolicyAgent(models[0])
    n_steps = models[0].n_steps
    iterations = max(1, -(-args.total_timesteps // (E * n_steps)))
    if args.env == "RPS-v0":
        alt = VecOnPolicyAgent(models[1])
        env = VecRPS(E, models[0].policy.ctx, models[0].device)
        for _ in range(iterations):
            selfplay_iteration(env, ego, alt, n_steps)
    else:
        alt = RaggedVecOnPolicyAgent(models[1])
        # the dice get a Philox key of their own: keyed by the bare seed they would BE the ego's sampling uniforms (same key,
        # same step counter, same row); without --seed every run deals a fresh sequence
        import random as _random
        base = args.seed if args.seed is not None else _random.SystemRandom().randrange(2 ** 31)
        dice_seed = ((base * 0x9E3779B97F4A7C15) ^ 0xD1CE0D1CE0D1CE) & 0x7FFFFFFFFFFFFFFF
        env = VecLiarSelfPlay(E, ego, alt, seed=dice_seed, **args.env_config)
        if iterations >= 4 and env.native:
            # two launch-by-launch iterations size the workspaces, the rest replay one hipGraph per iteration
            from .envs.vec import LiarIterationGraph
            graph = LiarIterationGraph(env, n_steps, warmup=2)
            for _ in range(iterations - 2):
                graph.launch()
        else:
            for _ in range(iterations):
                env.rollout_and_learn(n_steps)
    th.cuda.synchronize()
    print(f"vectorised self-play: {iterations} iterations x {E} envs x {n_steps} steps; "
          f"ego updates {ego.iteration}, partner updates {alt.iteration}")
    if args.ego_save:
        models[0].save(args.ego_save)
    if args.alt_save:
        models[1].save(args.alt_save)
    return models[0], [alt], env


def parse_cli(argv=None):
    """argv -> the checked and completed arguments, in the order of the reference's main (trainer.py:395-403): parse, preset,
    input_check, latent_check when --share-latent"""
    args = build_parser().parse_args(argv)
    args.ego_config, args.env_config = dict(args.ego_config), dict(args.env_config)     # (argparse hands out its default objects)
    if args.preset:
        args = preset(args, args.preset)
    input_check(args)
    if args.share_latent:
        latent_check(args)
    return args


def run(argv=None):
    """parse, build the graph, learn, save -- returns (ego, partners, env) for callers and tests"""
    args = parse_cli(argv)
    print(f"Arguments: {args}")
    if args.n_envs > 1:
        return run_vectorised(args)
    env, altenv = generate_env(args)
    print(f"Environment: {env}; Partner env: {altenv}")
    ego = generate_ego(env, args)
    print(f"Ego: {ego}")
    partners = generate_partners(altenv, env, ego, args)
    learn_config = {"total_timesteps": args.total_timesteps}
    if args.tensorboard_log:
        learn_config["tb_log_name"] = args.tensorboard_name
    ego.learn(**learn_config)
    if args.record is not None:
        env.get_transitions().write_transition(args.record)   # trainer.py:415-417
    if args.ego_save:
        ego.save(args.ego_save)
    if args.alt_save:
        multiple = len(partners) > 1
        for i, partner in enumerate(partners):
            model = getattr(partner, "model", None)
            if model is None:           # DEFAULT / FIXED partners have nothing to save (trainer.py:423-432)
                continue
            model.save(f"{args.alt_save}/{i}" if multiple else args.alt_save)
    return ego, partners, env


if __name__ == "__main__":
    run()
