#!/usr/bin/env python
"""bench.py -- headline benchmark: env-steps/sec (all agents), Overcooked-simple PPO self-play, on N MI355X.

A "step" (--steps K) is ONE whole PPO iteration of every learning agent on this node: a rollout of n_steps=128
environment steps over n_envs=1024 synthetic Overcooked-shaped environments (policy forward + rollout-buffer row write
+ late reward `+=` per step), the GAE pass, and PPO.train() with n_epochs=10 over minibatches of n_envs*n_steps/4
rows (SURVEY.md 8d "throughput mode").  Inputs are synthetic, seeded, and already resident in HBM when the timed
region starts.  value = agent-steps/s summed over all agents on all GPUs.  At N=1 the 128 steps of a learner's rollout run
as ONE launch by default (--rollout scripted: the synthetic transitions are a script in HBM; bitwise the per-step walk) and
the line also carries the figure with one launch per environment step (`stepwise_rollout`, --rollout stepwise), which is
how the N>1 layouts -- they exchange actions every step -- run.

Launch: `python bench.py` (N=1), `python bench.py --gpus N` (spawns its own N ranks through torch.distributed.run), or
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W
One rank per GPU (rank r -> device LOCAL_RANK); `--gpus` larger than the visible device count is an error, and so is a
WORLD_SIZE that disagrees with `--gpus`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# the host driver of the GPU boxes supports dmabuf IPC only: without this RCCL / cross-process device-memory sharing fails with
# `hipIpcGetMemHandle: invalid argument`.  Set before the first HIP call (the ROCm runtime reads it when it initialises).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch as th  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# synthetic shape sets of the other BASELINE configs; the default and the headline is "overcooked"
WORKLOADS = {
    "overcooked": dict(obs=("box", 62), act=[6], horizon=400, n_envs=1024,
                       name="OvercookedMultiEnv-v0 layout=simple, PPO self-play"),
    "liar": dict(obs=("multidiscrete", [7] * 6 + [7, 12] * 12), act=[7, 12], horizon=6, n_envs=256,
                 name="LiarsDice-v0 PPO-vs-PPO (synthetic transitions of the game's shapes)"),
    "mpe8": dict(obs=("box", 48), act=[5], horizon=25, n_envs=1024,
                 name="PettingZoo MPE simple_spread_v3 N=8 shapes, PPO learners"),
    "rps": dict(obs=("multidiscrete", [1]), act=[3], horizon=1, n_envs=1024, name="RPS-v0 PPO-vs-PPO shapes"),
}
_T0 = time.perf_counter()


def log(msg: str) -> None:
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


def usable_cores() -> int:
    """host cores this process may actually use: affinity mask, capped by a cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:  # noqa: BLE001
            pass
    return max(1, n)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed PPO iterations (K)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="overcooked")
    ap.add_argument("--n-envs", type=int, default=0, help="0 = the workload's BASELINE value (1024 for overcooked)")
    ap.add_argument("--n-steps", type=int, default=128)
    ap.add_argument("--n-epochs", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=0, help="0 = n_envs*n_steps/4")
    ap.add_argument("--agents-per-gpu", type=int, default=2,
                    help="2 = a self-play pair per GPU (ego PPO + partner PPO, trainer.py ... PPO PPO: the headline config); "
                         "1 = north_star's one agent per GPU (N agents on N GPUs, every step's partner is on another rank)")
    ap.add_argument("--mode", choices=("auto", "graph", "jointgraph", "eager", "fusedstep", "roundrobin"), default="auto",
                    help="graph: one hipGraph per agent-iteration (N=1 default); fusedstep: one fused launch of all local "
                         "agents + one action exchange per env step (the N>1 path; at N=1 the exchange is a local copy); "
                         "eager: per-agent launches; roundrobin: BASELINE config 4 -- one ego (rank 0, hosting the environments) "
                         "against --gpus - 1 partners, one agent per GPU, per-environment round-robin partner ids, partner "
                         "observations routed from the ego's rank (pantheonrl_amd/roundrobin.py)")
    ap.add_argument("--rollout", choices=("stepwise", "scripted"), default="scripted",
                    help="N=1 graph mode only.  scripted (default): the synthetic transitions are a script resident in HBM, so the "
                         "n_steps steps of a learner run as ONE launch (ph_scripted_rollout; bitwise the stepwise walk, test "
                         "test_scripted_rollout_is_bitwise_the_per_step_walk); the line then also carries the stepwise figure "
                         "(`stepwise_rollout`).  stepwise: one launch per environment step (get_action / update per step, as an "
                         "external environment or the N>1 per-step action exchange drives them)")
    ap.add_argument("--action-masks", choices=("none", "env", "policy+env"), default="none",
                    help="fusedstep layouts: SURVEY.md 8(d)'s config-5 variant -- Bernoulli(0.8) action masks with at least one legal "
                         "action.  env: the reference's plain PPO partner (the policy never sees the mask, agents.py:162; the "
                         "environment replaces an illegal sample by the first legal index, pettingzoo.py:81-82; the buffer keeps the "
                         "sample).  policy+env: ModularPolicy's -30 logit offset (modular/policies.py:330-333) as well")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to "
                    "exercise the N>1 code path with several ranks on ONE GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the extra figures measured after the timed region "
                    "(stepwise_rollout, one_agent_per_gpu): a kernel trace of the run then holds the headline's launches only")
    return ap.parse_args()


def build_agents(args, device):
    from pantheonrl_amd import PPO, spaces as sp
    from pantheonrl_amd.vec import SyntheticRollouts, VecOnPolicyAgent
    wl = WORKLOADS[args.workload]
    obs_space = sp.Box(-np.inf, np.inf, (wl["obs"][1],)) if wl["obs"][0] == "box" else (
        sp.Discrete(wl["obs"][1][0]) if len(wl["obs"][1]) == 1 else sp.MultiDiscrete(wl["obs"][1]))
    act_space = sp.Discrete(wl["act"][0]) if len(wl["act"]) == 1 else sp.MultiDiscrete(wl["act"])
    env = type("SpacesOnly", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()
    rank = int(os.environ.get("RANK", "0"))
    agents, datas = [], []
    for i in range(args.agents_per_gpu):
        seed = 1000 * rank + i
        model = PPO("MlpPolicy", env, n_steps=args.n_steps, n_envs=args.n_envs, batch_size=args.batch_size,
                    n_epochs=args.n_epochs, seed=seed, device=device)
        model.device_permutations = True
        shared = int(os.environ.get("WORLD_SIZE", "1")) > max(th.cuda.device_count(), 1)   # several ranks on one device (functional tests)
        if args.agents_per_gpu == 1 and not shared and os.environ.get("PH_EXCLUSIVE_DEVICE", "1") != "0":
            # one learner per GPU (north_star's layout, config 5): its update launches have the device to themselves, so the slab
            # reduction may use the wide blocks (same summation tree, same results: include/pantheon_hip.h)
            model.policy.ctx.set_exclusive_device(True)
        agents.append(VecOnPolicyAgent(model))
        datas.append(SyntheticRollouts(obs_space, args.n_envs, args.n_steps, wl["horizon"], seed % 3, device))
    return agents, datas


def cpu_baseline(args):
    """the oracle executed SB3-style (per-step add with host copies, Python-loop GAE, eager autograd) on the host
    cores of this box: whole iterations of ONE agent at the same sizes for ~10 s (a bounded sample of the workload)."""
    from oracle.sb3_oracle import (MlpPolicyOracle, PPOHyper, RolloutBufferOracle, SpaceSpec, synthetic_iteration)
    cores = usable_cores()
    th.set_num_threads(cores)
    log(f"cpu_baseline: oracle on {cores} host threads (os.cpu_count()={os.cpu_count()})")
    T, E = args.n_steps, args.n_envs
    th.manual_seed(0)
    wl = WORKLOADS[args.workload]
    obs_spec = SpaceSpec("box", dim=wl["obs"][1]) if wl["obs"][0] == "box" else SpaceSpec(
        "multidiscrete", nvec=tuple(wl["obs"][1]))
    act_spec = SpaceSpec("discrete" if len(wl["act"]) == 1 else "multidiscrete", nvec=tuple(wl["act"]))
    pol = MlpPolicyOracle(obs_spec, act_spec)
    buf = RolloutBufferOracle(T, E, obs_spec.stored_len, act_spec.stored_len)
    rng = np.random.default_rng(0)
    if obs_spec.kind == "box":
        obs = rng.standard_normal((T, E, obs_spec.dim), dtype=np.float32)
    else:
        obs = (rng.random((T, E, len(obs_spec.nvec))) * np.asarray(obs_spec.nvec)).astype(np.int64).astype(np.float32)
    rew = rng.standard_normal((T, E), dtype=np.float32)
    done = rng.random((T, E)) < 1.0 / wl["horizon"]
    hp = PPOHyper(batch_size=args.batch_size, n_epochs=args.n_epochs)
    # a bounded sample of the same workload: whole iterations of one agent until ~10 s of CPU work (at most 16 iterations)
    t0 = time.perf_counter()
    n_it = 0
    while n_it < 16 and (n_it == 0 or time.perf_counter() - t0 < 10.0):
        synthetic_iteration(pol, buf, hp, obs, rew, done)
        n_it += 1
    dt = time.perf_counter() - t0
    out = {"value": n_it * T * E / dt, "unit": "agent-steps/s", "cores": cores, "kind": "port",
           "os_cpu_count": os.cpu_count(),
           "sample": f"{n_it} PPO iteration(s) of 1 agent (n_envs={E}, n_steps={T}, batch={args.batch_size}, "
                     f"n_epochs={args.n_epochs}) = {n_it * T * E} agent-steps in {dt:.2f}s, torch threads={cores}"}
    # SURVEY.md 8d also asks for (i) the reference's own semantics -- E = 1, n_steps = 2048, batch 64, 10 epochs: batch-1
    # forwards and 320 Adam steps per 2048 transitions (agents.py:111-203 on SB3 defaults) -- and for one host thread.
    # Bounded samples: a quarter rollout of (i) on all threads and on one thread (the update dominates and scales with it).
    try:
        def timed_e1(threads):
            th.set_num_threads(threads)
            T1 = 512
            pol1 = MlpPolicyOracle(obs_spec, act_spec)
            buf1 = RolloutBufferOracle(T1, 1, obs_spec.stored_len, act_spec.stored_len)
            idx = np.arange(T1) % T
            o1, r1, d1 = obs[idx][:, :1], rew[idx][:, :1], done[idx][:, :1]
            t1 = time.perf_counter()
            synthetic_iteration(pol1, buf1, PPOHyper(batch_size=64, n_epochs=10), o1, r1, d1)
            return T1 / (time.perf_counter() - t1)
        out["reference_semantics_E1"] = {"value": timed_e1(cores), "cores": cores, "unit": "agent-steps/s",
                                         "sample": "n_envs=1, 512 of the 2048 default n_steps, batch 64, 10 epochs"}
        out["reference_semantics_E1_single_thread"] = {"value": timed_e1(1), "cores": 1, "unit": "agent-steps/s",
                                                       "sample": "same, torch.set_num_threads(1)"}
        th.set_num_threads(cores)
    except Exception as exc:  # noqa: BLE001 -- extra figures, never fatal
        out["reference_semantics_E1"] = {"error": str(exc)}
    return out


def gpu_reference_semantics_e1(args, device):
    """The engine driven exactly as the reference drives a partner (agents.py:111-203 on SB3's defaults): n_envs = 1, one
    get_action / update pair per environment step with the action copied back to the host every step (the environment is a
    host program), n_steps rows, batch 64, 10 epochs = 10 * n_steps / 64 dependent Adam steps per rollout.  Same bounded
    sample as cpu_baseline's reference_semantics_E1 (512 of the 2048 default steps).  This is the launch-latency-bound corner
    of the engine: printed so that nobody has to guess what E = 1 costs on the GPU next to the CPU figure."""
    from pantheonrl_amd import PPO, spaces as sp
    from pantheonrl_amd.common import Observation, OnPolicyAgent
    wl = WORKLOADS[args.workload]
    obs_space = sp.Box(-np.inf, np.inf, (wl["obs"][1],)) if wl["obs"][0] == "box" else (
        sp.Discrete(wl["obs"][1][0]) if len(wl["obs"][1]) == 1 else sp.MultiDiscrete(wl["obs"][1]))
    act_space = sp.Discrete(wl["act"][0]) if len(wl["act"]) == 1 else sp.MultiDiscrete(wl["act"])
    env = type("SpacesOnly", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()
    T1 = 512
    model = PPO("MlpPolicy", env, n_steps=T1, n_envs=1, batch_size=64, n_epochs=10, seed=0, device=device)
    agent = OnPolicyAgent(model)
    rng = np.random.default_rng(0)
    obs = (rng.standard_normal((2 * T1 + 1, wl["obs"][1])).astype(np.float32) if wl["obs"][0] == "box" else
           (rng.random((2 * T1 + 1, len(wl["obs"][1]))) * np.asarray(wl["obs"][1])).astype(np.int64).astype(np.float32))
    rew = rng.standard_normal(2 * T1 + 1).astype(np.float32)

    def run(lo, hi):
        for t in range(lo, hi):
            agent.get_action(Observation(obs[t]))
            agent.update(float(rew[t]), False)
    run(0, T1 + 1)                    # fills the buffer and triggers the first train() (warm-up: workspaces, first launches)
    th.cuda.synchronize(device)
    # where the time goes: the one train() of the sample is bracketed by device synchronisations of its own (the steps before it
    # end with the action read back, so nothing is in flight when it starts); the rest is the per-step host round trip
    train_s, inner_train = [0.0, 0], model.train

    def timed_train(*a, **k):
        th.cuda.synchronize(device)
        t = time.perf_counter()
        r = inner_train(*a, **k)
        th.cuda.synchronize(device)
        train_s[0] += time.perf_counter() - t
        train_s[1] += 1
        return r
    model.train = timed_train
    t0 = time.perf_counter()
    run(T1 + 1, 2 * T1 + 1)           # T1 steps including one whole train() of 80 dependent Adam steps
    th.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    model.train = inner_train
    n_adam = 10 * T1 // 64
    return {"value": T1 / dt, "unit": "agent-steps/s", "ms_per_env_step_incl_update": 1e3 * dt / T1,
            "split": {"train_calls": train_s[1], "train_ms": 1e3 * train_s[0],
                      "us_per_adam_step": 1e6 * train_s[0] / max(1, n_adam * train_s[1]),
                      "us_per_env_step_host_round_trip": 1e6 * (dt - train_s[0]) / T1,
                      "note": "get_action (H2D observation, one fused forward + row write, D2H action) + update (reward +=) per "
                              "step from Python; train() = GAE + 10 epochs of batch-64 minibatches, timed between two device "
                              "synchronisations"},
            "sample": f"n_envs=1, {T1} get_action/update pairs from the host (action read back every step) + one train() "
                      f"(batch 64, 10 epochs = {10 * T1 // 64} Adam steps)"}


def hbm_kernels(model, pol, rb, lay, hp, gm):
    """ph_bench_train_kernels: every kernel of a train() call but the gradient kernel, plus RolloutBuffer.add's row write"""
    from pantheonrl_amd import _native as nat
    scratch = [pol.params.clone(), pol.adam_m.clone(), pol.adam_v.clone(), pol.opt_step.clone()]
    opt = nat.PhOptState()
    opt.params, opt.adam_m, opt.adam_v, opt.step = (t.data_ptr() for t in scratch)
    n_slots = 9
    us = (C.c_float * n_slots)()
    nat.check(pol.ctx.lib.ph_bench_train_kernels(pol.ctx.handle, C.byref(pol.spec), C.byref(opt), C.byref(rb.c_struct()),
                                                 C.byref(hp), int(model.n_epochs), int(model.batch_size), 20, gm, us))
    th.cuda.synchronize()
    N = rb.buffer_size * rb.n_envs
    E, D, A, P = rb.n_envs, lay.D, lay.A, lay.P
    slab_bytes = 4.0 * us[8]
    split = us[1] > 0
    per_elem_w = 32 if split else 8
    alg = {
        "weight_image_kernel": (0, 4.0 * P + 6.0 * 2 * P, "reads the parameters, writes three bf16 planes of (at most) two fragment elements each"),
        "obs_planes_kernel": (1, N * (4.0 * D + 20.0 + 384.0 + 32.0), "per buffer row: reads 4 D + 5 scalars, writes 3 planes x 128 B + a 32 B scalar record"),
        "adv_stats_kernel": (2, model.n_epochs * N * (32.0 + per_elem_w), "per (epoch, position): one random 32 B read of the per-row table, two 16 B records written (the materialised order as well where a kernel reads it)"),
        "ppo_reduce_kernel": (3, slab_bytes + 4.0 * P, "reads every gradient slab of the minibatch once, writes the gradient"),
        "ppo_reduce_kernel (16-byte loads, a learner alone)": (4, slab_bytes + 4.0 * P, "same bytes"),
        "ppo_adam_kernel": (5, 28.0 * P, "gradient, two moments and the parameter in; moments and parameter out"),
        "ppo_step_kernel": (6, slab_bytes + 28.0 * P, "reduce + clip + Adam as one launch"),
        "buffer_add_kernel": (7, 2 * 4.0 * E * (D + A + 3) + 4.0 * E, "RolloutBuffer.add of one step: the row's observations, actions, three scalars in and out, the reward row zeroed"),
    }
    res = {"peak": 8000.0, "unit": "GB/s", "reps": 20, "source": "HIP events on the context's stream, this run (ph_bench_train_kernels)"}
    for name, (slot, nbytes, what) in alg.items():
        if us[slot] <= 0:
            continue
        gbs = nbytes / (us[slot] * 1e-6) / 1e9
        res[name] = {"launch_us": us[slot], "bytes_per_launch": nbytes, "achieved": gbs, "frac": gbs / 8000.0, "bytes": what}
    return res


def roofline(args, agent):
    """dominant kernel = ppo_grad_kernel (gather + forward + loss + backward of one minibatch); MFMA-bound.
    achieved = algorithmic FLOPs per launch (6*M per row, M = forward MACs) / mean launch duration measured
    with HIP events on the kernel's own stream."""
    from pantheonrl_amd import _native as nat
    model = agent.model
    pol, rb = model.policy, model.rollout_buffer
    lay = pol.layout
    hp = model.hyper()
    ms = C.c_float(0)
    pol._bind()
    gm = int(getattr(pol, "gemm_mode", 0))
    nat.check(pol.ctx.lib.ph_bench_ppo_grad(pol.ctx.handle, C.byref(pol.spec), pol.params.data_ptr(),
                                            C.byref(rb.c_struct()), C.byref(hp), int(model.batch_size), 20, gm,
                                            C.byref(ms)))
    macs = 2 * (lay.F * 64 + 64 * 64) + 64 * lay.L + 64
    nb = min(model.batch_size, rb.buffer_size * rb.n_envs)
    flops = 6.0 * macs * nb
    achieved = flops / (ms.value * 1e-3) / 1e12
    small = lay.F <= 64 and lay.A == 1 and lay.L <= 8 and os.environ.get("PH_GRAD_FAST", "1") != "0"
    box = type(pol.observation_space).__name__ == "Box"
    fold = "true" if (lay.F < 64 and box) else "false"
    split = small and box and gm == 2 and os.environ.get("PH_GRAD_SPLIT", "1") != "0"
    kernel = (f"ppo_grad_split_kernel<{lay.L}, {fold}>" if split else
              f"ppo_grad_fast_kernel<false, {lay.L}, {fold}>" if small else "ppo_grad_kernel<64,LP,false>")
    out = {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": 157.3,
           "unit": "TFLOP/s", "frac": achieved / 157.3, "traffic": None, "launch_ms": ms.value,
           "flops_per_launch": flops, "gemm_mode": gm, "frac_basis": "isolated (live HIP events; no committed trace of these sources)",
           "peak_basis": "dense f32 MFMA peak of gfx950 (v_mfma_f32_32x32x2_f32: 157.3 TFLOP/s = the f32 vector rate)"}
    if split:
        # what the matrix pipe actually executes: five 64x64x64 products per tile and net, six bf16 terms each, plus the
        # ones-products of the bias gradients -- against the dense bf16 peak
        tiles = (nb + 63) // 64
        executed = 2.0 * tiles * 4 * 246 * (16 * 16 * 32 * 2)
        out["matrix_pipe"] = {"executed_flops_per_launch": executed, "achieved": executed / (ms.value * 1e-3) / 1e12,
                              "peak": 2500.0, "unit": "TFLOP/s (bf16 dense)",
                              "frac": executed / (ms.value * 1e-3) / 1e12 / 2500.0,
                              "note": "float32 operands as three bf16 planes, six v_mfma_f32_16x16x32_bf16 terms per product, f32 "
                                      "accumulate; error vs a float64 gradient <= the exact-f32 kernel's (tests/test_gpu_parity.py); the "
                                      "kernel is bound by its VALU work (tanh, plane splits, head), not by the matrix pipe"}
    # secondary, HBM-bound: the GAE pass (20 algorithmic bytes per transition) at the bench size (launch-latency bound:
    # 2.6 MB) and at a saturating size (E=16384, T=2048: 671 MB), serial (bit-exact) and scan kernels
    lv = th.zeros(rb.n_envs, device=pol.device)
    gms = C.c_float(0)
    nat.check(pol.ctx.lib.ph_bench_gae(pol.ctx.handle, C.byref(rb.c_struct()), lv.data_ptr(), lv.data_ptr(), 0.99, 0.95,
                                       0, 20, C.byref(gms)))
    gb = 20.0 * rb.buffer_size * rb.n_envs
    out["gae"] = {"bound": "hbm", "achieved": gb / (gms.value * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                  "launch_ms": gms.value, "bytes_per_launch": gb}
    try:
        Tb, Eb = 2048, 16384
        big = nat.PhRollout()
        big.T, big.E = Tb, Eb
        keep = []
        gen = th.Generator(device=pol.device).manual_seed(0)
        for name in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages",
                     "returns"):
            if name in ("observations", "actions", "log_probs"):
                t = th.zeros(1, device=pol.device)
            elif name == "episode_starts":
                t = (th.rand((Tb, Eb), device=pol.device, generator=gen) < 0.0025).float()
            else:
                t = th.randn((Tb, Eb), device=pol.device, generator=gen)
            keep.append(t)
            setattr(big, name, t.data_ptr())
        lvb = th.zeros(Eb, device=pol.device)
        sat = {}
        for mode, label in ((1, "serial"), (2, "scan")):
            nat.check(pol.ctx.lib.ph_bench_gae(pol.ctx.handle, C.byref(big), lvb.data_ptr(), lvb.data_ptr(), 0.99, 0.95,
                                               mode, 5, C.byref(gms)))
            sat[label] = {"GB/s": 20.0 * Tb * Eb / (gms.value * 1e-3) / 1e9, "launch_ms": gms.value}
        out["gae_saturating"] = {"n_envs": Eb, "n_steps": Tb, "bytes_per_launch": 20.0 * Tb * Eb, "peak": 8000.0, **sat}
        del keep
    except Exception as exc:  # noqa: BLE001 -- measurement extra, never fatal
        out["gae_saturating"] = {"error": str(exc)}
    # HBM-side kernels of the update and the rollout buffer's row write (K1 / K3 / K6), each timed live with HIP
    # events around `reps` back-to-back launches on a SCRATCH copy of the optimizer state: algorithmic bytes / launch time / 8 TB/s
    try:
        out["hbm_kernels"] = hbm_kernels(model, pol, rb, lay, hp, gm)
    except Exception as exc:  # noqa: BLE001 -- measurement extra, never fatal
        out["hbm_kernels"] = {"error": str(exc)}
    # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 --pmc, separate runs).  The digest names the
    # hash of the kernel's sources at collection time; counters of another kernel are refused, not reported.
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from pmc_digest import kernel_source_sha256
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_ppo_grad.json")))
        now = kernel_source_sha256(ROOT)
        if kernel.split("<")[0] not in rec.get("kernel", "") or kernel not in rec.get("kernel", "") or nb != 32768:
            out["traffic_source"] = ("no committed PMC passes for this workload's kernel and minibatch size (the digest is of "
                                     f"{rec.get('kernel')} at 32768 rows)")
        elif rec.get("kernel_source_sha256") != now:
            out["traffic_source"] = (f"REFUSED: profiles/pmc_ppo_grad.json was collected on kernel sources {str(rec.get('kernel_source_sha256'))[:12]} "
                                     f"(commit {str(rec.get('git_head'))[:8]}), this tree's hash is {now[:12]}: re-run scripts/profile_round.sh "
                                     "+ scripts/pmc_digest.py")
        else:
            out["traffic"] = rec["hbm_bytes_per_launch"]
            out["traffic_algorithmic"] = rec["algorithmic_bytes_per_launch"]
            out["traffic_source"] = rec["source"] + f" @ {str(rec.get('git_head'))[:8]}"
            ig = rec.get("in_graph")
            if ig:   # the same kernel inside the whole-iteration graphs (committed rocprofv3 kernel trace of this command)
                out["in_graph"] = {"avg_launch_ms": ig["avg_us"] * 1e-3, "achieved": flops / (ig["avg_us"] * 1e-6) / 1e12,
                                   "frac": flops / (ig["avg_us"] * 1e-6) / 1e12 / 157.3, "source": ig["source"]}
                # The headline fraction is the one that governs the iteration: the kernel between the other learner's launches,
                # from the committed kernel trace of this very command on these very sources (hash checked above).  The figure
                # HIP events give for the kernel alone on the device, measured live in this run, stays beside it.
                # The top-level `achieved` / `frac` stay what THIS run measured with HIP events on the kernel's stream; the
                # figure the committed rocprofv3 kernel trace of this command gives for the same sources (hash checked above)
                # sits beside it, with its file.
                out["isolated"] = {"launch_ms": ms.value, "achieved": achieved, "frac": achieved / 157.3,
                                   "source": "HIP events on the kernel's stream, this run (ph_bench_ppo_grad)"}
                # Top level = the figure the committed rocprofv3 kernel trace of THIS tree's kernel gives (the trace's
                # average duration must agree with what the line claims): the lower of the two.  What this run measured live with HIP
                # events around 20 back-to-back launches stays beside it under "isolated".
                out["achieved"], out["frac"] = out["in_graph"]["achieved"], out["in_graph"]["frac"]
                out["launch_ms_trace"] = ig["avg_us"] * 1e-3
                out["frac_basis"] = ("in_graph: the kernel's average duration in the committed rocprofv3 kernel trace of `bench.py --headline-only` on "
                                     "these very sources (hash-checked; the two learners' launches serialised by the profiler); isolated = live HIP "
                                     "events of this run around 20 back-to-back launches")
                if "matrix_pipe" in out:
                    ex = out["matrix_pipe"]["executed_flops_per_launch"]
                    out["matrix_pipe"]["in_graph_frac"] = ex / (ig["avg_us"] * 1e-6) / 1e12 / 2500.0
                    out["matrix_pipe"]["frac_basis"] = "frac: isolated (live); in_graph_frac: the committed trace"
            if "sq" in rec:
                out["mfma_util_percent"] = rec["sq"].get("MfmaUtil_percent")
    except Exception as exc:  # noqa: BLE001
        out["traffic_source"] = f"unavailable ({exc})"
    return out


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run with N ranks on this node."""
    import socket
    import subprocess
    n_dev = th.cuda.device_count() if th.cuda.is_available() else 0
    if args.backend == "nccl" and args.gpus > n_dev:
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"self-launch: {' '.join(cmd)}")
    env = dict(os.environ, PANTHEON_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def bench_roundrobin(args, device, rank, world, json_fd, tdist):
    """BASELINE config 4: ego vs (world - 1) on-policy partners, exactly one partner active per episode and environment,
    one agent per GPU.  value = agent-steps/s of the ego plus the partners (every environment step has one ego action and one
    partner action, so 2 * n_envs * n_steps per iteration, whatever the number of partners)."""
    from pantheonrl_amd import PPO, roundrobin as rr, spaces as sp
    from pantheonrl_amd.vec import SyntheticRollouts
    if world < 2:
        raise SystemExit("bench.py --mode roundrobin needs --gpus >= 2 (ego + at least one partner)")
    wl = WORKLOADS[args.workload]
    if len(wl["act"]) != 1:
        raise SystemExit("bench.py --mode roundrobin runs the single-Discrete-action workloads (overcooked, mpe8, rps)")
    K = world - 1
    obs_space = sp.Box(-np.inf, np.inf, (wl["obs"][1],)) if wl["obs"][0] == "box" else (
        sp.Discrete(wl["obs"][1][0]) if len(wl["obs"][1]) == 1 else sp.MultiDiscrete(wl["obs"][1]))
    act_space = sp.Discrete(wl["act"][0])
    env = type("SpacesOnly", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()
    model = PPO("MlpPolicy", env, n_steps=args.n_steps, n_envs=args.n_envs, batch_size=args.batch_size,
                n_epochs=args.n_epochs, seed=1000 * rank, device=device)
    model.device_permutations = True
    data_ego = SyntheticRollouts(obs_space, args.n_envs, args.n_steps, wl["horizon"], 0, device)
    data_alt = SyntheticRollouts(obs_space, args.n_envs, args.n_steps, wl["horizon"], 1, device) if rank == 0 else None
    side = rr.make_rank(model, K, args.n_steps, data_ego=data_ego, obs_alt=None if data_alt is None else data_alt.obs)

    def barrier():
        th.cuda.synchronize(device)
        tdist.barrier()
        th.cuda.synchronize(device)
    for _ in range(args.warmup):
        side.run_iteration()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        side.run_iteration()
    barrier()
    dt = time.perf_counter() - t0
    tmax = th.tensor([dt], dtype=th.float64, device=device if args.backend == "nccl" else "cpu")
    tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
    dt = float(tmax.item())
    upd = th.tensor([float(getattr(side, "updates", 0))], dtype=th.float64, device=device if args.backend == "nccl" else "cpu")
    tdist.all_reduce(upd)
    # in-kernel waits that timed out anywhere invalidate the run on every rank (a partner's last iteration is only checked at
    # the start of its next one: roundrobin.RoundRobinLink.check)
    tmo = th.tensor([float(side.link.timeouts()) if getattr(side, "native", False) else 0.0], dtype=th.float64,
                    device=device if args.backend == "nccl" else "cpu")
    tdist.all_reduce(tmo)
    if tmo.item() != 0:
        raise SystemExit(f"bench.py: rank {rank}: {int(tmo.item())} round-robin waits timed out across the ranks -- the run is "
                         f"invalid; here: {side.link.timeout_record() if getattr(side, 'native', False) else None}")
    if rank == 0:
        value = 2.0 * args.n_envs * args.n_steps * args.steps / dt
        result = {"metric": f"env-steps/sec (all agents) {args.workload} shapes, ego vs {K} round-robin partners",
                  "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                  "dtype": "f32", "data": "synthetic", "ranks_seen": world,
                  "config": {"workload": wl["name"] + f", ego vs {K} OnPolicy partners, round-robin per environment",
                             "n_envs": args.n_envs, "n_steps": args.n_steps, "batch_size": args.batch_size,
                             "n_epochs": args.n_epochs, "partner_updates": int(upd.item()),
                             "parallelism": f"one agent per gpu x{world} (ego + {K} partners; per step the routing block to "
                                            "the partners and their actions back: " + (
                                                "direct stores into IPC-mapped receive areas + stamps, ONE native call per "
                                                "iteration and rank" if getattr(side, "native", False) else
                                                f"one broadcast + one all-gather of torch.distributed {args.backend} per step") + ")",
                             "carrier": "engine-side" if getattr(side, "native", False) else "torch.distributed",
                             "p2p_timeouts": int(tmo.item()),
                             "launch_mode": "roundrobin"}}
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    tdist.barrier()
    tdist.destroy_process_group()


def bench_liar(args, device, json_fd):
    """BASELINE config 2 as a device-resident game: LiarsDice-v0 PPO-vs-PPO, n_envs tables dealt, played and scored on the GPU
    (`ph_liar_selfplay_rollout`: one launch per rollout; rules pantheonrl/envs/liargym/liar.py:18-19,53-102, bit-exact against the
    reference-generated traces in tests/golden/game_traces.npz), the ego's rectangular buffer and the partner's ragged one (a game may
    end on either player's move), both learners training -- the ego's update from a hipGraph, the partner's beside it on a second
    stream whenever all its columns are full (pantheonrl_amd/envs/vec.py: LiarIterationGraph).  One "step" = one such iteration."""
    from pantheonrl_amd import PPO, _native as nat
    from pantheonrl_amd.envs.vec import LiarIterationGraph, RaggedVecOnPolicyAgent, VecLiarsDice, VecLiarSelfPlay
    from pantheonrl_amd.vec import VecOnPolicyAgent
    E, T = args.n_envs, args.n_steps
    spaces = type("S", (), dict(observation_space=VecLiarsDice.observation_space, action_space=VecLiarsDice.action_space,
                                _is_dummy_space_env=True))()
    models = [PPO("MlpPolicy", spaces, n_steps=T, n_envs=E, batch_size=args.batch_size, n_epochs=args.n_epochs, seed=sd, device=device)
              for sd in (0, 1)]
    for m in models:
        m.device_permutations = True
    ego, alt = VecOnPolicyAgent(models[0]), RaggedVecOnPolicyAgent(models[1])
    sp = VecLiarSelfPlay(E, ego, alt, seed=3, native=True)
    sp.rollout_and_learn(T)                       # launch by launch once: workspaces, first launches
    th.cuda.synchronize(device)
    g = LiarIterationGraph(sp, T)
    for _ in range(max(args.warmup, 1)):
        g.launch()
    th.cuda.synchronize(device)

    def partner_rows():
        return alt.iteration * E * T + int(alt.pos.sum().item())
    rows0, ep0, it0 = partner_rows(), sp.episodes, alt.iteration
    evs = [th.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    th.cuda.synchronize(device)
    t0 = time.perf_counter()
    evs[0].record(g.stream)
    for k in range(args.steps):
        g.launch()
        evs[k + 1].record(g.stream)
    th.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    per = sorted(evs[k].elapsed_time(evs[k + 1]) for k in range(args.steps))
    rows1, ep1 = partner_rows(), sp.episodes
    ego_steps, alt_steps = E * T * args.steps, rows1 - rows0
    result = {
        "metric": "env-steps/sec (all agents) LiarsDice-v0 PPO-vs-PPO self-play", "value": (ego_steps + alt_steps) / dt,
        "unit": "agent-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "ego_steps_per_s": ego_steps / dt, "partner_steps_per_s": alt_steps / dt,
        "episodes_per_iteration": (ep1 - ep0) / args.steps, "partner_updates": alt.iteration - it0,
        "iteration_ms": {"min": per[0], "median": per[len(per) // 2], "max": per[-1], "source": "HIP events on the iteration stream"},
        "config": {"workload": "LiarsDice-v0 PPO-vs-PPO self-play, the game itself on the device (dice Philox-keyed, rules bit-exact "
                               "against reference-generated traces), random-init policies",
                   "n_envs": E, "n_steps": T, "obs_dim": 30, "features": 270, "n_logits": 19, "batch_size": args.batch_size,
                   "n_epochs": args.n_epochs, "agents_per_gpu": 2, "parallelism": "ego + partner on one gpu (single process)",
                   "launch_mode": "persistent rollout launch + ego-update graph || partner update on a second stream",
                   "gemm_mode": int(getattr(models[0].policy, "gemm_mode", 0)), "rollout": "persistent" if sp.persistent else "stepwise",
                   "rollout_requested": "persistent"},
    }
    # the rollout alone (one launch for n_steps vector steps)
    if sp.persistent:
        alt_pos = alt.pos.clone()
        th.cuda.synchronize(device)
        t1 = time.perf_counter()
        sp.rollout_persistent(T, 1, 0)
        th.cuda.synchronize(device)
        result["rollout_ms"] = 1e3 * (time.perf_counter() - t1)
        alt.pos.copy_(alt_pos)
        ego.model.rollout_buffer.pos = 0
    if not args.no_roofline:
        pol, rb, model = models[0].policy, models[0].rollout_buffer, models[0]
        lay, hp, ms = pol.layout, models[0].hyper(), C.c_float(0)
        gm = int(getattr(pol, "gemm_mode", 0))
        rb.pos = T
        pol._bind()
        nat.check(pol.ctx.lib.ph_bench_ppo_grad(pol.ctx.handle, C.byref(pol.spec), pol.params.data_ptr(), C.byref(rb.c_struct()),
                                                C.byref(hp), int(model.batch_size), 20, gm, C.byref(ms)))
        rb.pos = 0
        macs = 2 * (lay.F * 64 + 64 * 64) + 64 * lay.L + 64
        nb = min(model.batch_size, E * T)
        flops = 6.0 * macs * nb
        tiles = (nb + 63) // 64
        # what the matrix pipe executes (hipcc's assembly of ppo_grad_split_oh_kernel<5, 2, false>: 480 v_mfma_f32_16x16x32_bf16 per
        # wave and tile -- X is ONE bf16 plane, so the two first-layer products are three terms per block, the rest six)
        executed = 2.0 * tiles * 4 * 480 * (16 * 16 * 32 * 2)
        result["roofline"] = {
            "bound": "mfma", "kernel": "ppo_grad_split_oh_kernel<5, 2, false>", "achieved": flops / (ms.value * 1e-3) / 1e12,
            "peak": 157.3, "unit": "TFLOP/s", "frac": flops / (ms.value * 1e-3) / 1e12 / 157.3, "traffic": None,
            "launch_ms": ms.value, "flops_per_launch": flops, "gemm_mode": gm,
            "frac_basis": "dense convention (SURVEY.md 8d: 6 M per row with the 270-feature first layer counted as SB3 executes it), "
                          "isolated launches, live HIP events",
            "executed": {"flops_per_launch": executed, "achieved": executed / (ms.value * 1e-3) / 1e12, "peak": 2500.0,
                         "unit": "TFLOP/s (bf16 dense)", "frac": executed / (ms.value * 1e-3) / 1e12 / 2500.0,
                         "note": "the matrix-pipe work the kernel issues: one-hot X as one bf16 plane (three terms per first-layer "
                                 "block), six terms elsewhere"}}
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args)
    os.write(json_fd, (json.dumps(result) + "\n").encode())
    return 0


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={os.environ.get('WORLD_SIZE')} "
                         "(launch with --nproc-per-node equal to --gpus)")
    # RCCL prints a version banner to stdout at communicator creation: keep fd 1 clean for the one JSON line
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.n_envs <= 0:
        args.n_envs = WORKLOADS[args.workload]["n_envs"]
    if args.batch_size <= 0:
        args.batch_size = args.n_envs * args.n_steps // 4
    from pantheonrl_amd import dist as pdist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not th.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    if args.backend != "nccl":
        local_rank = local_rank % th.cuda.device_count()   # several ranks per GPU: functional test only
    elif local_rank >= th.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but only {th.cuda.device_count()} are visible")
    th.cuda.set_device(local_rank)
    device = th.device("cuda", local_rank)
    distributed = pdist.init_from_env(args.backend)
    import torch.distributed as tdist

    if args.mode == "roundrobin":
        return bench_roundrobin(args, device, rank, world, json_fd, tdist)
    if args.workload == "liar" and world == 1 and args.mode == "auto":
        # BASELINE config 2: the game itself runs on the device (an explicit --mode keeps the synthetic-transition path of the shapes)
        return bench_liar(args, device, json_fd)
    from pantheonrl_amd.vec import IterationGraph, run_iteration_eager
    log(f"building {args.agents_per_gpu} agents, n_envs={args.n_envs}, n_steps={args.n_steps}, batch={args.batch_size}")
    agents, datas = build_agents(args, device)
    log("agents built")
    streams = [th.cuda.Stream(device=device) for _ in agents]
    mode = args.mode
    if mode == "auto":
        mode = "fusedstep" if distributed else "graph"
    if distributed:
        mode = "fusedstep"
    exchange = pdist.ActionExchange(len(agents), args.n_envs, device) if (distributed or mode == "fusedstep") else None
    if exchange is not None:
        # per-step all-gather route: auto = time engine-side RCCL against direct peer-to-peer stores on this node, keep the faster
        exchange.requested_route = os.environ.get("PANTHEON_EXCHANGE", "auto")

    rollout_requested = args.rollout
    if mode == "graph":
        if args.rollout == "scripted":
            lay = agents[0].model.policy.layout
            if not (lay.F <= 64 and lay.A == 1 and lay.L <= 8 and args.n_envs < 16384):
                # the one-launch rollout exists for the 16-row forward's shape class only (ph_scripted_rollout refuses others)
                log(f"workload {args.workload}: shapes outside the one-launch rollout's class -> --rollout stepwise")
                args.rollout = "stepwise"
        graphs = [IterationGraph(a, d, s, scripted=args.rollout == "scripted") for a, d, s in zip(agents, datas, streams)]

        def iteration():
            for g in graphs:
                g.launch()
    elif mode == "jointgraph":
        # all local learners in ONE hipGraph per iteration; the update is the chained joint call (ph_ppo_train_multi)
        from pantheonrl_amd.vec import JointIterationGraph
        joint = JointIterationGraph(agents, datas, streams)

        def iteration():
            joint.launch()
    elif mode == "eager":
        def iteration():
            for a, d, s in zip(agents, datas, streams):
                with th.cuda.stream(s):
                    run_iteration_eager(a, d)
    else:
        # agent-per-GPU layout: every environment step all-gathers the actions of all seats (RCCL over xGMI) between
        # the policy forwards and the reward updates; the (synthetic) transition consumes the joint action: a shared
        # coordination bonus when a seat's action equals its round-robin partner's (Overcooked's reward is shared).
        # One fused launch (all local agents' forwards + the previous step's joint-action reward) and one collective per
        # step (vec.FusedSelfPlayRollout); per-step hipGraphs measured 2x slower than direct launches here.
        from pantheonrl_amd.vec import FusedSelfPlayRollout
        th.cuda.set_stream(streams[0])
        masks = None
        if args.action_masks != "none":
            L = agents[0].model.policy.layout.L
            masks = []
            for i in range(len(agents)):
                rng = np.random.default_rng(7000 + 10 * rank + i)
                mk = (rng.random((args.n_steps, args.n_envs, L)) < 0.8).astype(np.uint8)
                dead = mk.sum(-1) == 0
                mk[dead, rng.integers(0, L, size=int(dead.sum()))] = 1      # at least one legal action
                masks.append(th.as_tensor(mk).to(device))
        steps = FusedSelfPlayRollout(agents, datas, exchange, streams[0], masks=masks,
                                     mask_mode=2 if args.action_masks == "env" else 1)
        it_counter = [0]

        def iteration():
            steps.run_iteration(it_counter[0])
            it_counter[0] += 1

    def barrier():
        th.cuda.synchronize(device)
        if distributed:
            tdist.barrier()
            th.cuda.synchronize(device)

    # per-iteration durations from events on every agent stream (the engine's launches go to these very streams), recorded
    # inside the timed region: a few host microseconds each, no synchronisation
    marks = [[th.cuda.Event(enable_timing=True) for _ in streams] for _ in range(args.steps + 1)]

    def mark(i):
        for ev, st in zip(marks[i], streams if mode != "fusedstep" else [streams[0]] * len(streams)):
            ev.record(st)

    log(f"mode={mode}; warmup x{args.warmup}")
    for _ in range(args.warmup):
        iteration()
    barrier()
    log(f"timing x{args.steps}")
    t0 = time.perf_counter()
    mark(0)
    for i in range(args.steps):
        iteration()
        mark(i + 1)
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed region {dt:.3f}s")
    per_iter = [max(marks[i][k].elapsed_time(marks[i + 1][k]) for k in range(len(streams))) for i in range(args.steps)]
    ranks_seen, devices_seen = 1, 1
    if distributed:
        tmax = th.tensor([dt], dtype=th.float64, device=device)
        tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
        dt = float(tmax.item())
        one = th.ones(1, dtype=th.float64, device=device)
        tdist.all_reduce(one)                        # communicator size as the collective itself sees it
        ranks_seen = int(one.item())
        ids = [None] * world
        tdist.all_gather_object(ids, str(th.cuda.get_device_properties(device).uuid) if hasattr(
            th.cuda.get_device_properties(device), "uuid") else f"{os.uname().nodename}:{local_rank}")
        devices_seen = len(set(ids))

    steps_per_iter = args.n_envs * args.n_steps * len(agents) * world
    value = steps_per_iter * args.steps / dt
    result = {
        "metric": "env-steps/sec (all agents) Overcooked-simple PPO self-play" if args.workload == "overcooked"
                  else f"env-steps/sec (all agents) {args.workload} shapes",
        "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "ranks_seen": ranks_seen, "devices_seen": devices_seen,
        "iteration_ms": {"min": float(np.min(per_iter)), "median": float(np.median(per_iter)),
                         "max": float(np.max(per_iter)), "source": "HIP events on the agents' streams, rank 0"},
        "config": {"workload": WORKLOADS[args.workload]["name"] + (
                       " (two independent PPO learners per GPU)" if len(agents) == 2 else
                       f" ({len(agents)} independent PPO learner(s) per GPU)") + ", synthetic (n_envs, n_steps, obs_dim) rollouts",
                   "n_envs": args.n_envs, "n_steps": args.n_steps, "obs_dim": agents[0].model.policy.layout.D,
                   "features": agents[0].model.policy.layout.F, "n_logits": agents[0].model.policy.layout.L,
                   "batch_size": args.batch_size, "n_epochs": args.n_epochs, "agents_per_gpu": len(agents),
                   "parallelism": f"{len(agents)} agent(s) per gpu x{world} gpu(s) = {len(agents) * world} learners (" + (
                       f"per-step action all-gather over xGMI, route {getattr(exchange, 'route', '?')}" if distributed
                       else "single process") + ")",
                   "exchange": ({"route": exchange.route, **exchange.route_log, "p2p_timeouts": exchange.p2p_timeouts()}
                                if exchange is not None and hasattr(exchange, "route") else None),
                   "launch_mode": mode,
                   # arithmetic of the update's 64x64 products: see include/pantheon_hip.h (gemm_mode) -- 2 = float32 operands as three
                   # bf16 planes, six matrix-pipe terms per product, f32 accumulate (float32 accuracy, measured against float64 in
                   # tests/test_gpu_parity.py); 0 = exact-f32 MFMA (PH_GEMM_MODE=0 python bench.py reproduces that line)
                   "gemm_mode": int(getattr(agents[0].model.policy, "gemm_mode", 0)),
                   "action_masks": args.action_masks if mode == "fusedstep" else "none",
                   # how the n_steps steps of a rollout are launched: "scripted" / "persistent" = ONE launch per rollout (N = 1
                   # graph mode; the exchange layouts with the per-step action hand-off done in-kernel), "stepwise" / "p2p" = one
                   # launch per environment step
                   "rollout": (args.rollout if mode == "graph" else getattr(steps, "last_rollout_mode", "stepwise")
                               if mode == "fusedstep" else "stepwise"),
                   # what --rollout asked for: differs from "rollout" when the workload's shapes are outside the one-launch
                   # rollout's class (F <= 64, one action component, <= 8 logits) and the run fell back to one launch per step
                   "rollout_requested": rollout_requested,
                   # Weak-scaling efficiency is value(N) / (N * value(1)) of the DEFAULT invocations: N = 1 runs the scripted
                   # one-launch rollout without any exchange, N > 1 the persistent one-launch exchange rollout, so both ends of the
                   # ratio launch a rollout once; `stepwise_rollout` is the base to use if a run fell back to one launch per step.
                   "efficiency_base": "python bench.py (N=1, launch_mode graph, rollout scripted); per-GPU work is fixed "
                                      "(agents_per_gpu learners x n_envs x n_steps per iteration)"},
    }
    sweep_errors = sum(a.model.policy.ctx.step_errors() for a in agents)
    if sweep_errors:
        raise SystemExit(f"bench.py: rank {rank}: {sweep_errors} waits of the one-launch optimizer step expired -- the run is invalid")
    if exchange is not None and hasattr(exchange, "route") and exchange.p2p_timeouts() != 0:
        raise SystemExit(f"bench.py: rank {rank}: {exchange.p2p_timeouts()} peer-to-peer polls timed out -- the run is invalid; "
                         f"first: {exchange.p2p_timeout_record()}")
    if distributed and ranks_seen != args.gpus:
        raise SystemExit(f"bench.py: the collective saw {ranks_seen} ranks, --gpus is {args.gpus}")
    if mode == "graph" and args.rollout == "scripted" and not distributed:
        # How the iteration divides, measured without a profiler: the learners' one-launch rollouts alone (side by side, as in the
        # iteration), timed over 10 repetitions; the rest of the iteration is the update -- GAE, the gradient pack and n_epochs x
        # minibatches of (gradient launch, reduce, Adam) per learner -- whose gradient launches take turns on the device, so
        # update time / gradient launches is what ONE gradient launch costs the iteration, everything else of the update included.
        try:
            for a, d, st in zip(agents, datas, streams):
                with th.cuda.stream(st):
                    a.bind_stream()
                    a.rollout_scripted(d)
                    a.finish_update()
            barrier()
            kr = 10
            tr = time.perf_counter()
            for _ in range(kr):
                for a, d, st in zip(agents, datas, streams):
                    with th.cuda.stream(st):
                        a.bind_stream()
                        a.rollout_scripted(d)
                        a.finish_update()
            barrier()
            rollout_ms = 1e3 * (time.perf_counter() - tr) / kr
            n_mb = args.n_epochs * ((args.n_envs * args.n_steps + args.batch_size - 1) // args.batch_size) * len(agents)
            upd_ms = 1e3 * dt / args.steps - rollout_ms
            result["iteration_split"] = {
                "rollout_ms": rollout_ms, "update_ms": upd_ms, "gradient_launches_per_iteration": n_mb,
                "us_per_gradient_launch_in_iteration": 1e3 * upd_ms / n_mb,
                "note": "rollouts of all learners side by side, timed alone (10 repetitions); update = iteration - rollouts (GAE, gradient "
                        "pack, every minibatch's gradient launch + reduce + Adam); no profiler involved -- compare with roofline.isolated "
                        "(the gradient kernel alone) and roofline.in_graph (the same kernel under rocprofv3, which serialises the learners)"}
        except Exception as exc:  # noqa: BLE001 -- an extra figure, never fatal
            result["iteration_split"] = {"error": str(exc)}
    if mode == "graph" and args.rollout == "scripted" and not args.headline_only:
        # the same iteration with one launch per environment step, timed in the same process on the same agents: what the
        # scripted rollout saves is launch boundaries, nothing else (the two walks are bitwise equal)
        sgraphs = [IterationGraph(a, d, s) for a, d, s in zip(agents, datas, streams)]
        for _ in range(2):
            for g in sgraphs:
                g.launch()
        barrier()
        k2 = max(1, min(args.steps, 10))
        t1 = time.perf_counter()
        for _ in range(k2):
            for g in sgraphs:
                g.launch()
        barrier()
        dt2 = time.perf_counter() - t1
        result["stepwise_rollout"] = {"value": steps_per_iter * k2 / dt2, "unit": "agent-steps/s", "ms_per_step": 1e3 * dt2 / k2,
                                      "steps": k2, "note": "one launch per environment step inside the iteration graphs "
                                      "(--rollout stepwise): what an environment that lives on the host or on another node forces; the N>1 layouts of this bench "
                                              "hand actions over in-kernel and launch a rollout once, like the default"}
    if mode == "graph" and world == 1 and not args.headline_only:
        # The same iteration when the boundary hands over HOST buffers (the owning-handle ABI's situation): every learner's rollout
        # inputs -- observations, rewards, dones of the n_steps x n_envs transitions -- cross PCIe from pinned host memory before its
        # graph runs, its sampled actions cross back after.  Never `value`: that is measured with the inputs resident in HBM.
        try:
            hosts = [(d.obs.cpu().pin_memory(), d.rewards.cpu().pin_memory(), d.dones.cpu().pin_memory()) for d in datas]
            acts_host = [th.empty_like(a.model.rollout_buffer.actions, device="cpu").pin_memory() for a in agents]

            def host_iteration():
                for a, d, st, h in zip(agents, datas, streams, hosts):
                    with th.cuda.stream(st):
                        d.obs.copy_(h[0], non_blocking=True)
                        d.rewards.copy_(h[1], non_blocking=True)
                        d.dones.copy_(h[2], non_blocking=True)
                iteration()
                for a, st, ah in zip(agents, streams, acts_host):
                    with th.cuda.stream(st):
                        ah.copy_(a.model.rollout_buffer.actions, non_blocking=True)
            for _ in range(2):
                host_iteration()
            barrier()
            kh = max(1, min(args.steps, 10))
            t1 = time.perf_counter()
            for _ in range(kh):
                host_iteration()
            barrier()
            dth = time.perf_counter() - t1
            h2d = sum(sum(t.numel() * t.element_size() for t in h) for h in hosts)
            d2h = sum(t.numel() * t.element_size() for t in acts_host)
            result["host_buffers_pcie_inclusive"] = {
                "value": steps_per_iter * kh / dth, "unit": "agent-steps/s", "ms_per_step": 1e3 * dth / kh, "steps": kh,
                "h2d_bytes_per_step": h2d, "d2h_bytes_per_step": d2h,
                "note": "the timed iteration with every learner's rollout inputs copied from pinned host memory first and its actions "
                        "copied back after (async copies on the learner's stream): the rate when the caller owns host arrays; not `value`"}
        except Exception as exc:  # noqa: BLE001 -- an extra figure, never fatal
            result["host_buffers_pcie_inclusive"] = {"error": str(exc)}
    if mode == "graph" and world == 1 and args.agents_per_gpu != 1 and args.rollout in ("scripted", "stepwise") and not args.headline_only:
        # north_star's literal layout -- ONE learner per GPU ("8 agents on 8 GPUs") -- on this GPU: the N = 1 base a scaling curve of
        # `--agents-per-gpu 1` runs is measured against.  Nothing overlaps the learner's reduce / Adam then, so each minibatch's
        # reduce + clip + Adam runs as one launch (ppo_step_kernel) between two gradient launches.
        try:
            one_args = argparse.Namespace(**{**vars(args), "agents_per_gpu": 1})
            a1, d1 = build_agents(one_args, device)
            s1 = th.cuda.Stream(device=device)
            g1 = IterationGraph(a1[0], d1[0], s1, scripted=args.rollout == "scripted")
            for _ in range(2):
                g1.launch()
            barrier()
            k1 = max(1, min(args.steps, 20))
            t1 = time.perf_counter()
            for _ in range(k1):
                g1.launch()
            barrier()
            dt1 = time.perf_counter() - t1
            result["one_agent_per_gpu"] = {"value": args.n_envs * args.n_steps * k1 / dt1, "unit": "agent-steps/s",
                                           "ms_per_step": 1e3 * dt1 / k1, "steps": k1,
                                           "note": "python bench.py --agents-per-gpu 1: one learner alone on the device (exclusive "
                                                   "hint: reduce + clip + Adam of a minibatch as one launch)"}
        except Exception as exc:  # noqa: BLE001 -- an extra figure, never fatal
            result["one_agent_per_gpu"] = {"error": str(exc)}
    if rank == 0:
        if not args.no_roofline:
            result["roofline"] = roofline(args, agents[0])
            log("roofline measured")
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(args)
            result["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
            try:
                result["gpu_reference_semantics_E1"] = gpu_reference_semantics_e1(args, device)
            except Exception as exc:  # noqa: BLE001 -- an extra figure, never fatal
                result["gpu_reference_semantics_E1"] = {"error": str(exc)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if distributed:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
