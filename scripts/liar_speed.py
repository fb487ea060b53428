"""steady-state throughput of the device-resident Liar's Dice self-play (BASELINE config 2: PPO-vs-PPO, n_envs = 256)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

from pantheonrl_amd import PPO  # noqa: E402
from pantheonrl_amd.envs.vec import RaggedVecOnPolicyAgent, VecLiarsDice, VecLiarSelfPlay  # noqa: E402
from pantheonrl_amd.vec import VecOnPolicyAgent  # noqa: E402

E, T = 256, 128
native = os.environ.get("LIAR_NATIVE", "1") != "0"
spaces = type("S", (), dict(observation_space=VecLiarsDice.observation_space, action_space=VecLiarsDice.action_space,
                            _is_dummy_space_env=True))()
models = [PPO("MlpPolicy", spaces, n_steps=T, n_envs=E, batch_size=E * T // 4, n_epochs=10, seed=s) for s in (0, 1)]
for m in models:
    m.device_permutations = True
ego, alt = VecOnPolicyAgent(models[0]), RaggedVecOnPolicyAgent(models[1])
sp = VecLiarSelfPlay(E, ego, alt, seed=3, native=native)
sp.rollout_and_learn(T)
th.cuda.synchronize()
iters = 5
t0 = time.perf_counter()
for _ in range(iters):
    sp.rollout_and_learn(T)
th.cuda.synchronize()
dt = time.perf_counter() - t0
t1 = time.perf_counter()
for _ in range(T):
    sp.step()
th.cuda.synchronize()
dr = time.perf_counter() - t1
sp.ego.learn_from_buffer()
if native and sp.persistent:
    alt.pos.zero_()
    th.cuda.synchronize()
    t3 = time.perf_counter()
    sp.rollout_persistent(T, 1, 0)
    th.cuda.synchronize()
    dp = time.perf_counter() - t3
    print(f"persistent rollout (one launch for {T} steps): {dp * 1e3:.2f} ms = {dp / T * 1e6:.1f} us per vector step")
    alt.pos.zero_()
if native and os.environ.get("LIAR_GRAPH", "1") != "0":
    from pantheonrl_amd.envs.vec import LiarIterationGraph  # noqa: E402
    g = LiarIterationGraph(sp, T)
    for _ in range(2):
        g.launch()
    th.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(10):
        g.launch()
    th.cuda.synchronize()
    dg = (time.perf_counter() - t2) / 10
    print(f"iteration (persistent rollout launch, ego update graph || partner update on a second stream): {dg * 1e3:.1f} ms each -> "
          f"{E * T / dg:,.0f} ego steps/s")
print(f"native={native}: {iters} iterations (rollout + updates) {dt / iters * 1e3:.1f} ms each -> {E * T * iters / dt:,.0f} ego "
      f"steps/s; rollout alone {dr / T * 1e6:.0f} us per vector step; episodes {sp.episodes}, partner updates {alt.iteration}")
