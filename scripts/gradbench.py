"""ph_bench_ppo_grad timing at the bench size (overcooked): prints us per launch.  Same-box A/B of kernel variants:
PANTHEON_HIP_LIB=<other build> (scripts/build_variants.sh) or the PH_GRAD_FAST switch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as th
from pantheonrl_amd import PPO, _native as nat, spaces as sp
from pantheonrl_amd.vec import SyntheticRollouts, VecOnPolicyAgent, run_iteration_eager
E, T = 1024, 128
obs_space, act_space = sp.Box(-np.inf, np.inf, (62,)), sp.Discrete(6)
env = type("S", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()
model = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=E * T // 4, n_epochs=1, seed=0)
model.device_permutations = True
agent = VecOnPolicyAgent(model)
data = SyntheticRollouts(obs_space, E, T, 400, 0, model.device)
run_iteration_eager(agent, data)
th.cuda.synchronize()
pol, rb = model.policy, model.rollout_buffer
hp = model.hyper(); ms = C.c_float(0); rb.pos = T
for mode in (int(m) for m in os.environ.get("GRADBENCH_MODES", "0,2").split(",")):   # 0 = f32 MFMA kernel, 2 = split-bf16 kernel
    for rep in range(3):
        nat.check(pol.ctx.lib.ph_bench_ppo_grad(pol.ctx.handle, C.byref(pol.spec), pol.params.data_ptr(), C.byref(rb.c_struct()),
                                                C.byref(hp), int(model.batch_size), 200, mode, C.byref(ms)))
    print(os.environ.get("PANTHEON_HIP_LIB", "default").split("/")[-1], "gemm_mode", mode, "us/launch %.2f" % (ms.value * 1e3))
