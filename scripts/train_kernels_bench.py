"""ph_bench_train_kernels at the bench size (overcooked) and ph_bench_gae at the saturating size: prints us per launch of every
kernel of a train() call but the gradient kernel.  Same-box A/B: PANTHEON_HIP_LIB=<other build> (scripts/build_variants.sh),
PH_GAE_LC=8|16|32 (steps per lane of the GAE scan)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as th
from pantheonrl_amd import PPO, _native as nat, spaces as sp
from pantheonrl_amd.vec import SyntheticRollouts, VecOnPolicyAgent, run_iteration_eager
E, T = 1024, 128
obs_space, act_space = sp.Box(-np.inf, np.inf, (62,)), sp.Discrete(6)
env = type("S", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()
model = PPO("MlpPolicy", env, n_steps=T, n_envs=E, batch_size=E * T // 4, n_epochs=10, seed=0)
model.device_permutations = True
agent = VecOnPolicyAgent(model)
data = SyntheticRollouts(obs_space, E, T, 400, 0, model.device)
run_iteration_eager(agent, data)
th.cuda.synchronize()
pol, rb = model.policy, model.rollout_buffer
hp = model.hyper(); rb.pos = T
tag = os.environ.get("PANTHEON_HIP_LIB", "default").split("/")[-1]
names = ["weight_image", "obs_planes", "adv_stats", "reduce", "reduce_wide", "adam", "step_fused", "buffer_add", "slab_floats"]
if os.environ.get("TKB_SKIP_TRAIN", "0") != "1":
    scratch = [pol.params.clone(), pol.adam_m.clone(), pol.adam_v.clone(), pol.opt_step.clone()]
    opt = nat.PhOptState()
    opt.params, opt.adam_m, opt.adam_v, opt.step = (t.data_ptr() for t in scratch)
    us = (C.c_float * 9)()
    for rep in range(2):
        nat.check(pol.ctx.lib.ph_bench_train_kernels(pol.ctx.handle, C.byref(pol.spec), C.byref(opt), C.byref(rb.c_struct()),
                                                     C.byref(hp), 10, int(model.batch_size), 50, int(pol.gemm_mode), us))
    print(tag, " ".join(f"{n} {us[i]:.2f}" for i, n in enumerate(names)))
# GAE at the saturating size
Tb, Eb = 2048, 16384
big = nat.PhRollout(); big.T, big.E = Tb, Eb
keep = []
gen = th.Generator(device=pol.device).manual_seed(0)
for name in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns"):
    if name in ("observations", "actions", "log_probs"):
        t = th.zeros(1, device=pol.device)
    elif name == "episode_starts":
        t = (th.rand((Tb, Eb), device=pol.device, generator=gen) < 0.0025).float()
    else:
        t = th.randn((Tb, Eb), device=pol.device, generator=gen)
    keep.append(t); setattr(big, name, t.data_ptr())
lvb = th.zeros(Eb, device=pol.device)
gms = C.c_float(0)
for rep in range(2):
    nat.check(pol.ctx.lib.ph_bench_gae(pol.ctx.handle, C.byref(big), lvb.data_ptr(), lvb.data_ptr(), 0.99, 0.95, 2, 10, C.byref(gms)))
print(tag, "PH_GAE_LC", os.environ.get("PH_GAE_LC", "-"), "gae scan E=16384 T=2048: %.1f us = %.2f TB/s" % (gms.value * 1e3, 20.0 * Tb * Eb / (gms.value * 1e-3) / 1e12))
