"""GAE scan at the saturating size of SURVEY.md 8(d) (E = 16384, T = 2048: 671 MB of algorithmic traffic): us per launch and TB/s.
Same-box A/B of kernel variants: PANTHEON_HIP_LIB=<other build> (scripts/build_variants.sh); PH_GAE_LC / PH_GAE_NCH override the shape."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as th
from pantheonrl_amd import _native as nat, spaces as sp
from pantheonrl_amd.ppo import ActorCriticPolicy
pol = ActorCriticPolicy(sp.Box(-1, 1, (2,)), sp.Discrete(2))
Eb, Tb = int(os.environ.get("E", 16384)), int(os.environ.get("T", 2048))
big = nat.PhRollout()
big.T, big.E = Tb, Eb
keep = []
gen = th.Generator(device=pol.device).manual_seed(0)
for name in ("observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns"):
    if name in ("observations", "actions", "log_probs"):
        t = th.zeros(1, device=pol.device)
    elif name == "episode_starts":
        t = (th.rand((Tb, Eb), device=pol.device, generator=gen) < 0.0025).float()
    else:
        t = th.randn((Tb, Eb), device=pol.device, generator=gen)
    keep.append(t)
    setattr(big, name, t.data_ptr())
lvb = th.zeros(Eb, device=pol.device)
gms = C.c_float(0)
pol._bind()
for rep in range(3):
    nat.check(pol.ctx.lib.ph_bench_gae(pol.ctx.handle, C.byref(big), lvb.data_ptr(), lvb.data_ptr(), 0.99, 0.95, 2, 10, C.byref(gms)))
print(os.environ.get("PANTHEON_HIP_LIB", "default").split("/")[-1], "gae scan E=%d T=%d: %.1f us = %.2f TB/s" % (Eb, Tb, gms.value * 1e3, 20.0 * Tb * Eb / (gms.value * 1e-3) / 1e12))
