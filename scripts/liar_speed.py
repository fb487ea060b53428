import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from pantheonrl_amd.trainer import run
t0 = time.perf_counter()
ego, partners, env = run(["LiarsDice-v0", "PPO", "PPO", "--n-envs", "256", "-t", str(256 * 128 * 6), "--seed", "1"])
th.cuda.synchronize()
dt = time.perf_counter() - t0
print("liar on-device self-play: %.2f s for %d ego steps -> %.0f ego steps/s; episodes %d; partner updates %d" % (
    dt, 256 * 128 * 6, 256 * 128 * 6 / dt, env.episodes, partners[0].iteration))
