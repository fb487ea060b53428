#!/usr/bin/env python
"""Per-phase shader-clock breakdown of one behavioural-cloning step (bc_train_kernel built with -DPH_BC_PROF:
scripts/build_variants.sh bcprof:"-DPH_BC_PROF"; run with PANTHEON_HIP_LIB=.../variants/bcprof.so)."""
import os, sys, numpy as np, torch as th
sys.path.insert(0, os.getcwd())
from pantheonrl_amd.bc import BC
from pantheonrl_amd.common import TransitionsMinimal
from pantheonrl_amd.spaces import Box, Discrete
N, D, L = 32768, 62, 6
rng = np.random.default_rng(0)
clone = BC(Box(-np.inf, np.inf, (D,)), Discrete(L), expert_data=TransitionsMinimal(rng.standard_normal((N, D)).astype(np.float32), rng.integers(0, L, N).astype(np.float32)))
st = clone.train(n_epochs=2)
names = ["P0 rows -> LDS, next gather issued", "P1 layer1", "P2 layer2", "P3 logits", "P4 loss", "P5 dZ2", "P6 dZ1",
         "P7 weight-gradient tiles + biases", "P8 stats", "Adam"]
if os.environ.get("PH_BC_MFMA", "1") == "0":
    names = ["adam(prev) + rows + gather", "layer1", "layer2", "logits", "loss", "dZ2", "dZ1", "owner W1/W2", "owner rest", "stats"]
for n, v in zip(names, st.reshape(-1)[:10]): print(f"{n:<14} {v:9.0f} cycles/step")
print("sum", st.reshape(-1)[:10].sum())
