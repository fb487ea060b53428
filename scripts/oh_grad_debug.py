"""gemm_mode 2 vs autograd on the one-hot shapes, per parameter block: where does ppo_grad_split_oh_kernel disagree?
usage: oh_grad_debug.py [name T E nb]..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import sb3_oracle as orc
from tests import test_gpu_parity as G
cases = [("liar", 16, 6, 77), ("discrete20", 16, 8, 64)]
for name, T, E, nb in cases:
    idx = np.random.default_rng(nb).permutation(T * E)[:nb]
    for mode in (2, 0):
        g, g_ref, st, st_ref, lay = G._grad_pair(name, T, E, idx, orc.PPOHyper(), gemm_mode=mode)
        blocks = [("pi_W1", lay.pi_W1, lay.pi_b1), ("pi_b1", lay.pi_b1, lay.pi_W2), ("pi_W2", lay.pi_W2, lay.pi_b2), ("pi_b2", lay.pi_b2, lay.vf_W1),
                  ("vf_W1", lay.vf_W1, lay.vf_b1), ("vf_b1", lay.vf_b1, lay.vf_W2), ("vf_W2", lay.vf_W2, lay.vf_b2), ("vf_b2", lay.vf_b2, lay.act_W),
                  ("act_W", lay.act_W, lay.act_b), ("act_b", lay.act_b, lay.val_W), ("val_W", lay.val_W, lay.val_b), ("val_b", lay.val_b, lay.P)]
        scale = np.abs(g_ref).max()
        line = " ".join(f"{n}:{np.abs(g[a:b] - g_ref[a:b]).max() / scale:.1e}" + ("!" if not np.isfinite(g[a:b]).all() else "") for n, a, b in blocks)
        print(f"{name} nb={nb} mode={mode} scale={scale:.2e} | {line}")
        if mode == 2:
            for nm, a0 in (("pi_W2", lay.pi_W2), ("vf_W2", lay.vf_W2)):
                G2, R2 = g[a0:a0 + 4096].reshape(64, 64), g_ref[a0:a0 + 4096].reshape(64, 64)
                er = np.abs(G2 - R2)
                print(f"   {nm}: max|g| {np.abs(G2).max():.3e} max|ref| {np.abs(R2).max():.3e} err {er.max():.3e} errT {np.abs(G2.T - R2).max():.3e} "
                      f"sorted-set err {np.abs(np.sort(G2.ravel()) - np.sort(R2.ravel())).max():.3e} corr {np.corrcoef(G2.ravel(), R2.ravel())[0, 1]:.4f}")
                print("     err by 16x16 block (in-unit block rows, out-unit block cols):")
                print("     " + str((er.reshape(4, 16, 4, 16).max(axis=(1, 3)) / max(np.abs(R2).max(), 1e-30)).round(3).tolist()))
                rows_bad = (er.max(axis=1) > 1e-4 * np.abs(R2).max()).nonzero()[0]
                cols_bad = (er.max(axis=0) > 1e-4 * np.abs(R2).max()).nonzero()[0]
                print("     bad in-units:", rows_bad.tolist()[:70], " bad out-units:", cols_bad.tolist()[:70])
                ratio = G2[np.abs(R2) > 0.1 * np.abs(R2).max()] / R2[np.abs(R2) > 0.1 * np.abs(R2).max()]
                print("     ratio g/ref on large entries: min %.3f median %.3f max %.3f" % (ratio.min(), np.median(ratio), ratio.max()))
        print("   stats", " ".join(f"{st[i]:.5f}/{st_ref[k]:.5f}" for i, k in enumerate(("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss"))))
