#!/usr/bin/env python
"""Builds profiles/pmc_ppo_grad.json -- the digest bench.py reports as roofline.traffic / in_graph / mfma_util_percent -- from a
round's rocprofv3 summaries (scripts/profile_round.sh -> scripts/rocprof_summary.py):

    python scripts/pmc_digest.py gpurun_out/r03_x profiles/r03_x        # <dir with pmc_{f,w,m,g}.txt + bench_graph_kernel_stats.txt> <tag>

The digest carries `kernel_source_sha256`: the hash of the sources the dominant kernel is compiled from, at the time the
counters were collected.  bench.py recomputes it from the tree it runs in and REFUSES the committed counters on a mismatch
(they describe another kernel), reporting traffic = null and the reason instead of stale numbers."""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("pantheonrl_amd/csrc/ph_ppo_split.hip", "pantheonrl_amd/csrc/ph_split_tile.h", "pantheonrl_amd/csrc/ph_split.h",
                  "pantheonrl_amd/csrc/ph_head.h", "pantheonrl_amd/csrc/ph_device.h", "pantheonrl_amd/csrc/ph_launch.h:struct GradArgs")


def kernel_source_sha256(root: str = ROOT) -> str:
    """hash of what ppo_grad_split_kernel (the default gradient kernel of the bench workload, gemm_mode 2) is compiled from: its translation unit, the two headers with its device code, and its
    argument record (the rest of ph_launch.h -- other kernels' records and launcher prototypes -- does not enter)"""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        path, _, part = rel.partition(":")
        with open(os.path.join(root, path), "rb") as fh:
            data = fh.read()
        if part:
            text = data.decode()
            i = text.index(part + " {")
            data = text[i:text.index("\n};", i) + 3].encode()
        h.update(rel.encode() + b"\0" + data)
    return h.hexdigest()


def _counters(path, kernel_substr):
    out = {}
    if not os.path.exists(path):
        return out
    seen_pmc = False
    for line in open(path):
        if line.startswith("# PMC counters"):
            seen_pmc = True
            continue
        if not seen_pmc or kernel_substr not in line:
            continue
        m = re.match(r"^(.{60}) (\S+)\s+(\d+)\s+([\d.eE+-]+)\s+([\d.eE+-]+)\s*$", line.rstrip("\n"))
        if m:
            out[m.group(2)] = {"n": int(m.group(3)), "sum": float(m.group(4)), "mean": float(m.group(5))}
    return out


def _kernel_row(path, kernel_substr):
    if not os.path.exists(path):
        return None
    for line in open(path):
        if kernel_substr in line and not line.startswith("#"):
            parts = line.rstrip("\n")[78:].split()
            if len(parts) >= 6:
                return {"name": line[:78].strip(), "calls": int(parts[0]), "avg_us": float(parts[2]), "min_us": float(parts[3]),
                        "max_us": float(parts[4])}
    return None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    K = "ppo_grad_split_kernel"
    f, w = _counters(os.path.join(src, "pmc_f.txt"), K), _counters(os.path.join(src, "pmc_w.txt"), K)
    m, g = _counters(os.path.join(src, "pmc_m.txt"), K), _counters(os.path.join(src, "pmc_g.txt"), K)
    cal_f = _counters(os.path.join(src, "pmc_f.txt"), "gae_serial_kernel")
    cal_w = _counters(os.path.join(src, "pmc_w.txt"), "gae_serial_kernel")
    known_r, known_w = 16384 * 2048 * 12, 16384 * 2048 * 8
    fc = known_r / (cal_f["FETCH_SIZE"]["mean"] * 1024) if cal_f else 2.0
    wc = known_w / (cal_w["WRITE_SIZE"]["mean"] * 1024) if cal_w else 1.0
    fetch_kb, write_kb = f["FETCH_SIZE"]["mean"], w["WRITE_SIZE"]["mean"]
    row_iso = _kernel_row(os.path.join(src, "pmc_m.txt"), K)
    row_graph = _kernel_row(os.path.join(src, "bench_graph_kernel_stats.txt"), K)
    nb, D, A, M = 32768, 62, 1, 2 * (62 * 64 + 64 * 64) + 64 * 6 + 64
    waves = 2 * 256 * 4
    sq = {k: v["mean"] for k, v in {**m, **g}.items()}
    digest = {
        "kernel": (re.search(r"ph::" + K + r"<[^>]*>", row_iso["name"]).group(0) if row_iso else "ph::" + K),
        "round": tag,
        "kernel_source_sha256": kernel_source_sha256(),
        "kernel_sources": list(KERNEL_SOURCES),
        "git_head": subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
        "shape": "batch 32768 rows, D=62, L=6 (bench config), 256 workgroups x 2 nets, 2 row tiles per workgroup",
        "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
        "fetch_correction": round(fc, 4), "write_correction": round(wc, 4),
        "calibration": "same PMC passes, gae_serial_kernel at E=16384, T=2048 (exactly 402,653,184 B read / 268,435,456 B written): "
                       f"FETCH_SIZE x {fc:.4f}, WRITE_SIZE x {wc:.4f} give the known byte counts (MI355X_MICROARCH.md HBM section: "
                       "FETCH_SIZE reports half of a wide coalesced read on gfx950)",
        "hbm_bytes_per_launch": int(fetch_kb * 1024 * fc + write_kb * 1024 * wc),
        "algorithmic_bytes_per_launch": nb * 4 * (D + A + 4),
        "in_graph": None if not row_graph else {"avg_us": row_graph["avg_us"], "min_us": row_graph["min_us"],
                                                "max_us": row_graph["max_us"], "calls": row_graph["calls"],
                                                "source": f"{tag}_bench_graph_kernel_stats.txt (rocprofv3 --kernel-trace --stats -- "
                                                          "python bench.py --no-cpu-baseline --no-roofline --headline-only: the launches of the two-learner headline only; the profiler serialises the two learners' kernels)"},
        "isolated_under_pmc": row_iso,
        "sq": {**sq, "waves_per_launch": waves},
        "flops_per_launch_6M_convention": 6.0 * M * nb,
        "source": f"{tag}_pmc_f.txt + _w.txt + _m.txt + _g.txt (rocprofv3 --kernel-trace --pmc ... in separate passes over "
                  "scripts/pmc_workload.py; scripts/profile_round.sh; digest: scripts/pmc_digest.py)",
    }
    if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "GRBM_GUI_ACTIVE" in sq:
        # busy cycles are summed over the 4 SIMDs of 256 CUs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        digest["sq"]["MfmaUtil_percent"] = round(100.0 * sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * sq["GRBM_GUI_ACTIVE"] / 8), 1)
    if "SQ_WAVE_CYCLES" in sq:
        digest["sq"]["wave_lifetime_cycles"] = round(4 * sq["SQ_WAVE_CYCLES"] / waves)
        if "SQ_WAIT_INST_ANY" in sq:
            digest["sq"]["SQ_WAIT_INST_ANY_over_SQ_WAVE_CYCLES"] = round(sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"], 3)
    for k in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS"):
        if k in sq:
            digest["sq"][k + "_per_wave"] = round(sq[k] / waves, 1)
    json.dump(digest, open(os.path.join(ROOT, "profiles", "pmc_ppo_grad.json"), "w"), indent=1)
    print(json.dumps(digest, indent=1))


if __name__ == "__main__":
    main()
