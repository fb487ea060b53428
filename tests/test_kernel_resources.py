"""Register hygiene of the bench path's kernels, checked where they are compiled (no GPU): hipcc's own resource-usage remarks for
gfx950.  The gradient kernels must not touch scratch, and the split gradient kernel's register count is part of the iteration's
schedule: two of its waves plus one wave of the OTHER learner's reduce / Adam kernels have to fit into the 512
registers of a SIMD lane, or every reduce block displaces a gradient workgroup for its whole life (measured: -13 % on the bench)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pantheonrl_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")


def _usage(source, tmp_path_factory):
    out = tmp_path_factory.mktemp("res") / "x.o"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c",
                        os.path.join(CSRC, source), "-o", str(out), "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name:\s+(\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|LDS Size \[bytes/block\]):\s+(\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return kernels


@pytest.fixture(scope="module")
def split(tmp_path_factory):
    return _usage("ph_ppo_split.hip", tmp_path_factory)


@pytest.fixture(scope="module")
def ppo(tmp_path_factory):
    return _usage("ph_ppo.hip", tmp_path_factory)


def _alloc(k):     # registers a wave of the kernel occupies: VGPRs + AGPRs in the unified file, in granules of 8
    return (k["VGPRs"] + k.get("AGPRs", 0) + 7) // 8 * 8


def test_split_gradient_kernel_has_no_scratch_and_leaves_room_for_the_update_kernels(split, ppo):
    inst = {n: k for n, k in split.items() if "ppo_grad_split_kernel" in n}
    assert len(inst) == 16, sorted(inst)                      # NK = 1..8 x FOLD
    for n, k in inst.items():
        assert k["ScratchSize"] == 0 and k["VGPRs Spill"] == 0, (n, k)
        assert _alloc(k) <= 240, (n, k)
    bench = inst["_ZN2ph21ppo_grad_split_kernelILi6ELb1EEEvNS_8GradArgsE"]    # Overcooked: 6 logits, bias folded
    reduce_k = ppo["_ZN2ph17ppo_reduce_kernelILi2EEEvNS_10ReduceArgsE"]     # the 8-byte-load shape learners that share a device launch
    adam_k = ppo["_ZN2ph15ppo_adam_kernelENS_8AdamArgsE"]
    grad_lds = 3 * 3 * 64 * 128 + 4 * (8 * (64 + 8) + 64 * 8 + 2 * 64 + 16 + 3 * 64 + 64)   # grad_split_lds_bytes(): 79 680 B, dynamic
    for other in (reduce_k, adam_k):
        assert other["ScratchSize"] == 0
        assert 2 * _alloc(bench) + _alloc(other) <= 512, (bench, other)
        # ... and its static LDS into what two gradient workgroups leave of the CU's 160 KB (a 5 KB reduce block once cost 13 %)
        assert 2 * grad_lds + other["LDS Size"] <= 160 * 1024, (grad_lds, other)


def test_one_hot_gradient_kernel_leaves_a_wave_of_the_update_kernels_on_every_simd(ppo, tmp_path_factory):
    """ppo_grad_split_oh_kernel holds ONE workgroup per CU (its LDS), i.e. one wave per SIMD: the other learner's reduce / Adam
    launches overlap its gradient launch only while one of THEIR waves still fits into the SIMD's 512 registers beside it -- all five
    chunks of W1 fragments in flight took Liar's Dice's instantiation to 499 registers and the iteration from 4.0 to 4.2 ms
    (profiles/r05_z_liar_grad_w1_slots_ab.txt).  No scratch in any one-hot instantiation."""
    oh = _usage("ph_ppo_split_oh.hip", tmp_path_factory)
    inst = {n: k for n, k in oh.items() if "ppo_grad_split_oh_kernel" in n}
    assert len(inst) == 18, sorted(inst)                      # NCH = 1..5 x LB = 1, 2 one-hot, NCH = 1..4 x LB Box
    reduce_k = ppo["_ZN2ph17ppo_reduce_kernelILi2EEEvNS_10ReduceArgsE"]
    adam_k = ppo["_ZN2ph15ppo_adam_kernelENS_8AdamArgsE"]
    for n, k in inst.items():
        assert k["ScratchSize"] == 0, (n, k)
        if n.endswith("Lb0EEEvNS_8GradArgsE"):               # the one-hot form (config 2's is <5, 2, false>)
            assert k["VGPRs Spill"] == 0, (n, k)
            for other in (reduce_k, adam_k):
                assert _alloc(k) + _alloc(other) <= 512, (n, k, other)


def test_general_and_fast_gradient_kernels_have_no_scratch(ppo, tmp_path_factory):
    fast = _usage("ph_ppo_fast.hip", tmp_path_factory)
    for n, k in list(fast.items()) + [(n, k) for n, k in ppo.items() if "ppo_grad_kernel" in n]:
        if "ppo_grad" not in n:
            continue
        # some general-kernel instantiations reserve an SGPR-scavenging slot (<= 24 bytes of private segment) without a single
        # scratch instruction; a spilled VGPR is what this guards against
        assert k["VGPRs Spill"] == 0 and k["ScratchSize"] <= 24, (n, k)


def test_update_step_kernels_have_no_one_load_per_iteration_loops(tmp_path_factory):
    """Round 4's lesson as a guard: `for (...) v += p[i]` compiles to load, s_waitcnt vmcnt(0), add, branch -- one memory round trip
    per iteration -- and sixteen of those made block 0 of ppo_reduce_kernel the long pole of its launch (CHANGELOG.md).  The reduce,
    step and Adam kernels of the update must not contain an inner loop that waits for ALL of at most two loads before it branches
    back (scripts/serial_load_loops.py reads hipcc's assembly)."""
    import sys
    out = tmp_path_factory.mktemp("asm") / "ppo.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-S",
                        "--cuda-device-only", os.path.join(CSRC, "ph_ppo.hip"), "-o", str(out)], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    scan = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "serial_load_loops.py"), str(out), "2"], capture_output=True,
                          text=True, timeout=300)
    assert scan.returncode == 0, scan.stderr[-2000:]
    hits = [l for l in scan.stdout.splitlines() if any(k in l for k in ("ppo_reduce_kernel", "ppo_step_kernel", "ppo_adam_kernel",
                                                                         "adv_stats_kernel", "obs_planes_kernel"))]
    assert not hits, "\n".join(hits)


def test_rollout_kernels_have_no_scratch(tmp_path_factory):
    """The persistent rollout kernels hold their state for thousands of dependent steps: a spilled register is a scratch round trip
    per step.  (The Liar's Dice rollout once kept 31 spilled registers: per-lane mirror addresses hoisted out of its loop.)"""
    pol = _usage("ph_policy.hip", tmp_path_factory)
    want = ["liar_rollout_kernel", "policy_fwd16_rollout_kernelILb0ELb0E", "policy_fwd16_rollout_kernelILb0ELb1E", "policy_fwd16_exchange_rollout_kernel",
            "policy_fwd16_multi_kernel", "policy_fwd16_kernelILb0ELb0E", "policy_fwd16_kernelILb0ELb1E", "policy_fwd16h_kernelILb0E"]
    for w in want:
        hit = {n: k for n, k in pol.items() if w in n}
        assert hit, (w, sorted(pol))
        for n, k in hit.items():
            if "exchange_rollout" in w:
                # built for THREE waves per SIMD since round 5 (168 registers: 512 resident workgroups on the chip instead of 256, the
                # N > 1 layout's residency margin): eight registers -- pointer pairs the row tails reload -- live in scratch, measured
                # at no cost (same-box A/B, profiles/r05_r_exchange_launch_bounds_ab.txt).  More than that would be a regression.
                assert _alloc(k) <= 168 and k["VGPRs Spill"] <= 8 and k["ScratchSize"] <= 64, (n, k)
                continue
            assert k["ScratchSize"] == 0 and k["VGPRs Spill"] == 0, (n, k)

