"""Build libpantheon_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["ph_abi.hip", "ph_policy.hip", "ph_gae.hip", "ph_ppo.hip", "ph_ppo_fast.hip", "ph_ppo_split.hip", "ph_envs.hip", "ph_agent.hip", "ph_bc.hip", "ph_adap.hip", "ph_modular.hip", "ph_adapmult.hip"]
HEADERS = ["ph_device.h", "ph_launch.h", "ph_liar.h", "ph_head.h", "ph_split.h", "ph_rowtail.h", os.path.join(ROOT, "include", "pantheon_hip.h")]
LIB = os.path.join(HERE, "libpantheon_hip.so")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the gfx950 engine)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "include"), "-I", HERE, "-Wall", "-Wno-unused-function",
           "-o", LIB] + [os.path.join(HERE, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print("[pantheonrl_amd] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
