"""Build libpantheon_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per translation unit under csrc/build/ (git-ignored), compiled in parallel and only when the unit or a header it
includes -- found by following its `#include "…"` lines, so a new header can not be forgotten -- is newer than its object."""
from __future__ import annotations

import os
import re
import shlex
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
INCLUDE = os.path.join(ROOT, "include")
SOURCES = ["ph_abi.hip", "ph_policy.hip", "ph_gae.hip", "ph_ppo.hip", "ph_ppo_fast.hip", "ph_ppo_split.hip", "ph_ppo_split_oh.hip", "ph_envs.hip", "ph_agent.hip", "ph_bc.hip", "ph_adap.hip", "ph_modular.hip", "ph_adapmult.hip"]
LIB = os.path.join(HERE, "libpantheon_hip.so")
OBJDIR = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", HERE, "-Wall", "-Wno-unused-function"]

_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the gfx950 engine)")


def dependencies(path: str, seen: set | None = None) -> set:
    """`path` and every project header reachable from it through `#include "…"` (searched beside the file and in include/)."""
    seen = set() if seen is None else seen
    path = os.path.abspath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for name in _INC.findall(open(path).read()):
        for base in (os.path.dirname(path), HERE, INCLUDE):
            cand = os.path.join(base, name)
            if os.path.exists(cand):
                dependencies(cand, seen)
                break
    return seen


def _object(src: str) -> str:
    return os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")


def _stale(src: str, extra: str) -> bool:
    obj = _object(src)
    if not os.path.exists(obj):
        return True
    stamp = obj + ".flags"
    if not os.path.exists(stamp) or open(stamp).read() != extra:
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in dependencies(os.path.join(HERE, src)))


def needs_build() -> bool:
    extra = os.environ.get("PH_EXTRA_HIPCC_FLAGS", "")
    return not os.path.exists(LIB) or any(_stale(s, extra) or os.path.getmtime(_object(s)) > os.path.getmtime(LIB) for s in SOURCES)


def build(force: bool = False, verbose: bool = True, lib: str | None = None, extra_flags: str | None = None) -> str:
    """Compile what is stale (everything with `force`) and link.  `lib` / `extra_flags` build an A/B variant elsewhere
    (objects of a variant live in their own directory)."""
    extra = os.environ.get("PH_EXTRA_HIPCC_FLAGS", "") if extra_flags is None else extra_flags
    out = LIB if lib is None else lib
    global OBJDIR
    saved = OBJDIR
    if lib is not None:
        OBJDIR = os.path.join(HERE, "build", "variant_" + os.path.splitext(os.path.basename(lib))[0])
    try:
        if not force and os.path.exists(out) and not any(_stale(s, extra) or os.path.getmtime(_object(s)) > os.path.getmtime(out) for s in SOURCES):
            return out
        os.makedirs(OBJDIR, exist_ok=True)
        hipcc = _hipcc()
        todo = [s for s in SOURCES if force or _stale(s, extra)]

        def compile_one(src: str) -> None:
            cmd = [hipcc] + FLAGS + shlex.split(extra) + ["-c", os.path.join(HERE, src), "-o", _object(src)]
            if verbose:
                print("[pantheonrl_amd] " + " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(_object(src) + ".flags", "w") as f:
                f.write(extra)

        with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as pool:
            list(pool.map(compile_one, todo))
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + [_object(s) for s in SOURCES] + ["-ldl"]
        if verbose:
            print("[pantheonrl_amd] " + " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return out
    finally:
        OBJDIR = saved


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
