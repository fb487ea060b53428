#!/bin/bash
# A/B builds of libpantheon_hip.so with extra -D flags: scripts/build_variants.sh name1:"-DFOO -DBAR" name2:"" ...
# -> pantheonrl_amd/csrc/variants/<name>.so (git-ignored, travels to the GPU box); select with PANTHEON_HIP_LIB=<path>.
# One variant after the other: each one's translation units already compile in parallel (csrc/build.py).
set -e
cd "$(dirname "$0")/.."
mkdir -p pantheonrl_amd/csrc/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  python - "$name" "$flags" <<'PY'
import sys
from pantheonrl_amd.csrc import build as b
name, flags = sys.argv[1], sys.argv[2]
b.build(lib=b.HERE + "/variants/" + name + ".so", extra_flags=flags, verbose=False)
print("built", name, "[" + flags + "]")
PY
done
