#!/bin/bash
# A/B builds of libpantheon_hip.so with extra -D flags: scripts/build_variants.sh name1:"-DFOO -DBAR" name2:"" ...
# -> pantheonrl_amd/csrc/variants/<name>.so (git-ignored, travels to the GPU box); select with PANTHEON_HIP_LIB=<path>
set -e
cd "$(dirname "$0")/../pantheonrl_amd/csrc"
mkdir -p variants
SRCS="ph_abi.hip ph_policy.hip ph_gae.hip ph_ppo.hip ph_ppo_fast.hip ph_ppo_split.hip ph_envs.hip ph_agent.hip ph_bc.hip ph_adap.hip ph_modular.hip ph_adapmult.hip"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I ../../include -I . -Wall -Wno-unused-function $flags \
      -o variants/$name.so $SRCS -ldl > variants/$name.log 2>&1 && echo "built $name [$flags]" || (echo "FAILED $name"; tail -5 variants/$name.log) ) &
done
wait
