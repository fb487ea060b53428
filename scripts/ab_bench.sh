#!/bin/bash
# on the GPU box: the bench line (headline, stepwise, one agent per GPU, isolated gradient launch) for every built variant
# (scripts/build_variants.sh, pantheonrl_amd/csrc/variants/<name>.so), interleaved REPS times:  scripts/ab_bench.sh name1 name2 ...
cd "$(dirname "$0")/.."
V=pantheonrl_amd/csrc/variants
for rep in $(seq 1 ${REPS:-2}); do
  for so in "$@"; do
    PANTHEON_HIP_LIB=$PWD/$V/$so.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
one, sw, r = d.get('one_agent_per_gpu', {}), d.get('stepwise_rollout', {}), d['roofline']
iso = r.get('isolated', r)
print('$so: %.2f M, %.3f ms/iter (min %.3f) | stepwise %.2f M | one agent %.2f M, %.3f ms | grad isolated %.2f us' % (
    d['value'] / 1e6, d['ms_per_step'], d['iteration_ms']['min'], sw.get('value', 0) / 1e6, one.get('value', 0) / 1e6,
    one.get('ms_per_step', 0), iso['launch_ms'] * 1e3))"
  done
done
