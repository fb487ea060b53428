#!/bin/bash
# on the GPU box: python bench.py --headline-only for every built variant (pantheonrl_amd/csrc/variants/<name>.so), interleaved twice:
# value, ms per iteration, the rollout / update split and the in-iteration cost of one gradient launch
cd "$(dirname "$0")/.."
V=pantheonrl_amd/csrc/variants
for rep in 1 2; do
  for so in "$@"; do
    PANTHEON_HIP_LIB=$PWD/$V/$so.so python bench.py --steps 20 --warmup 3 --headline-only --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d.get('iteration_split', {})
print('$so: %.2f M agent-steps/s, %.3f ms/iter (min %.3f) | rollout %.3f ms | update %.3f ms = %.2f us per gradient launch' % (
    d['value'] / 1e6, d['ms_per_step'], d['iteration_ms']['min'], s.get('rollout_ms', 0), s.get('update_ms', 0), s.get('us_per_gradient_launch_in_iteration', 0)))"
  done
done
