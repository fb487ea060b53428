#!/bin/bash
# On the GPU box, for the round's last tree: the whole GPU suite, the default bench line, and the rocprofv3 kernel trace of the
# headline (no counter passes: the gradient kernel's digest stays valid while its source hash does).  Output under gpurun_out/$1/.
set -u
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1
grep -E "passed|failed|error" $OUT/gpu_tests.log | tail -2
timeout 150 python bench.py > $OUT/bench_stdout.json 2> $OUT/bench.log
tail -c 300 $OUT/bench_stdout.json | head -c 0
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline --no-roofline --headline-only > $OUT/bench_under_rocprof.json 2> $OUT/kt.log
python scripts/rocprof_summary.py $(find $OUT/kt -name "*_results.db" | head -1) > $OUT/bench_graph_kernel_stats.txt 2>&1
rm -rf $OUT/kt
python -c "
import json
j=json.loads([l for l in open('$OUT/bench_stdout.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('iteration_split',{}).get('rollout_ms'), j.get('stepwise_rollout',{}).get('value'), j.get('one_agent_per_gpu',{}).get('value'), j.get('gpu_reference_semantics_E1',{}).get('value'))"
head -12 $OUT/bench_graph_kernel_stats.txt
