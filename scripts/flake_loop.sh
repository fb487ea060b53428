#!/bin/bash
# on the GPU box: the 8-rank config-5 bench invocation of tests/test_gpu_zz_bench.py, N times per rollout form, every run's exit
# code, wall time and (on failure) stderr kept: scripts/flake_loop.sh [N=10] [budget_seconds=600] -> gpurun_out/flake/
cd "$(dirname "$0")/.."
N=${1:-10}; BUDGET=${2:-600}
OUT=gpurun_out/flake; mkdir -p $OUT
START=$(date +%s)
export PANTHEON_EXCHANGE=p2p
for i in $(seq 1 $N); do
  for envs in 96 1024; do
    NOW=$(date +%s); if [ $((NOW - START)) -gt $BUDGET ]; then echo "budget spent after run $i" | tee -a $OUT/summary.txt; exit 0; fi
    t0=$(date +%s%N)
    timeout 300 python bench.py --gpus 8 --workload mpe8 --agents-per-gpu 1 --n-envs $envs --n-steps 16 --n-epochs 2 --steps 2 \
      --warmup 1 --action-masks env --backend gloo --no-roofline > $OUT/run_${envs}_$i.out 2> $OUT/run_${envs}_$i.err
    rc=$?
    t1=$(date +%s%N)
    echo "envs=$envs run=$i rc=$rc ms=$(( (t1 - t0) / 1000000 )) rollout=$(grep -o '"rollout": "[a-z0-9]*"' $OUT/run_${envs}_$i.out | head -1)" | tee -a $OUT/summary.txt
    if [ $rc -eq 0 ]; then rm -f $OUT/run_${envs}_$i.err $OUT/run_${envs}_$i.out; else cp $OUT/run_${envs}_$i.err $OUT/fail_${envs}_$i.err; rm -f $OUT/run_${envs}_$i.err; fi
  done
done
