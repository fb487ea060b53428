mkdir -p gpurun_out/r04_g; O=gpurun_out/r04_g
for rep in 1 2 3; do
  REPS=1 bash scripts/ab_bench.sh pre hoist
  PH_ROLLOUT_SPREAD=1 REPS=1 bash scripts/ab_bench.sh hoist | sed 's/^hoist/hoist+spread/'
done > $O/ab_hoist_spread.txt 2>&1
(timeout 400 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "train or gradient or fused or rollout or scripted" 2>&1 | tail -8) > $O/tests.txt
(PH_ROLLOUT_SPREAD=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "rollout or scripted" 2>&1 | tail -8) > $O/tests_spread.txt
cat $O/ab_hoist_spread.txt; grep -h "passed\|failed" $O/tests.txt $O/tests_spread.txt
