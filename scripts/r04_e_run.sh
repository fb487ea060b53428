mkdir -p gpurun_out/r04_e; O=gpurun_out/r04_e
REPS=3 bash scripts/ab_bench.sh base vec > $O/ab_reduce.txt 2>&1
(timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adap.py tests/test_gpu_modular.py -q -x -p no:cacheprovider 2>&1 | tail -15) > $O/tests.txt
python scripts/rollout_phase.py > $O/rollout_phase.txt 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --no-cpu-baseline --no-roofline --headline-only > $O/bench_headline_under_rocprof.json 2> $O/kt.log
python scripts/rocprof_summary.py $(find $O/kt -name "*_results.db" | head -1) > $O/headline_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt -- python bench.py --no-cpu-baseline --no-roofline --headline-only --agents-per-gpu 1 > $O/bench_one_agent_under_rocprof.json 2> $O/kt1.log
python scripts/rocprof_summary.py $(find $O/kt1 -name "*_results.db" | head -1) > $O/one_agent_kernel_stats.txt 2>&1
rm -rf $O/kt $O/kt1
cat $O/ab_reduce.txt; tail -3 $O/tests.txt
