mkdir -p gpurun_out/r04_f; O=gpurun_out/r04_f
REPS=3 bash scripts/ab_bench.sh vec pre > $O/ab_stats_block_pretrain.txt 2>&1
(timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adap.py tests/test_gpu_modular.py tests/test_gpu_handle_abi.py -q -x -p no:cacheprovider 2>&1 | tail -15) > $O/tests.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --no-cpu-baseline --no-roofline --headline-only > $O/bench_headline_under_rocprof.json 2> $O/kt.log
python scripts/rocprof_summary.py $(find $O/kt -name "*_results.db" | head -1) > $O/headline_kernel_stats.txt 2>&1
python scripts/trace_gaps.py $(find $O/kt -name "*_results.db" | head -1) > $O/headline_trace_gaps.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt -- python bench.py --no-cpu-baseline --no-roofline --headline-only --agents-per-gpu 1 > $O/bench_one_agent_under_rocprof.json 2> $O/kt1.log
python scripts/rocprof_summary.py $(find $O/kt1 -name "*_results.db" | head -1) > $O/one_agent_kernel_stats.txt 2>&1
python scripts/trace_gaps.py $(find $O/kt1 -name "*_results.db" | head -1) > $O/one_agent_trace_gaps.txt 2>&1
rm -rf $O/kt $O/kt1
python scripts/modular_speed.py > $O/modular_speed.txt 2>&1
python scripts/liar_speed.py > $O/liar_speed.txt 2>&1
cat $O/ab_stats_block_pretrain.txt; tail -3 $O/tests.txt
