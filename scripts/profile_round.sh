#!/bin/bash
# On the GPU box: the round's profiles -- bench line, rocprofv3 kernel trace of the same command, three PMC passes over
# scripts/pmc_workload.py (counters in their own runs, kernel-trace only beside them).  Output under gpurun_out/$1/.
set -u
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_stdout.json 2> $OUT/bench.log
# the kernel trace holds the headline's launches only (two learners per GPU): the extra figures of the bench line (stepwise rollout,
# one learner alone) would mix their launches into the same kernel names
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline --no-roofline --headline-only > $OUT/bench_under_rocprof.json 2> $OUT/kt.log
python scripts/rocprof_summary.py $(find $OUT/kt -name "*_results.db" | head -1) > $OUT/bench_graph_kernel_stats.txt 2>&1
python scripts/trace_gaps.py $(find $OUT/kt -name "*_results.db" | head -1) > $OUT/bench_graph_trace_gaps.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt1 -o kt -- python bench.py --no-cpu-baseline --no-roofline --headline-only --agents-per-gpu 1 > $OUT/bench_one_agent_under_rocprof.json 2> $OUT/kt1.log
python scripts/rocprof_summary.py $(find $OUT/kt1 -name "*_results.db" | head -1) > $OUT/one_agent_kernel_stats.txt 2>&1
for pass in "f:FETCH_SIZE" "w:WRITE_SIZE" "m:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "g:GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  PMC_GEMM_MODES=2 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/pmc_$name -o p -- python scripts/pmc_workload.py > $OUT/pmc_$name.log 2>&1
  python scripts/rocprof_summary.py $(find $OUT/pmc_$name -name "*_results.db" | head -1) > $OUT/pmc_$name.txt 2>&1
done
rm -rf $OUT/kt $OUT/kt1 $OUT/pmc_f $OUT/pmc_w $OUT/pmc_m $OUT/pmc_g
ls -la $OUT
