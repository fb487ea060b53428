set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcs
mkdir -p $OUT
for pass in "m:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "g:GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/pmc_$name -o p -- python scripts/pmc_workload.py > $OUT/pmc_$name.log 2>&1
  python scripts/rocprof_summary.py $(find $OUT/pmc_$name -name "*_results.db" | head -1) > $OUT/pmc_$name.txt 2>&1
  rm -rf $OUT/pmc_$name
done
grep -A12 "PMC counters" $OUT/pmc_m.txt | grep "ppo_grad" ; grep "ppo_grad" $OUT/pmc_g.txt
