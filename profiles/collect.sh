#!/bin/bash
# Collects the rocprofv3 evidence quoted in DESIGN.md / bench.py on a GPU box (run from the repo root through gpurun):
#   gpurun --timeout 900 -- 'bash profiles/collect.sh'
# Outputs land in gpurun_out/prof/ (scratch); the summaries are then copied into profiles/ (tracked).
# Passes are separate, as MI355X_MICROARCH.md prescribes: kernel trace + stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; SQ counters.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-secondary"   # (without the 4-stream leg: its overlapped kernels would enter the per-kernel averages)
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- $BENCH > $OUT/stats.log 2>&1 || echo "stats pass failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p --output-format csv -- $BENCH > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p --output-format csv -- $BENCH > $OUT/write.log 2>&1 || echo "write pass failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d $OUT/sq1 -o p --output-format csv -- $BENCH > $OUT/sq1.log 2>&1 || echo "sq1 pass failed"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq2 -o p --output-format csv -- $BENCH > $OUT/sq2.log 2>&1 || echo "sq2 pass failed"
rocprofv3 --kernel-trace --stats -d $OUT/c2 -o s --output-format csv -- python3 $R/profiles/c2_run.py 20 > $OUT/c2.log 2>&1 || echo "c2 pass failed"
C2="python3 $R/profiles/c2_run.py 6"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d $OUT/c2sq1 -o p --output-format csv -- $C2 > $OUT/c2sq1.log 2>&1 || echo "c2 sq1 pass failed"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $OUT/c2sq2 -o p --output-format csv -- $C2 > $OUT/c2sq2.log 2>&1 || echo "c2 sq2 pass failed"
cd $R
python3 profiles/pmc_summary.py k_superpose_uniform4 $OUT/c2sq1/p_counter_collection.csv $OUT/c2sq2/p_counter_collection.csv > $OUT/c2_pmc_sq_uniform.txt
cp $OUT/stats/s_kernel_stats.csv $OUT/kernel_stats.csv
cp $OUT/c2/s_kernel_stats.csv $OUT/c2_kernel_stats.csv
grep "^{" $OUT/c2.log | tail -1 > $OUT/c2_run.json
python3 profiles/pmc_summary.py $OUT/fetch/p_counter_collection.csv $OUT/write/p_counter_collection.csv > $OUT/pmc_fetch_write_kb.txt
python3 profiles/pmc_summary.py k_superpose_sweep $OUT/sq1/p_counter_collection.csv $OUT/sq2/p_counter_collection.csv > $OUT/pmc_sq_superpose.txt
RTD_NO_SWEEP=1 rocprofv3 --kernel-trace --stats -d $OUT/nosweep -o s --output-format csv -- $BENCH > $OUT/nosweep.log 2>&1 || echo "no-sweep pass failed"
cp $OUT/nosweep/s_kernel_stats.csv $OUT/nosweep_kernel_stats.csv
grep "^{" $OUT/stats.log | tail -1 > $OUT/bench_under_rocprof.json
python3 profiles/pmc_summary.py k_fill $OUT/sq1/p_counter_collection.csv $OUT/sq2/p_counter_collection.csv > $OUT/pmc_sq_fill.txt
python3 profiles/pmc_summary.py --traffic-json $OUT/fetch/p_counter_collection.csv $OUT/write/p_counter_collection.csv $OUT/sq1/p_counter_collection.csv $OUT/sq2/p_counter_collection.csv > $OUT/traffic.json
tail -1 $OUT/stats.log | cut -c1-400
