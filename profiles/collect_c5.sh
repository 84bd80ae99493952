#!/bin/bash
# C5 (768^3 CT, voxel 1/3 mm; one field of the eight-angle plan per GPU) on ONE GPU: bench line, rocprofv3 kernel stats and the counters
# SURVEY.md 8(d) names for C5 (HBM read / write bytes, LDS bank conflicts, VALU busy) for the field at 0 degrees.
#   gpurun --timeout 900 -- 'bash profiles/collect_c5.sh'      -> gpurun_out/prof_c5/ (then copied to profiles/r03_c5_*)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --size 768 --steps 10 --warmup 3 --no-cpu --no-secondary"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- $BENCH > $OUT/stats.log 2>&1 || echo "stats pass failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p --output-format csv -- $BENCH > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p --output-format csv -- $BENCH > $OUT/write.log 2>&1 || echo "write pass failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/sq -o p --output-format csv -- $BENCH > $OUT/sq.log 2>&1 || echo "sq pass failed"
cd $R
cp $OUT/stats/s_kernel_stats.csv $OUT/kernel_stats.csv
grep "^{" $OUT/stats.log | tail -1 > $OUT/bench.json
python3 profiles/pmc_summary.py $OUT/fetch/p_counter_collection.csv $OUT/write/p_counter_collection.csv > $OUT/pmc_fetch_write_kb.txt
python3 profiles/pmc_summary.py k_superpose_sweep $OUT/sq/p_counter_collection.csv > $OUT/pmc_sq_superpose.txt
python3 profiles/pmc_summary.py k_fill $OUT/sq/p_counter_collection.csv > $OUT/pmc_sq_fill.txt
tail -1 $OUT/bench.json | cut -c1-300
