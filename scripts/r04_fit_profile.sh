#!/bin/bash
# Round 4: the GP rebuild per size -- rocprofv3 kernel trace of rebuilds at n = 4096 ONLY and at n = 2048 ONLY, one PMC
# pass of the step kernel (matrix-pipe busy cycles against the active cycles), and the acquisition lock-step (kernel trace
# + FETCH_SIZE of its four launches).   usage (repo root): bash scripts/r04_fit_profile.sh <outdir under gpurun_out>
set -u
OUT=${1:-gpurun_out/r4fit}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
for n in 4096 2048; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/t_$n -o fit -- python $R/scripts/fit_once.py $n 10 0 20 > $R/$OUT/fit_$n.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/p_fit -o pmc -- python $R/scripts/fit_once.py 4096 10 0 5 > $R/$OUT/pmc_fit.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/t_lcb -o lcb -- python $R/scripts/lcb_loop.py 4096 10 > $R/$OUT/lcb.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/p_lcb -o pmc -- python $R/scripts/lcb_loop.py 4096 10 > $R/$OUT/pmc_lcb.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/p_lcb2 -o pmc -- python $R/scripts/lcb_loop.py 4096 10 > $R/$OUT/pmc_lcb2.log 2>&1
cd $R
for n in 4096 2048; do
  for f in $(find $OUT/t_$n -name "*.db"); do python scripts/rocprof_summary.py $f "scripts/fit_once.py $n 10 0 20 (21 rebuilds at n = $n, nothing else)" > $OUT/fit_${n}_trace.md; done
  grep "per rebuild" $OUT/fit_$n.log
done
for f in $(find $OUT/t_lcb -name "*.db"); do python scripts/rocprof_summary.py $f "scripts/lcb_loop.py 4096 10 (200 lock-steps of 10 points at n = 4096, d = 10)" > $OUT/lcb_trace.md; done
f=$(find $OUT/p_fit -name "*counter_collection.csv" | head -1); python scripts/pmc_summary.py $f step_kernel > $OUT/pmc_fit.txt 2>&1
f=$(find $OUT/p_lcb -name "*counter_collection.csv" | head -1); python scripts/pmc_summary.py $f > $OUT/pmc_lcb.txt 2>&1
f=$(find $OUT/p_lcb2 -name "*counter_collection.csv" | head -1); python scripts/pmc_summary.py $f >> $OUT/pmc_lcb.txt 2>&1
rm -rf $OUT/t_4096 $OUT/t_2048 $OUT/t_lcb $OUT/p_fit $OUT/p_lcb $OUT/p_lcb2
head -30 $OUT/fit_4096_trace.md; cat $OUT/pmc_fit.txt; head -16 $OUT/lcb_trace.md; cat $OUT/pmc_lcb.txt
