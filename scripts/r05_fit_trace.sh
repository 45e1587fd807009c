#!/bin/bash
# Round 5: kernel traces of the GP rebuild, the three-launch step (schedule 2, the default) against the merged step (schedule 4:
# panel solve + diagonal tile in one launch), at n = 4096 and n = 1024.   usage: bash scripts/r05_fit_trace.sh <outdir>
set -u
OUT=${1:-gpurun_out/r5fit}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
for n in 4096 1024; do
  for s in 2 4; do
    timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/t_${n}_$s -o fit -- python $R/scripts/fit_once.py $n 10 $s 20 > $R/$OUT/fit_${n}_$s.log 2>&1
  done
done
cd $R
for n in 4096 1024; do
  for s in 2 4; do
    for f in $(find $OUT/t_${n}_$s -name "*.db"); do python scripts/rocprof_summary.py $f "scripts/fit_once.py $n 10 $s 20 (21 rebuilds at n = $n, schedule $s)" > $OUT/fit_${n}_s${s}_trace.md; done
    grep "per rebuild" $OUT/fit_${n}_$s.log
    head -14 $OUT/fit_${n}_s${s}_trace.md | tail -8
  done
done
rm -rf $OUT/t_*
