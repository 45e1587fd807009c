#!/bin/bash
# Round-2 PMC passes for the GP rebuild kernels (separate runs per counter group, kernel trace only).
# usage (repo root): bash scripts/r02_gp_pmc.sh <outdir under gpurun_out>
set -u
OUT=${1:-gpurun_out/r2pmc}; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
for shape in "4096 10" "8192 20"; do
  tag=$(echo $shape | tr ' ' 'x')
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$OUT/p_${tag}_$i -o pmc -- python $R/scripts/fit_once.py $shape 2 5 > $R/$OUT/log_${tag}_$i.txt 2>&1
    f=$(find $R/$OUT/p_${tag}_$i -name "*counter_collection.csv" | head -1)
    echo "== n x d = $tag : $grp" >> $R/$OUT/pmc.txt
    python $R/scripts/pmc_summary.py $f >> $R/$OUT/pmc.txt 2>&1
    rm -rf $R/$OUT/p_${tag}_$i
  done
done
cd $R
cat $OUT/pmc.txt | grep -A4 "step_kernel\|^==" | head -80
