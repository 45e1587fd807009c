#!/bin/bash
# rocprofv3 counter passes of one command, summarised per kernel.
# usage (repo root): bash scripts/r04_pmc.sh <outdir> <kernel-substring> "<counters pass 1>" ["<counters pass 2>" ...] -- <python args>
set -u
OUT=$1; KER=$2; shift 2
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
i=0
: > $R/$OUT/pmc.txt
for grp in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$OUT/p_$i -o pmc -- python $R/"$@" > $R/$OUT/p_$i.log 2>&1
  f=$(find $R/$OUT/p_$i -name "*counter_collection.csv" | head -1)
  echo "== $* : $grp" >> $R/$OUT/pmc.txt
  python $R/scripts/pmc_summary.py $f $KER >> $R/$OUT/pmc.txt 2>&1
  rm -rf $R/$OUT/p_$i
done
cd $R
cat $OUT/pmc.txt
