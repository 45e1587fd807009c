#!/bin/bash
set -u
OUT=${1:-gpurun_out/r5e}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 300 python scripts/dist_once.py > $OUT/dist_once.txt 2>&1
cat $OUT/dist_once.txt
timeout 300 python scripts/dist_once.py 1250000 64 100 > $OUT/dist_once_64.txt 2>&1
cat $OUT/dist_once_64.txt
timeout 300 python scripts/dist_once.py 2000000 16 100 > $OUT/dist_once_16.txt 2>&1
cat $OUT/dist_once_16.txt
