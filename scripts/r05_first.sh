#!/bin/bash
# Round-5 first GPU call: the LDS-DMA probe of the row-distance stream and the per-kernel table with honest rotations.
set -u
OUT=${1:-gpurun_out/r5a}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 scripts/native/glds_probe > $OUT/glds_probe.txt 2>&1
cat $OUT/glds_probe.txt
timeout 400 python scripts/bench_kernels.py > $OUT/kernel_table.md 2> $OUT/kernel_table.err
tail -3 $OUT/kernel_table.err
head -30 $OUT/kernel_table.md
