#!/bin/bash
# Round-5 record run on the GPU box: the default bench line (all legs), the rocprofv3 kernel trace of the same command
# without the end-to-end legs, PMC passes of the distance kernel (HBM bytes; one counter per pass), the per-kernel table.
# usage (repo root): bash scripts/r05_record.sh <outdir under gpurun_out>
set -u
OUT=${1:-gpurun_out/r5rec}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 300 $OUT/bench_default.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --cfg3 off --e2e off > $R/$OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-bolfi --no-cfg4 --e2e off > $R/$OUT/pmc_$c.log 2>&1
done
cd $R
for f in $(find $OUT/trace -name "*.db"); do python scripts/rocprof_summary.py $f "bench.py --steps 50 --warmup 5 --no-cpu-baseline --cfg3 off --e2e off" > $OUT/trace_summary.md; done
for c in FETCH_SIZE WRITE_SIZE; do for f in $(find $OUT/pmc_$c -name "*counter_collection.csv"); do python scripts/pmc_summary.py $f dist_ > $OUT/pmc_$c.txt; done; done
cat $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt
timeout 300 python scripts/bench_kernels.py > $OUT/kernel_table.md 2> $OUT/kernel_table.err
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
head -16 $OUT/trace_summary.md
