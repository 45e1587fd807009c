#!/bin/bash
# Round-2 record run on the GPU box: the default bench line, rocprofv3 kernel trace + PMC passes of it, the developer
# tables kept under profiles/.   usage (from the repo root): bash scripts/r02_record.sh <outdir under gpurun_out>
set -u
OUT=${1:-gpurun_out/r2rec}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.err
timeout 600 python scripts/time_schedules.py > $OUT/schedules.txt 2>&1
timeout 600 python scripts/bench_kernels.py > $OUT/kernels.md 2> $OUT/kernels.err
timeout 60 ./scripts/native/step_probe > $OUT/step_probe.txt 2>&1
timeout 60 ./scripts/native/potf2_probe > $OUT/potf2_probe.txt 2>&1
timeout 60 ./scripts/native/mfma_probe > $OUT/mfma_probe.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/$OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-bolfi > $R/$OUT/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/tl -o tl -- python $R/scripts/fit_once.py 4096 10 2 > $R/$OUT/tl.log 2>&1
cd $R
for f in $(find $OUT/trace -name "*.db"); do python scripts/rocprof_summary.py $f "bench.py --steps 50 --warmup 5 --no-cpu-baseline" > $OUT/trace_summary.md; done
for c in FETCH_SIZE WRITE_SIZE; do for f in $(find $OUT/pmc_$c -name "*counter_collection.csv"); do python scripts/pmc_summary.py $f dist_ > $OUT/pmc_$c.txt; done; done
cat $OUT/pmc_*.txt
f=$(find $OUT/tl -name "*kernel_trace.csv" | head -1); python scripts/timeline_gp.py $f -1 140 > $OUT/timeline.txt 2>&1
rm -rf $OUT/tl $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
