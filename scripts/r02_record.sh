#!/bin/bash
# Round-2 record run on the GPU box: new parity tests, the default bench line, rocprofv3 kernel trace + PMC passes of it.
# usage (from the repo root): bash scripts/r02_record.sh <outdir under gpurun_out>
set -u
OUT=${1:-gpurun_out/r2rec}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.err
timeout 600 python -m pytest tests/test_gp_hyper_gpu.py tests/test_selection_gpu.py -m gpu -x -q > $OUT/pytest_new.log 2>&1
tail -3 $OUT/pytest_new.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/$OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-bolfi > $R/$OUT/pmc_$c.log 2>&1
done
cd $R
find $OUT -name "*.db" | head; find $OUT -name "*counter_collection.csv" | head
for f in $(find $OUT/trace -name "*.db"); do python scripts/rocprof_summary.py $f "bench.py --steps 50 --warmup 5 --no-cpu-baseline" > $OUT/trace_summary.md; done
for c in FETCH_SIZE WRITE_SIZE; do for f in $(find $OUT/pmc_$c -name "*counter_collection.csv"); do python scripts/pmc_summary.py $f dist_ > $OUT/pmc_$c.txt; done; done
cat $OUT/pmc_*.txt
# keep the merge-back small
find $OUT -name "*.db" -size +20M -delete
