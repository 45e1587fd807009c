#!/bin/bash
# Round 4: configs[3] (one adaptive-distance round of 10^7 x 64 per step) -- bench line, rocprofv3 kernel trace and the
# HBM-traffic counters of the same command.   usage (repo root): bash scripts/r04_cfg4.sh <outdir under gpurun_out> [pmc]
set -u
OUT=${1:-gpurun_out/r4cfg4}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
CMD="bench.py --workload adaptive --steps 10 --warmup 3 --no-cpu-baseline --no-bolfi"
timeout 300 python $CMD > $OUT/bench_adaptive.json 2> $OUT/bench_adaptive.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o bench -- python $R/$CMD > $R/$OUT/trace.log 2>&1
if [ "${2:-}" = "pmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/$CMD > $R/$OUT/pmc_$c.log 2>&1
  done
fi
cd $R
for f in $(find $OUT/trace -name "*.db"); do python scripts/rocprof_summary.py $f "$CMD" > $OUT/trace_summary.md; done
if [ "${2:-}" = "pmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do for f in $(find $OUT/pmc_$c -name "*counter_collection.csv"); do python scripts/pmc_summary.py $f adaptive_pass > $OUT/pmc_$c.txt; done; done
  cat $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt
fi
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
cat $OUT/trace_summary.md | head -40
python -c "
import json
d = json.loads(open('$OUT/bench_adaptive.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
