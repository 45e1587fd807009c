#!/bin/bash
# reject_merge_kernel in its two habitats: the distance bench (a few hundred candidates every 8th step) and configs[3]'s round
# (the ~2000 candidates of a fresh state).  usage: bash scripts/r06_merge_trace.sh <outdir>
set -u
OUT=${1:-gpurun_out/r6merge}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
python bench.py --steps 50 --warmup 5 --no-bolfi --no-cfg4 --e2e off --cfg3 off --no-cpu-baseline > $OUT/bench_dist.json 2> $OUT/bench_dist.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-bolfi --no-cfg4 --e2e off --cfg3 off --no-cpu-baseline > $R/$OUT/prof_dist.json 2> $R/$OUT/prof_dist.err
cd $R
for f in $(find $OUT/trace -name "*.db"); do python scripts/rocprof_summary.py $f "bench.py distance only" > $OUT/trace_summary.md; done
rm -rf $OUT/trace
grep -n "merge\|dist_rows" $OUT/trace_summary.md | cut -c1-200
python - $OUT/bench_dist.json <<'P'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('distance bench: ms_per_step %.5f kernel_ms %.5f value %.4g' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['value']))
P
