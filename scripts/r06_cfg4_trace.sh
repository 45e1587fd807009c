#!/bin/bash
# rocprofv3 kernel trace of configs[3]'s step alone (bench.py --workload adaptive --scaling strong --total 10^7 --m 64) and of
# the 8-rank shard (--n 1250000 --scaling weak): which launches a step holds beside the fused pass, with the gaps between
# them (scripts/step_timeline.py).   usage: bash scripts/r06_cfg4_trace.sh <outdir>
set -u
OUT=${1:-gpurun_out/r6cfg4}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for tag in full shard; do
  if [ $tag = full ]; then A="--scaling strong --total 10000000"; else A="--scaling weak --n 1250000"; fi
  python bench.py --workload adaptive $A --m 64 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace_$tag -o bench -- python $R/bench.py --workload adaptive $A --m 64 --steps 30 --warmup 5 --no-cpu-baseline > $R/$OUT/prof_$tag.json 2> $R/$OUT/prof_$tag.err
  cd $R
  f=$(find $OUT/trace_$tag -name "*kernel_trace.csv" | head -1)
  python scripts/step_timeline.py $f reject_init_kernel 20 2 > $OUT/timeline_$tag.txt
  rm -rf $OUT/trace_$tag
  python - $OUT/bench_$tag.json $OUT/prof_$tag.json <<'P'
import json, sys
for p in sys.argv[1:]:
    r = json.loads(open(p).read().strip().splitlines()[-1])
    print(p, 'ms_per_step %.4f kernel_ms %.4f value %.4g' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['value']))
P
  cat $OUT/timeline_$tag.txt
done
