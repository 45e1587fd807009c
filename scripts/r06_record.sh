#!/bin/bash
# Round-6 record run on the GPU box: the default bench line (compact stdout line + detail record), the rocprofv3 kernel trace of
# the same command without the CPU / end-to-end legs, PMC passes of the distance kernel (HBM bytes; one counter per pass), the
# MFMA-utilisation counters of the GP kernels (rebuild at n = 4096 / 2048, lock-steps, configs[4]-shaped dense acquisition), the
# per-kernel table.   usage (repo root): bash scripts/r06_record.sh <outdir under gpurun_out>
set -u
OUT=${1:-gpurun_out/r6rec}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
cp bench_detail.json $OUT/bench_default_detail.json
wc -c $OUT/bench_default.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --cfg3 off --e2e off > $R/$OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-bolfi --no-cfg4 --e2e off > $R/$OUT/pmc_$c.log 2>&1
done
: > $R/$OUT/gp_pmc.txt
for tgt in "scripts/fit_once.py 4096 10 2 5" "scripts/fit_once.py 2048 10 2 5" "scripts/lcb_loop.py 4096 10" "scripts/cfg5_trace.py"; do
  tag=$(echo $tgt | tr ' /.' '___')
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/g_$tag -o pmc -- python $R/$tgt > $R/$OUT/g_$tag.log 2>&1
  f=$(find $R/$OUT/g_$tag -name "*counter_collection.csv" | head -1)
  echo "== $tgt : SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" >> $R/$OUT/gp_pmc.txt
  python $R/scripts/pmc_summary.py $f >> $R/$OUT/gp_pmc.txt 2>&1
  rm -rf $R/$OUT/g_$tag
done
cd $R
for f in $(find $OUT/trace -name "*.db"); do python scripts/rocprof_summary.py $f "bench.py --steps 50 --warmup 5 --no-cpu-baseline --cfg3 off --e2e off" > $OUT/trace_summary.md; done
for c in FETCH_SIZE WRITE_SIZE; do for f in $(find $OUT/pmc_$c -name "*counter_collection.csv"); do python scripts/pmc_summary.py $f dist_ > $OUT/pmc_$c.txt; done; done
cat $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt
timeout 300 python scripts/bench_kernels.py > $OUT/kernel_table.md 2> $OUT/kernel_table.err
timeout 300 python scripts/lockstep_forms.py > $OUT/lockstep_forms.txt 2>&1
timeout 300 env ONLY_SCHEDULES=2 python scripts/time_schedules.py > $OUT/time_schedules.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
head -24 $OUT/trace_summary.md
tail -5 $OUT/lockstep_forms.txt
