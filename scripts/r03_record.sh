#!/bin/bash
# Round-3 record run on the GPU box: the default bench line (with the configs[2] end-to-end leg), rocprofv3 kernel trace of
# the same command, PMC passes (HBM bytes of the distance kernel; matrix-pipe busy cycles of the dense predictor), the
# developer probes kept under profiles/.   usage (repo root): bash scripts/r03_record.sh <outdir under gpurun_out>
set -u
OUT=${1:-gpurun_out/r3rec}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
( time timeout 600 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 300 $OUT/bench_default.err
timeout 200 python scripts/r3_probe.py lockstep > $OUT/probe_lockstep_fused.json 2>/dev/null
R3_LOCKSTEP_FORM=1 timeout 200 python scripts/r3_probe.py lockstep > $OUT/probe_lockstep_six.json 2>/dev/null
timeout 200 python scripts/r3_probe.py dense > $OUT/probe_dense.json 2>/dev/null
timeout 200 python scripts/r3_probe.py tiles > $OUT/probe_tiles.json 2>/dev/null
ELFIHIP_ACQ_TRACE=2 timeout 200 python scripts/r3_probe.py cfg5 > $OUT/probe_cfg5.json 2> $OUT/probe_cfg5.err
timeout 200 python scripts/time_topk.py > $OUT/topk.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --cfg3 off > $R/$OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-bolfi --no-cfg4 > $R/$OUT/pmc_$c.log 2>&1
done
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$OUT/pd_$i -o pmc -- python $R/scripts/dense_once.py 8192 20 256 5 1 > $R/$OUT/pd_$i.log 2>&1
  f=$(find $R/$OUT/pd_$i -name "*counter_collection.csv" | head -1)
  echo "== dense predictor, n x d = 8192x20, S = 256 : $grp" >> $R/$OUT/pmc_dense.txt
  python $R/scripts/pmc_summary.py $f >> $R/$OUT/pmc_dense.txt 2>&1
  rm -rf $R/$OUT/pd_$i
done
cd $R
for f in $(find $OUT/trace -name "*.db"); do python scripts/rocprof_summary.py $f "bench.py --steps 50 --warmup 5 --no-cpu-baseline --cfg3 off" > $OUT/trace_summary.md; done
for c in FETCH_SIZE WRITE_SIZE; do for f in $(find $OUT/pmc_$c -name "*counter_collection.csv"); do python scripts/pmc_summary.py $f dist_ > $OUT/pmc_$c.txt; done; done
cat $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt
grep -A6 "dense_tri" $OUT/pmc_dense.txt | head -60
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
