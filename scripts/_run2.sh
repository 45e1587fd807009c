set -u
OUT=gpurun_out/r2j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gp_gpu.py -m gpu -x -q -k "not large_n" 2>&1 | tail -3
timeout 600 python scripts/time_schedules.py 1024:2 2048:10 3072:10 4096:10 5120:10 6144:10 8192:20 > $OUT/sched.txt 2>&1; cat $OUT/sched.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$OUT/tl -o tl -- python /root/repo/scripts/fit_once.py 4096 10 > /root/repo/$OUT/tl.log 2>&1
cd /root/repo
f=$(find $OUT/tl -name "*kernel_trace.csv" | head -1); python scripts/timeline_gp.py $f -1 120 > $OUT/timeline.txt 2>&1; grep step_kernel $OUT/timeline.txt | awk '{print $3}' | tr '\n' ' '; echo; tail -5 $OUT/timeline.txt
rm -rf $OUT/tl
