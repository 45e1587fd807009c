set -u
OUT=gpurun_out/r2l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gp_gpu.py -m gpu -x -q -k "not large_n" 2>&1 | tail -3
timeout 600 python scripts/time_schedules.py 2048:10 4096:10 6144:10 8192:20 2>&1 | grep fused
timeout 60 ./scripts/native/step_probe | awk 'NR<3 || /^ *(2|4|13|26) /'
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$OUT/tl -o tl -- python /root/repo/scripts/fit_once.py 4096 10 2 > /root/repo/$OUT/tl.log 2>&1
cd /root/repo
f=$(find $OUT/tl -name "*kernel_trace.csv" | head -1); python scripts/timeline_gp.py $f -1 120 > $OUT/timeline.txt 2>&1; grep step_kernel $OUT/timeline.txt | awk '{print $3}' | tr '\n' ' '; echo; grep trsm $OUT/timeline.txt | awk '{print $3}' | tr '\n' ' '; echo
rm -rf $OUT/tl
