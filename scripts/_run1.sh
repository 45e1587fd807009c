set -u
OUT=gpurun_out/r2i; mkdir -p $OUT
timeout 300 ./scripts/native/potf2_probe > $OUT/potf2_probe.txt 2>&1; tail -45 $OUT/potf2_probe.txt
timeout 600 python -m pytest tests/test_selection_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-bolfi > $OUT/bench_dist.json 2> $OUT/bench_dist.err; python -c "
import json; r=json.load(open('$OUT/bench_dist.json')); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac'])"
timeout 300 python bench.py --no-cpu-baseline --no-bolfi --steps 200 > $OUT/bench_dist200.json 2>> $OUT/bench_dist.err; python -c "
import json; r=json.load(open('$OUT/bench_dist200.json')); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac'])"
