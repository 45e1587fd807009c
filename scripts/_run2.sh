set -u
OUT=gpurun_out/r2q; mkdir -p $OUT
( time timeout 900 python bench.py --no-cpu-baseline ) > $OUT/bench.json 2> $OUT/bench.err; tail -4 $OUT/bench.err
python - <<'P'
import json
r=json.load(open('gpurun_out/r2q/bench.json'))
print(r['value'], r['bolfi']['value'], r['bolfi']['ms_fit'], r['bolfi']['ms_acquire'])
print(r['bolfi']['cfg5'])
P
