#!/bin/bash
# Round-5 GPU check: tests named on the command line (default: the ones this round touched) + a short default bench line.
set -u
OUT=${1:-gpurun_out/r5c}
shift || true
mkdir -p $OUT
export TMPDIR=/tmp
TESTS=${@:-tests/test_distance_gpu.py tests/test_selection_gpu.py tests/test_gp_gpu.py tests/test_gp_hyper_gpu.py}
timeout 1500 python -m pytest $TESTS -x -q -m gpu > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
( time timeout 600 python bench.py --no-cpu-baseline --cfg3 off --e2e off ) > $OUT/bench_short.json 2> $OUT/bench_short.err
tail -3 $OUT/bench_short.err
python - <<PY
import json
d = json.loads(open('$OUT/bench_short.json').read().strip().splitlines()[-1])
r = d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'])
print({k: r[k] for k in r if not isinstance(r[k], (dict, list))})
PY
