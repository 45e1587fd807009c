set -u
timeout 1200 python -m pytest tests/test_gp_gpu.py tests/test_acquisition_gpu.py tests/test_maxvar_gpu.py tests/test_posterior_gpu.py tests/test_bolfi_trace_gpu.py -m gpu -x -q -k "not large_n" 2>&1 | tail -3
timeout 300 python scripts/time_step.py 4096 10 10 2>&1 | tail -3
timeout 300 python scripts/time_predict.py 2>&1 | tail -12
python - <<'P'
import sys; sys.path.insert(0,'.')
from elfi_amd import bolfi_bench
print(bolfi_bench.cfg5_leg())
P
