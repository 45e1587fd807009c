set -u
timeout 900 python -m pytest tests/test_gp_gpu.py tests/test_acquisition_gpu.py tests/test_bolfi_trace_gpu.py tests/test_maxvar_gpu.py tests/test_posterior_gpu.py -m gpu -x -q -k "not large_n and not cfg5" 2>&1 | tail -3
timeout 300 python scripts/time_step.py 4096 10 10 2>&1 | tail -3
timeout 300 python scripts/time_step.py 8192 20 16 2>&1 | tail -3
