set -u
timeout 900 python -m pytest tests/test_gp_gpu.py tests/test_gp_hyper_gpu.py tests/test_acquisition_gpu.py tests/test_bolfi_trace_gpu.py -m gpu -x -q -k "not large_n and not cfg5" 2>&1 | tail -3
timeout 600 python scripts/time_schedules.py 512:2 1024:2 4096:10 8192:20 2>&1 | grep fused
