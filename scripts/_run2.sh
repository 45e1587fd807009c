set -u
timeout 600 python -m pytest tests/test_comm_gpu.py tests/test_gp_gpu.py -m gpu -x -q -k "not large_n" 2>&1 | tail -8
