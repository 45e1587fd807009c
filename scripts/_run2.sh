set -u
timeout 600 python -m pytest tests/test_gm_gpu.py -m gpu -x -q 2>&1 | tail -8
