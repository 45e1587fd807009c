set -u
timeout 900 python -m pytest tests/test_distance_gpu.py -m gpu -x -q 2>&1 | tail -8
timeout 600 python scripts/bench_kernels.py 2>/dev/null | grep -i "mahal"
