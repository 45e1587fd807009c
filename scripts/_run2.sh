set -u
timeout 600 python -m pytest tests/test_summaries_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python scripts/bench_kernels.py 2>/dev/null | grep -i "MA2\|row mean\|row var"
