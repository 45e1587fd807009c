set -u
timeout 900 python -m pytest tests/test_gp_gpu.py tests/test_acquisition_gpu.py -m gpu -x -q -k "not large_n and not cfg5" 2>&1 | tail -3
echo fused; timeout 300 python scripts/time_step.py 4096 10 10 2>&1 | tail -3
echo separate; ELFIHIP_NO_FUSE=1 timeout 300 python scripts/time_step.py 4096 10 10 2>&1 | tail -3
