timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -4
timeout 2400 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -6
