timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -3
timeout 2400 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
