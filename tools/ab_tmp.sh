export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
timeout 1000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -8 > gpurun_out/r02e_gpu_pytest.log
cat gpurun_out/r02e_gpu_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
