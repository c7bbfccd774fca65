export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 420 python bench.py > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err
tail -c 300 gpurun_out/r02f_bench.json
rm -f gpurun_out/parity_report.txt
timeout 560 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -8 > gpurun_out/r02f_gpu_pytest.log
tail -2 gpurun_out/r02f_gpu_pytest.log
