export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -8 > gpurun_out/r02e_gpu_pytest.log
tail -2 gpurun_out/r02e_gpu_pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python bench.py > gpurun_out/r02e_bench.json 2> gpurun_out/r02e_bench.err
tail -c 400 gpurun_out/r02e_bench.json
bash tools/profile_stats_only.sh r02e
head -8 gpurun_out/r02e_kernel_stats.txt | cut -c1-180
