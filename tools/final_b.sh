export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r02e_bench.json 2> gpurun_out/r02e_bench.err
tail -c 600 gpurun_out/r02e_bench.json
bash tools/profile_stats_only.sh r02e
head -12 gpurun_out/r02e_kernel_stats.txt
