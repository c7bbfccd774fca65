export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 170 python bench.py --arch swinb --steps 2 --no-side-configs --no-cpu-baseline --no-host-fed > gpurun_out/r02f_bench_swinb.json 2> gpurun_out/r02f_bench_swinb.err
tail -c 200 gpurun_out/r02f_bench_swinb.json
timeout 150 python bench.py --sample-step 4 --steps 2 --no-side-configs --no-cpu-baseline --no-host-fed > gpurun_out/r02f_bench_x4.json 2> gpurun_out/r02f_bench_x4.err
tail -c 200 gpurun_out/r02f_bench_x4.json
