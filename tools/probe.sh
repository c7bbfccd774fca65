DVID_IGEMM_TUNE_LOG=1 python tools/bench_igemm.py --iters 10 --batch 8 > gpurun_out/layers_tuned_b8.txt 2> gpurun_out/tune_log_b8.txt
DVID_IGEMM_TUNE_LOG=1 python tools/bench_igemm.py --iters 10 --batch 16 > gpurun_out/layers_tuned_b16.txt 2> gpurun_out/tune_log_b16.txt
DVID_IGEMM_TUNE_LOG=1 python tools/bench_igemm.py --iters 10 --batch 24 > gpurun_out/layers_tuned_b24.txt 2> gpurun_out/tune_log_b24.txt
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/t_kernels.txt
for la in 1 2 3 4; do python bench.py --steps 3 --warmup 1 --lookahead $la --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_tuned_la$la.json; done
