DVID_IGEMM_TUNE_LOG=1 python tools/bench_igemm.py --iters 10 --batch 24 > gpurun_out/layers_tuned2_b24.txt 2> gpurun_out/tune_log2_b24.txt
DVID_IGEMM_TUNE_LOG=1 python tools/bench_igemm.py --iters 10 --batch 8 > gpurun_out/layers_tuned2_b8.txt 2> gpurun_out/tune_log2_b8.txt
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/t_kernels.txt
