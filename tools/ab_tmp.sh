timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -s -k "conv3x3_halo" 2>&1 | grep -v "^$" | tail -8
DVID_CONV3X3_HALO=2 python tools/bench_igemm.py --batch 104 2>&1 | grep -E "res2.*conv2|total" > gpurun_out/halo_on.txt
DVID_CONV3X3_HALO=0 python tools/bench_igemm.py --batch 104 2>&1 | grep -E "res2.*conv2|total" > gpurun_out/halo_off.txt
paste -d'\n' gpurun_out/halo_on.txt gpurun_out/halo_off.txt
DVID_CONV3X3_HALO=2 python tools/bench_igemm.py --batch 8 2>&1 | grep -E "res2.*conv2|total" > gpurun_out/halo_on.txt
DVID_CONV3X3_HALO=0 python tools/bench_igemm.py --batch 8 2>&1 | grep -E "res2.*conv2|total" > gpurun_out/halo_off.txt
paste -d'\n' gpurun_out/halo_on.txt gpurun_out/halo_off.txt
