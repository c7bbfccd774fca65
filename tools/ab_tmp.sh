timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -s -k "conv4x4 or conv3x3_halo or stem" 2>&1 | grep -v "^$" | tail -12
DVID_IGEMM_TUNE_LOG=0 python tools/bench_igemm.py --batch 104 2>&1 | grep -E "stem|total" > gpurun_out/halo_on.txt
DVID_CONV3X3_HALO=0 python tools/bench_igemm.py --batch 104 2>&1 | grep -E "stem|total" > gpurun_out/halo_off.txt
paste -d'\n' gpurun_out/halo_on.txt gpurun_out/halo_off.txt
for v in 1 0 1 0; do
  DVID_CONV3X3_HALO=$v python bench.py --no-side-configs --no-host-fed --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ab.json
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('halo', $v, d['value'], d['roofline']['achieved'])"
done
