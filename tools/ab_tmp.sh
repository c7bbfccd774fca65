export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -k "swin or igemm_configs" 2>&1 | grep -v "^$" | tail -6
timeout 300 python tools/bench_launch_order.py swin 32 2>&1 | grep Swin
timeout 600 python bench.py --arch swinb --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed --no-side-configs > gpurun_out/ab_swin_gelu2.json 2> gpurun_out/ab_swin_gelu2.err; python -c "import json;d=json.load(open('gpurun_out/ab_swin_gelu2.json'));print('swinb packed gelu', d['value'])"
