set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "add_layernorm or swin or rcnn_head" 2>&1 | tail -3
for v in 0 1 0 1; do DVID_LN_ROWS=$v timeout 300 python tools/bench_launch_order.py swin 32 2>&1 | grep Swin; done
for v in 0 1; do DVID_LN_ROWS=$v timeout 600 python bench.py --arch swinb --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed --no-side-configs > gpurun_out/ab_swin_ln$v.json 2> gpurun_out/ab_swin_ln$v.err; python -c "import json;d=json.load(open('gpurun_out/ab_swin_ln$v.json'));print('swinb ln_rows $v', d['value'])"; done
