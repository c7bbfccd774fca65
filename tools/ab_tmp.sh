export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "add_layernorm or swin or rcnn_head or dynamic_head or xattn" 2>&1 | grep -v "^$" | tail -15
