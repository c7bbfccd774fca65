export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -x -k "swin" 2>&1 | grep -v "^$" | tail -3
for v in 1 4; do DVID_SWIN_ATTN_WPB=$v timeout 60 python tools/bench_launch_order.py swin 32 2>&1 | grep Swin | sed "s/^/WPB=$v /"; done
