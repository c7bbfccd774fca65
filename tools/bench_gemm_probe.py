"""Latency probe: one GEMM shape, timed under the env-selected igemm variant (DVID_IGEMM_TILE/BK/STAGES/BIG).
    python tools/bench_gemm_probe.py M N K [M N K ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd._lib import call, ptr, stream_ptr  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


args = [int(a) for a in sys.argv[1:]]
tag = " ".join(f"{k[11:]}={v}" for k, v in os.environ.items() if k.startswith("DVID_IGEMM_"))
for i in range(0, len(args), 3):
    M, N, K = args[i:i + 3]
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K) * 0.05).half().cuda()
    out = torch.empty(M, 1, 1, N, dtype=torch.float16, device="cuda")
    fn = lambda: call("dvid_conv2d_nhwc_f16", ptr(x), ptr(w), None, None, ptr(out), M, 1, 1, K, N, 1, 1, 1, 0, K, 0, 0, 0, stream_ptr())
    ms = timeit(fn)
    print("[%s] GEMM %6d x %5d x %5d : %7.2f us  %6.1f TFLOP/s" % (tag, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
