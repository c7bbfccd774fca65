import sys, torch
sys.path.insert(0, "/root/repo")
from diffusionvid_amd import ops
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (19456, 256, 2304), (19456, 1024, 2304), (77824, 1024, 1024)]:
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K) * 0.05).half().cuda()
    out = torch.empty(M, 1, 1, N, dtype=torch.float16, device="cuda")
    from diffusionvid_amd._lib import call, ptr, stream_ptr
    fn = lambda: call("dvid_conv2d_nhwc_f16", ptr(x), ptr(w), None, None, ptr(out), M, 1, 1, K, N, 1, 1, 1, 0, K, 0, 0, 0, stream_ptr())
    ms = timeit(fn)
    print("GEMM %6d x %5d x %5d : %.3f ms  %.1f TFLOP/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
