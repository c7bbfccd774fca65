"""What the vendor GEMM library reaches on the backbone's GEMM shapes (torch.matmul -> hipBLASLt/rocBLAS), as a
yardstick for igemm2 on the same [M,K]x[N,K]^T fp16 problems.  Measurement aid only; the product never calls it."""
import torch


def timeit(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for (M, N, K) in [(16384, 256, 1024), (19456, 256, 1024), (19456, 256, 2304), (19456, 1024, 256), (77824, 128, 1152),
                  (77824, 512, 128), (4864, 512, 4608), (4864, 2048, 512), (2400, 32768, 256), (2400, 256, 12544), (4096, 4096, 4096)]:
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * 0.05).half()
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    ms = timeit(lambda: torch.matmul(x, w.t(), out=out))
    print("hipBLASLt GEMM %6d x %5d x %5d : %7.2f us  %6.1f TFLOP/s" % (M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
