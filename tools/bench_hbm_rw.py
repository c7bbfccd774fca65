#!/usr/bin/env python
"""Plain HBM streams as yardsticks for the store-bound kernels: fill (write only), copy (read + write), sum (read only)."""
import torch

def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

for mb in (256, 1024, 2048):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.float16, device="cuda")
    y = torch.empty(n, dtype=torch.float16, device="cuda")
    t = timeit(lambda: x.zero_())
    print(f"fill  {mb:5d} MB: {t:.4f} ms  {mb * 1.048576e-3 / t:7.2f} TB/s written")
    t = timeit(lambda: y.copy_(x))
    print(f"copy  {mb:5d} MB: {t:.4f} ms  {2 * mb * 1.048576e-3 / t:7.2f} TB/s read + written")
    t = timeit(lambda: torch.relu_(x))
    print(f"relu_ {mb:5d} MB: {t:.4f} ms  {2 * mb * 1.048576e-3 / t:7.2f} TB/s read + written (in place)")
