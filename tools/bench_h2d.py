#!/usr/bin/env python
"""Host -> HBM copy rate of pinned frames on this box (what bounds bench.py's host_fed mode): per-frame copies vs one large
copy, fp32 padded frames vs uint8 native frames.  Measurement aid."""
import time

import torch

dev = torch.device("cuda")
for name, shape, dtype in (("fp32 608x1024 frame", (3, 608, 1024), torch.float32), ("uint8 720x1280 frame", (720, 1280, 3), torch.uint8)):
    n = 128
    host = [torch.empty(shape, dtype=dtype).pin_memory() for _ in range(n)]
    big = torch.empty((n,) + shape, dtype=dtype).pin_memory()
    dst = torch.empty((n,) + shape, dtype=dtype, device=dev)
    nbytes = dst.numel() * dst.element_size()
    for label, fn in (("per-frame copies", lambda: [dst[i].copy_(host[i], non_blocking=True) for i in range(n)]),
                      ("one copy", lambda: dst.copy_(big, non_blocking=True))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{name:24s} {label:18s} {nbytes / 1e6:8.1f} MB in {dt * 1e3:7.2f} ms = {nbytes / dt / 1e9:6.1f} GB/s", flush=True)
