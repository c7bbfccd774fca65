#!/usr/bin/env python
"""The weight-stationary kernel (csrc/wstat.hip) against igemm2 on the layers it takes over, at the bench's 104-frame shapes:

    python tools/bench_wstat.py [--iters 20] [--frames 104]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd import _lib, ops  # noqa: E402


def timeit(fn, iters):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--frames", type=int, default=104)
    args = ap.parse_args()
    lib = _lib.load()
    f = args.frames
    shapes = [("res3.conv3+res", f * 76 * 128, 128, 512, True, True), ("res4.conv3+res", f * 38 * 64, 256, 1024, True, True),
              ("res4.conv3 (no res)", f * 38 * 64, 256, 1024, False, True), ("dynamic_layer", f * 300, 256, 32768, False, False),
              ("linear1", f * 300, 256, 2048, False, True)]
    g = torch.Generator().manual_seed(0)
    for name, m, k, n, res, relu in shapes:
        x = (torch.randn(m, k, generator=g) * 0.5).to(torch.float16).cuda().view(m, 1, 1, k)
        wt = torch.randn(n, k, generator=g) * (1.0 / k ** 0.5)
        wp, kpad = ops.pack_conv_weight(wt)
        wp = wp.cuda()
        bias = torch.randn(n, generator=g).cuda()
        r = torch.randn(m, 1, 1, n, device="cuda", dtype=torch.float16) if res else None

        def run():
            return ops.conv2d_nhwc(x, wp, kpad, bias, n, 1, 1, 1, 0, relu=relu, residual=r, residual_mode=1 if res else 0)
        out = {}
        for mode, tag in ((2, "wstat"), (0, "igemm2")):
            _lib.check(lib.dvid_igemm_set_wstat(mode), "set_wstat")
            out[tag] = run()
            ms = timeit(run, args.iters)
            byt = (m * k + n * k + m * n * (2 if res else 1)) * 2
            print(f"{name:22s} M {m:8d} K {k:4d} N {n:6d}  {tag:7s} {ms:8.4f} ms  {2.0 * m * n * k / ms / 1e9:7.1f} TFLOP/s  {byt / ms / 1e6:7.0f} GB/s", flush=True)
        lib.dvid_igemm_set_wstat(-1)
        print(f"{'':22s} identical: {torch.equal(out['wstat'], out['igemm2'])}", flush=True)


if __name__ == "__main__":
    main()
