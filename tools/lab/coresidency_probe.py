#!/usr/bin/env python
"""What would a PERFECT co-resident partner gain on the res4 kernels?   (round 5; review item 1a)

    python tools/lab/coresidency_probe.py [--frames 152] [--reps 8]

For each res4 layer (conv1 1024 -> 256, conv2 3x3 256 -> 256, conv3 256 -> 1024 + residual; the product kernels, untouched) the layer is
run `reps` times on stream A while stream B runs a synthetic partner that fits in what the layer leaves of a CU (tools/lab/partner_kernels.hip:
no LDS, ~40 VGPRs): a pure HBM stream (read, or read + write) or a pure MFMA stream, sized to take about as long as the layer alone.  Reported:
the layer alone, the partner alone, both together (first start to last end), and the overlap efficiency
    (T_layer + T_partner - T_both) / min(T_layer, T_partner)        1 = the shorter one is free, 0 = they serialise.
A partner cut from a real layer can only do worse than these: it needs LDS, more registers and both pipes.
"""
import argparse
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diffusionvid_amd import ops  # noqa: E402


def load_partner():
    so = os.path.join(ROOT, "tools", "lab", "libpartner.so")
    if not os.path.exists(so):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(ROOT, "tools", "lab", "partner_kernels.hip"), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.partner_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int]
    lib.partner_mfma.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=152)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--grid", type=int, default=2048, help="workgroups of the HBM partner (256 threads each, grid-stride): 256 = one per CU, "
                    "so it cannot crowd the layer's workgroups out of a CU's registers; 2048 = it takes every slot it is offered")
    ap.add_argument("--mgrid", type=int, default=1024, help="workgroups of the MFMA partner")
    args = ap.parse_args()
    dev = "cuda"
    lib = load_partner()
    n, h, w = args.frames, 38, 64
    M = n * h * w

    def conv(cin, cout, k, res):
        x = torch.randn(n, h, w, cin, device=dev).half()
        wp, kpad = ops.pack_conv_weight(torch.randn(cout, cin, k, k) * 0.05)
        wp = wp.to(dev)
        b = torch.randn(cout, device=dev)
        r = torch.randn(n, h, w, cout, device=dev).half() if res else None
        return lambda: ops.conv2d_nhwc(x, wp, kpad, b, cout, k, k, 1, k // 2, relu=True, residual=r, residual_mode=1 if res else 0)

    layers = {
        "conv1 1x1 1024->256 (HBM-paced)": (conv(1024, 256, 1, False), 2.0 * M * 256 * 1024, M * (1024 + 256) * 2),
        "conv2 3x3 256->256 (MFMA-paced)": (conv(256, 256, 3, False), 2.0 * M * 256 * 2304, M * (256 + 256) * 2),
        "conv3 1x1 256->1024 +res (HBM-paced)": (conv(256, 1024, 1, True), 2.0 * M * 1024 * 256, M * (256 + 1024 + 1024) * 2),
    }
    big = 2 << 30
    src = torch.empty(big, dtype=torch.uint8, device=dev).fill_(1)
    dst = torch.empty(big, dtype=torch.uint8, device=dev)
    sink = torch.zeros(4, device=dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fa, fb):
        """fa on stream A, fb on stream B (either may be None) -> ms from the first start to the last end"""
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        sa.wait_event(e0)
        sb.wait_event(e0)
        if fa:
            with torch.cuda.stream(sa):
                fa()
                e1.record(sa)
        if fb:
            with torch.cuda.stream(sb):
                fb()
                e2.record(sb)
        torch.cuda.synchronize()
        return max(e0.elapsed_time(e1) if fa else 0.0, e0.elapsed_time(e2) if fb else 0.0)

    def best(fa, fb, k=5):
        timed(fa, fb)
        return min(timed(fa, fb) for _ in range(k))

    def copy_partner(nbytes, write, grid):
        def f():
            left = nbytes
            while left > 0:
                c = min(left, big)
                lib.partner_copy(torch.cuda.current_stream().cuda_stream, src.data_ptr(), dst.data_ptr(), c, write, grid)
                left -= c
        return f

    def mfma_partner(iters, grid):
        return lambda: lib.partner_mfma(torch.cuda.current_stream().cuda_stream, iters, grid, sink.data_ptr())

    print(f"# res4 shapes at {n} frames (M = {M} rows), {args.reps} launches of the layer per measurement; times in ms")
    # partners alone, to size them
    t_copy = best(copy_partner(big, 1, args.grid), None)
    t_read = best(copy_partner(big, 0, args.grid), None)
    t_mfma = best(mfma_partner(4096, args.mgrid), None)
    print(f"# partner rates alone: copy {2 * big / t_copy / 1e9:.2f} TB/s (r+w), read {big / t_read / 1e9:.2f} TB/s, "
          f"mfma {4096 * 2.0 * 32 * 32 * 16 * 16 * 4 * args.mgrid / t_mfma / 1e9:.0f} TFLOP/s (4 waves x {args.mgrid} workgroups); HBM partner grid {args.grid}")
    for name, (fn, flops, nbytes) in layers.items():
        fn()
        run = lambda: [fn() for _ in range(args.reps)]
        t_l = best(run, None)
        print(f"{name:40s} alone {t_l / args.reps:7.3f} ms per launch  {flops * args.reps / t_l / 1e9:7.0f} TFLOP/s  {nbytes * args.reps / t_l / 1e9:6.2f} TB/s")
        for pname, mk, t_unit in (("HBM read", lambda s: copy_partner(int(big * s) // 4096 * 4096, 0, args.grid), t_read),
                                  ("HBM copy", lambda s: copy_partner(int(big * s) // 4096 * 4096, 1, args.grid), t_copy),
                                  ("MFMA", lambda s: mfma_partner(int(4096 * s), args.mgrid), t_mfma)):
            scale = t_l / t_unit                      # partner as long as the layer's reps
            pf = mk(scale)
            t_p = best(None, pf)
            t_b = best(run, pf)
            eff = (t_l + t_p - t_b) / min(t_l, t_p)
            print(f"    + {pname:9s} partner: partner alone {t_p:7.3f}  layer alone {t_l:7.3f}  together {t_b:7.3f}   overlap efficiency {eff:5.2f}   "
                  f"(serial {t_l + t_p:7.3f}, ideal {max(t_l, t_p):7.3f})")


if __name__ == "__main__":
    main()
