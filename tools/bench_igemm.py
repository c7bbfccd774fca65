#!/usr/bin/env python
"""Per-layer micro-benchmark of the implicit-GEMM kernel over the distinct layer shapes of the
R101-FPN backbone (batch 8, 608x1024) and of the decoder head (2400 rows).  Prints achieved TFLOP/s
and the algorithmic HBM GB/s (input + weights + output + residual, each touched once) per layer.

    python tools/bench_igemm.py [--iters 20] [--batch 8]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd import ops  # noqa: E402


def layers(n):
    L = []
    H, W = 608, 1024
    L.append(("stem7x7s2", n, H, W, 8, 64, 7, 2, 3, 0, 3))
    h, w = H // 4, W // 4
    cin = 64
    for s, nb in enumerate((3, 4, 23, 3)):
        width, cout = 64 << s, 256 << s
        for b in (0, 1):
            if b >= nb:
                continue
            stride = 2 if (b == 0 and s > 0) else 1
            cnt = 1 if b == 0 else nb - 1
            L.append((f"res{s+2}.{b}.conv1 x{cnt}", n, h, w, cin, width, 1, 1, 0, 0, cin))
            L.append((f"res{s+2}.{b}.conv2 x{cnt}", n, h, w, width, width, 3, stride, 1, 0, width))
            ho, wo = h // stride, w // stride
            if b == 0:
                L.append((f"res{s+2}.{b}.shortcut x1", n, h, w, cin, cout, 1, stride, 0, 0, cin))
            L.append((f"res{s+2}.{b}.conv3+res x{cnt}", n, ho, wo, width, cout, 1, 1, 0, 1, width))
            h, w, cin = ho, wo, cout
    for lvl, c, sc in ((5, 2048, 32), (4, 1024, 16), (3, 512, 8)):
        L.append((f"fpn_lateral{lvl}", n, H // sc, W // sc, c, 256, 1, 1, 0, 2 if lvl < 5 else 0, c))
        L.append((f"fpn_output{lvl}", n, H // sc, W // sc, 256, 256, 3, 1, 1, 0, 256))
    return L


def head_layers(rows):
    return [("in_proj", rows, 256, 768), ("out_proj", rows, 256, 256), ("dynamic_layer", rows, 256, 32768),
            ("out_layer", rows, 12544, 256), ("linear1", rows, 256, 2048), ("linear2", rows, 2048, 256),
            ("tower", rows, 256, 256), ("class_logits", rows, 256, 30)]


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
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--only", type=str, default="", help="substring filter on layer names")
    ap.add_argument("--vendor", action="store_true",
                    help="also time torch.matmul (hipBLASLt / rocBLAS) on the same [M,K] x [N,K]^T fp16 problem for every 1x1 stride-1 "
                         "layer -- the vendor library as a yardstick (plain GEMM: no bias / residual / ReLU epilogue, so it does less)")
    args = ap.parse_args()
    dev = "cuda"
    tot_ms = tot_fl = 0.0
    print(f"{'layer':28s} {'M':>8s} {'N':>6s} {'K':>6s} {'ms':>8s} {'TFLOP/s':>8s} {'GB/s':>8s} {'MB':>8s}")
    for name, n, h, w, cin, cout, k, stride, pad, res, cin_real in layers(args.batch):
        if args.only and args.only not in name:
            continue
        x = torch.randn(n, h, w, cin, device=dev).half()
        wt = torch.randn(cout, cin, k, k) * 0.05
        wp, kpad = ops.pack_conv_weight(wt)
        wp = wp.to(dev)
        bias = torch.randn(cout, device=dev)
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        r = None
        if res == 1:
            r = torch.randn(n, ho, wo, cout, device=dev).half()
        elif res == 2:
            r = torch.randn(n, ho // 2, wo // 2, cout, device=dev).half()
        ms = timeit(lambda: ops.conv2d_nhwc(x, wp, kpad, bias, cout, k, k, stride, pad, relu=True, residual=r, residual_mode=res),
                    args.iters)
        M = n * ho * wo
        fl = 2.0 * M * cout * k * k * cin_real
        by = x.numel() * 2 + wp.numel() * 2 + M * cout * 2 + (r.numel() * 2 if r is not None else 0)
        mult = int(name.split(" x")[1]) if " x" in name else 1
        tot_ms += ms * mult
        tot_fl += fl * mult
        extra = ""
        if args.vendor and k == 1 and stride == 1:
            a2 = x.view(-1, cin)
            w2 = (torch.randn(cout, cin, device=dev) * 0.05).half()
            o2 = torch.empty(a2.shape[0], cout, dtype=torch.float16, device=dev)
            vms = timeit(lambda: torch.matmul(a2, w2.t(), out=o2), args.iters)
            extra = f"   vendor GEMM {vms:8.4f} ms {fl/vms/1e9:8.1f} TFLOP/s  (igemm2 / vendor time {ms/vms:5.2f})"
        print(f"{name:28s} {M:8d} {cout:6d} {k*k*cin:6d} {ms:8.4f} {fl/ms/1e9:8.1f} {by/ms/1e6:8.0f} {by/1e6:8.1f}{extra}")
    if tot_ms:
        print(f"backbone total (weighted): {tot_ms:.3f} ms per {args.batch} frames, {tot_fl/tot_ms/1e9:.1f} TFLOP/s")
    rows = args.batch * 300
    for name, m, k, nout in head_layers(rows):
        if args.only and args.only not in "head." + name:
            continue
        x = torch.randn(m, k, device=dev).half()
        wp, kpad = ops.pack_conv_weight(torch.randn(nout, k) * 0.05)
        wp = wp.to(dev)
        bias = torch.randn(nout, device=dev)
        f32 = name != "dynamic_layer" and name != "linear1"
        ms = timeit(lambda: ops.linear(x, wp, kpad, bias, out_f32=f32), args.iters)
        fl = 2.0 * m * k * nout
        by = x.numel() * 2 + wp.numel() * 2 + m * nout * (4 if f32 else 2)
        extra = ""
        if args.vendor:
            w2 = (torch.randn(nout, k, device=dev) * 0.05).half()
            o2 = torch.empty(m, nout, dtype=torch.float16, device=dev)
            vms = timeit(lambda: torch.matmul(x, w2.t(), out=o2), args.iters)
            extra = f"   vendor GEMM {vms:8.4f} ms {fl/vms/1e9:8.1f} TFLOP/s  (igemm2 / vendor time {ms/vms:5.2f})"
        print(f"{'head.'+name:28s} {m:8d} {nout:6d} {k:6d} {ms:8.4f} {fl/ms/1e9:8.1f} {by/ms/1e6:8.0f} {by/1e6:8.1f}{extra}")


if __name__ == "__main__":
    main()
