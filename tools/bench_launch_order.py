#!/usr/bin/env python
"""Launch-order A/B of the two gather kernels (run once per setting; the switches are read at the first launch):
   DVID_ROI_XCD=0|1         roialign: boxes round-robin over the XCDs | an XCD takes whole images
   DVID_SWIN_ATTN_ORDER=0|1 swin_window_attn: window-major per head | heads of a window together on one XCD
usage: bench_launch_order.py roi [frames] | swin [frames]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd import ops
from diffusionvid_amd.utils import synthetic


def timeit(fn, iters=10):
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


what = sys.argv[1] if len(sys.argv) > 1 else "roi"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (104 if what == "roi" else 32)
H, W, M = 608, 1024, 300
g = torch.Generator().manual_seed(0)
if what == "roi":
    feats = [torch.randn(n, H // s, W // s, 256, generator=g).half().cuda() for s in (8, 16, 32)]
    for kind in ("noise", "refined"):
        if kind == "noise":          # boxes of the first head: clamp(randn * scale) as centre / size (diffusion_det.py:536-540)
            x = torch.clamp(torch.randn(n, M, 4, generator=g) * 2.0, -2.0, 2.0)
            x = (x / 2.0 + 1) / 2
        else:                        # object-sized boxes
            x = torch.rand(n, M, 4, generator=g)
            x[..., 2:] = 0.05 + 0.3 * x[..., 2:]
        cx, cy, w, h = x.unbind(-1)
        boxes = torch.stack([(cx - w / 2) * 1000, (cy - h / 2) * 600, (cx + w / 2) * 1000, (cy + h / 2) * 600], -1).cuda()
        ms = timeit(lambda: ops.roialign(feats, boxes, H, W, want_mean=True))
        print("DVID_ROI_XCD=%s roialign %s boxes, %d frames x %d: %.3f ms" % (os.environ.get("DVID_ROI_XCD", "default"), kind, n, M, ms))
else:
    sd = synthetic.make_state_dict(0, swin=dict(embed_dim=128, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32), window=7))
    m = ops.Model(sd, res_blocks=(0, 0, 0, 0), backbone="swin")
    m.reserve(n, H, W, M)
    x = torch.rand(n, 3, H, W, device="cuda")
    ms = timeit(lambda: m.backbone(x), iters=5)
    print("DVID_SWIN_ATTN_ORDER=%s Swin-B+FPN backbone: %.2f ms per %d frames -> %.1f frames/s" % (os.environ.get("DVID_SWIN_ATTN_ORDER", "default"), ms, n, n / ms * 1e3))
