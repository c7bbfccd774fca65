#!/usr/bin/env python
"""Backbone and RCNNHead time per frame as a function of the frames per launch (8/16/24/32) and chains."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd import ops
from diffusionvid_amd.utils import synthetic
sd = synthetic.make_state_dict(0)
m = ops.Model(sd)
M, H, W = 300, 608, 1024
g = torch.Generator().manual_seed(0)
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for n in (8, 16, 24, 32, 56):
    m.reserve(n, H, W, M)
    x = torch.rand(n, 3, H, W, device="cuda")
    feats = [torch.randn(n, H // s, W // s, 256, generator=g).half().cuda() for s in (8, 16, 32)]
    boxes = torch.rand(n, M, 4, generator=g) * 300
    boxes[..., 2:] += boxes[..., :2] + 20
    boxes = boxes.cuda()
    pro = torch.randn(n * M, 256, generator=g).cuda()
    t = torch.full((n,), 999, dtype=torch.long)
    for ch in (1, 2, 4):
        m.set_chains(ch)
        ms = timeit(lambda: m.backbone(x))
        mh = timeit(lambda: m.rcnn_head(1, feats, H, W, boxes, pro, t), 20)
        print("frames=%2d chains=%d: backbone %.3f ms (%.3f per 8 frames, %.0f TFLOP/s); RCNNHead %.3f ms (%.3f per 8 frames)" % (
            n, ch, ms, ms * 8 / n, 2 * 106.54e9 * n / ms / 1e9, mh, mh * 8 / n), flush=True)
