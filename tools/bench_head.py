#!/usr/bin/env python
"""One RCNNHead pass (8 frames x 300 boxes, 608x1024 pyramids) for 1 vs 2 sub-batch chains, plus global attention."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd import ops
from diffusionvid_amd.utils import synthetic
sd = synthetic.make_head_state_dict(0)
m = ops.Model(sd, res_blocks=(0, 0, 0, 0))
n, M, H, W = 8, 300, 608, 1024
m.reserve(n, H, W, M)
g = torch.Generator().manual_seed(0)
feats = [torch.randn(n, H // s, W // s, 256, generator=g).half().cuda() for s in (8, 16, 32)]
boxes = torch.rand(n, M, 4, generator=g) * 300
boxes[..., 2:] += boxes[..., :2] + 20
boxes = boxes.cuda()
pro = torch.randn(n * M, 256, generator=g).cuda()
t = torch.full((n,), 999, dtype=torch.long)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for ch in (1, 2, 1, 2):
    m.set_chains(ch)
    ms = timeit(lambda: m.rcnn_head(1, feats, H, W, boxes, pro, t))
    ms0 = timeit(lambda: m.rcnn_head(0, feats, H, W, boxes, None, t))
    print("chains=%d: RCNNHead %.3f ms, first head (pro=None) %.3f ms per %d frames" % (ch, ms, ms0, n))
mem = torch.randn(900, 256, generator=g).cuda()
print("global_xattn: %.3f ms" % timeit(lambda: m.global_xattn(pro, mem)))
lg = torch.randn(n, M, 30, generator=g).cuda()
print("postproc: %.3f ms" % timeit(lambda: ops.postproc_topk_nms(lg, boxes, 1000., 600.)))
print("select_topk: %.3f ms" % timeit(lambda: ops.select_topk_features(lg, pro, 75, 25)))
