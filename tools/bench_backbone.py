#!/usr/bin/env python
"""R101-FPN backbone timing (8 frames 608x1024) for 1..4 sub-batch chains."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd import ops
from diffusionvid_amd.utils import synthetic
sd = synthetic.make_state_dict(0)
m = ops.Model(sd)
n = 8
m.reserve(n, 608, 1024, 300)
x = torch.rand(n, 3, 608, 1024, device="cuda")
ref = None
for ch in (1, 2, 4, 1, 2):
    m.set_chains(ch)
    for _ in range(2): p = m.backbone(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): p = m.backbone(x)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    if ref is None: ref = [t.clone() for t in p]
    same = all(torch.equal(t, r) for t, r in zip(p, ref))
    print("chains=%d: %.3f ms per %d frames (%.0f frames/s, %.0f TFLOP/s); identical to chains=1: %s" % (ch, ms, n, n / ms * 1e3, 2 * 106.54e9 * n / ms / 1e9, same))
