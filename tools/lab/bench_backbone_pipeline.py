#!/usr/bin/env python
"""R101-FPN backbone at the bench's launch size (104 frames 608x1024): sequential / parallel chains / the front-back software
pipeline (ops.Model.set_pipeline) over sub-batch counts and split points.  Prints ms per pass and whether the feature maps
are bit-identical to the sequential schedule."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd import ops  # noqa: E402
from diffusionvid_amd.utils import synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 104
sd = synthetic.make_state_dict(0)
m = ops.Model(sd)
m.reserve(n, 608, 1024, 300)
x = torch.rand(n, 3, 608, 1024, device="cuda")
ref = None


def run(label):
    global ref
    for _ in range(2):
        p = m.backbone(x)       # first pass: tuner sees the launch shapes of this schedule
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        p = m.backbone(x)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    if ref is None:
        ref = [t.clone() for t in p]
    same = all(torch.equal(t, r) for t, r in zip(p, ref))
    print("%-34s %8.3f ms per %d frames  %6.0f frames/s  %5.0f TFLOP/s  identical: %s" % (label, ms, n, n / ms * 1e3, 2 * 106.54e9 * n / ms / 1e9, same),
          flush=True)


m.set_pipeline(0)
m.set_chains(1)
run("sequential")
m.set_chains(2)
run("2 parallel chains")
for parts in (2, 3, 4, 6, 8):
    for st, blk in ((2, 0), (2, 6), (1, 0), (2, 11)):
        m.set_pipeline(parts, st, blk)
        run("pipeline %d parts, split res%d.%d" % (parts, st + 2, blk))
