#!/usr/bin/env python
"""R101-FPN backbone with and without the conv3 -> conv1 fusion (csrc/c3c1.hip) at a given launch size; prints ms per pass,
bit-identity, and the per-launch table of the fused kernel from the library's event profile."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd import _lib, ops  # noqa: E402
from diffusionvid_amd.utils import synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 104
sd = synthetic.make_state_dict(0)
m = ops.Model(sd)
m.reserve(n, 608, 1024, 300)
x = torch.rand(n, 3, 608, 1024, device="cuda")
lib = _lib.load()
ref = None
for chains in (1, 2):
    m.set_chains(chains)
    for fuse in (False, True):
        m.set_fusion(fuse)
        for _ in range(2):
            p = m.backbone(x)
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
        print("chains %d fusion %-5s %8.3f ms per %d frames  %6.0f frames/s  identical: %s" % (chains, fuse, ms, n, n / ms * 1e3, same), flush=True)
m.set_chains(1)
for fuse in (False, True):
    m.set_fusion(fuse)
    lib.dvid_profile_reset()
    lib.dvid_profile_enable(1)
    m.backbone(x)
    torch.cuda.synchronize()
    lib.dvid_profile_enable(0)
    path = "/tmp/prof_fuse_%d.csv" % fuse
    lib.dvid_profile_dump(path.encode())
    rows = [l.strip().split(",") for l in open(path)][1:]
    tot = sum(float(r[6]) for r in rows)
    print("fusion %s: %d launches, %.3f ms in GEMM kernels" % (fuse, len(rows), tot))
    agg = {}
    for r in rows:
        key = (r[8], r[0], r[1], r[2], r[3], r[4], r[5])
        t = agg.setdefault(key, [0, 0.0])
        t[0] += 1
        t[1] += float(r[6])
    for key, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   kind %s M %8s N %5s K %5s taps %s stride/N2 %4s res %s : x%-3d %8.3f ms total %8.1f us each" % (key + (cnt, t, t / cnt * 1e3)))
    lib.dvid_profile_reset()
