#!/usr/bin/env python
"""Time the res4 conv3->conv1 fused launches of one backbone pass under DVID_C3C1_ABLATE (set in the environment):
1 no residual loads, 2 no Y stores, 4 no second product, 8 no first product (sums allowed)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd import _lib, ops
from diffusionvid_amd.utils import synthetic
n = 104
m = ops.Model(synthetic.make_state_dict(0))
m.reserve(n, 608, 1024, 300)
m.set_chains(1)
m.set_fusion(True)
x = torch.rand(n, 3, 608, 1024, device="cuda")
lib = _lib.load()
m.backbone(x); m.backbone(x)
torch.cuda.synchronize()
lib.dvid_profile_reset(); lib.dvid_profile_enable(1)
m.backbone(x)
torch.cuda.synchronize()
lib.dvid_profile_enable(0)
lib.dvid_profile_dump(b"/tmp/abl.csv")
rows = [l.strip().split(",") for l in open("/tmp/abl.csv")][1:]
t = [float(r[6]) for r in rows if r[8] == "1" and r[1] == "1024"]
print("ABLATE=%s: res4 fused launch %.1f us (x%d)" % (os.environ.get("DVID_C3C1_ABLATE", "0"), sum(t) / len(t) * 1e3, len(t)))
