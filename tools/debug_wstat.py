#!/usr/bin/env python
"""Where does wstat differ from igemm2?  Mismatch statistics by row / column / tile for a few shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd import _lib, ops  # noqa: E402

lib = _lib.load()
g = torch.Generator().manual_seed(0)
for (m, k, n, res, relu, bias_on) in [(32, 256, 256, False, False, False), (32, 128, 256, False, False, False), (256, 256, 256, False, False, False),
                                      (2048, 256, 256, False, False, False), (2048, 256, 256, False, False, True), (2048, 256, 256, True, False, False),
                                      (2048, 256, 1024, False, False, False), (10253, 256, 1024, True, True, True)]:
    x = (torch.randn(m, k, generator=g)).to(torch.float16).cuda().view(m, 1, 1, k)
    wt = torch.randn(n, k, generator=g) * (1.0 / k ** 0.5)
    wp, kpad = ops.pack_conv_weight(wt)
    wp = wp.cuda()
    bias = (torch.randn(n, generator=g) if bias_on else torch.zeros(n)).cuda()
    r = torch.randn(m, 1, 1, n, device="cuda", dtype=torch.float16) if res else None
    out = {}
    for mode, tag in ((2, "wstat"), (0, "igemm2")):
        _lib.check(lib.dvid_igemm_set_wstat(mode), "set_wstat")
        out[tag] = ops.conv2d_nhwc(x, wp, kpad, bias, n, 1, 1, 1, 0, relu=relu, residual=r, residual_mode=1 if res else 0).view(m, n).float()
    lib.dvid_igemm_set_wstat(-1)
    torch.cuda.synchronize()
    bad = (out["wstat"] != out["igemm2"])
    print(f"M {m} K {k} N {n} res {res} relu {relu} bias {bias_on}: mismatching {bad.float().mean().item():.4f}, max |d| {(out['wstat'] - out['igemm2']).abs().max().item():.3e}")
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("   bad rows: n =", rows.numel(), " row%32 histogram:", torch.bincount(rows % 32, minlength=32).tolist())
        print("   bad row blocks:", torch.unique(rows // 32)[:40].tolist())
        print("   bad cols: n =", cols.numel(), " col%32 histogram:", torch.bincount(cols % 32, minlength=32).tolist())
        print("   bad col groups of 32:", torch.unique(cols // 32)[:40].tolist())
        i = bad.nonzero()[0]
        print("   first:", i.tolist(), out["wstat"][i[0], i[1]].item(), out["igemm2"][i[0], i[1]].item())
        # is the wstat value present elsewhere in igemm2's row (a permutation)?
        rr = i[0].item()
        a, b = out["wstat"][rr], out["igemm2"][rr]
        srt = torch.equal(torch.sort(a).values, torch.sort(b).values)
        print("   row", rr, "is a permutation of the right row:", srt)
        if srt and n <= 1024:
            perm = [(b == v).nonzero().flatten()[:1].tolist() for v in a[:64]]
            print("   wstat col j holds igemm2 col:", [p[0] if p else -1 for p in perm])
