"""The fused block kernels stand-alone, repeated while a second stream runs farthest-point sweeps (a single long-lived workgroup that
shares a CU with whatever else is scheduled there): outputs of every repetition against the first, bit for bit."""
import sys

import torch

sys.path.insert(0, ".")
from diffusionvid_amd import ops as dv  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    n, hh, ww = 24, 152, 256
    g = torch.Generator().manual_seed(0)
    x256 = torch.randn(n, hh, ww, 256, generator=g, dtype=torch.float16).cuda()
    t1 = x256[..., 64:128].clamp_min(0).contiguous()
    mk = lambda *s, sc=0.1: torch.randn(*s, generator=g) * sc
    (w2p, _), (w3p, _), (w1p64, _), (w1p128, _) = (dv.pack_conv_weight(w) for w in (mk(64, 64, 3, 3), mk(256, 64), mk(64, 256), mk(128, 256)))
    w2d, w3d, w1d64, w1d128 = w2p.cuda(), w3p.cuda(), w1p64.cuda(), w1p128.cuda()
    b2, b3, b164, b1128 = (mk(c, sc=0.3).cuda() for c in (64, 256, 64, 128))
    mem = torch.randn(1800, 256, generator=g).cuda()
    d0 = dv.cdist(mem)
    side = torch.cuda.Stream()
    cases = {"next64": lambda: dv.bottleneck64_tail(t1, w2d, b2, w3d, b3, x256, None, None, w1d64, b164),
             "next128": lambda: dv.bottleneck64_tail(t1, w2d, b2, w3d, b3, x256, None, None, w1d128, b1128),
             "none": lambda: dv.bottleneck64_tail(t1, w2d, b2, w3d, b3, x256)}
    for name, fn in cases.items():
        base = [o.clone() for o in fn() if o is not None]
        torch.cuda.synchronize()
        bad = 0
        for r in range(reps):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dv.fps_greedy(d0, 900)
            outs = []
            for _ in range(6):          # several launches beside one sweep
                outs.append([o.clone() for o in fn() if o is not None])
            torch.cuda.synchronize()
            for k, got in enumerate(outs):
                for nm, a, b in zip(("out", "t1_next"), got, base):
                    if not torch.equal(a, b):
                        bad += 1
                        d = a != b
                        rows = d.reshape(n * hh, -1).any(dim=1).nonzero().flatten()
                        print("%s run %d.%d: %s differs in %d values (max %.3e); batch rows %d..%d, %d rows" % (
                            name, r, k, nm, int(d.sum()), (a.float() - b.float()).abs().max().item(), int(rows.min()), int(rows.max()), rows.numel()), flush=True)
        print("%s: %d x 6 launches, %d differing outputs" % (name, reps, bad), flush=True)


if __name__ == "__main__":
    main()
