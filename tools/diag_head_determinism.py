"""Run-to-run determinism of one RCNNHead / RCNNHead_cond pass at several launch sizes: the same inputs R times, logits / boxes /
object features compared bit for bit with the first run (and the 8-frame launch's rows against the same frames inside larger ones)."""
import sys

import torch

sys.path.insert(0, ".")
from diffusionvid_amd import ops as dv  # noqa: E402
from diffusionvid_amd.utils import synthetic  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    sd = synthetic.make_state_dict(7, blocks=(1, 1, 1, 1))
    nmax, h, w, M = 128, 608, 1024, 300
    g = torch.Generator().manual_seed(3)
    feats = [torch.randn(nmax, h >> s, w >> s, 256, generator=g).half().cuda() for s in (3, 4, 5)]
    cx = torch.rand(nmax, M, 2, generator=g) * torch.tensor([1000., 600.])
    wh = torch.rand(nmax, M, 2, generator=g) * 300 + 8
    boxes = torch.cat([(cx - wh / 2).clamp_min(0), torch.minimum(cx + wh / 2, torch.tensor([999., 599.]))], dim=-1).cuda()
    pro = torch.randn(nmax * M, 256, generator=g).cuda()
    cond = torch.randn(nmax * M, 256, generator=g).cuda()
    model = dv.Model(sd, res_blocks=(1, 1, 1, 1))
    model.reserve(nmax, h, w, M)
    small = {}
    for n in (8, 24, 104, 128):
        t = torch.full((n,), 999, dtype=torch.int64)
        f = [x[:n] for x in feats]
        for name, hi, c in (("head0", 0, None), ("head2", 2, None), ("cond", 0, cond[:n * M])):
            base = None
            bad = 0
            for r in range(reps):
                out = model.rcnn_head(hi, f, h, w, boxes[:n], pro[:n * M], t, cond=c)
                torch.cuda.synchronize()
                out = [o.clone() for o in out[:3]]
                if base is None:
                    base = out
                    continue
                for nm, a, b in zip(("logits", "boxes", "obj"), out, base):
                    if not torch.equal(a, b):
                        bad += 1
                        d = (a != b).view(n, -1).any(dim=1).nonzero().flatten().tolist()
                        print("%s n=%d run %d: %s differs in %d values (max %.3e), frames %s" % (name, n, r, nm, int((a != b).sum()),
                              (a - b).abs().max().item(), d[:10]), flush=True)
            if n == 8:
                small[name] = base
            else:
                same = all(torch.equal(a[:8 * (a.shape[0] // n)] if a.dim() == 2 else a[:8], b) for a, b in zip(base, small[name]))
                print("%s n=%d: frames 0-7 equal to the 8-frame launch: %s" % (name, n, same), flush=True)
            print("%s n=%d: %d runs, %d differing outputs" % (name, n, reps, bad), flush=True)
    model.close()


if __name__ == "__main__":
    main()
