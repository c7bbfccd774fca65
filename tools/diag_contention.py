"""Determinism under contention: a launch sequence of 24 frames (backbone + one extraction head pass) repeated on the main stream while a
second stream runs the global-memory build's kernels (1800 x 1800 distances + farthest-point sweeps) back to back -- the situation of
the second launch sequence of a video's first call.  Outputs of every repetition against the first, bit for bit."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, ".")
from diffusionvid_amd import ops as dv  # noqa: E402
from diffusionvid_amd.utils import synthetic  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    blocks = (3, 4, 23, 3)
    sd = synthetic.make_state_dict(5, blocks=blocks)
    g = torch.Generator().manual_seed(0)
    imgs = torch.rand(n, 3, 608, 1024, generator=g).cuda()
    M = 300
    cx = torch.rand(n, M, 2, generator=g) * torch.tensor([1000., 600.])
    wh = torch.rand(n, M, 2, generator=g) * 300 + 8
    boxes = torch.cat([(cx - wh / 2).clamp_min(0), torch.minimum(cx + wh / 2, torch.tensor([999., 599.]))], dim=-1).cuda()
    t = torch.full((n,), 999, dtype=torch.int64)
    mem = torch.randn(1800, 256, generator=g).cuda()
    model = dv.Model(sd, res_blocks=blocks)
    model.reserve(n, 608, 1024, M)
    side = torch.cuda.Stream()
    if os.environ.get("SIDE", "").startswith("spin"):
        import ctypes
        parts = os.environ["SIDE"].split(":")
        spin_lds, spin_threads, spin_cycles, spin_mode = int(parts[1]), int(parts[2]), int(parts[3]), int(parts[4])
        spin_n = int(parts[5]) if len(parts) > 5 else 1
        spin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab", "libspin.so"))
        sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    d0 = dv.cdist(mem)
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    big2 = torch.empty_like(big)
    contend = os.environ.get("CONTEND", "1") != "0"

    def one():
        feats = model.backbone(imgs)
        outs = []
        pro = None
        for hi in range(3):          # the three extraction heads, each fed by the previous one, then the conditioned head
            lg, bx, obj = model.rcnn_head(hi, feats, 608, 1024, boxes if hi == 0 else bx.view(n, M, 4), pro, t)[:3]
            pro = obj
            outs += [lg, bx, obj]
        lg, bx, obj = model.rcnn_head(0, feats, 608, 1024, bx.view(n, M, 4), pro, t, cond=pro)[:3]
        outs += [lg, bx, obj]
        return [f.clone() for f in feats] + [o.clone() for o in outs]

    base = one()
    torch.cuda.synchronize()
    names = ("p3", "p4", "p5") + tuple("%s.%s" % (h, x) for h in ("head0", "head1", "head2", "cond") for x in ("logits", "boxes", "obj"))
    bad = 0
    for r in range(reps):
        if contend:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                kind = os.environ.get("SIDE", "both")
                if kind.startswith("spin"):          # SIDE=spin:<lds bytes>:<threads>:<cycles>:<mode> (tools/lab/spin_kernel.hip), several workgroups' worth
                    for _ in range(spin_n):
                        spin.spin_launch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), spin_lds, ctypes.c_long(spin_cycles), spin_threads,
                                         ctypes.c_void_p(sink.data_ptr()), spin_mode)
                for _ in range(0 if kind.startswith("spin") else 3):
                    if kind in ("both", "cdist"):
                        d = dv.cdist(mem)
                    if kind in ("both", "fps"):
                        dv.fps_greedy(d0, 900)
                    if kind == "copy":          # plain device copies: memory traffic without another kernel's workgroups
                        big2.copy_(big)
        got = one()
        torch.cuda.synchronize()
        for nm, a, b in zip(names, got, base):
            if not torch.equal(a, b):
                bad += 1
                diff = (a != b)
                fr = diff.reshape(n, -1).any(dim=1).nonzero().flatten().tolist()
                print("run %d: %s differs in %d values (max %.3e), frames %s" % (r, nm, int(diff.sum()), (a.float() - b.float()).abs().max().item(), fr[:10]), flush=True)
    print("contention %s, %d frames: %d runs, %d differing outputs" % (contend, n, reps, bad), flush=True)
    model.close()


if __name__ == "__main__":
    main()
