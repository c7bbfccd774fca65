"""Backbone features of a frame must not depend on how many frames share its launch: frames [0, 8) alone against the same frames inside
launches of 16 / 40 / 72 frames (two chains from 32), with the fused res2 / res3 blocks on and off."""
import sys

import torch

sys.path.insert(0, ".")
from diffusionvid_amd import _lib, ops as dv  # noqa: E402
from diffusionvid_amd.utils import synthetic  # noqa: E402


def main():
    lib = _lib.load()
    blocks = (3, 4, 2, 1)
    sd = synthetic.make_state_dict(5, blocks=blocks)
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (608, 1024)
    g = torch.Generator().manual_seed(0)
    imgs = torch.rand(72, 3, h, w, generator=g).cuda()
    model = dv.Model(sd, res_blocks=blocks)
    model.reserve(72, h, w, 300)
    for mode in (1, 0):
        _lib.check(lib.dvid_igemm_set_bottleneck_fusion(mode), "set")
        base = [t.clone() for t in model.backbone(imgs[:8])]
        for n in (16, 40, 72):
            got = model.backbone(imgs[:n])
            torch.cuda.synchronize()
            print("fusion %d: 8 frames alone vs inside %d:" % (mode, n),
                  " ".join("%s identical %.6f" % (nm, (a[:8] == b).float().mean().item()) for nm, a, b in zip(("p3", "p4", "p5"), got, base)), flush=True)
    lib.dvid_igemm_set_bottleneck_fusion(-1)
    model.close()


if __name__ == "__main__":
    main()
