"""Run-to-run determinism of the ResNet-FPN backbone at a two-chain launch size (full R-101 depth): the same 144 frames K times, p3 / p4 /
p5 compared bit for bit with the first run; with the fused res2 / res3 blocks on and off (dvid_igemm_set_bottleneck_fusion) and one or two chains."""
import sys

import torch

sys.path.insert(0, ".")
from diffusionvid_amd import _lib, ops as dv  # noqa: E402
from diffusionvid_amd.utils import synthetic  # noqa: E402


def main():
    lib = _lib.load()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 144
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    blocks = (3, 4, 23, 3)
    sd = synthetic.make_state_dict(5, blocks=blocks)
    g = torch.Generator().manual_seed(0)
    imgs = torch.rand(n, 3, 608, 1024, generator=g).cuda()
    model = dv.Model(sd, res_blocks=blocks)
    model.reserve(n, 608, 1024, 300)
    for mode in (1, 0):
        _lib.check(lib.dvid_igemm_set_bottleneck_fusion(mode), "set")
        for chains in (2, 1):
            _lib.check(lib.dvid_set_chains(model.handle, chains), "chains")
            base = None
            bad = 0
            for r in range(reps):
                got = [t.clone() for t in model.backbone(imgs)]
                torch.cuda.synchronize()
                if base is None:
                    base = got
                    continue
                for nm, a, b in zip(("p3", "p4", "p5"), got, base):
                    if not torch.equal(a, b):
                        d = (a != b)
                        fr = d.view(n, -1).any(dim=1).nonzero().flatten().tolist()
                        print("fusion %d chains %d run %d: %s differs in %d values, frames %s" % (mode, chains, r, nm, int(d.sum()), fr[:12]), flush=True)
                        bad += 1
            print("fusion %d chains %d: %d runs, %d differing outputs" % (mode, chains, reps, bad), flush=True)
    lib.dvid_igemm_set_bottleneck_fusion(-1)
    model.close()


if __name__ == "__main__":
    main()
