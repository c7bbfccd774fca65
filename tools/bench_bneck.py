"""Stand-alone timing of the fused res2 block tail (csrc/bneck.hip) against the layer-by-layer launches on the bench's res2 map
(152 x 256 per 608 x 1024 frame):  python tools/bench_bneck.py [frames]   -> one line per variant, ms per launch (chain) and
the HBM rate of the bytes each form touches once."""
import sys

import torch

sys.path.insert(0, ".")
from diffusionvid_amd import ops as dv  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 304
    hh, ww = 152, 256
    g = torch.Generator().manual_seed(0)
    px = n * hh * ww
    x256 = (torch.randn(px, 256, generator=g, dtype=torch.float16)).cuda().view(n, hh, ww, 256)
    x64 = x256[..., :64].contiguous()
    t1 = x256[..., 64:128].clamp_min(0).contiguous()
    w2 = torch.randn(64, 64, 3, 3, generator=g) * (1.5 / 24)
    w3, wsc = (torch.randn(256, 64, generator=g) * (1.5 / 8) for _ in range(2))
    w1n = torch.randn(64, 256, generator=g) * (1.5 / 16)
    (w2p, k2), (w3p, k3), (wsp, ks), (w1p, k1) = (dv.pack_conv_weight(w) for w in (w2, w3, wsc, w1n))
    w2d, w3d, wsd, w1d = w2p.cuda(), w3p.cuda(), wsp.cuda(), w1p.cuda()
    b2, b3, bs, b1 = (torch.randn(c, generator=g).cuda() * 0.3 for c in (64, 256, 256, 64))
    w1n128 = torch.randn(128, 256, generator=g) * (1.5 / 16)
    w1p128, k1b = dv.pack_conv_weight(w1n128)
    w1d128, b1128 = w1p128.cuda(), torch.randn(128, generator=g).cuda() * 0.3
    for sc, tail in ((False, True), (False, False), (True, True), (False, 128)):
        xin = x64 if sc else x256
        if tail == 128:
            w1d, b1, k1 = w1d128, b1128, k1b

        def layers():
            t2 = dv.conv2d_nhwc(t1, w2d, k2, b2, 64, 3, 3, 1, 1, relu=True)
            res = dv.conv2d_nhwc(xin, wsd, ks, bs, 256, 1, 1, 1, 0) if sc else xin
            out = dv.conv2d_nhwc(t2, w3d, k3, b3, 256, 1, 1, 1, 0, relu=True, residual=res, residual_mode=1)
            return out, (dv.conv2d_nhwc(out, w1d, k1, b1, int(w1d.shape[0]), 1, 1, 1, 0, relu=True) if tail else None)

        def fused():
            return dv.bottleneck64_tail(t1, w2d, b2, w3d, b3, xin, wsd if sc else None, bs if sc else None, w1d if tail else None,
                                        b1 if tail else None)

        ol, tl = layers()
        of, tf = fused()
        torch.cuda.synchronize()
        same = torch.equal(ol, of) and (not tail or torch.equal(tl, tf))
        del ol, tl, of, tf
        ms_l, ms_f = timed(layers), timed(fused)
        nn = int(w1d.shape[0]) if tail else 0
        by_f = px * 2.0 * (64 + (64 if sc else 256) + 256 + nn)
        by_l = px * 2.0 * (64 + 64 + 64 + 256 + 256 + (64 + 256 if sc else 256) + ((256 + nn) if tail else 0))
        print("res2 frames %d shortcut %d next_conv1 %d: layer by layer %.3f ms (%.0f GB/s of its bytes), one launch %.3f ms (%.0f GB/s of its "
              "bytes), x%.2f, identical %s" % (n, sc, tail, ms_l, by_l / ms_l / 1e6, ms_f, by_f / ms_f / 1e6, ms_l / ms_f, same), flush=True)


def main128():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 304
    hh, ww = 76, 128
    g = torch.Generator().manual_seed(1)
    px = n * hh * ww
    res = torch.randn(px, 512, generator=g, dtype=torch.float16).cuda().view(n, hh, ww, 512)
    t1 = res[..., :128].clamp_min(0).contiguous()
    w2 = torch.randn(128, 128, 3, 3, generator=g) * (1.5 / 34)
    w3 = torch.randn(512, 128, generator=g) * (1.5 / 11.3)
    w1n = torch.randn(128, 512, generator=g) * (1.5 / 22.6)
    (w2p, k2), (w3p, k3), (w1p, k1) = (dv.pack_conv_weight(w) for w in (w2, w3, w1n))
    w2d, w3d, w1d = w2p.cuda(), w3p.cuda(), w1p.cuda()
    b2, b3, b1 = (torch.randn(c, generator=g).cuda() * 0.3 for c in (128, 512, 128))
    for c2, tail in ((True, True), (True, False), (False, True)):
        def layers():
            t2 = dv.conv2d_nhwc(t1, w2d, k2, b2, 128, 3, 3, 1, 1, relu=True) if c2 else t1
            out = dv.conv2d_nhwc(t2, w3d, k3, b3, 512, 1, 1, 1, 0, relu=True, residual=res, residual_mode=1)
            return out, (dv.conv2d_nhwc(out, w1d, k1, b1, 128, 1, 1, 1, 0, relu=True) if tail else None)

        def fused():
            return dv.bottleneck128_tail(t1, w2d if c2 else None, b2 if c2 else None, w3d, b3, res, w1d if tail else None, b1 if tail else None)

        ol, tl = layers()
        of, tf = fused()
        torch.cuda.synchronize()
        close = (ol.float() - of.float()).abs().max().item()
        del ol, tl, of, tf
        ms_l, ms_f = timed(layers), timed(fused)
        by_f = px * 2.0 * (128 + 512 + 512 + (128 if tail else 0))
        print("res3 frames %d conv2 %d next_conv1 %d: layer by layer %.3f ms, one launch %.3f ms (%.0f GB/s of its bytes), x%.2f, max |diff| "
              "to the patch-kernel chain %.3e" % (n, c2, tail, ms_l, ms_f, by_f / ms_f / 1e6, ms_l / ms_f, close), flush=True)


if __name__ == "__main__":
    main128()
    main()
