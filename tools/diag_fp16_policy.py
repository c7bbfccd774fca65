#!/usr/bin/env python
"""How closely does oracle/precision.py's "fp16" policy mirror the HIP path's storage rounding?  Stage by stage, GPU vs
fp16-policy oracle on identical inputs: fraction of outputs that are bit-equal after rounding, and the error tail.
Measurement aid (GPU box); the tests that gate on these numbers are in tests/test_gpu_kernels.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd import ops  # noqa: E402
from diffusionvid_amd.utils import synthetic  # noqa: E402
from oracle import backbone_r101, head as ohead, precision, schedule as osch  # noqa: E402


def stats(name, got, ref, as_f16=False):
    got, ref = got.float().cpu(), ref.float().cpu()
    if as_f16:
        eq = (got.half() == ref.half()).float().mean().item()
    else:
        eq = (got == ref).float().mean().item()
    e = (got - ref).abs().flatten()
    rms = ref.pow(2).mean().sqrt().item()
    k = min(e.numel(), 2000000)
    es = e[torch.randperm(e.numel())[:k]]
    print(f"{name:36s} rms {rms:9.3e}  equal {eq:.5f}  |err| median {es.median():.2e} p99 {es.quantile(0.99):.2e} "
          f"p99.9 {es.quantile(0.999):.2e} max {e.max():.2e}", flush=True)


def main():
    torch.set_num_threads(32)
    g = torch.Generator().manual_seed(14)
    blocks = (1, 2, 2, 1)
    sd = synthetic.make_state_dict(0, blocks=blocks)
    imgs = torch.rand(2, 3, 128, 192, generator=g)
    mean, std = (123.675, 116.280, 103.530), (58.395, 57.120, 57.375)
    model = ops.Model(sd, res_blocks=blocks)
    model.reserve(2, 160, 256, 300)
    p = model.backbone(imgs.cuda())
    with precision.use("fp16"):
        ref = backbone_r101.backbone_r101_fpn(backbone_r101.normalizer(imgs, mean, std), sd, "backbone.", blocks)
    ref32 = backbone_r101.backbone_r101_fpn(backbone_r101.normalizer(imgs, mean, std), sd, "backbone.", blocks)
    for name, got in zip(("p3", "p4", "p5"), p):
        stats("backbone " + name + " vs fp16-policy", ops.nchw_from_nhwc(got), ref[name], as_f16=True)
        stats("backbone " + name + " vs fp32 oracle", ops.nchw_from_nhwc(got), ref32[name])
    # one head, identical inputs
    n, M, H, W = 2, 300, 160, 256
    feats = [(torch.randn(n, 256, H // s, W // s, generator=g) * 0.5).half().float() for s in (8, 16, 32)]
    cxcy = torch.rand(n, M, 2, generator=g) * torch.tensor([W, H])
    wh = torch.exp(torch.rand(n, M, 2, generator=g) * 4.5 + 0.8)
    boxes = torch.cat([cxcy - wh / 2, cxcy + wh / 2], dim=-1)
    t = torch.tensor([999, 499], dtype=torch.long)
    time = osch.time_mlp(sd, "head.", t, 256)
    fd = [ops.nhwc_from_nchw(f.cuda()) for f in feats]
    cfg = ohead.HeadCfg()
    for tag, pfx, idx, cond in (("head0 (pro=None)", "head.head_series.0", 0, False), ("head1", "head.head_series.1", 1, False),
                                ("cond head", "head.head_series_cond.0", 0, True)):
        pro = None if idx == 0 and not cond else torch.randn(1, n * M, 256, generator=g)
        cnd = torch.randn(n * M, 256, generator=g) if cond else None
        taps = {}
        with precision.use("fp16"):
            cl, bx, of = ohead.rcnn_head(sd, pfx, feats, boxes, pro, time, cfg, cond=cnd, taps=taps)
        cl32, bx32, of32 = ohead.rcnn_head(sd, pfx, feats, boxes, pro, time, cfg, cond=cnd)
        gl, gb, go = model.rcnn_head(idx, fd, H, W, boxes.cuda(), None if pro is None else pro[0].cuda(), t,
                                     cond=None if cnd is None else cnd.cuda())
        stats(tag + " obj_features vs fp16-policy", go, of[0])
        stats(tag + " obj_features vs fp32", go, of32[0])
        stats(tag + " logits vs fp16-policy", gl, cl)
        stats(tag + " logits vs fp32", gl, cl32)
        stats(tag + " boxes vs fp16-policy", gb, bx)
        stats(tag + " boxes vs fp32", gb, bx32)
    # global attention
    q = torch.randn(600, 256, generator=g)
    mem = torch.randn(900, 256, generator=g)
    with precision.use("fp16"):
        r16 = ohead.global_attention(sd, "head.", q[None], [mem, None], cfg)
    r32 = ohead.global_attention(sd, "head.", q[None], [mem, None], cfg)
    out = model.global_xattn(q.cuda(), mem.cuda())
    stats("global_xattn vs fp16-policy", out, r16)
    stats("global_xattn vs fp32", out, r32)


if __name__ == "__main__":
    main()
