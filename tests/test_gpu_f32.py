"""DTYPE float32 (the reference's default precision, mega_core/config/defaults.py:582): per-kernel and per-stage parity of csrc/f32.hip
against the fp32 CPU oracle / plain torch fp32 on identical un-rounded inputs, through the C ABI.

Tolerances.  Both sides are fp32 fmaf chains that differ in summation order only (MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 is
bit-for-bit a k-ordered fmaf chain): per layer ~1e-6 of the output's RMS, a few 1e-6 through a chain of layers.  Bounds below are
2e-5 of RMS + 2e-5 relative for single kernels and 2e-4 for the ~15-layer head / ~20-layer backbone chains -- two orders of
magnitude inside what the fp16 path is allowed (tests/test_gpu_kernels.py).
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import backbone_r101, head as ohead, roi_align as oroi, schedule as osch  # noqa: E402
from test_gpu_kernels import _boxes, check  # noqa: E402


@pytest.fixture(scope="module")
def dv():
    from diffusionvid_amd import ops
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return ops


def nhwc(x, pad_to=None):
    x = x.permute(0, 2, 3, 1).contiguous()
    if pad_to and x.shape[-1] < pad_to:
        x = F.pad(x, (0, pad_to - x.shape[-1]))
    return x.cuda()


@pytest.mark.parametrize("cfg", [
    dict(n=2, h=20, w=28, cin=64, cout=128, k=3, stride=1, pad=1, relu=1, res=1),
    dict(n=2, h=21, w=27, cin=128, cout=64, k=3, stride=2, pad=1, relu=0, res=0),          # ragged M, BN 64 tiles
    dict(n=3, h=16, w=24, cin=256, cout=256, k=1, stride=1, pad=0, relu=1, res=1),
    dict(n=1, h=16, w=24, cin=128, cout=512, k=1, stride=2, pad=0, relu=0, res=0),          # strided 1x1 (shortcut)
    dict(n=1, h=12, w=16, cin=512, cout=256, k=1, stride=1, pad=0, relu=0, res=2),          # FPN lateral + nearest-x2 top-down sum
    dict(n=2, h=64, w=96, cin=3, cout=64, k=7, stride=2, pad=3, relu=1, res=0),             # the stem: 3 channels padded to 4, K 196 -> 208
    dict(n=1, h=9, w=11, cin=256, cout=30, k=1, stride=1, pad=0, relu=0, res=0),            # N tail (class_logits)
    dict(n=1, h=5, w=7, cin=64, cout=130, k=3, stride=1, pad=1, relu=2, res=0),             # exact GELU, N tail across two tiles
    # 3x3 / stride 1 without a residual: csrc/f32_conv3x3.hip (halo staged once per 32-channel chunk) when split = 1
    dict(n=3, h=19, w=32, cin=256, cout=256, k=3, stride=1, pad=1, relu=1, res=0),          # res5-like map: 19 rows = two full patches + 3 rows
    dict(n=2, h=9, w=70, cin=32, cout=64, k=3, stride=1, pad=1, relu=1, res=0),             # BN 64, ragged width (2 x 32 + 6), one chunk
    dict(n=2, h=8, w=32, cin=128, cout=96, k=3, stride=1, pad=1, relu=0, res=0),            # N tail inside one tile
    dict(n=1, h=38, w=64, cin=64, cout=64, k=3, stride=1, pad=1, relu=1, res=0),
])
@pytest.mark.parametrize("split", [1, 0])
def test_f32_conv(dv, cfg, split):
    """split = 1 (default): (hi, lo) fp16 operands, three passes of the fp16 MFMA; 0: exact fp32 products on the fp32 MFMA.  Same bound."""
    g = torch.Generator().manual_seed(1)
    n, h, w, cin, cout, k = cfg["n"], cfg["h"], cfg["w"], cfg["cin"], cfg["cout"], cfg["k"]
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), wt.double(), bias.double(), stride=cfg["stride"], padding=cfg["pad"])
    res = None
    if cfg["res"] == 1:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res.double()
    elif cfg["res"] == 2:
        res = torch.randn(n, cout, ref.shape[2] // 2, ref.shape[3] // 2, generator=g)
        ref = ref + F.interpolate(res.double(), scale_factor=2.0, mode="nearest")
    if cfg["relu"] == 1:
        ref = F.relu(ref)
    elif cfg["relu"] == 2:
        ref = F.gelu(ref)
    wp, kpad, rs = dv.pack_conv_weight_f32(wt, scale_rows=True)
    ws = tuple(t.cuda() for t in dv.split_f16(wp))
    dv.set_option("f32_split", split)
    try:
        out = dv.conv2d_nhwc_f32(nhwc(x, (cin + 3) // 4 * 4), wp.cuda(), kpad, bias.cuda(), cout, k, k, cfg["stride"], cfg["pad"], relu=cfg["relu"],
                                 residual=nhwc(res) if res is not None else None, residual_mode=cfg["res"], row_scale=rs.cuda(), w_split=ws)
    finally:
        dv.reset_options()
    check(f"f32_conv[split {split}]{cfg}", out.permute(0, 3, 1, 2), ref.float(), 2e-5, 2e-5)


@pytest.mark.parametrize("rows,k,nout", [(600, 256, 768), (300, 256, 4), (257, 12544, 256), (600, 256, 32768), (1, 1024, 256)])
@pytest.mark.parametrize("split", [1, 0])
def test_f32_linear(dv, rows, k, nout, split):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, k, generator=g)
    wt = torch.randn(nout, k, generator=g) / math.sqrt(k)
    bias = torch.randn(nout, generator=g)
    ref = F.linear(x.double(), wt.double(), bias.double()).float()
    wp, kpad, rs = dv.pack_conv_weight_f32(wt, scale_rows=True)
    ws = tuple(t.cuda() for t in dv.split_f16(wp))
    dv.set_option("f32_split", split)
    try:
        out = dv.linear_f32(x.cuda(), wp.cuda(), kpad, bias.cuda(), row_scale=rs.cuda(), w_split=ws)
        # small values too: the split keeps an absolute 3e-8 per operand below |v| = 2^-3, so a tensor of 1e-3-sized activations still meets the bound
        out_small = dv.linear_f32((x * 1e-3).cuda(), wp.cuda(), kpad, None, row_scale=rs.cuda(), w_split=ws)
    finally:
        dv.reset_options()
    check(f"f32_linear[split {split}][{rows}x{k}->{nout}]", out, ref, 2e-5, 2e-5)
    check(f"f32_linear[split {split}][{rows}x{k}->{nout}] x 1e-3", out_small, (ref - bias) * 1e-3, 1e-4, 1e-4)


@pytest.mark.parametrize("rows,k,nout,relu,res", [
    (2400 + 17, 256, 1024, 1, 1),          # res4 conv3 + residual + ReLU; ragged tail of 17 rows on the tiled kernel
    (4096, 128, 512, 1, 1),                # res3 conv3
    (2400, 256, 32768, 0, 0),              # dynamic_layer: 128 slabs, every workgroup walks four
    (1504, 256, 2048, 1, 0),               # linear1
    (3200, 128, 512, 2, 0),                # Swin fc1: exact GELU
    (2080, 256, 768, 0, 0),                # three slabs (Swin qkv): 30 of an XCD's 32 workgroups work
    (32, 256, 256, 0, 1),                  # one row block: 255 workgroups have nothing to do
    (4096 + 40, 64, 256, 1, 1),            # res2 conv3: K = 64 runs tiles of 64 rows; tail of 40 rows on the tiled kernel
    (64, 64, 512, 0, 0),
])
def test_f32_wstat_matches_tiled(dv, rows, k, nout, relu, res):
    """csrc/f32_wstat.hip (weight-stationary split-operand kernel) against csrc/f32.hip's tiled split-operand kernel: the same split, the
    same three products per K step in the same order, the same epilogue arithmetic -- bit for bit (and both inside the fp32 bound)."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(rows, k, generator=g)
    wt = torch.randn(nout, k, generator=g) / math.sqrt(k)
    bias = torch.randn(nout, generator=g)
    resid = torch.randn(rows, nout, generator=g) if res else None
    ref = F.linear(x.double(), wt.double(), bias.double())
    if res:
        ref = ref + resid.double()
    ref = F.relu(ref) if relu == 1 else F.gelu(ref) if relu == 2 else ref
    wp, kpad, rs = dv.pack_conv_weight_f32(wt, scale_rows=True)
    ws = tuple(t.cuda() for t in dv.split_f16(wp))
    outs = {}
    for mode in (0, 2):
        dv.set_option("f32_wstat", mode)
        try:
            outs[mode] = dv.conv2d_nhwc_f32(x.view(rows, 1, 1, k).cuda(), wp.cuda(), kpad, bias.cuda(), nout, 1, 1, 1, 0, relu=relu,
                                            residual=resid.view(rows, 1, 1, nout).cuda() if res else None, residual_mode=res, row_scale=rs.cuda(),
                                            w_split=ws).view(rows, nout)
        finally:
            dv.reset_options()
    check(f"f32_wstat[{rows}x{k}->{nout}]", outs[2], ref.float(), 2e-5, 2e-5)
    assert torch.equal(outs[0], outs[2]), f"weight-stationary and tiled kernels differ: max |d| {(outs[0] - outs[2]).abs().max().item():.3e}"


def test_f32_roialign(dv):
    """zero-area, oversize and edge boxes, all three levels; against oracle/roi_align.py on the same fp32 maps"""
    g = torch.Generator().manual_seed(4)
    n, M, H, W = 2, 300, 160, 256
    feats = [torch.randn(n, 256, H // s, W // s, generator=g) for s in (8, 16, 32)]
    boxes = _boxes(g, n, M, H, W)
    ref = oroi.roi_pooler(feats, boxes, 7, (1 / 8., 1 / 16., 1 / 32.), 2)          # [n*M, 256, 7, 7]
    roi, mean = dv.roialign_f32([nhwc(f) for f in feats], boxes.cuda(), H, W, want_mean=True)
    check("f32_roialign.tiles", roi.view(n * M, 49, 256).permute(0, 2, 1), ref.reshape(n * M, 256, 49), 1e-5, 1e-5)
    check("f32_roialign.mean", mean, ref.reshape(n * M, 256, 49).mean(-1), 1e-5, 1e-5)


@pytest.mark.parametrize("B,lq,lk", [(2, 300, 300), (1, 777, 900), (1, 64, 37), (3, 17, 1)])
def test_f32_mha(dv, B, lq, lk):
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, L, 256, generator=g) for L in (lq, lk, lk))
    qh, kh, vh = (t.double().view(B, -1, 8, 32).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(32.0), dim=-1)
    ref = (p @ vh).transpose(1, 2).reshape(B, lq, 256).float()
    out = dv.mha_f32(q.cuda(), k.cuda(), v.cuda(), 8)
    check(f"f32_mha[{B},{lq},{lk}]", out, ref, 2e-5, 2e-5)


def _head_state(seed=0):
    from diffusionvid_amd.utils import synthetic
    sd = synthetic.make_head_state_dict(seed)
    return sd, {k: v.float() for k, v in sd.items()}


@pytest.mark.parametrize("split", [1, 0])
def test_f32_dynconv(dv, split):
    """box_head.py:687-711 up to (not including) out_layer, on un-rounded parameters.  split = 1 (default): both per-box products on split
    (hi, lo) fp16 operands (f32x3_dynconv_kernel); 0: on the fp32 MFMA.  Same bound."""
    sd, _ = _head_state()
    g = torch.Generator().manual_seed(6)
    R, d, dd = 77, 256, 64
    roi = torch.randn(R, 49, d, generator=g)
    params = torch.randn(R, 2 * d * dd, generator=g) / 8.0
    pfx = "head.head_series.0.inst_interact"
    p1 = params[:, :d * dd].view(R, d, dd)
    p2 = params[:, d * dd:].view(R, dd, d)
    f = torch.bmm(roi.double(), p1.double())
    f = F.relu(F.layer_norm(f, (dd,), sd[pfx + ".norm1.weight"].double(), sd[pfx + ".norm1.bias"].double()))
    f = torch.bmm(f, p2.double())
    ref = F.relu(F.layer_norm(f, (d,), sd[pfx + ".norm2.weight"].double(), sd[pfx + ".norm2.bias"].double())).float()
    packed = torch.cat([p1.transpose(1, 2).reshape(R, -1), p2.transpose(1, 2).reshape(R, -1)], dim=1).contiguous()      # P1T | P2T (model.hip: make_head)
    dv.set_option("f32_split", split)
    try:
        out = dv.dynconv_f32(roi.cuda(), packed.cuda(), *(sd[pfx + k].cuda() for k in (".norm1.weight", ".norm1.bias", ".norm2.weight", ".norm2.bias")))
    finally:
        dv.reset_options()
    check(f"f32_dynconv[split {split}]", out, ref, 3e-5, 3e-5)


@pytest.mark.parametrize("cond", [False, True])
def test_f32_rcnn_head(dv, cond):
    """One RCNNHead / RCNNHead_cond pass (box_head.py:495-548, :605-664) with DTYPE float32 against the fp32 oracle: the same test as
    test_gpu_kernels.py::test_rcnn_head with bounds 100 x tighter (2e-4 against 2e-2)."""
    sd, sdo = _head_state()
    g = torch.Generator().manual_seed(7)
    n, M, H, W = 2, 300, 160, 256
    feats = [torch.randn(n, 256, H // s, W // s, generator=g) * 0.5 for s in (8, 16, 32)]
    boxes = _boxes(g, n, M, H, W)
    boxes[0, 0] = torch.tensor([10.0, 10.0, 14.0, 13.0])
    cfg = ohead.HeadCfg()
    t = torch.tensor([999, 499], dtype=torch.long)
    time = osch.time_mlp(sdo, "head.", t, 256)
    pfx = "head.head_series_cond.0" if cond else "head.head_series.1"
    pro = torch.randn(1, n * M, 256, generator=g)
    cnd = torch.randn(n * M, 256, generator=g) if cond else None
    cl, bx, of = ohead.rcnn_head(sdo, pfx, feats, boxes, pro, time, cfg, cond=cnd)
    model = dv.Model(sd, res_blocks=(0, 0, 0, 0), precision="float32")
    model.reserve(n, H, W, M)
    fd = [nhwc(f) for f in feats]
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    gl, gb, go = model.rcnn_head(0 if cond else 1, fd, H, W, boxes.cuda(), pro[0].cuda(), t, cond=None if cnd is None else cnd.cuda(), bad_flag=flag)
    tag = "cond" if cond else "plain"
    check(f"f32_rcnn_head[{tag}].obj_features", go, of[0], 2e-4, 2e-4)
    check(f"f32_rcnn_head[{tag}].logits", gl, cl, 2e-4, 2e-4)
    bw = (boxes[..., 2:] - boxes[..., :2]).clamp(min=1.0).max(-1).values
    err = ((gb.cpu() - bx).abs().max(-1).values / bw).max().item()
    print(f"f32_rcnn_head[{tag}].boxes rel-to-size err max={err:.3e}")
    assert err < 3e-4
    assert int(flag.item()) == 0
    # first head: pro_features None -> mean of the RoI tiles
    cl0, bx0, of0 = ohead.rcnn_head(sdo, "head.head_series.0", feats, boxes, None, time, cfg)
    gl0, gb0, go0 = model.rcnn_head(0, fd, H, W, boxes.cuda(), None, t)
    check(f"f32_rcnn_head[{tag}].first.obj_features", go0, of0[0], 2e-4, 2e-4)
    check(f"f32_rcnn_head[{tag}].first.logits", gl0, cl0, 2e-4, 2e-4)
    # an fp16 map handed to an fp32 model is refused, not reinterpreted
    from diffusionvid_amd._lib import DvidError
    with pytest.raises(DvidError):
        model.rcnn_head(0, [f.half() for f in fd], H, W, boxes.cuda(), None, t)
    model.close()


def test_f32_head_against_full_dimension_reference_fixture(dv):
    """DTYPE float32 against the REFERENCE's own RCNNHead / RCNNHead_cond at the kernels' dimensions, no oracle in between: golden g16
    (tests/golden/make_golden.py: g16_full_dim_head) holds inputs and fp32 outputs of the reference modules run in fp32 on
    synthetic.make_head_state_dict(0) with fp16-representable matrices (box_head.py:495-548, :605-664).  The fp16 path is held to
    2e-3 .. 6e-3 on this fixture (test_gpu_kernels.py); with fp32 storage and products the logits agree to 2e-4 of their RMS and the
    boxes to 3e-4 of the box size -- what is left is the fp32 summation order.  (The fixture's pooler is oracle.roi_align standing in
    for detectron2's; tests/test_known_answers.py pins that function and the kernels against closed-form answers.)"""
    from conftest import golden
    from diffusionvid_amd.utils import synthetic
    z = golden("g16_full_dim_head")
    n, M, H, W = (int(z[k]) for k in ("n", "M", "H", "W"))
    sd = {k: (v.half().float() if v.dim() > 1 else v.clone()) for k, v in synthetic.make_head_state_dict(int(z["weights_seed"])).items()}
    T32 = lambda k: torch.from_numpy(z[k].astype(np.float32))
    fd = [nhwc(T32(k)) for k in ("p3", "p4", "p5")]
    t = torch.from_numpy(z["t"])
    model = dv.Model(sd, res_blocks=(0, 0, 0, 0), precision="float32")
    model.reserve(n, H, W, M)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    stages = (("head_series.0", 0, False, "boxes", None, "0"), ("head_series.1", 1, False, "bx0", "of0", "1"), ("head_series_cond.0", 0, True, "bx1", "of1", "2"))
    for name, idx, is_cond, kb, kf, o in stages:
        bin_ = T32(kb)
        pro = None if kf is None else T32(kf)[0].cuda()
        gl, gb, go = model.rcnn_head(idx, fd, H, W, bin_.cuda(), pro, t, cond=T32("cond").cuda() if is_cond else None, bad_flag=flag)
        check(f"f32_reference_fixture[{name}].logits", gl, T32("cl" + o), 2e-4, 2e-4)
        check(f"f32_reference_fixture[{name}].obj_features", go, T32("of" + o)[0], 1.5e-3, 1.5e-3)          # the fixture stores them as fp16
        bw = (bin_[..., 2:] - bin_[..., :2]).clamp(min=1.0).max(-1).values
        err = ((gb.cpu() - T32("bx" + o)).abs().max(-1).values / bw).max().item()
        print(f"f32_reference_fixture[{name}]: boxes rel-to-size err max={err:.3e}")
        assert err < 3e-4, name
    assert int(flag.item()) == 0
    model.close()


def test_f32_global_xattn(dv):
    sd, sdo = _head_state()
    g = torch.Generator().manual_seed(8)
    rows, lk = 600, 900
    q = torch.randn(1, rows, 256, generator=g)
    mem = torch.randn(lk, 256, generator=g)
    ref = ohead.global_attention(sdo, "head.", q, [mem, None], ohead.HeadCfg())
    model = dv.Model(sd, res_blocks=(0, 0, 0, 0), precision="float32")
    model.reserve(2, 64, 64, 300)
    memd = mem.cuda()
    out = model.global_xattn(q[0].cuda(), memd)
    check("f32_global_xattn", out, ref, 1e-4, 1e-4)
    model.close()


def test_f32_backbone_small(dv):
    """Reduced-depth ResNet-FPN on 2 frames of 128 x 192 with DTYPE float32 against the fp32 oracle (the fp16 path's bound here is 3e-2)."""
    from diffusionvid_amd.utils import synthetic
    blocks = (1, 2, 2, 1)
    sd = synthetic.make_state_dict(0, blocks=blocks)
    g = torch.Generator().manual_seed(14)
    imgs = torch.rand(2, 3, 128, 192, generator=g)
    cfg_mean, cfg_std = (123.675, 116.280, 103.530), (58.395, 57.120, 57.375)
    ref = backbone_r101.backbone_r101_fpn(backbone_r101.normalizer(imgs, cfg_mean, cfg_std), sd, "backbone.", blocks)
    model = dv.Model(sd, res_blocks=blocks, precision="float32")
    model.reserve(2, 128, 192, 300)
    p3, p4, p5 = model.backbone(imgs.cuda())
    assert p3.dtype == torch.float32
    for name, got in (("p3", p3), ("p4", p4), ("p5", p5)):
        check(f"f32_backbone_small.{name}", dv.nchw_from_nhwc(got), ref[name], 2e-4, 2e-4)
    # the pointer-table entry point gives the same maps
    q3, _, _ = model.backbone_frames([imgs[i:i + 1].cuda() for i in range(2)])
    assert torch.equal(q3, p3)
    model.close()


def test_f32_backbone_swin_small(dv):
    """Swin-Transformer + FPN with DTYPE float32 (reduced widths / depths; odd token maps, padded windows, shifted-window masks, GELU MLP,
    PatchMerging) against the fp32 oracle (oracle/swin.py, itself bit-exact against the reference's module: golden g8).  The fp16 path's
    bound on the same test is 3e-2."""
    from diffusionvid_amd.utils import synthetic
    from oracle import swin as oswin
    sw = dict(embed_dim=64, depths=(2, 2, 2, 1), heads=(2, 4, 8, 16), window=7)
    sd = synthetic.make_state_dict(0, swin=sw)
    g = torch.Generator().manual_seed(15)
    imgs = torch.rand(2, 3, 160, 224, generator=g)        # tokens 40x56 -> 20x28 -> 10x14 -> 5x7 (pads to 42x56, 21x28, 14x14, 7x7)
    mean, std = (123.675, 116.280, 103.530), (58.395, 57.120, 57.375)
    ref = oswin.backbone_swin_fpn(backbone_r101.normalizer(imgs, mean, std), sd, "backbone.", embed_dim=64, depths=sw["depths"], num_heads=sw["heads"])
    model = dv.Model(sd, res_blocks=(0, 0, 0, 0), backbone="swin", swin_embed_dim=64, swin_depths=sw["depths"], swin_heads=sw["heads"], precision="float32")
    model.reserve(2, 160, 224, 300)
    p3, p4, p5 = model.backbone(imgs.cuda())
    assert p3.dtype == torch.float32
    for name, got in (("p3", p3), ("p4", p4), ("p5", p5)):
        check(f"f32_backbone_swin_small.{name}", dv.nchw_from_nhwc(got), ref[name], 2e-4, 2e-4)
    model.close()


def test_f32_split_range_is_reported_not_silent(dv):
    """The split-operand products (option f32_split = 1) cannot represent an activation beyond the fp16 range: such a launch sets the
    model's range flag (dvid_model_take_range_flag), which the detector turns into an error at the batch's host synchronisation.  With
    f32_split = 0 (the fp32 MFMA) the same input runs clean -- and gives finite results."""
    sd, _ = _head_state()
    g = torch.Generator().manual_seed(11)
    n, M, H, W = 1, 300, 96, 160
    feats = [nhwc(torch.randn(n, 256, H // s, W // s, generator=g)) for s in (8, 16, 32)]
    boxes = _boxes(g, n, M, H, W).cuda()
    pro = torch.randn(n * M, 256, generator=g).cuda()
    t = torch.tensor([499], dtype=torch.long)
    model = dv.Model(sd, res_blocks=(0, 0, 0, 0), precision="float32")
    model.reserve(n, H, W, M)
    model.rcnn_head(1, feats, H, W, boxes, pro, t)
    assert model.take_range_flag() is False
    big = pro * 1e6                                                   # in_proj's operand: |v| up to ~4e6 > 65504
    model.rcnn_head(1, feats, H, W, boxes, big, t)
    assert model.take_range_flag() is True and model.take_range_flag() is False          # reported once, then cleared
    dv.set_option("f32_split", 0)
    try:
        gl, gb, go = model.rcnn_head(1, feats, H, W, boxes, big, t)
        assert model.take_range_flag() is False and torch.isfinite(go).all() and torch.isfinite(gl).all()
    finally:
        dv.reset_options()
    model.close()


def test_unknown_precision_is_refused(dv):
    from diffusionvid_amd._lib import DvidError
    sd, _ = _head_state()
    with pytest.raises(DvidError, match="precision"):
        dv.Model(sd, res_blocks=(0, 0, 0, 0), precision="bfloat16")
