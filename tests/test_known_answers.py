"""Known-answer tests that pin the restated THIRD-PARTY arithmetic independently of the oracle.

detectron2 / torchvision are absent from /root/reference and un-pinned by it; `oracle/roi_align.py`,
`oracle/backbone_r101.py` restate their published algorithms, the HIP kernels are compared with those
restatements, and the reference-generated head fixtures (g4 / g5 / g16) were produced THROUGH `oracle.roi_align` as the pooler
stand-in (tests/golden/_ref_shims.py:157-168) -- a shared misreading would pass everything.  The answers below come from closed
forms, not from the oracle:

* RoIAlignV2 (Appendix A.1) on an AFFINE feature map f(y, x) = a + b x + c y.  Bilinear interpolation reproduces an affine
  function exactly, so a sample at (y, x) contributes 0 when y < -1 or y > H or x < -1 or x > W and otherwise
  f(clamp(y), clamp(x)) with clamp(t) = 0 for t <= 0 and L - 1 when (int) t >= L - 1; a bin is the mean of its 2 x 2 samples at
  start + (p + (i + 0.5) / 2) * bin, start = coord * scale - 0.5 (aligned = True), bin = (end - start) / 7 (no clamp to 1).
  Cases: interior box; the -0.5 shift; boxes straddling each border (dropped samples, samples clamped at 0, the `low >= H - 1`
  clamp); a zero-area box; an oversize box; the level map at sqrt(area) = 112 / 224 / 448 and one pixel below
  (level = clamp(floor(4 + log2(sqrt(area) / 224 + 1e-8)), 3, 5): 224 and 448 are the first sizes of levels 4 and 5, everything
  below 224 -- including 112 -- is level 3), with a different affine map per level so that a wrong level shows.
* FrozenBatchNorm2d (Appendix A.3): y = x w rsqrt(var + 1e-5) + (b - mean w rsqrt(var + 1e-5)) on hand-picked numbers, against
  the oracle's conv + FrozenBN in both of its forms (F.batch_norm, and the folded weights of the fp16 policy).
* FPN top-down (Appendix A.3): lateral 1x1 + nearest-x2 upsample + sum, 3x3 output convolution, on a 2-level toy with
  hand-computed sums.
The same vectors then go through the C ABI on the GPU (`-m gpu`): dvid_roialign_v2_multilevel (fp16 maps whose values are exactly
representable) and its fp32 form, and the implicit-GEMM kernels' nearest-x2 residual epilogue.
"""
import math

import numpy as np
import pytest
import torch

from oracle import backbone_r101 as obb
from oracle import roi_align as oroi

H_IMG = W_IMG = 512
C = 256


def _coeffs():
    """per (level, channel): f = a + bx * x + by * y on the level's pixel grid; every value a multiple of 1/16 below 128 (exact in fp16)"""
    c = np.arange(C)
    a = (c % 16) / 8.0
    bx = ((c * 3) % 8) / 16.0
    by = ((c * 5 + 1) % 8) / 16.0
    return a, bx, by


def _pyramid(n):
    a, bx, by = _coeffs()
    feats = []
    for l, s in enumerate((8, 16, 32)):
        h, w = H_IMG // s, W_IMG // s
        yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        f = (a + 16.0 * l)[:, None, None] + bx[:, None, None] * xx[None] + by[:, None, None] * yy[None]
        f = np.stack([f + 4.0 * i for i in range(n)])          # frame i: + 4 i (the batch index must pick the right frame)
        feats.append(torch.from_numpy(f.astype(np.float32)))
    return feats


BOXES = [
    ("interior, level 3", [100.3, 80.7, 180.9, 150.2], 0),
    ("sqrt(area) = 224 exactly: first size of level 4", [32.0, 32.0, 256.0, 256.0], 1),
    ("223 x 224: level 3", [32.0, 32.0, 255.0, 256.0], 0),
    ("sqrt(area) = 448 exactly: first size of level 5", [16.0, 16.0, 464.0, 464.0], 2),
    ("447 x 448: level 4", [16.0, 16.0, 463.0, 464.0], 1),
    ("sqrt(area) = 112: level 3 (the minimum level)", [200.0, 200.0, 312.0, 312.0], 0),
    ("straddles the top / left border: samples below -1 dropped, (-1, 0] clamped to 0", [-40.0, -30.0, 60.0, 50.0], 0),
    ("straddles the bottom / right border: samples beyond H dropped, low >= H - 1 clamped", [450.0, 460.0, 560.0, 540.0], 0),
    ("zero area: every bin is the same point, level 3", [100.0, 100.0, 100.0, 100.0], 0),
    ("larger than the image on level 5", [-100.0, -100.0, 700.0, 700.0], 2),
    ("thin box across a level-4 border", [-20.0, 100.0, 500.0, 200.0], 1),
]


def _closed_form(n):
    """[n * M, C, 7, 7] float64 from the closed form in the module docstring"""
    a, bx, by = _coeffs()
    out = np.zeros((n * len(BOXES), C, 7, 7))
    for i in range(n):
        for j, (_, (x1, y1, x2, y2), level) in enumerate(BOXES):
            area = (x2 - x1) * (y2 - y1)
            lv = int(min(max(math.floor(4 + math.log2(math.sqrt(area) / 224 + 1e-8)), 3), 5)) - 3
            assert lv == level, (BOXES[j][0], lv)
            s = (8, 16, 32)[lv]
            Hl, Wl = H_IMG // s, W_IMG // s
            sx, sy = x1 / s - 0.5, y1 / s - 0.5
            bw, bh = (x2 / s - 0.5 - sx) / 7, (y2 / s - 0.5 - sy) / 7

            def clamp(t, L):
                if t <= 0:
                    return 0.0
                return float(L - 1) if int(t) >= L - 1 else t
            for ph in range(7):
                for pw in range(7):
                    acc = np.zeros(C)
                    for iy in range(2):
                        for ix in range(2):
                            y = sy + ph * bh + (iy + 0.5) * bh / 2
                            x = sx + pw * bw + (ix + 0.5) * bw / 2
                            if y < -1 or y > Hl or x < -1 or x > Wl:
                                continue
                            acc += a + 16.0 * lv + 4.0 * i + bx * clamp(x, Wl) + by * clamp(y, Hl)
                    out[i * len(BOXES) + j, :, ph, pw] = acc / 4
    return out


def _boxes(n):
    return torch.tensor([[b for _, b, _ in BOXES]] * n, dtype=torch.float32)


def test_roialign_v2_affine_known_answers_oracle():
    n = 2
    want = _closed_form(n)
    got = oroi.roi_pooler(_pyramid(n), _boxes(n)).double().numpy()
    err = np.abs(got - want).reshape(n * len(BOXES), -1).max(1)
    for j, e in enumerate(err):
        assert e <= 2e-4, f"{BOXES[j % len(BOXES)][0]}: oracle differs from the closed form by {e:.3e}"
    # the dropped-sample cases really drop something, and the clamp cases really clamp (the vectors exercise what they claim)
    assert np.abs(want[6, :, 0, 0]).max() < np.abs(want[6, :, 6, 6]).max() * 0.9
    assert np.allclose(want[8, :, 0, 0], want[8, :, 6, 6])


def test_level_assignment_known_answers():
    sizes = torch.tensor([[0, 0, 112, 112], [0, 0, 223, 224], [0, 0, 224, 224], [0, 0, 447, 448], [0, 0, 448, 448], [0, 0, 1000, 600], [5, 5, 5, 5]],
                         dtype=torch.float32)
    assert oroi.assign_boxes_to_levels(sizes, 3, 5).tolist() == [0, 0, 1, 1, 2, 2, 0]


def test_frozen_bn_known_answer():
    """1x1 convolution with w = 2 on x = 3, FrozenBN (gamma 0.5, beta -1, mean 4, var 0.25 - 1e-5): y = (6 - 4) * 0.5 / 0.5 - 1 = 1; a
    second channel with var 4 - 1e-5, gamma 3, beta 0.25, mean -2, w = -1: (-3 + 2) * 3 / 2 + 0.25 = -1.25.  Both forms of the oracle."""
    from oracle import precision
    sd = {"c.weight": torch.tensor([[[[2.0]]], [[[-1.0]]]]), "c.norm.weight": torch.tensor([0.5, 3.0]), "c.norm.bias": torch.tensor([-1.0, 0.25]),
          "c.norm.running_mean": torch.tensor([4.0, -2.0]), "c.norm.running_var": torch.tensor([0.25 - 1e-5, 4.0 - 1e-5])}
    x = torch.full((1, 1, 2, 2), 3.0)
    want = torch.tensor([1.0, -1.25]).view(1, 2, 1, 1).expand(1, 2, 2, 2)
    assert torch.allclose(obb._conv_bn(x, sd, "c"), want, atol=1e-6)
    assert torch.allclose(obb._conv_bn(x, sd, "c", relu=True), want.clamp(min=0), atol=1e-6)
    with precision.use("fp16"):          # the folded form (what csrc/model.hip: make_conv_bn packs), weights rounded to fp16 (exact here)
        assert torch.allclose(obb._conv_bn(x, sd, "c"), want, atol=2e-3)
    res = torch.full((1, 2, 2, 2), 0.5)
    assert torch.allclose(obb._conv_bn(x, sd, "c", relu=True, residual=res), (want + 0.5).clamp(min=0), atol=1e-6)          # relu(conv3 + shortcut)


def _fpn_toy():
    """2 channels, res4 2x2 = [[1,2],[3,4]] (+10 in channel 1), res5 1x1 = 7 (+10); laterals = identity, output convs = centre tap 1 (+ bias 0.5)"""
    sd = {}
    eye = torch.eye(2).view(2, 2, 1, 1)
    ctr = torch.zeros(2, 2, 3, 3)
    ctr[0, 0, 1, 1] = ctr[1, 1, 1, 1] = 1.0
    for s in (4, 5):
        sd[f"backbone.fpn_lateral{s}.weight"] = eye.clone()
        sd[f"backbone.fpn_lateral{s}.bias"] = torch.zeros(2)
        sd[f"backbone.fpn_output{s}.weight"] = ctr.clone()
        sd[f"backbone.fpn_output{s}.bias"] = torch.full((2,), 0.5)
    r4 = torch.tensor([[1.0, 2.0], [3.0, 4.0]])
    feats = {"res4": torch.stack([r4, r4 + 10]).unsqueeze(0), "res5": torch.tensor([7.0, 17.0]).view(1, 2, 1, 1)}
    want4 = torch.stack([r4 + 7 + 0.5, r4 + 10 + 17 + 0.5]).unsqueeze(0)          # lateral + nearest-x2(top) + output bias
    want5 = torch.tensor([7.5, 17.5]).view(1, 2, 1, 1)
    return sd, feats, want4, want5


def test_fpn_top_down_known_answer():
    sd, feats, want4, want5 = _fpn_toy()
    out = obb.fpn(feats, sd, "backbone.", in_features=("res4", "res5"))
    assert torch.equal(out["p5"], want5) and torch.equal(out["p4"], want4)
    assert out["p6"].shape == (1, 2, 1, 1) and torch.equal(out["p6"], want5)          # LastLevelMaxPool: max_pool2d(p5, 1, 2, 0)
    # a 3x3 output convolution that SUMS its window (all taps 1) on the 2x2 map: zero padding -> every pixel = the map's total
    sd["backbone.fpn_output4.weight"] = torch.zeros(2, 2, 3, 3)
    sd["backbone.fpn_output4.weight"][0, 0] = 1.0
    sd["backbone.fpn_output4.weight"][1, 1] = 1.0
    out = obb.fpn(feats, sd, "backbone.", in_features=("res4", "res5"))
    assert torch.equal(out["p4"][0, 0], torch.full((2, 2), 1 + 2 + 3 + 4 + 4 * 7 + 0.5))


# ---- the same vectors through the C ABI ----------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float16", "float32"])
def test_roialign_v2_affine_known_answers_gpu(dtype):
    from diffusionvid_amd import ops
    n = 2
    want = _closed_form(n)
    feats = [f.permute(0, 2, 3, 1).contiguous().cuda() for f in _pyramid(n)]
    if dtype == "float16":
        feats = [f.half() for f in feats]          # every value a multiple of 1/16 below 128: exact
        roi, mean = ops.roialign(feats, _boxes(n).cuda(), H_IMG, W_IMG, want_mean=True)
        tol = 1.5e-3          # one fp16 rounding of the stored tile (2^-11 relative) on values up to ~110
    else:
        roi, mean = ops.roialign_f32(feats, _boxes(n).cuda(), H_IMG, W_IMG, want_mean=True)
        tol = 2e-5
    got = roi.float().view(n * len(BOXES), 7, 7, C).permute(0, 3, 1, 2).double().cpu().numpy()
    for j in range(n * len(BOXES)):
        e = np.abs(got[j] - want[j]).max() / max(1.0, np.abs(want[j]).max())
        assert e <= tol, f"{BOXES[j % len(BOXES)][0]} (frame {j // len(BOXES)}): kernel differs from the closed form by {e:.3e} (relative)"
    e = np.abs(mean.double().cpu().numpy() - want.reshape(n * len(BOXES), C, 49).mean(-1)).max()
    assert e <= 110 * tol, e


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float16", "float32"])
def test_fpn_top_down_known_answer_gpu(dtype):
    """the lateral 1x1 + nearest-x2 top-down sum as the implicit-GEMM kernels compute it (residual mode 2), and a 3x3 output convolution
    that sums its window, on a 64-channel version of the toy (identity lateral weights): hand-computed integers"""
    from diffusionvid_amd import ops
    cch = 64
    r4 = torch.arange(1, 17, dtype=torch.float32).view(1, 4, 4, 1).expand(1, 4, 4, cch) + torch.arange(cch).view(1, 1, 1, cch)          # pixel value + channel
    top = torch.tensor([[7.0, 8.0], [9.0, 11.0]]).view(1, 2, 2, 1).expand(1, 2, 2, cch).contiguous()
    want = r4 + top.repeat_interleave(2, 1).repeat_interleave(2, 2)
    eye = torch.eye(cch)
    if dtype == "float16":
        wp, kpad = ops.pack_conv_weight(eye)
        lat = ops.conv2d_nhwc(r4.contiguous().cuda().half(), wp.cuda(), kpad, torch.zeros(cch).cuda(), cch, 1, 1, 1, 0, residual=top.cuda().half(), residual_mode=2)
        ones = torch.zeros(cch, cch, 3, 3)
        ones[torch.arange(cch), torch.arange(cch)] = 1.0
        wp3, k3 = ops.pack_conv_weight(ones)
        summed = ops.conv2d_nhwc(lat, wp3.cuda(), k3, torch.full((cch,), 0.5).cuda(), cch, 3, 3, 1, 1)
    else:
        wp, kpad = ops.pack_conv_weight_f32(eye)
        lat = ops.conv2d_nhwc_f32(r4.contiguous().cuda(), wp.cuda(), kpad, torch.zeros(cch).cuda(), cch, 1, 1, 1, 0, residual=top.cuda(), residual_mode=2)
        ones = torch.zeros(cch, cch, 3, 3)
        ones[torch.arange(cch), torch.arange(cch)] = 1.0
        wp3, k3 = ops.pack_conv_weight_f32(ones)
        summed = ops.conv2d_nhwc_f32(lat, wp3.cuda(), k3, torch.full((cch,), 0.5).cuda(), cch, 3, 3, 1, 1)
    assert torch.equal(lat.float().cpu(), want)          # small integers: exact in fp16 and fp32
    win = torch.nn.functional.avg_pool2d(want.permute(0, 3, 1, 2), 3, 1, 1, divisor_override=1).permute(0, 2, 3, 1) + 0.5          # zero-padded 3x3 window sums
    assert torch.equal(summed.float().cpu(), win)
