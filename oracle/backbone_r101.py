"""Oracle: detectron2 `build_resnet_fpn_backbone` for configs/vid_R_101_DiffusionVID.yaml:4-16.

Third-party, un-vendored (detectron2, un-pinned: INSTALL.md:68-69).  Call sites in the
reference: mega_core/modeling/detector/diffusion_det.py:219 (build), :427 (forward),
:301-303 + :422 (normalizer).  Restated from the published architecture (SURVEY.md
Appendix A.3): BasicStem, BottleneckBlock with STRIDE_IN_1X1=False, FrozenBatchNorm2d
(eps 1e-5), FPN (lateral 1x1 + nearest x2 top-down sum + 3x3 output), LastLevelMaxPool.
Parameter names follow detectron2 (`backbone.bottom_up.*`, `backbone.fpn_*`).
"""
import torch
import torch.nn.functional as F

from . import precision
from .precision import a16, w16

R101_BLOCKS = (3, 4, 23, 3)
R50_BLOCKS = (3, 4, 6, 3)


def normalizer(x, pixel_mean, pixel_std):
    """diffusion_det.py:301-303: (x - mean/255) / (std/255) on [0,1] RGB input."""
    mean = torch.tensor(pixel_mean, dtype=x.dtype).view(3, 1, 1) / 255.0
    std = torch.tensor(pixel_std, dtype=x.dtype).view(3, 1, 1) / 255.0
    return (x - mean) / std


def _frozen_bn(x, sd, name, eps=1e-5):
    # detectron2 FrozenBatchNorm2d.forward (no-grad branch) == F.batch_norm(training=False)
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training=False, eps=eps)


def _conv_bn(x, sd, name, stride=1, padding=0, relu=False, residual=None):
    if precision.is_fp16():
        # the runtime's repack (csrc/model.hip make_conv_bn): FrozenBN folded into the weights before the fp16 rounding,
        # fp32 bias; one fp16 store after bias (+ residual) (+ ReLU)
        eps = 1e-5
        sc = sd[name + ".norm.weight"] / torch.sqrt(sd[name + ".norm.running_var"] + eps)
        w = w16(sd[name + ".weight"] * sc.view(-1, 1, 1, 1))
        b = sd[name + ".norm.bias"] - sd[name + ".norm.running_mean"] * sc
        x = F.conv2d(x, w, b, stride=stride, padding=padding)
        if residual is not None:
            x = x + residual
        return a16(F.relu(x) if relu else x)
    x = F.conv2d(x, sd[name + ".weight"], None, stride=stride, padding=padding)
    x = _frozen_bn(x, sd, name + ".norm")
    if residual is not None:
        x = x + residual
    return F.relu(x) if relu else x


def bottleneck(x, sd, pfx, stride, has_shortcut):
    """detectron2 BottleneckBlock, stride on the 3x3 conv (STRIDE_IN_1X1: False)."""
    out = _conv_bn(x, sd, pfx + ".conv1", 1, 0, relu=True)
    out = _conv_bn(out, sd, pfx + ".conv2", stride, 1, relu=True)
    sc = _conv_bn(x, sd, pfx + ".shortcut", stride, 0) if has_shortcut else x
    return _conv_bn(out, sd, pfx + ".conv3", 1, 0, relu=True, residual=sc)      # relu(conv3 + shortcut)


def resnet_bottom_up(x, sd, pfx="backbone.bottom_up.", blocks=R101_BLOCKS):
    """Returns dict res2..res5 (NCHW fp32)."""
    x = _conv_bn(a16(x), sd, pfx + "stem.conv1", 2, 3, relu=True)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = {}
    for si, nb in enumerate(blocks):
        stage = f"res{si + 2}"
        for bi in range(nb):
            stride = 2 if (bi == 0 and si > 0) else 1
            x = bottleneck(x, sd, f"{pfx}{stage}.{bi}", stride, has_shortcut=(bi == 0))
        outs[stage] = x
    return outs


def fpn(feats, sd, pfx="backbone.", in_features=("res3", "res4", "res5")):
    """detectron2 FPN.forward, FUSE_TYPE 'sum', NORM ''.  Returns dict p3,p4,p5,p6."""
    stages = [int(f[3:]) for f in in_features]            # [3,4,5]
    results = {}
    prev = None
    for s in reversed(stages):
        lat = F.conv2d(feats[f"res{s}"], w16(sd[f"{pfx}fpn_lateral{s}.weight"]), sd[f"{pfx}fpn_lateral{s}.bias"])
        if prev is not None:
            lat = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        prev = a16(lat)
        results[f"p{s}"] = a16(F.conv2d(prev, w16(sd[f"{pfx}fpn_output{s}.weight"]), sd[f"{pfx}fpn_output{s}.bias"], padding=1))
    top = stages[-1]
    results[f"p{top + 1}"] = F.max_pool2d(results[f"p{top}"], kernel_size=1, stride=2, padding=0)  # LastLevelMaxPool
    return results


def backbone_r101_fpn(images_norm, sd, pfx="backbone.", blocks=R101_BLOCKS):
    """images_norm: normalised NCHW fp32, H and W multiples of 32.  Returns {p3,p4,p5,p6}."""
    with precision.stage("backbone"):
        return fpn(resnet_bottom_up(images_norm, sd, pfx + "bottom_up.", blocks), sd, pfx)
