"""Oracle: detectron2 `ROIPooler(pooler_type="ROIAlignV2")` restated on CPU.

Third-party, un-vendored: detectron2 (un-pinned; INSTALL.md:68-69 clones HEAD) ->
`detectron2/modeling/poolers.py` (assign_boxes_to_levels, ROIPooler.forward) ->
`torchvision.ops.roi_align(..., aligned=True)` (roi_align_kernel.cpp
`bilinear_interpolate` / `pre_calc_for_bilinear_interpolate`).  Call sites in the
reference: mega_core/modeling/roi_heads/box_head/box_head.py:250-271 (construction),
:507 and :617 (use).  Published algorithm restated; see SURVEY.md Appendix A.1.
"""
import math

import torch


def assign_boxes_to_levels(boxes, min_level, max_level, canonical_box_size=224, canonical_level=4):
    """detectron2 poolers.assign_boxes_to_levels.  boxes [K,4] xyxy -> int64 [K] in [0, L)."""
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    box_sizes = torch.sqrt(area)
    lvl = torch.floor(canonical_level + torch.log2(box_sizes / canonical_box_size + 1e-8))
    lvl = torch.clamp(lvl, min=min_level, max=max_level)
    return lvl.to(torch.int64) - min_level


def _axis_samples(start, bin_size, pooled, grid, limit):
    """Per-axis sample coordinates and their bilinear taps.

    start, bin_size: [K]; returns (low [K,P*G] int64, high, w_low, w_high, valid) following
    torchvision's bilinear_interpolate for one axis.
    """
    ph = torch.arange(pooled, dtype=start.dtype).view(1, pooled, 1)
    ig = torch.arange(grid, dtype=start.dtype).view(1, 1, grid)
    # y = roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / roi_bin_grid_h
    y = start.view(-1, 1, 1) + ph * bin_size.view(-1, 1, 1) + (ig + 0.5) * bin_size.view(-1, 1, 1) / grid
    y = y.reshape(y.shape[0], -1)
    valid = ~((y < -1.0) | (y > limit))
    y = torch.where(y <= 0, torch.zeros_like(y), y)
    low = y.to(torch.int64)  # (int) truncation; y >= 0 here
    clamp = low >= limit - 1
    high = torch.where(clamp, torch.full_like(low, limit - 1), low + 1)
    low = torch.where(clamp, torch.full_like(low, limit - 1), low)
    y = torch.where(clamp, low.to(y.dtype), y)
    l = y - low.to(y.dtype)
    h = 1.0 - l
    # samples outside are zeroed through `valid`; keep indices in range for the gather
    low = low.clamp(0, limit - 1)
    high = high.clamp(0, limit - 1)
    return low, high, h, l, valid


def roi_align_v2(feat, rois, output_size, spatial_scale, sampling_ratio, chunk=256):
    """torchvision.ops.roi_align(feat, rois, output_size, spatial_scale, sampling_ratio, aligned=True).

    feat [N,C,H,W] fp32; rois [K,5] = (batch_idx, x1, y1, x2, y2).  Returns [K,C,P,P].
    """
    assert sampling_ratio > 0
    N, C, H, W = feat.shape
    K = rois.shape[0]
    P, G = output_size, sampling_ratio
    out = feat.new_zeros((K, C, P, P))
    if K == 0:
        return out
    feat_l = feat.permute(0, 2, 3, 1).contiguous()  # [N,H,W,C]
    for s in range(0, K, chunk):
        r = rois[s:s + chunk]
        b = r[:, 0].to(torch.int64)
        offset = 0.5
        x1 = r[:, 1] * spatial_scale - offset
        y1 = r[:, 2] * spatial_scale - offset
        x2 = r[:, 3] * spatial_scale - offset
        y2 = r[:, 4] * spatial_scale - offset
        roi_w = x2 - x1
        roi_h = y2 - y1
        bin_w = roi_w / P
        bin_h = roi_h / P
        yl, yh, hy, ly, vy = _axis_samples(y1, bin_h, P, G, H)
        xl, xh, hx, lx, vx = _axis_samples(x1, bin_w, P, G, W)
        bb = b.view(-1, 1, 1)

        def g(yi, xi):
            return feat_l[bb, yi.unsqueeze(2), xi.unsqueeze(1)]  # [k, PG, PG, C]

        w1 = (hy.unsqueeze(2) * hx.unsqueeze(1)).unsqueeze(-1)
        w2 = (hy.unsqueeze(2) * lx.unsqueeze(1)).unsqueeze(-1)
        w3 = (ly.unsqueeze(2) * hx.unsqueeze(1)).unsqueeze(-1)
        w4 = (ly.unsqueeze(2) * lx.unsqueeze(1)).unsqueeze(-1)
        val = w1 * g(yl, xl) + w2 * g(yl, xh) + w3 * g(yh, xl) + w4 * g(yh, xh)
        valid = (vy.unsqueeze(2) & vx.unsqueeze(1)).unsqueeze(-1)
        val = torch.where(valid, val, torch.zeros_like(val))
        k = val.shape[0]
        # accumulate the G*G samples of each bin in (iy, ix) order, then divide by count
        val = val.view(k, P, G, P, G, C)
        acc = torch.zeros((k, P, P, C), dtype=feat.dtype)
        for iy in range(G):
            for ix in range(G):
                acc = acc + val[:, :, iy, :, ix, :]
        acc = acc / float(G * G)
        out[s:s + chunk] = acc.permute(0, 3, 1, 2)
    return out


def roi_pooler(features, boxes, output_size=7, scales=(1 / 8., 1 / 16., 1 / 32.), sampling_ratio=2):
    """detectron2 ROIPooler.forward for per-image box tensors.

    features: list of [N,C,Hl,Wl]; boxes: [N, M, 4] absolute xyxy (one Boxes per image, as
    built at box_head.py:504-507).  Returns [N*M, C, P, P] in image-major order.
    """
    N, M = boxes.shape[:2]
    flat = boxes.reshape(N * M, 4)
    bidx = torch.arange(N, dtype=flat.dtype).repeat_interleave(M).view(-1, 1)
    rois = torch.cat([bidx, flat], dim=1)
    min_level = int(round(-math.log2(scales[0])))
    max_level = int(round(-math.log2(scales[-1])))
    C = features[0].shape[1]
    out = features[0].new_zeros((N * M, C, output_size, output_size))
    if len(features) == 1:
        return roi_align_v2(features[0], rois, output_size, scales[0], sampling_ratio)
    lvl = assign_boxes_to_levels(flat, min_level, max_level)
    for l, (f, sc) in enumerate(zip(features, scales)):
        inds = torch.nonzero(lvl == l).squeeze(1)
        if inds.numel():
            out[inds] = roi_align_v2(f, rois[inds], output_size, sc, sampling_ratio)
    return out
