"""Oracle: DynamicHead / RCNNHead / RCNNHead_cond / DynamicConv, functional, CPU fp32.

Follows mega_core/modeling/roi_heads/box_head/box_head.py:
  DynamicConv.forward            :687-711
  RCNNHead.forward               :495-548   apply_deltas :550-590
  RCNNHead_cond.forward          :605-664
  DynamicHead.forward            :273-435
Weights come from a state_dict with the reference's parameter names
(`head.head_series.{i}.*`, `head.head_series_cond.0.*`, `head.global_attention.0.0.*`,
`head.time_mlp.{1,3}.*`).
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import precision
from .precision import a16, w16
from .roi_align import roi_pooler
from .schedule import time_mlp

_DEFAULT_SCALE_CLAMP = math.log(100000.0 / 16)  # box_head.py:19


@dataclass
class HeadCfg:
    hidden_dim: int = 256
    nheads: int = 8
    dim_dynamic: int = 64
    num_dynamic: int = 2
    num_classes: int = 30
    num_cls: int = 1
    num_reg: int = 3
    num_heads: int = 3           # MODEL.DiffusionDet.NUM_HEADS (head_series)
    num_heads_local: int = 1     # MODEL.DiffusionDet.NUM_HEADS_LOCAL (head_series_cond)
    pooler_resolution: int = 7
    sampling_ratio: int = 2
    scales: tuple = (1 / 8., 1 / 16., 1 / 32.)
    top_k: tuple = (75, 25)      # box_head.py:235
    sampling_timesteps: int = 1
    bbox_weights: tuple = (2.0, 2.0, 1.0, 1.0)
    scale_clamp: float = _DEFAULT_SCALE_CLAMP


def _ln(x, sd, name, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _lin(x, sd, name, bias=True):
    """F.linear with the policy's weight rounding (fp32 policy: the plain call)"""
    return F.linear(x, w16(sd[name + ".weight"]), sd[name + ".bias"] if bias else None)


def _mha_fp16_policy(sd, pfx, query, key, value, nheads):
    """nn.MultiheadAttention under the fp16 storage policy (see precision.py): fp16 inputs / q / k / v / probabilities /
    per-head outputs, fp32 scores, softmax sums and accumulation.  The probabilities that enter P.V are exp(s - rowmax)
    rounded to fp16, normalised afterwards by the fp32 sum of the un-rounded ones (csrc/attention.hip)."""
    Lq, B, d = query.shape
    Lk = key.shape[0]
    hd = d // nheads
    W, bvec = w16(sd[pfx + ".in_proj_weight"]), sd[pfx + ".in_proj_bias"]
    q = a16(F.linear(a16(query), W[:d], bvec[:d]))
    k = a16(F.linear(a16(key), W[d:2 * d], bvec[d:2 * d]))
    v = a16(F.linear(a16(value), W[2 * d:], bvec[2 * d:]))
    q = q.view(Lq, B * nheads, hd).transpose(0, 1)
    k = k.view(Lk, B * nheads, hd).transpose(0, 1)
    v = v.view(Lk, B * nheads, hd).transpose(0, 1)
    s_ = torch.bmm(q, k.transpose(1, 2)) * (1.0 / math.sqrt(hd))
    p_ = torch.exp(s_ - s_.amax(-1, keepdim=True))
    o = torch.bmm(a16(p_), v) / p_.sum(-1, keepdim=True)
    o = a16(o).transpose(0, 1).reshape(Lq, B, d)
    return F.linear(o, w16(sd[pfx + ".out_proj.weight"]), sd[pfx + ".out_proj.bias"])


def _mha(sd, pfx, query, key, value, nheads):
    """nn.MultiheadAttention(d, nheads)(query, key, value)[0]; inputs [L, B, d]."""
    if precision.is_fp16():
        return _mha_fp16_policy(sd, pfx, query, key, value, nheads)
    out, _ = F.multi_head_attention_forward(
        query, key, value, query.shape[-1], nheads,
        sd[pfx + ".in_proj_weight"], sd[pfx + ".in_proj_bias"],
        None, None, False, 0.0,
        sd[pfx + ".out_proj.weight"], sd[pfx + ".out_proj.bias"],
        training=False, need_weights=False)
    return out


def dynamic_conv(sd, pfx, pro_features, roi_features, cfg):
    """box_head.py:687-711.  pro_features [1, R, d]; roi_features [P*P, R, d]."""
    d, dd = cfg.hidden_dim, cfg.dim_dynamic
    num_params = d * dd
    features = roi_features.permute(1, 0, 2)
    parameters = a16(_lin(a16(pro_features), sd, pfx + ".dynamic_layer")).permute(1, 0, 2)
    param1 = parameters[:, :, :num_params].reshape(-1, d, dd)
    param2 = parameters[:, :, num_params:].reshape(-1, dd, d)
    features = torch.bmm(features, param1)
    features = a16(F.relu(_ln(features, sd, pfx + ".norm1")))
    features = torch.bmm(features, param2)
    features = a16(F.relu(_ln(features, sd, pfx + ".norm2")))
    features = features.flatten(1)
    features = _lin(features, sd, pfx + ".out_layer")
    features = F.relu(_ln(features, sd, pfx + ".norm3"))
    return features


def apply_deltas(deltas, boxes, cfg):
    """box_head.py:550-590."""
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = cfg.bbox_weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = deltas[:, 2::4] / ww
    dh = deltas[:, 3::4] / wh
    dw = torch.clamp(dw, max=cfg.scale_clamp)
    dh = torch.clamp(dh, max=cfg.scale_clamp)
    pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
    pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
    pred_w = torch.exp(dw) * widths[:, None]
    pred_h = torch.exp(dh) * heights[:, None]
    pred_boxes = torch.zeros_like(deltas)
    pred_boxes[:, 0::4] = pred_ctr_x - 0.5 * pred_w
    pred_boxes[:, 1::4] = pred_ctr_y - 0.5 * pred_h
    pred_boxes[:, 2::4] = pred_ctr_x + 0.5 * pred_w
    pred_boxes[:, 3::4] = pred_ctr_y + 0.5 * pred_h
    assert (pred_boxes[:, 2:] >= pred_boxes[:, :2]).all()
    return pred_boxes


def rcnn_head(sd, pfx, features, bboxes, pro_features, time_emb, cfg, cond=None, taps=None):
    """RCNNHead.forward (cond is None, box_head.py:495-548) / RCNNHead_cond.forward (:605-664).

    features: list of NCHW levels; bboxes [N, nr, 4]; pro_features [1, N*nr, d] or None;
    time_emb [N, 4d]; cond [N*nr, d] or None.  `taps`: optional dict that receives
    intermediates (used by the per-kernel GPU parity tests).
    """
    d = cfg.hidden_dim
    N, nr = bboxes.shape[:2]
    # stage tags for a scoped precision policy (precision.use(..., only=...)): "head.<module prefix>.<part>"
    tag = "head." + pfx.split("head.", 1)[-1] + "."
    stage = precision.stage
    roi = roi_pooler(features, bboxes, cfg.pooler_resolution, cfg.scales, cfg.sampling_ratio)
    if pro_features is None:
        pro_features = roi.view(N, nr, d, -1).mean(-1)          # fp16 policy: mean of the un-rounded bins (csrc/roialign.hip)
    with stage(tag + "roi"):
        roi = a16(roi)
    roi_features = roi.view(N * nr, d, -1).permute(2, 0, 1)
    if taps is not None:
        taps["roi"] = roi
        taps["pro_in"] = pro_features.reshape(N * nr, d).clone()

    # self_att.
    pro_features = pro_features.view(N, nr, d).permute(1, 0, 2)
    with stage(tag + "attn"):
        pro_features2 = _mha(sd, pfx + ".self_attn", pro_features, pro_features, pro_features, cfg.nheads)
    pro_features = _ln(pro_features + pro_features2, sd, pfx + ".norm1")
    # inst_interact.
    pro_features = pro_features.view(nr, N, d).permute(1, 0, 2).reshape(1, N * nr, d)
    if taps is not None:
        taps["after_attn"] = pro_features[0].clone()
    with stage(tag + "dynconv"):
        pro_features2 = dynamic_conv(sd, pfx + ".inst_interact", pro_features, roi_features, cfg)
    if taps is not None:
        taps["dynconv"] = pro_features2.clone()
    obj_features = _ln(pro_features + pro_features2, sd, pfx + ".norm2")
    # obj_feature.
    with stage(tag + "ffn"):
        obj_features2 = _lin(a16(F.relu(_lin(a16(obj_features), sd, pfx + ".linear1"))), sd, pfx + ".linear2")
    obj_features = _ln(obj_features + obj_features2, sd, pfx + ".norm3")

    fc_feature = obj_features.transpose(0, 1).reshape(N * nr, -1)
    if cond is None:
        scale_shift = F.linear(F.silu(time_emb), sd[pfx + ".block_time_mlp.1.weight"], sd[pfx + ".block_time_mlp.1.bias"])
        scale_shift = torch.repeat_interleave(scale_shift, nr, dim=0)
        scale, shift = scale_shift.chunk(2, dim=1)
    else:
        with stage(tag + "mod"):
            shift = _lin(a16(F.silu(cond)), sd, pfx + ".c_mlp.1")
        scale = F.linear(F.silu(time_emb), sd[pfx + ".block_time_mlp.1.weight"], sd[pfx + ".block_time_mlp.1.bias"])
        scale = torch.repeat_interleave(scale, nr, dim=0)
    with stage(tag + "mod"):
        fc_feature = a16(fc_feature * (scale + 1) + shift)
    if taps is not None:
        taps["fc_feature"] = fc_feature.clone()

    cls_feature = fc_feature
    reg_feature = fc_feature
    with stage(tag + "cls_tower"):
        for i in range(cfg.num_cls):
            cls_feature = a16(F.relu(_ln(_lin(cls_feature, sd, f"{pfx}.cls_module.{3 * i}", bias=False), sd, f"{pfx}.cls_module.{3 * i + 1}")))
    with stage(tag + "reg_tower"):
        for i in range(cfg.num_reg):
            reg_feature = a16(F.relu(_ln(_lin(reg_feature, sd, f"{pfx}.reg_module.{3 * i}", bias=False), sd, f"{pfx}.reg_module.{3 * i + 1}")))
    with stage(tag + "class_logits"):
        class_logits = _lin(cls_feature, sd, pfx + ".class_logits")
    with stage(tag + "bboxes_delta"):
        bboxes_deltas = _lin(reg_feature, sd, pfx + ".bboxes_delta")
    if taps is not None:
        taps["deltas"] = bboxes_deltas.clone()
    pred_bboxes = apply_deltas(bboxes_deltas, bboxes.reshape(-1, 4), cfg)
    return class_logits.view(N, nr, -1), pred_bboxes.view(N, nr, -1), obj_features


def select_topk_features(class_logits, proposal_features, cfg):
    """box_head.py:304-317: top-k1 / top-k2 boxes per frame by max logit, returned in MASK
    (box-index) order, not score order."""
    N, nr = class_logits.shape[:2]
    d = proposal_features.shape[-1]
    class_logits_max, _ = torch.max(class_logits, dim=-1)
    _, topk_idx = class_logits_max.topk(k=cfg.top_k[0], dim=-1)
    m1 = torch.zeros_like(class_logits_max, dtype=torch.bool)
    m1.scatter_(1, topk_idx, 1)
    m2 = torch.zeros_like(class_logits_max, dtype=torch.bool)
    m2.scatter_(1, topk_idx[:, :cfg.top_k[1]], 1)
    pf = proposal_features.view(-1, nr, d)
    return pf[m1], pf[m2]


def head_extract(sd, pfx, features, init_bboxes, t, cfg):
    """DynamicHead.forward with box_extract>0 (box_head.py:286-317).

    Returns ([class_logits, bboxes, proposal_features], top_k1 feats, top_k2 feats)."""
    time = time_mlp(sd, pfx, t, cfg.hidden_dim)
    bboxes = init_bboxes
    proposal_features = None
    for i in range(cfg.num_heads):
        class_logits, pred_bboxes, proposal_features = rcnn_head(
            sd, f"{pfx}head_series.{i}", features, bboxes, proposal_features, time, cfg)
        bboxes = pred_bboxes
    k1, k2 = select_topk_features(class_logits, proposal_features, cfg)
    return [class_logits, bboxes, proposal_features], k1, k2


def global_attention(sd, pfx, proposal_features, memory, cfg):
    """box_head.py:349,366-394 with adaptive_norm=True and one global stage:
    query [R,1,d] against kv = memory[0] [Lk,1,d]; returns cond [R, d]."""
    query_ = proposal_features.permute(1, 0, 2)
    kv = memory[0].unsqueeze(1)
    with precision.stage("head.global_attention"):
        attn_ = _mha(sd, pfx + "global_attention.0.0", query_, kv, kv, cfg.nheads)
    return attn_.reshape(-1, proposal_features.shape[-1])


def head_final(sd, pfx, features, init_bboxes, t, cfg, cached=None, memory=None):
    """DynamicHead.forward with box_extract == 0 at test time (box_head.py:286-302, :319-432).

    cached = (class_logits, bboxes, proposal_features) popped from proposals_feat_cur when
    sampling_timesteps == 1; memory = proposal_feats_global ([mem900, mem150])."""
    time = time_mlp(sd, pfx, t, cfg.hidden_dim)
    if cfg.sampling_timesteps > 1:
        bboxes = init_bboxes
        proposal_features = None
        for i in range(cfg.num_heads):
            class_logits, pred_bboxes, proposal_features = rcnn_head(
                sd, f"{pfx}head_series.{i}", features, bboxes, proposal_features, time, cfg)
            bboxes = pred_bboxes
    else:
        class_logits, bboxes, proposal_features = cached
    attn_ = global_attention(sd, pfx, proposal_features, memory, cfg)
    query_ = proposal_features.permute(1, 0, 2)
    bboxes2 = bboxes
    for i in range(cfg.num_heads_local):
        proposal_features2 = query_.permute(1, 0, 2)
        class_logits2, pred_bboxes2, proposal_features2 = rcnn_head(
            sd, f"{pfx}head_series_cond.{i}", features, bboxes2, proposal_features2, time, cfg, cond=attn_)
        bboxes2 = pred_bboxes2
        query_ = proposal_features2.permute(1, 0, 2)
    return class_logits2[None], pred_bboxes2[None]
