"""ImageNet-VID AP50 evaluator (CPU, numpy) and the predictions.pth writer.

Restates the non-motion branch of mega_core/data/datasets/evaluation/vid/vid_eval.py:
`eval_detection_vid` :130-161, `calc_detection_vid_prec_rec` :164-299 (greedy per-frame, per-class
matching in descending score order; "VID evaluation follows integer typed bounding boxes": +1 on
x2,y2 at :220-224 on top of boxlist_iou's own +1, structures/boxlist_ops.py:53-89) and
`calc_detection_vid_ap` :302-354 (area under the monotone precision envelope).  Used to report the
AP50 delta between the GPU path and the CPU oracle on identical inputs.
`predictions.pth` is what mega_core/engine/inference.py:168 saves: a list of BoxList indexed by
dataset image id.
"""
from collections import defaultdict

import numpy as np
import torch


def _iou_vid(pred, gt):
    """IoU matrix [P,G] with the evaluator's double +1 convention (see module docstring)."""
    p = pred.astype(np.float32).copy()
    g = gt.astype(np.float32).copy()
    p[:, 2:] += 1
    g[:, 2:] += 1
    area_p = (p[:, 2] - p[:, 0] + 1) * (p[:, 3] - p[:, 1] + 1)
    area_g = (g[:, 2] - g[:, 0] + 1) * (g[:, 3] - g[:, 1] + 1)
    lt = np.maximum(p[:, None, :2], g[None, :, :2])
    rb = np.minimum(p[:, None, 2:], g[None, :, 2:])
    wh = np.clip(rb - lt + 1, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_p[:, None] + area_g[None, :] - inter)


def _fields(bl):
    return (bl.bbox.detach().cpu().numpy(), bl.get_field("labels").detach().cpu().numpy(),
            bl.get_field("scores").detach().cpu().numpy() if bl.has_field("scores") else None)


def calc_prec_rec(pred_boxlists, gt_boxlists, iou_thresh=0.5):
    n_pos, score, match = defaultdict(int), defaultdict(list), defaultdict(list)
    for gt_bl, pr_bl in zip(gt_boxlists, pred_boxlists):
        pb, pl, ps = _fields(pr_bl)
        gb, gl, _ = _fields(gt_bl)
        for l in np.unique(np.concatenate((pl, gl)).astype(int)):
            pm = pl == l
            pb_l, ps_l = pb[pm], ps[pm]
            order = ps_l.argsort()[::-1]
            pb_l, ps_l = pb_l[order], ps_l[order]
            gb_l = gb[gl == l]
            n_pos[l] += gb_l.shape[0]
            score[l].extend(ps_l)
            if len(pb_l) == 0:
                continue
            if len(gb_l) == 0:
                match[l].extend((0,) * pb_l.shape[0])
                continue
            iou = _iou_vid(pb_l, gb_l)
            taken = np.zeros(gb_l.shape[0], dtype=bool)
            for j in range(iou.shape[0]):
                # best not-yet-taken ground truth with IoU >= threshold; first one wins ties
                best, arg = iou_thresh, -1
                for k in range(iou.shape[1]):
                    if taken[k] or iou[j, k] < best:
                        continue
                    if iou[j, k] == best and arg >= 0:
                        continue
                    best, arg = iou[j, k], k
                if arg >= 0:
                    taken[arg] = True
                    match[l].append(1)
                else:
                    match[l].append(0)
    n_fg = max(n_pos.keys()) + 1 if n_pos else 0
    prec, rec = [None] * n_fg, [None] * n_fg
    for l in n_pos.keys():
        s = np.array(score[l])
        m = np.array(match[l], dtype=np.int8)
        m = m[s.argsort()[::-1]]
        tp = np.cumsum(m == 1)
        fp = np.cumsum(m == 0)
        prec[l] = tp / (fp + tp + np.spacing(1))
        if n_pos[l] > 0:
            rec[l] = tp / n_pos[l]
    return prec, rec


def calc_ap(prec, rec):
    ap = np.empty(len(prec))
    for l in range(len(prec)):
        if prec[l] is None or rec[l] is None:
            ap[l] = np.nan
            continue
        mpre = np.concatenate(([0], np.nan_to_num(prec[l]), [0]))
        mrec = np.concatenate(([0], rec[l], [1]))
        mpre = np.maximum.accumulate(mpre[::-1])[::-1]
        i = np.where(mrec[1:] != mrec[:-1])[0]
        ap[l] = np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])
    return ap


def eval_detection_vid(pred_boxlists, gt_boxlists, iou_thresh=0.5):
    """-> {"ap": per-class array (index = label, background nan), "map": AP50}"""
    assert len(gt_boxlists) == len(pred_boxlists), "Length of gt and pred lists need to be same."
    prec, rec = calc_prec_rec(pred_boxlists, gt_boxlists, iou_thresh)
    ap = calc_ap(prec, rec)
    return {"ap": ap, "map": float(np.nanmean(ap)) if len(ap) else float("nan")}


def save_predictions(predictions, path):
    """predictions: list[BoxList] indexed by image id (engine/inference.py:101-115, :168)."""
    torch.save([p.to(torch.device("cpu")) for p in predictions], path)


def load_predictions(path):
    return torch.load(path, weights_only=False)
