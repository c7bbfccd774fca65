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


def calc_prec_rec(pred_boxlists, gt_boxlists, iou_thresh=0.5, motion_ious=None, motion_range=(0.0, 1.0)):
    """vid_eval.py:164-299.  `motion_ious` (motion-specific AP, :39-50): per frame, one motion IoU per ground-truth box; ground
    truth outside `motion_range` is IGNORED -- it does not count as a positive, a detection matched to it is neither a true
    nor a false positive, and an unmatched detection counts as a false positive with a weight: 0 / 1 by whether its best
    overlap is with a counted / an ignored box, the ignored fraction of the frame's boxes of that class on a draw, and the
    data set's in-range fraction on frames without any box of the class (zero weights count fully, :283-285)."""
    n_pos, score, match, weight = defaultdict(int), defaultdict(list), defaultdict(list), defaultdict(list)
    lo, hi = motion_range
    empty_w = 0.0
    if motion_ious is not None:
        flat = np.concatenate([np.asarray(m, dtype=np.float64).reshape(-1) for m in motion_ious]) if len(motion_ious) else np.zeros(0)
        empty_w = float(((flat >= lo) & (flat <= hi)).sum() / float(len(flat))) if len(flat) else 0.0
        if empty_w == 1:
            empty_w = 0.0
    for f, (gt_bl, pr_bl) in enumerate(zip(gt_boxlists, pred_boxlists)):
        pb, pl, ps = _fields(pr_bl)
        gb, gl, _ = _fields(gt_bl)
        ignored = np.zeros(len(gb))
        if motion_ious is not None and len(motion_ious[f]):
            mi = np.asarray(motion_ious[f], dtype=np.float64).reshape(-1)
            ignored = ((mi < lo) | (mi > hi)).astype(np.float64)
        for l in np.unique(np.concatenate((pl, gl)).astype(int)):
            pm = pl == l
            pb_l, ps_l = pb[pm], ps[pm]
            order = ps_l.argsort()[::-1]
            pb_l, ps_l = pb_l[order], ps_l[order]
            gb_l, ig_l = gb[gl == l], ignored[gl == l]
            n_pos[l] += gb_l.shape[0] - ig_l.sum()
            score[l].extend(ps_l)
            if len(pb_l) == 0:
                continue
            if len(gb_l) == 0:
                match[l].extend((0,) * pb_l.shape[0])
                weight[l].extend((empty_w,) * pb_l.shape[0])
                continue
            iou = _iou_vid(pb_l, gb_l)
            taken = np.zeros(gb_l.shape[0], dtype=bool)
            for j in range(iou.shape[0]):
                # best not-yet-taken ground truth with IoU >= threshold; on an exact tie the earlier one stays unless it is ignored
                best, arg = iou_thresh, -1
                top_ig = top_cnt = -1.0
                for k in range(iou.shape[1]):
                    v = iou[j, k]
                    if ig_l[k] == 1:
                        top_ig = max(top_ig, v)
                    else:
                        top_cnt = max(top_cnt, v)
                    if taken[k] or v < best:
                        continue
                    if v == best and arg >= 0 and not ig_l[arg]:
                        continue
                    best, arg = v, k
                if arg >= 0:
                    taken[arg] = True
                    match[l].append(1)
                    weight[l].append(ig_l[arg])
                else:
                    match[l].append(0)
                    weight[l].append(0.0 if top_cnt > top_ig else (1.0 if top_ig > top_cnt else ig_l.sum() / float(len(ig_l))))
    n_fg = max(n_pos.keys()) + 1 if n_pos else 0
    prec, rec = [None] * n_fg, [None] * n_fg
    for l in n_pos.keys():
        s = np.array(score[l])
        order = s.argsort()[::-1]
        m = np.array(match[l], dtype=np.int8)[order]
        w = np.array(weight[l], dtype=np.float64)[order]
        counted = w != 1
        tp = np.cumsum((m == 1) & counted)
        fp = np.cumsum(((m == 0) & counted) * np.where(w == 0, 1.0, w))
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


MOTION_RANGES = ((0.0, 1.0), (0.0, 0.7), (0.7, 0.9), (0.9, 1.0))       # all / fast / medium / slow (vid_eval.py:39-41)
MOTION_NAMES = ("all", "fast", "medium", "slow")


def eval_detection_vid(pred_boxlists, gt_boxlists, iou_thresh=0.5, motion_ious=None, motion_ranges=None):
    """-> {"ap": per-class array (index = label, background nan), "map": AP50}; with `motion_ious` (one list of per-box motion
    IoUs per frame) -> one such dict per range of `motion_ranges` (default MOTION_RANGES), as vid_eval.py:130-161."""
    assert len(gt_boxlists) == len(pred_boxlists), "Length of gt and pred lists need to be same."
    if motion_ious is None:
        prec, rec = calc_prec_rec(pred_boxlists, gt_boxlists, iou_thresh)
        ap = calc_ap(prec, rec)
        return {"ap": ap, "map": float(np.nanmean(ap)) if len(ap) else float("nan")}
    out = []
    for rng in (motion_ranges or MOTION_RANGES):
        prec, rec = calc_prec_rec(pred_boxlists, gt_boxlists, iou_thresh, motion_ious, rng)
        ap = calc_ap(prec, rec)
        out.append({"ap": ap, "map": float(np.nanmean(ap)) if len(ap) else float("nan")})
    return out


def load_motion_ious(mat_file):
    """The data set's `vid_groundtruth_motion_iou.mat` (vid_eval.py:143-148): per frame, the motion IoU of each ground-truth
    box (missing entries count as 0)."""
    import scipy.io as sio
    cells = sio.loadmat(mat_file)["motion_iou"]
    return [[float(c[0]) if len(c) else 0.0 for c in cells[i][0]] for i in range(len(cells))]


def result_string(results, class_names=None):
    """the `result.txt` text of do_vid_evaluation (vid_eval.py:58-70) for one or four motion ranges"""
    results = results if isinstance(results, (list, tuple)) else [results]
    text = "".join("AP50 | motion={:>6s} = {:0.4f}\n".format(MOTION_NAMES[i], r["map"]) for i, r in enumerate(results))
    text += "Category AP:\n"
    for i, ap in enumerate(results[0]["ap"]):
        if i == 0:
            continue
        text += "{:<16}: {:.4f}\n".format(class_names[i] if class_names else str(i), ap)
    return text


def save_predictions(predictions, path):
    """predictions: list[BoxList] indexed by image id (engine/inference.py:101-115, :168)."""
    torch.save([p.to(torch.device("cpu")) for p in predictions], path)


def load_predictions(path):
    return torch.load(path, weights_only=False)
