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
import contextlib
import importlib
import os
import pickle
import sys
import types
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


def corloc_eval_detection_vid(pred_boxlists, gt_boxlists, iou_thresh=0.5):
    """CorLoc of vid_eval.py:356-443: per frame only the single highest-scoring detection is looked at; for every ground-truth
    entry of label l the frame counts as an image of class l, and as correctly localised when that detection carries label
    l and overlaps a class-l box by MORE than the threshold (double +1 convention as in the AP).  A frame with k boxes of one
    class counts k times (the reference loops over the label list, not over its set).  -> ({label: corloc}, mean over the
    labels seen, info text)."""
    assert len(gt_boxlists) == len(pred_boxlists), "Length of gt and pred lists need to be same."
    n_img, n_hit = defaultdict(int), defaultdict(int)
    for gt_bl, pr_bl in zip(gt_boxlists, pred_boxlists):
        pb, pl, ps = _fields(pr_bl)
        gb, gl, _ = _fields(gt_bl)
        top = ps.argsort()[::-1][:1]
        pb, pl = pb[top], pl[top]
        for l in gl.astype(int):
            n_img[l] += 1
            cand = pb[pl == l]
            if len(cand) and (_iou_vid(cand, gb[gl == l]) > iou_thresh).any():
                n_hit[l] += 1
    corloc = {l: (n_hit[l] / n_img[l] if n_img[l] else float("nan")) for l in n_img}
    info = "".join("gt:{} correct:{}\n".format(n_img[l], n_hit[l]) for l in n_img)
    info += "total_gt:{}, total_correct:{}\n".format(sum(n_img.values()), sum(n_hit[l] for l in n_img))
    avg = sum(corloc.values()) / len(corloc) if corloc else float("nan")
    return corloc, avg, info + "Avg_CorLoc = {:.4f}".format(avg)


VID_CLASSES = ('__background__', 'airplane', 'antelope', 'bear', 'bicycle', 'bird', 'bus', 'car', 'cattle', 'dog', 'domestic_cat',
               'elephant', 'fox', 'giant_panda', 'hamster', 'horse', 'lion', 'lizard', 'monkey', 'motorcycle', 'rabbit', 'red_panda',
               'sheep', 'snake', 'squirrel', 'tiger', 'train', 'turtle', 'watercraft', 'whale', 'zebra')


class GroundTruthList:
    """a list of ground-truth BoxLists (one per image id) behind the dataset surface do_vid_evaluation reads"""

    def __init__(self, gts, name_of=None, motion=None):
        self.gts, self._motion = gts, motion
        if name_of is not None:
            self.map_class_id_to_class_name = name_of

    def get_img_info(self, i):
        w, h = self.gts[i].size
        return {"width": w, "height": h}

    def get_groundtruth(self, i):
        return self.gts[i]

    def motion_ious(self):
        if self._motion is None:
            raise ValueError("motion-specific evaluation needs the per-box motion IoUs")
        return self._motion


def do_vid_evaluation(dataset, predictions, output_folder=None, box_only=False, motion_specific=False, logger=None,
                      motion_ious=None):
    """The evaluator wrapper the reference's `inference` ends in (vid_eval.py:14-78; engine/inference.py:176-181 through
    `evaluate`).  `dataset` supplies what the reference's VIDDataset does: `get_img_info(i) -> {"width", "height"}` of the
    ORIGINAL frame, `get_groundtruth(i) -> BoxList` with "labels" in original-frame pixels, and (optionally)
    `map_class_id_to_class_name(i)`.  Every prediction is first mapped from the resized frame the detector saw (600 x 1000) to
    the original size with `BoxList.resize` (:17-21) -- the AP is computed in annotation pixels --, then AP50 (per motion range
    with `motion_specific`; `motion_ious` or `dataset.motion_ious()` supplies the per-box motion IoUs the reference reads from
    its .mat file, :143-148) and CorLoc; the text goes to the logger and to `output_folder/result.txt`.  Returns the AP
    result (a list with one dict per motion range, as the reference does)."""
    if box_only:
        raise NotImplementedError("proposal recall (RPN_ONLY) is outside the DiffusionVID path")
    preds, gts = [], []
    for image_id, prediction in enumerate(predictions):
        info = dataset.get_img_info(image_id)
        preds.append(prediction.resize((info["width"], info["height"])))
        gts.append(dataset.get_groundtruth(image_id))
    names, ranges = ("all",), ((0.0, 1.0),)
    if motion_specific:
        names, ranges = MOTION_NAMES, MOTION_RANGES
        if motion_ious is None:
            motion_ious = dataset.motion_ious()
        result = eval_detection_vid(preds, gts, 0.5, motion_ious=motion_ious, motion_ranges=ranges)
    else:
        result = [eval_detection_vid(preds, gts, 0.5)]
    corloc, corloc_avg, _ = corloc_eval_detection_vid(preds, gts, 0.5)
    name_of = getattr(dataset, "map_class_id_to_class_name", None) or (lambda i: VID_CLASSES[i] if i < len(VID_CLASSES) else str(i))
    text = "".join("AP50 | motion={:>6s} = {:0.4f}\n".format(n, r["map"]) for n, r in zip(names, result))
    text += "Category AP:\n"
    text += "".join("{:<16}: {:.4f}\n".format(name_of(i), ap) for i, ap in enumerate(result[0]["ap"]) if i != 0)
    text += "Mean CorLoc: {:.4f}\n".format(corloc_avg)
    text += "Category CorLoc:\n"
    text += "".join("{:<16}: {:.4f}\n".format(name_of(l), corloc[l]) for l in corloc)
    if logger is not None:
        logger.info("\n" + text)
    if output_folder:
        with open(os.path.join(output_folder, "result.txt"), "w") as fid:
            fid.write(text)
    return result


# ---- predictions.pth -------------------------------------------------------------------------------------------------------------
# The reference's file is `torch.save(list[mega_core.structures.bounding_box.BoxList])` (engine/inference.py:168): a pickle that
# names that class.  To hand files back and forth between the two code bases without either importing the other:
#   * save_predictions(..., class_module="mega_core.structures.bounding_box") writes this repo's detections under the
#     reference's class path (its real class when importable here, otherwise an alias registered for the duration of the save);
#   * load_predictions resolves `<any package>.structures.bounding_box.BoxList` to this repo's class when that package is absent.
REFERENCE_BOXLIST_MODULE = "mega_core.structures.bounding_box"


@contextlib.contextmanager
def _boxlist_class_at(module_name):
    from ...structures.bounding_box import BoxList
    if module_name in (None, BoxList.__module__):
        yield BoxList
        return
    try:
        yield getattr(importlib.import_module(module_name), "BoxList")
        return
    except ImportError:
        pass
    alias = type("BoxList", (BoxList,), {"__module__": module_name})
    added = []
    parts = module_name.split(".")
    for i in range(1, len(parts) + 1):
        name = ".".join(parts[:i])
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__path__ = []
            sys.modules[name] = mod
            added.append(name)
    setattr(sys.modules[module_name], "BoxList", alias)
    try:
        yield alias
    finally:
        for name in added:
            sys.modules.pop(name, None)


def _as_class(bl, cls):
    if type(bl) is cls:
        return bl.to(torch.device("cpu"))
    out = cls(bl.bbox.detach().cpu(), tuple(bl.size), mode=bl.mode)
    for k in bl.fields():
        v = bl.get_field(k)
        out.add_field(k, v.detach().cpu() if isinstance(v, torch.Tensor) else v)
    return out


def save_predictions(predictions, path, class_module=None):
    """predictions: list[BoxList] indexed by image id (engine/inference.py:101-115, :168).  class_module None keeps every
    object's own class; a module path (REFERENCE_BOXLIST_MODULE) writes them all as that module's BoxList."""
    if class_module is None:
        torch.save([p.to(torch.device("cpu")) for p in predictions], path)
        return
    with _boxlist_class_at(class_module) as cls:
        torch.save([_as_class(p, cls) for p in predictions], path)


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            if name == "BoxList" and module.endswith("structures.bounding_box"):
                from ...structures.bounding_box import BoxList
                return BoxList
            raise


_tolerant_pickle = types.SimpleNamespace(**{k: getattr(pickle, k) for k in dir(pickle) if not k.startswith("__")})
_tolerant_pickle.__name__ = "pickle"
_tolerant_pickle.Unpickler = _TolerantUnpickler
_tolerant_pickle.load = lambda f, **kw: _TolerantUnpickler(f, **kw).load()


def load_predictions(path):
    return torch.load(path, weights_only=False, pickle_module=_tolerant_pickle)
