#!/usr/bin/env python
"""Evaluate a saved `predictions.pth` without running the model -- the counterpart of the reference's
tools/test_prediction.py:1-86 (`inference_no_model`, mega_core/engine/inference.py:184-211 -> do_vid_evaluation,
vid_eval.py:14-78).

    python tools/test_prediction.py --prediction-folder OUT --dataset VID_val_videos --ground-truth GT.pth [--motion-specific]

OUT/inference/<dataset>/predictions.pth is what engine.inference() (or the reference) wrote: a list of BoxList indexed by
image id.  Ground truth comes from a file (annotation parsing is outside the hot path): a torch-saved dict
{"gt": list[BoxList with "labels"], "motion_ious": per-frame lists (optional)}; `--motion-iou-mat` reads the data set's own
vid_groundtruth_motion_iou.mat instead.  Predictions whose size differs from the ground truth's are rescaled first
(vid_eval.py:21-24).  Writes OUT/inference/<dataset>/result.txt and prints it.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd.data.evaluation import vid_eval  # noqa: E402
from diffusionvid_amd.structures.bounding_box import BoxList  # noqa: E402

CLASSES = ['__background__', 'airplane', 'antelope', 'bear', 'bicycle', 'bird', 'bus', 'car', 'cattle', 'dog', 'domestic_cat',
           'elephant', 'fox', 'giant_panda', 'hamster', 'horse', 'lion', 'lizard', 'monkey', 'motorcycle', 'rabbit', 'red_panda',
           'sheep', 'snake', 'squirrel', 'tiger', 'train', 'turtle', 'watercraft', 'whale', 'zebra']


def rescale(pred, size_wh):
    if tuple(pred.size) == tuple(size_wh):
        return pred
    sx, sy = size_wh[0] / pred.size[0], size_wh[1] / pred.size[1]
    out = BoxList(pred.bbox * torch.tensor([sx, sy, sx, sy]), size_wh, mode="xyxy")
    for k in ("scores", "labels"):
        out.add_field(k, pred.get_field(k))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prediction-folder", required=True)
    ap.add_argument("--dataset", default="VID_val_videos")
    ap.add_argument("--ground-truth", required=True)
    ap.add_argument("--motion-specific", "-ms", action="store_true")
    ap.add_argument("--motion-iou-mat", default=None)
    args = ap.parse_args()
    folder = os.path.join(args.prediction_folder, "inference", args.dataset)
    preds = vid_eval.load_predictions(os.path.join(folder, "predictions.pth"))
    gt = torch.load(args.ground_truth, weights_only=False)
    gts = gt["gt"]
    if len(gts) != len(preds):
        raise SystemExit("predictions.pth holds %d frames, the ground truth %d" % (len(preds), len(gts)))
    preds = [rescale(p, g.size) for p, g in zip(preds, gts)]
    motion = None
    if args.motion_specific:
        motion = vid_eval.load_motion_ious(args.motion_iou_mat) if args.motion_iou_mat else gt.get("motion_ious")
        if motion is None:
            raise SystemExit("--motion-specific needs --motion-iou-mat or 'motion_ious' in the ground-truth file")
    res = vid_eval.eval_detection_vid(preds, gts, motion_ious=motion)
    text = vid_eval.result_string(res, CLASSES)
    print(text)
    with open(os.path.join(folder, "result.txt"), "w") as f:
        f.write(text)


if __name__ == "__main__":
    main()
