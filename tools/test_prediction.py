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
    motion = None
    if args.motion_specific:
        motion = vid_eval.load_motion_ious(args.motion_iou_mat) if args.motion_iou_mat else gt.get("motion_ious")
        if motion is None:
            raise SystemExit("--motion-specific needs --motion-iou-mat or 'motion_ious' in the ground-truth file")
    # do_vid_evaluation maps every prediction to its ground truth's size (BoxList.resize), evaluates AP50 (+ CorLoc) and
    # writes folder/result.txt
    vid_eval.do_vid_evaluation(vid_eval.GroundTruthList(gts, motion=motion), preds, folder, motion_specific=args.motion_specific)
    with open(os.path.join(folder, "result.txt")) as f:
        print(f.read())


if __name__ == "__main__":
    main()
