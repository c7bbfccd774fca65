#!/usr/bin/env python
"""Which stage's fp16 stores account for the path's final-stage logit differences?   (round 5, review item 2b; CPU only)

    python tools/diag_logit_error_stages.py [--frames 8] [--global-frames 24] [--height 600 --width 1000] [--blocks 3,4,23,3]

The GPU path's final-stage logits differ from the fp32 oracle's by 1.2e-2 at the median in the trained-like regime (R101 x1, full
size), which at scores near 0.5 is a score difference of 3e-3 -- the reason the contract's |dscore| <= 5e-3 holds for 97.7 % of the
candidate slots, not 99 %.  The CPU oracle under the path's fp16 STORAGE POLICY (oracle/precision.py) reproduces that difference without
any HIP kernel.  This tool applies the policy to ONE stage at a time (precision.use("fp16", only=[stage-tag prefixes])) on the same
video and prints each stage's own |dlogit| against the fp32 oracle (median / p90 / p99 over the final stage's 8 x 300 x 30 logits) and
its share of the whole policy's squared error: what it would buy to keep that stage's operands wider (e.g. a split hi + lo fp16 A
operand for one K = 256 product).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--global-frames", type=int, default=24)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--blocks", type=str, default="3,4,23,3")
    ap.add_argument("--out", type=str, default="")
    args = ap.parse_args()
    from diffusionvid_amd.config import get_cfg
    from diffusionvid_amd.data.synthetic_video import SyntheticVIDDataset
    from diffusionvid_amd.modeling.detector import build_detection_model
    from diffusionvid_amd.utils import synthetic
    from oracle import backbone_r101, detector as odet, precision
    blocks = tuple(int(b) for b in args.blocks.split(","))
    cfg = get_cfg(os.path.join(ROOT, "configs/vid_R_101_DiffusionVID.yaml"), ["DTYPE", "float16", "MODEL.VID.MEGA.GLOBAL.SIZE", args.global_frames],
                  os.path.join(ROOT, "configs/BASE_RCNN_1gpu.yaml"))
    cfg.MODEL.RESNETS.BLOCKS_OVERRIDE = None if blocks == (3, 4, 23, 3) else blocks
    cfg.freeze()
    model = build_detection_model(cfg)           # host object only: the state_dict and the frame protocol (no engine is built)
    sd = synthetic.trained_like_scores(synthetic.tame_box_deltas(model.state_dict(), 0.1))          # tests/test_gpu_e2e.py::_weights("trained_like")
    sd = {k: v.detach().cpu() for k, v in sd.items()}
    L = args.frames
    ds = SyntheticVIDDataset([L], cfg, height=args.height, width=args.width, device="cpu", smooth=True)
    images = ds[0][0]                             # the frame dict of the video's first call -> the oracle's item (tests/test_gpu_e2e.py::_oracle_items)
    oitem = {k: v for k, v in images.items() if k not in ("cur", "ref_l", "ref_g")}
    oitem["cur"] = images["cur"].tensors.cpu()
    oitem["image_size"] = tuple(images["cur"].image_sizes[0])
    oitem["ref_l"] = [im.tensors.cpu() for im in images["ref_l"]]
    oitem["ref_g"] = [im.tensors.cpu() for im in images["ref_g"]]
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    feats_cache = {}

    def run(only, backbone_policy):
        """oracle forward with the fp16 policy on the stages in `only` (None: everywhere; (): nowhere = fp32); the backbone's features
        are computed once per backbone policy and re-used (they do not depend on the heads' policy)"""
        ocfg = odet.DetCfg(sample_step=1, infer_batch=L, all_frame_interval=L, blocks=blocks)

        def backbone_fn(x):
            key = (backbone_policy, tuple(x.shape), float(x.double().sum()))
            if key not in feats_cache:
                with precision.use("fp16" if backbone_policy else "fp32"):
                    feats_cache[key] = backbone_r101.backbone_r101_fpn(x, sd, "backbone.", blocks)
            return feats_cache[key]
        o = odet.OracleDiffusionDet(sd, ocfg, synthetic.noise_fn, backbone_fn=backbone_fn)
        with torch.no_grad():
            if only == ():
                o.forward(oitem)
            else:
                with precision.use("fp16", only=only):
                    o.forward(oitem)
        return o.taps["final_0"][0].reshape(-1).double(), o.taps["extract"][0].reshape(-1).double()

    t0 = time.time()
    ref_final, ref_ext = run((), False)
    say(f"# R101 {blocks} x1, {L} local + {args.global_frames} global frames of {args.width}x{args.height}, trained-like class layers; fp32 oracle in {time.time() - t0:.0f} s; "
        f"final-stage logits: {ref_final.numel()} values, rms {ref_final.pow(2).mean().sqrt():.2f}")
    heads = ["head.head_series.%d." % i for i in range(3)]
    cond = "head.head_series_cond.0."
    cases = [("everything (the path's policy)", None, True),
             ("backbone only", ["backbone"], True),
             ("all heads, fp32 backbone", ["head."], False),
             ("extraction heads 0-2", heads, False),
             ("  extraction head 0", [heads[0]], False),
             ("  extraction head 1", [heads[1]], False),
             ("  extraction head 2", [heads[2]], False),
             ("global attention", ["head.global_attention"], False),
             ("cond head (whole)", [cond], False)]
    for part in ("roi", "attn", "dynconv", "ffn", "mod", "cls_tower", "class_logits"):
        cases.append((f"  cond head: {part}", [cond + part], False))
    for part in ("roi", "attn", "dynconv", "ffn", "mod", "cls_tower", "class_logits"):
        cases.append((f"  every head: {part}", [h + part for h in heads + [cond]], False))
    whole = None
    say(f"{'fp16 storage policy applied to':44s} {'|dlogit| median':>16s} {'p90':>10s} {'p99':>10s} {'rms':>10s} {'share of the whole policy (rms^2)':>34s}")
    for name, only, bb in cases:
        t0 = time.time()
        fin, _ = run(only, bb)
        d = (fin - ref_final).abs()
        q = np.quantile(d.numpy(), [0.5, 0.9, 0.99])
        rms = float(d.pow(2).mean().sqrt())
        if whole is None:
            whole = rms
        say(f"{name:44s} {q[0]:16.3e} {q[1]:10.3e} {q[2]:10.3e} {rms:10.3e} {rms * rms / (whole * whole):34.2f}   ({time.time() - t0:.0f} s)")
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
