#!/usr/bin/env python
"""Where does a free-running x4 run of the GPU path leave the CPU oracles?   (round 5; GPU box)

    python tools/diag_x4_divergence.py [--global-frames 4] [--frames 4] [--video 0]

One x4 video (R101 full depth, 1000x600, trained-like scores) through the GPU path, the fp32 oracle and the fp16-policy oracle, all three
free-running, with every stage's outputs kept: extraction logits, the memory, and per DDIM step the final-stage logits and the keep
decisions (best sigmoid score > 0.5, diffusion_det.py:559-565) -- the first stage at which the GPU path is further from the fp32 oracle
than the policy oracle is says where to look.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--global-frames", type=int, default=4)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--video", type=int, default=0)
    args = ap.parse_args()
    os.makedirs("gpurun_out", exist_ok=True)
    import test_gpu_e2e as T
    from diffusionvid_amd.data.synthetic_video import SyntheticVIDDataset
    from diffusionvid_amd.utils import synthetic
    from oracle import detector as odet, precision
    L = args.frames
    cfg, model = T._build(4, None, "trained_like", extra=["MODEL.VID.MEGA.GLOBAL.SIZE", args.global_frames, "INPUT.INFER_BATCH", L, "MODEL.VID.MEGA.MAX_OFFSET", L - 1,
                                                        "MODEL.VID.MEGA.ALL_FRAME_INTERVAL", L])          # the oracle splits by L: the draws are keyed by (split, image)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ds = SyntheticVIDDataset([L], cfg, height=600, width=1000, device="cuda", smooth=True, video_base=args.video)
    model.noise_fn = synthetic.noise_fn
    model.debug_taps = {}
    images, oitem, _ = T._oracle_items(ds, 0)
    torch.set_num_threads(min(32, torch.get_num_threads()))

    def oracle_run(policy):
        ocfg = odet.DetCfg(sample_step=4, infer_batch=L, all_frame_interval=L)
        ocfg.head.sampling_timesteps = 4
        o = odet.OracleDiffusionDet(sd, ocfg, synthetic.noise_fn)
        with torch.no_grad():
            if policy:
                with precision.use("fp16"):
                    o.forward(oitem)
            else:
                o.forward(oitem)
        return o
    with torch.no_grad():
        model(images)
    o32, o16 = oracle_run(False), oracle_run(True)
    g = model.debug_taps

    def q(a, b):
        d = (a.float().cpu() - b.float().cpu()).abs().reshape(-1)
        return "median %.2e p99 %.2e max %.2e" % (d.median().item(), d.quantile(0.99).item() if d.numel() < 10_000_000 else float(np.quantile(d.numpy(), 0.99)), d.max().item())
    gcl = torch.cat([e[0] for e in g["extract"]]).cpu()
    print("extraction logits   GPU - fp32:", q(gcl, o32.taps["extract"][0]), "| policy - fp32:", q(o16.taps["extract"][0], o32.taps["extract"][0]))
    gm = [m.cpu() for m in g["memory"]]
    for i in range(2):
        a, b, c = gm[i], o32.mem[i], o16.mem[i]
        print(f"memory {i}: rows GPU {tuple(a.shape)} fp32 {tuple(b.shape)} policy {tuple(c.shape)}; "
              + (f"GPU - fp32 {q(a, b)} | policy - fp32 {q(c, b)}" if a.shape == b.shape == c.shape else "SHAPES DIFFER"))
    for step in range(4):
        key = f"final_{step}"
        if key not in g or key not in o32.taps:
            continue
        gl, l32, l16 = g[key][0].float().cpu().reshape(L, -1, 30), o32.taps[key][0].reshape(L, -1, 30), o16.taps[key][0].reshape(L, -1, 30)
        k = lambda x: (torch.sigmoid(x).max(-1).values > 0.5)          # noqa: E731
        kg, k32, k16 = k(gl), k(l32), k(l16)
        print(f"step {step}: final logits GPU - fp32: {q(gl, l32)} | policy - fp32: {q(l16, l32)} | kept boxes per frame GPU {kg.sum(-1).tolist()} fp32 {k32.sum(-1).tolist()} "
              f"policy {k16.sum(-1).tolist()} | keep decisions differing from fp32: GPU {(kg != k32).sum(-1).tolist()} policy {(k16 != k32).sum(-1).tolist()}")
        ik = f"img_{step + 1}"
        if ik in o32.taps and ik in o16.taps:
            print(f"         renewed boxes (oracle taps) policy - fp32: {q(o16.taps[ik], o32.taps[ik])}")


if __name__ == "__main__":
    main()
