"""Host-side view of one bench video: wall time, time in calls that do no GPU work, where the host blocks."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffusionvid_amd.config import get_cfg  # noqa: E402
from diffusionvid_amd.data.synthetic_video import SyntheticVIDDataset  # noqa: E402
from diffusionvid_amd.modeling.detector import build_detection_model  # noqa: E402
from diffusionvid_amd.utils import synthetic  # noqa: E402

la = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ss = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = get_cfg(os.path.join(ROOT, "configs/vid_R_101_DiffusionVID.yaml"), ["DTYPE", "float16", "INPUT.LOOKAHEAD_BATCHES", la, "MODEL.DiffusionDet.SAMPLE_STEP", ss], os.path.join(ROOT, "configs/BASE_RCNN_1gpu.yaml"))
cfg.freeze()
model = build_detection_model(cfg).to("cuda").eval()
model.noise_fn = synthetic.noise_fn
model.results_on_host = True
ds = SyntheticVIDDataset([304], cfg, device="cuda")
ds.preload()
with torch.no_grad():
    bench.run_video(model, ds, "cuda")
    bench.run_video(model, ds, "cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idle_calls = work_calls = 0.0
    per_call = []
    for idx in range(len(ds)):
        a = time.perf_counter()
        images, _, ids = ds[idx]
        b = time.perf_counter()
        out = model(images)
        c = time.perf_counter()
        if out:
            work_calls += c - a
            per_call.append((idx, (b - a) * 1e3, (c - b) * 1e3))
        else:
            idle_calls += c - a
    torch.cuda.synchronize()
    print("video wall %.1f ms; 266 no-op calls %.1f ms; 38 working calls %.1f ms" % ((time.perf_counter() - t0) * 1e3, idle_calls * 1e3, work_calls * 1e3))
    print("working calls (idx, dataset ms, model ms):", " ".join("%d:%.2f/%.2f" % p for p in per_call[:14]))
    pr = cProfile.Profile()
    pr.enable()
    bench.run_video(model, ds, "cuda")
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
