#!/usr/bin/env python
"""Where does the host-fed mode lose against resident frames?  Per 304-frame video: resident, host-fed, copies alone, and
compute on resident frames while the same copies run beside it."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd.config import get_cfg
from diffusionvid_amd.data.prefetch import HostFedVideo
from diffusionvid_amd.data.synthetic_video import SyntheticVIDDataset
from diffusionvid_amd.engine import inference as engine
from diffusionvid_amd.modeling.detector import build_detection_model
from diffusionvid_amd.utils import synthetic
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = get_cfg(os.path.join(ROOT, "configs/vid_R_101_DiffusionVID.yaml"), ["DTYPE", "float16", "INPUT.LOOKAHEAD_BATCHES", 13], os.path.join(ROOT, "configs/BASE_RCNN_1gpu.yaml"))
dev = torch.device("cuda")
model = build_detection_model(cfg).to(dev).eval()
model.noise_fn = synthetic.noise_fn
model.results_on_host = True
L = 304
ds = SyntheticVIDDataset([L], cfg, device=dev, emit_ref_ahead=False); ds.preload()
hds = SyntheticVIDDataset([L], cfg, device="cpu", emit_ref_ahead=False)
hf = HostFedVideo(hds, dev, 104, cyclic=True).pin()
def run(d):
    n = 0
    for idx, (images, _, ids) in engine.lookahead_items(d, range(len(d)), 8, 13):
        n += len(model(images))
    return n
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
with torch.no_grad():
    run(ds); run(ds)
    print("resident            %.1f ms per video" % t(lambda: run(ds)))
    print("host-fed            %.1f ms per video" % t(lambda: run(hf)))
    hf.attach(model)
    print("host-fed (prefetch issued behind the call's uploads) %.1f ms per video" % t(lambda: run(hf)))
    model.after_first_launch = None
    hf._model = None
    def copies():
        for g in range(3):
            hf._stage(0, g, g & 1)
    print("copies alone        %.1f ms per video" % t(copies))
    def both():
        for idx, (images, _, ids) in engine.lookahead_items(ds, range(len(ds)), 8, 13):
            if images["frame_id"] % 104 == 0:
                hf._stage(0, (images["frame_id"] // 104 + 1) % 3, (images["frame_id"] // 104 + 1) & 1)
            model(images)
    print("resident + copies   %.1f ms per video" % t(both))
    # per-group timeline of the host-fed pass
    for idx, (images, _, ids) in engine.lookahead_items(hf, range(len(hf)), 8, 13):
        if images["frame_id"] % 104 == 0:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model(images)
        if images["frame_id"] % 104 == 0:
            torch.cuda.synchronize(); print("  group at frame %3d: first call %.1f ms" % (images["frame_id"], (time.perf_counter() - t0) * 1e3))
