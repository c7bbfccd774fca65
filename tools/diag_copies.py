#!/usr/bin/env python
"""Where do the small copies of a steady-state video come from?  One bench video (R101 x1, look-ahead 13) under
torch.profiler: memcpy / memset activity by kind, and the aten ops by call count."""
import os, sys, collections
import torch
from torch.profiler import profile, ProfilerActivity
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from diffusionvid_amd.config import get_cfg
from diffusionvid_amd.data.synthetic_video import SyntheticVIDDataset
from diffusionvid_amd.modeling.detector import build_detection_model
from diffusionvid_amd.utils import synthetic

device = torch.device("cuda", 0)
cfg = get_cfg(os.path.join(ROOT, "configs/vid_R_101_DiffusionVID.yaml"), ["DTYPE", "float16", "INPUT.LOOKAHEAD_BATCHES", 38],
              os.path.join(ROOT, "configs/BASE_RCNN_1gpu.yaml"))
cfg.freeze()
model = build_detection_model(cfg).to(device).eval()
model.noise_fn = synthetic.DeviceNoise()
model.results_on_host = True
ds = SyntheticVIDDataset([304], cfg, height=600, width=1000, device=device, emit_ref_ahead=False)
ds.preload()
with torch.no_grad():
    bench.run_video(model, ds, device); bench.run_video(model, ds, device)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        bench.run_video(model, ds, device)
        torch.cuda.synchronize()
ka = prof.key_averages()
print("## device-side copies / fills")
for e in ka:
    if any(s in e.key for s in ("Memcpy", "Memset", "copyBuffer", "fillBuffer", "copy", "Copy")):
        print("%-90s calls %5d  device %9.1f us  cpu %9.1f us" % (e.key[:90], e.count, e.device_time_total, e.cpu_time_total))
print("## aten ops by count")
for e in sorted(ka, key=lambda e: -e.count)[:40]:
    print("%-60s calls %5d  cpu %9.1f us  device %9.1f us" % (e.key[:60], e.count, e.cpu_time_total, e.device_time_total))
print("## copy_ call sites (python stacks)")
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::_to_copy", "aten::fill_", "aten::zero_") and ev.stack:
        fr = [s for s in ev.stack if "diffusionvid_amd" in s or "bench.py" in s]
        sites[(ev.name, fr[0] if fr else ev.stack[0])] += 1
for (name, site), c in sites.most_common(40):
    print("%5d  %-16s %s" % (c, name, site))
