"""Streaming mode alone (INFER_BATCH 1, online memory update; bench.py's other_configs.r101_x1_streaming): frames/s and ms per frame
of one 304-frame video.   python tools/bench_streaming.py [frames]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusionvid_amd.config import get_cfg  # noqa: E402
from diffusionvid_amd.data.synthetic_video import SyntheticVIDDataset  # noqa: E402
from diffusionvid_amd.modeling.detector import build_detection_model  # noqa: E402
from diffusionvid_amd.utils import synthetic  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 304
cfg = get_cfg(os.path.join(ROOT, "configs/vid_R_101_DiffusionVID.yaml"),
              ["DTYPE", "float16", "INPUT.INFER_BATCH", 1, "MODEL.VID.MEGA.MAX_OFFSET", 0, "MODEL.VID.MEGA.MIN_OFFSET", 0,
               "MODEL.VID.MEGA.ALL_FRAME_INTERVAL", 1, "MODEL.VID.MEGA.KEY_FRAME_LOCATION", 0,
               "MODEL.VID.MEGA.GLOBAL.STOP_UPDATE_AFTER_INIT_TEST", False], os.path.join(ROOT, "configs/BASE_RCNN_1gpu.yaml"))
cfg.freeze()
model = build_detection_model(cfg).to("cuda").eval()
model.noise_fn = synthetic.DeviceNoise() if os.environ.get("DVID_HOST_NOISE", "0") != "1" else synthetic.noise_fn
model.results_on_host = True
ds = SyntheticVIDDataset([L], cfg, device="cuda")
ds.preload()


def run():
    n = 0
    for idx in range(len(ds)):
        images, _, ids = ds[idx]
        n += len(model(images))
    return n


with torch.no_grad():
    run()
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = run() + run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("streaming: %.1f frames/s, %.3f ms per frame (%d frames)" % (n / dt, dt / n * 1e3, n))
