import sys, os, time, cProfile, pstats, torch
sys.path.insert(0, "/root/repo")
import bench
from diffusionvid_amd.config import get_cfg
from diffusionvid_amd.data.synthetic_video import SyntheticVIDDataset
from diffusionvid_amd.modeling.detector import build_detection_model
from diffusionvid_amd.utils import synthetic
ROOT = "/root/repo"
cfg = get_cfg(os.path.join(ROOT, "configs/vid_R_101_DiffusionVID.yaml"), None, os.path.join(ROOT, "configs/BASE_RCNN_1gpu.yaml")); cfg.freeze()
model = build_detection_model(cfg).to("cuda").eval()
model.noise_fn = synthetic.noise_fn
model.results_on_host = True
ds = SyntheticVIDDataset([304], cfg, device="cuda"); ds.preload()
with torch.no_grad():
    bench.run_video(model, ds, "cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter(); bench.run_video(model, ds, "cuda"); torch.cuda.synchronize(); print("video wall %.1f ms" % ((time.perf_counter()-t0)*1e3))
    # host-only cost: time spent outside of waiting for the GPU = total - (time in .cpu()/.tolist sync)
    pr = cProfile.Profile(); pr.enable(); bench.run_video(model, ds, "cuda"); torch.cuda.synchronize(); pr.disable()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
