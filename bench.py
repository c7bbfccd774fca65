#!/usr/bin/env python
"""DiffusionVID-R101 x1 inference throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path over one synthetic video: L = 304 frames of 1000x600
(padded to 608x1024), i.e. 24 global + 8 local frames through backbone + 3 heads and the
farthest-point memory pruning once, then 38 batches of 8 frames through backbone -> 3x RCNNHead ->
global cross-attention -> RCNNHead_cond -> top-k/NMS, with the reference's per-item call protocol
(304 model() calls, 38 of which do work).  Frames are resident in HBM before the timed region.
value = frames emitted by all ranks / max-over-ranks wall time (barrier + device sync on both sides).
N > 1: one process per GPU, each rank owns whole videos (weak scaling, no data-path collective); the
single RCCL gather of the predictions to rank 0 is inside the timed region.

Schedule: INPUT.LOOKAHEAD_BATCHES (--lookahead, default 13 = 104 frames) 8-frame batches are processed as one group:
every stage is per-frame independent given the video's global memory, so the group shares its launches and its one
host sync; --lookahead 1 is the reference's schedule and gives the same detections (tests/test_gpu_e2e.py::test_lookahead_batches_do_not_change_results).

Extra objects on the JSON line:
  roofline     the dominant kernel is the implicit-GEMM MFMA conv/linear kernel (igemm2_kernel<...>, ~80 % of the
               GPU time, profiles/r01h_kernel_stats.txt).  An instrumented repeat of one step right after the timed
               region brackets every launch with HIP events on its launch stream (sub-batch chains off, so launches
               do not overlap) and sums durations, algorithmic FLOP (2*M*N*K) and algorithmic HBM bytes (input +
               weights + output + residual, each once).  The bound is the lower roof at the measured intensity
               (ridge = 2500 TFLOP/s / 8 TB/s = 312 FLOP/B, MI355X_MICROARCH.md): below the ridge
               achieved/peak are GB/s against 8000, above it TFLOP/s against 2500; both fractions are always
               printed (hbm_frac, mfma_frac).  traffic = measured HBM bytes per launch from the committed
               rocprofv3 --pmc passes (tools/profile_round.sh).
  cpu_baseline the CPU oracle (oracle/, PyTorch CPU fp32, a port of the reference path) timed on the host
               cores of this box on ONE steady-state call of 4 frames at the same size.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from diffusionvid_amd import _lib  # noqa: E402
from diffusionvid_amd.config import get_cfg  # noqa: E402
from diffusionvid_amd.data.synthetic_video import SyntheticVIDDataset  # noqa: E402
from diffusionvid_amd.engine import inference as engine  # noqa: E402
from diffusionvid_amd.modeling.detector import build_detection_model  # noqa: E402
from diffusionvid_amd.utils import comm, synthetic  # noqa: E402

TRAFFIC_FILE = "r01h_pmc_igemm_traffic.json"
PEAK_FP16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def run_video(model, ds, device):
    results = {}
    for idx in range(len(ds)):
        images, _, ids = ds[idx]
        out = model(images)
        if out:
            results.update({i: o for i, o in zip(ids, out)})
    return results


def cpu_baseline(cfg, sd, frames, height, width):
    """Oracle (CPU port) on one steady-state 8-frame batch; returns dict for the JSON line."""
    from oracle import backbone_r101, detector as odet
    # pick the thread count that is actually fastest on this host (256 OpenMP threads on small ops can
    # be an order of magnitude slower than 32): one-frame backbone probe per candidate
    best_t, best_dt = None, None
    with torch.no_grad():
        for nt in sorted({min(c, os.cpu_count()) for c in (16, 32, 64, 128, os.cpu_count())}):
            torch.set_num_threads(nt)
            t0 = time.perf_counter()
            backbone_r101.resnet_bottom_up(backbone_r101.normalizer(frames[0], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD), sd,
                                           "backbone.bottom_up.", (3, 4, 6, 3))
            d = time.perf_counter() - t0
            if best_dt is None or d < best_dt:
                best_t, best_dt = nt, d
    torch.set_num_threads(best_t)
    ocfg = odet.DetCfg()
    oracle = odet.OracleDiffusionDet(sd, ocfg, synthetic.noise_fn)
    from collections import deque
    g = torch.Generator().manual_seed(7)
    oracle.local_img_queue = []
    oracle.mem = [torch.randn(900, 256, generator=g), torch.randn(150, 256, generator=g)]
    oracle.feats = deque(maxlen=8)
    oracle.classes_300, oracle.proposals_300, oracle.proposals_feat_300 = deque(maxlen=8), deque(maxlen=8), deque(maxlen=8)
    nb = len(frames)
    item = {"cur": frames[0], "image_size": (height, width), "ref_l": frames, "ref_g": [], "frame_category": 1,
            "frame_id": 8, "start_id": 0, "end_id": 8 + nb - 1, "seg_len": 8 + nb, "last_queue_id": 15}
    t0 = time.perf_counter()
    with torch.no_grad():
        out = oracle.forward(item)
    dt = time.perf_counter() - t0
    assert len(out) == nb
    return {"value": round(nb / dt, 4), "unit": "frames/sec", "cores": torch.get_num_threads(), "host_cpus": os.cpu_count(),
            "kind": "port",
            "sample": "1 steady-state call on %d frames 1000x600 (R101-FPN + 3 RCNNHead + global attention + RCNNHead_cond "
                      "+ top-k/NMS; per-video init excluded), CPU oracle fp32, %.1f s wall, %d threads (fastest of a "
                      "16..%d probe)" % (nb, dt, torch.get_num_threads(), os.cpu_count())}


def launch_ranks(n):
    """`python bench.py --gpus N ...` -> N ranks of this same command line under torch.distributed.run (127.0.0.1
    rendezvous on a free port); returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL between processes on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """--dry: no GPU work.  Proves the launch + collective path: N processes, gloo group, each rank fabricates the
    predictions of its own synthetic video, one gather to rank 0, one JSON line."""
    from diffusionvid_amd.structures.bounding_box import BoxList
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        comm.init_dist("gloo")
        assert dist.get_world_size() == args.gpus
    L = args.frames
    res = {}
    for f in range(L):
        k = (f * 7 + rank) % 5
        bl = BoxList(torch.full((k, 4), float(rank)), (1000, 600))
        bl.add_field("scores", torch.linspace(0.9, 0.5, k))
        bl.add_field("labels", torch.full((k,), rank % 30 + 1))
        res[rank * L + f] = bl
    merged = engine.gather_predictions(res) if world > 1 else res
    if rank == 0:
        ranks_seen = sorted({int(k) // L for k in merged})
        print(json.dumps({"metric": "dry run (launcher + gather only)", "n_gpus": world, "ranks_seen": ranks_seen,
                          "frames_gathered": len(merged), "backend": "gloo" if world > 1 else "none"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=304, help="frames per synthetic video")
    ap.add_argument("--lookahead", type=int, default=0,
                    help="INPUT.LOOKAHEAD_BATCHES: INFER_BATCH groups whose backbone + extraction heads share one launch sequence "
                         "(1 = the reference's schedule; default 0 = 104 frames' worth: 13 for R101, 26 for Swin-B)")
    ap.add_argument("--arch", choices=("r101", "swinb"), default="r101",
                    help="r101 = the BASELINE.json headline configuration; swinb = configs/vid_Swin_B_DiffusionVID.yaml (INFER_BATCH 4)")
    ap.add_argument("--sample-step", type=int, default=1, help="MODEL.DiffusionDet.SAMPLE_STEP (4 = the x4 configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry", action="store_true",
                    help="launcher / collective check without a GPU: every rank fabricates its shard's predictions, the gather to "
                         "rank 0 runs over gloo, and the JSON line reports the ranks seen (tests/test_dist_gloo.py)")
    args = ap.parse_args()

    # One process per GPU (the reference: `python -m torch.distributed.launch --nproc_per_node N tools/test_net.py`,
    # README.md:100-106, mega_core/utils/dist_env.py:18-23).  Started bare with --gpus N > 1, this process becomes the
    # launcher: it re-executes itself N times through torch.distributed.run and relays the exit code.
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the DiffusionVID hot path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d has no GPU: %d visible device(s) for --gpus %d" % (rank, torch.cuda.device_count(), args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        comm.init_dist("nccl")
        assert dist.get_world_size() == args.gpus and dist.get_backend() == "nccl"

    headline = args.arch == "r101" and args.sample_step == 1
    yaml = "configs/vid_R_101_DiffusionVID.yaml" if args.arch == "r101" else "configs/vid_Swin_B_DiffusionVID.yaml"
    if args.lookahead <= 0:
        args.lookahead = 13 if args.arch == "r101" else 26
    cfg = get_cfg(os.path.join(ROOT, yaml), ["DTYPE", "float16", "INPUT.LOOKAHEAD_BATCHES", args.lookahead,
                                             "MODEL.DiffusionDet.SAMPLE_STEP", args.sample_step],
                  os.path.join(ROOT, "configs/BASE_RCNN_1gpu.yaml"))
    cfg.freeze()
    model = build_detection_model(cfg).to(device).eval()
    model.noise_fn = synthetic.noise_fn
    model.results_on_host = True      # one D2H copy per 8-frame batch (results end up on the host either way)
    H, W, L = 600, 1000, args.frames
    ds = SyntheticVIDDataset([L], cfg, height=H, width=W, device=device, video_base=rank)
    ds.preload()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # set-up, outside the step accounting: weight repack / upload and the per-shape tile tuner (it times every
        # configuration on the first launch of each GEMM shape; DVID_IGEMM_TUNE_CACHE makes that persistent)
        run_video(model, ds, device)
        for _ in range(args.warmup):
            run_video(model, ds, device)
        barrier()
        t0 = time.perf_counter()
        frames = 0
        results = {}
        for s in range(args.steps):
            r = run_video(model, ds, device)
            frames += len(r)
            results.update({k + s * L + rank * args.steps * L: v.to("cpu") for k, v in r.items()})
        merged = engine.gather_predictions(results, device=device) if world > 1 else results
        barrier()
        dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    ff = torch.tensor([frames], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ff, op=dist.ReduceOp.SUM)
    dt, total_frames = float(tt.item()), float(ff.item())

    # ---- roofline of the dominant kernel: instrumented repeat of one step -------------------------
    roofline = None
    lib = _lib.load()
    # per-launch HIP events on the library's stream; sub-batch chains are switched off for this pass so that launches
    # do not overlap and each event pair times one kernel alone (the same condition rocprofv3 summaries in profiles/
    # are taken under: DVID_CHAINS=1)
    engine_model = model._get_engine()
    engine_model.set_chains(1)
    with torch.no_grad():
        run_video(model, ds, device)      # un-instrumented: lets the per-shape tile tuner see the chains=1 launch shapes first
    torch.cuda.synchronize()
    lib.dvid_profile_reset()
    lib.dvid_profile_enable(1)
    with torch.no_grad():
        run_video(model, ds, device)
    torch.cuda.synchronize()
    engine_model.set_chains(int(os.environ.get("DVID_CHAINS", "2")))
    ms, fl, nl = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _lib.check(lib.dvid_profile_read(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(nl)), "dvid_profile_read")
    ab = ctypes.c_double()
    _lib.check(lib.dvid_profile_read_bytes(ctypes.byref(ab)), "dvid_profile_read_bytes")
    lib.dvid_profile_enable(0)
    if os.environ.get("DVID_PROFILE_DUMP") and rank == 0:
        lib.dvid_profile_dump(os.environ["DVID_PROFILE_DUMP"].encode())
    lib.dvid_profile_reset()
    traffic = mfma_busy = None
    tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if os.path.exists(tpath):      # HBM bytes per launch from the committed rocprofv3 --pmc passes (cannot be collected in-process)
        pmc = json.load(open(tpath))
        traffic = round(pmc["hbm_bytes_per_launch"])
        mfma_busy = pmc.get("mfma_busy_fraction")
    if ms.value > 0:
        sec = ms.value * 1e-3
        tflops = fl.value / sec / 1e12
        gbs = ab.value / sec / 1e9
        intensity = fl.value / max(ab.value, 1.0)                    # algorithmic FLOP per HBM byte over all igemm launches
        ridge = PEAK_FP16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)       # 312 FLOP/B: below it the HBM roof is the lower one
        hbm_bound = intensity < ridge
        roofline = {"bound": "hbm" if hbm_bound else "mfma", "kernel": "igemm2_kernel (implicit-GEMM conv/linear, fp16 MFMA)",
                    "achieved": round(gbs if hbm_bound else tflops, 2), "peak": PEAK_HBM_GBS if hbm_bound else PEAK_FP16_TFLOPS,
                    "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": round(gbs / PEAK_HBM_GBS if hbm_bound else tflops / PEAK_FP16_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of tools/profile_round.sh on this "
                                      "workload, bytes per launch; its own algorithmic figure is in the file)" % TRAFFIC_FILE,
                    "alg_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                    "mfma_tflops": round(tflops, 2), "mfma_frac": round(tflops / PEAK_FP16_TFLOPS, 4),
                    "mfma_busy_pmc": None if mfma_busy is None else round(mfma_busy, 4),
                    "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4),
                    "alg_mbytes_per_launch": round(ab.value / max(1, nl.value) / 1e6, 2),
                    "alg_gflop_per_launch": round(fl.value / max(1, nl.value) / 1e9, 3),
                    "launches_per_step": int(nl.value), "avg_launch_us": round(ms.value * 1e3 / max(1, nl.value), 2),
                    "kernel_ms_per_step": round(ms.value, 2)}

    if rank == 0:
        line = {
            "metric": "frames/sec (1000x600) DiffusionVID-%s x%d" % ("R101" if args.arch == "r101" else "SwinB", args.sample_step),
            "value": round(total_frames / dt, 2), "unit": "frames/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "%s DiffusionVID x%d fp16, 300 boxes, %d DDIM step(s); one step = one synthetic "
                                   "%d-frame 1000x600 video per GPU (24 global + %d local frames, %d batches of %d)"
                                   % ("ResNet-101" if args.arch == "r101" else "Swin-Base", args.sample_step, args.sample_step, L, L,
                                      -(-L // cfg.INPUT.INFER_BATCH), cfg.INPUT.INFER_BATCH),
                       "frames_per_step_per_gpu": L, "infer_batch": cfg.INPUT.INFER_BATCH, "lookahead_batches": args.lookahead,
                       "parallelism": "videos sharded across ranks (one process per GPU, %s)"
                                      % ("RCCL group of %d ranks: one gather of the predictions to rank 0" % dist.get_world_size()
                                         if world > 1 else "single rank, no collective"),
                       "ranks": world},
            "roofline": roofline,
        }
        if world == 1 and headline and not args.no_cpu_baseline:
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            frames8 = [ds.frame(0, i).tensors.cpu() for i in range(8, 12)]
            line["cpu_baseline"] = cpu_baseline(cfg, sd, frames8, H, W)
            line["gpu_over_cpu"] = round(line["value"] / max(line["cpu_baseline"]["value"], 1e-9), 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
