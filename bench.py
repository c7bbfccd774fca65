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

Schedule: INPUT.LOOKAHEAD_BATCHES (--lookahead, default 38 = the whole 304-frame video; 104-frame groups measured 3-5 % slower) 8-frame batches are processed as one group:
every stage is per-frame independent given the video's global memory, so the group shares its launches and its one
host sync; --lookahead 1 is the reference's schedule and gives the same detections (tests/test_gpu_e2e.py::test_lookahead_batches_do_not_change_results).

Extra objects on the JSON line:
  roofline     the dominant kernels are the implicit-GEMM MFMA conv/linear kernels (igemm2_kernel<...>, and conv3x3_* for the
               3x3 / stride-1 layers; ~80 % of the GPU time, profiles/r02d_kernel_stats.txt).  An instrumented repeat of one
               step right after the timed region brackets every such launch with HIP events on its launch stream (sub-batch
               chains off, so launches do not overlap) and sums durations, algorithmic FLOP (2*M*N*K) and the layer-wise
               byte model (input + weights + output + residual, each once).  bound = "mfma" (249.3 GFLOP
               against 60-90 MB of ideal-fusion HBM traffic per frame); achieved = FLOP / summed durations against the
               2500 TFLOP/s dense fp16 peak.  traffic = measured HBM bytes per launch from the committed rocprofv3 --pmc
               passes (tools/profile_round.sh), labelled as coming from that profile.
  host_fed     the same workload with the frames in pinned host memory, H2D inside the timed region.
  other_configs  the reference call protocol without look-ahead, R101 x4, Swin-B x1 (each its own model, timed the same way).
  cpu_baseline the CPU oracle (oracle/, PyTorch CPU fp32, a port of the reference path) timed on the host cores of this
               box: one steady-state 8-frame call after a warm-up call, thread count chosen by a probe.
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

from diffusionvid_amd import _lib, ops  # noqa: E402
from diffusionvid_amd.config import get_cfg  # noqa: E402
from diffusionvid_amd.data.synthetic_video import SyntheticVIDDataset  # noqa: E402
from diffusionvid_amd.engine import inference as engine  # noqa: E402
from diffusionvid_amd.modeling.detector import build_detection_model  # noqa: E402
from diffusionvid_amd.utils import comm, synthetic  # noqa: E402

# measured HBM bytes per implicit-GEMM launch (rocprofv3 --pmc passes of tools/profile_round.sh), one file PER CONFIGURATION: a line
# only ever carries the traffic collected on its own workload, never another configuration's
TRAFFIC_FILES = {("r101", 1): "r06_pmc_igemm_traffic_r101_x1.json", ("r101", 4): "r06_pmc_igemm_traffic_r101_x4.json",
                 ("swinb", 1): "r06_pmc_igemm_traffic_swinb_x1.json", ("r101", 1, "lookahead1"): "r06_pmc_igemm_traffic_r101_x1_lookahead1.json",
                 ("r101", 1, "float32"): "r06_pmc_igemm_traffic_r101_x1_float32.json"}


def _md5(path):
    import hashlib
    try:
        with open(path, "rb") as f:
            return hashlib.md5(f.read()).hexdigest()
    except OSError:
        return None


# the build this run measures: committed PMC traffic files carry the md5 of the library they were collected on, and a line whose
# traffic comes from another build says so
LIB_MD5 = _md5(os.path.join(ROOT, "diffusionvid_amd", "libdvid_hip.so"))
PEAK_FP16_TFLOPS = 2500.0
PEAK_FP32_TFLOPS = 157.3          # v_mfma_f32_32x32x2_f32, dense (MI355X_MICROARCH.md): the DTYPE float32 path's roof with library option f32_split = 0
# ... and with f32_split = 1 (default): every fp32-grade product is three passes of the fp16 MFMA over (hi, lo) operand pairs (csrc/f32.hip:
# f32x3_igemm_kernel), so the roof of that kernel in ALGORITHMIC (fp32-grade) FLOP is a third of the dense fp16 peak
PEAK_FP32_SPLIT_TFLOPS = round(PEAK_FP16_TFLOPS / 3.0, 1)
# the library's option table at its defaults (csrc/options.h): the configuration every number of this file is quoted on unless --option says otherwise;
# tests/test_host_logic.py::test_default_library_configuration_is_the_benchmarked_one holds the library to this string
DEFAULT_LIBRARY_CONFIG = "conv3x3=1 wstat=1 bneck_fuse=1 stem_pool=1 head_tail=1 roi_fuse=1 ln_rows=1 igemm_cfg=-1 igemm_tune=-1 igemm_generic=0 f32_split=1 f32_wstat=1 f32_conv3x3=1 bneck_lds=0"
PEAK_HBM_GBS = 8000.0
# algorithmic work per output frame: backbone + heads + global attention, faithful pass counts
ALG_GFLOP_PER_FRAME = {("r101", 1): 249.3, ("r101", 4): 385.0, ("swinb", 1): 459.6}


def top_kernels(csv_path, n=5, mfma_peak=PEAK_FP16_TFLOPS):
    """The per-launch table of the library (dvid_profile_dump: HIP events around every launch of the instrumented pass, chains off) grouped
    by (kernel, shape); the `n` heaviest groups, each against its own roof: algorithmic intensity below the ridge of the part (2500 TFLOP/s /
    8 TB/s = 312 FLOP/B) -> HBM bound, achieved = algorithmic bytes / time; else MFMA bound, achieved = algorithmic FLOP / time."""
    import csv
    groups, total = {}, 0.0
    with open(csv_path) as f:
        for r in csv.DictReader(f):
            ms = float(r["ms"])
            total += ms
            key = (r["kernel"], int(r["M"]), int(r["N"]), int(r["K"]), int(r["taps"]), int(r["stride"]), int(r["res_mode"]))
            g = groups.setdefault(key, [0, 0.0, 0.0, 0.0])
            g[0] += 1
            g[1] += ms
            g[2] += float(r["tflops"]) * ms * 1e9          # FLOP
            g[3] += float(r["alg_mbytes"]) * 1e6
    out = []
    for key, (calls, ms, flop, nbytes) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:n]:
        kernel, M, N, K, taps, stride, res = key
        hbm = nbytes > 0 and flop / nbytes < mfma_peak * 1e12 / (PEAK_HBM_GBS * 1e9)
        ach = nbytes / (ms * 1e-3) / 1e9 if hbm else flop / (ms * 1e-3) / 1e12
        out.append({"name": "%s [M %d, N %d, K %d%s%s%s]" % (kernel, M, N, K, ", %d taps" % taps if taps > 1 else "", ", stride %d" % stride if stride > 1 else "",
                                                             {0: "", 1: ", + residual", 2: ", + upsampled residual", 3: ", fused block tail", 4: ", fused block tail + shortcut"}.get(res, "")),
                    "launches": calls, "ms": round(ms, 3), "share": round(ms / total, 4) if total else None, "bound": "hbm" if hbm else "mfma",
                    "achieved": round(ach, 1), "unit": "GB/s" if hbm else "TFLOP/s", "peak": PEAK_HBM_GBS if hbm else mfma_peak,
                    "frac": round(ach / (PEAK_HBM_GBS if hbm else mfma_peak), 4)})
    return {"recorded_ms": round(total, 2), "what": "share = of the recorded kernel time of one step (implicit-GEMM family + RoIAlign, DynamicConv, attention, head tail, max pool; chains off)",
            "kernels": out}


def run_video(model, ds, device):
    """the reference's per-item loop (mega_core/engine/inference.py:22-94); the dataset emits the reference's unchanged item
    dict, the look-ahead hand-over is built by the engine from the items it reads ahead (engine.lookahead_items)"""
    results = {}
    for idx, (images, _, ids) in engine.lookahead_items(ds, range(len(ds)), model.infer_batch, model.lookahead):
        out = model(images)
        if out:
            results.update({i: o for i, o in zip(ids, out)})
    return results


def cpu_baseline(cfg, sd, frames, height, width):
    """Oracle (CPU port) on one steady-state 8-frame batch after a warm-up batch; returns dict for the JSON line."""
    from oracle import backbone_r101, detector as odet
    # the thread count that is actually fastest on this host FOR THE TIMED CALL'S SHAPE (256 OpenMP threads on small ops can be an
    # order of magnitude slower than 32): the full R101 bottom-up pass on the call's 8 frames per candidate (round 4 probed one frame
    # of a shallower net and landed on 16 threads of 256 CPUs; SURVEY.md 8(d) asks for os.cpu_count() -- both are reported)
    best_t, best_dt, probe = None, None, {}
    x8 = backbone_r101.normalizer(torch.cat(list(frames[:8])), cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD)
    with torch.no_grad():
        for nt in sorted({min(c, os.cpu_count()) for c in (16, 32, 64, 128, os.cpu_count())}):
            torch.set_num_threads(nt)
            t0 = time.perf_counter()
            backbone_r101.resnet_bottom_up(x8, sd, "backbone.bottom_up.")
            d = time.perf_counter() - t0
            probe[nt] = round(d, 2)
            if best_dt is None or d < best_dt:
                best_t, best_dt = nt, d
    del x8
    torch.set_num_threads(best_t)
    ocfg = odet.DetCfg()
    oracle = odet.OracleDiffusionDet(sd, ocfg, synthetic.noise_fn)
    from collections import deque
    g = torch.Generator().manual_seed(7)
    oracle.local_img_queue = []
    oracle.mem = [torch.randn(900, 256, generator=g), torch.randn(150, 256, generator=g)]
    oracle.feats = deque(maxlen=8)
    oracle.classes_300, oracle.proposals_300, oracle.proposals_feat_300 = deque(maxlen=8), deque(maxlen=8), deque(maxlen=8)
    nb = 8
    assert len(frames) >= 2 * nb

    def call(fr, frame_id):
        item = {"cur": fr[0], "image_size": (height, width), "ref_l": fr, "ref_g": [], "frame_category": 1,
                "frame_id": frame_id, "start_id": 0, "end_id": 303, "seg_len": 304, "last_queue_id": frame_id + 7}
        t0 = time.perf_counter()
        with torch.no_grad():
            out = oracle.forward(item)
        assert len(out) == len(fr)
        return time.perf_counter() - t0
    warm = call(frames[:nb], 8)             # warm-up call: thread pools, allocator, oneDNN primitive caches
    dt = call(frames[nb:2 * nb], 16)
    # the same call on every CPU of the host (os.cpu_count() threads), unless the probe already says it is far slower
    all_cpus = None
    if os.cpu_count() != best_t and probe.get(os.cpu_count(), 1e9) <= 4 * best_dt:
        torch.set_num_threads(os.cpu_count())
        call(frames[:nb], 8)
        all_cpus = round(nb / call(frames[nb:2 * nb], 16), 4)
        torch.set_num_threads(best_t)
    return {"value": round(nb / dt, 4), "unit": "frames/sec", "cores": best_t, "host_cpus": os.cpu_count(),
            # the container's CPU-time quota in CPUs (cgroup cpu.max; null = none): the GPU boxes of this pool show 256 CPUs and grant 16, which
            # is why the probe lands on 16 threads and every larger pool is slower
            "cpu_quota_cpus": comm.cpu_quota(),
            "value_all_host_cpus": all_cpus if os.cpu_count() != best_t else round(nb / dt, 4),
            "thread_probe_s": {str(k): v for k, v in probe.items()},
            "kind": "port",
            "sample": "frames 16-23 of the bench video (one steady-state call of the reference's per-batch protocol: R101-FPN + 3 "
                      "RCNNHead + global attention + RCNNHead_cond + top-k/NMS on 8 frames 1000x600; the per-video global-memory "
                      "initialisation is excluded), after one warm-up call on frames 8-15 (%.1f s); CPU oracle fp32, %.1f s wall, "
                      "%d threads (fastest of a 16..%d probe on the 8-frame R101 bottom-up pass: thread_probe_s; value_all_host_cpus = the same call on os.cpu_count() "
                      "threads, null when its probe was over 4x slower)" % (warm, dt, best_t, os.cpu_count())}


def _decode_u8(path):
    """what a DataLoader worker of the reference does per frame up to the tensor hand-over: open + decode (PIL, vid.py:96 /
    vid_mega.py:223-233) -> uint8 HWC array"""
    import numpy as np
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


_FEED_SHM = None


def _feed_attach(name):
    """pool initializer: every decode worker maps the parent's frame ring"""
    global _FEED_SHM
    from multiprocessing import shared_memory
    _FEED_SHM = shared_memory.SharedMemory(name=name)


def _decode_into(job):
    """decode `path` into slot `slot` of the shared frame ring and hand back the slot number only -- what a DataLoader worker does
    with a tensor (shared-memory hand-over, torch/utils/data/_utils/worker.py), instead of pickling 2.7 MB per frame through a pipe
    (which caps a pool at the parent's ~1 GB/s unpickling rate: 400-630 frames/s whatever the worker count, rounds 2-3)"""
    import numpy as np
    path, slot = job
    arr = _decode_u8(path)
    ring = np.ndarray((arr.size,), dtype=np.uint8, buffer=_FEED_SHM.buf, offset=slot * arr.size)
    ring[:] = arr.reshape(-1)
    return slot


def real_data_feed_rate(device, n_files=48, passes=4, feed_workers=None):
    """Can a host keep ONE GPU fed with real frames?  n_files synthetic 1280x720 JPEG files (quality 90, ~ImageNet-VID's 720p
    snippets) on local disk -> a pool of decode workers (the reference uses 16 PIL workers, configs/vid_R_101_DiffusionVID.yaml:69)
    -> uint8 [720, 1280, 3] frames -> pinned staging + H2D + the device's Pillow-exact resize / ToTensor / padding
    (dvid_resize_u8_to_f32).  Reports the decode rate of the pool and the rate of the upload + resize leg alone; the slower of
    the two is what a host-fed pipeline delivers per GPU (JPEG decode itself stays on the host: SURVEY.md 2, out of scope)."""
    import multiprocessing as mp
    import shutil
    import tempfile

    import numpy as np
    from PIL import Image

    from diffusionvid_amd.data import transforms as T
    d = tempfile.mkdtemp(prefix="dvid_feed_")
    try:
        rng = np.random.RandomState(0)
        paths = []
        for i in range(n_files):
            yy, xx = np.mgrid[0:720, 0:1280]
            img = np.stack([(128 + 100 * np.sin(xx / (37.0 + i) + c) * np.cos(yy / (53.0 + 2 * i))) for c in range(3)], -1)
            img = np.clip(img + rng.randn(720, 1280, 3) * 6, 0, 255).astype(np.uint8)
            paths.append(os.path.join(d, "%06d.JPEG" % i))
            Image.fromarray(img).save(paths[-1], format="JPEG", quality=90)
        # decode workers: the reference's 16 (configs/vid_R_101_DiffusionVID.yaml:69) and, when the host has them, 32 / 64 / 128 --
        # the knee is where the pool stops scaling; --feed-workers N measures one setting only
        cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2)
        sweep = [w for w in (feed_workers or (16, 32, 64, 128)) if w <= max(1, cpus - 1)] or [max(1, cpus - 1)]
        rates = {}
        from multiprocessing import shared_memory
        frame_bytes = 720 * 1280 * 3
        shm = shared_memory.SharedMemory(create=True, size=frame_bytes * n_files)
        try:
            jobs = [(pth, i) for i, pth in enumerate(paths)]
            for workers in sweep:
                with mp.get_context("fork").Pool(workers, initializer=_feed_attach, initargs=(shm.name,)) as pool:
                    pool.map(_decode_into, jobs[:min(workers, len(jobs))])                      # start-up
                    t0 = time.perf_counter()
                    n = 0
                    reps = max(passes, (6 * workers) // n_files + 1)                  # every worker gets several files
                    for _ in range(reps):
                        for _slot in pool.imap_unordered(_decode_into, jobs, chunksize=1):
                            n += 1
                    rates[workers] = n / (time.perf_counter() - t0)
        finally:
            shm.close()
            shm.unlink()
        workers = max(rates, key=rates.get)
        decode_fps = rates[workers]
        # the knee: the smallest pool within 10 % of the best rate
        knee = min(w for w in rates if rates[w] >= 0.9 * decode_fps)
        arrs = [_decode_u8(p) for p in paths[:16]]
        tf = T.ResizeToTensorDevice(device, 600, 1000, 32)
        for a in arrs[:4]:
            tf(a, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = 0
        for _ in range(passes * 3):
            for a in arrs:
                tf(a, False)
                m += 1
        torch.cuda.synchronize()
        dev_fps = m / (time.perf_counter() - t0)
        return {"decode_workers": workers, "host_cpus": os.cpu_count(), "cpu_quota_cpus": comm.cpu_quota(), "decode_frames_per_sec": round(decode_fps, 1),
                "decode_frames_per_sec_by_workers": {str(w): round(r, 1) for w, r in rates.items()}, "decode_workers_knee": knee,
                "upload_resize_frames_per_sec": round(dev_fps, 1), "frame": "1280x720 JPEG q90 -> uint8 HWC -> fp32 CHW 576x1000 padded to 576x1024",
                "delivered_frames_per_sec": round(min(decode_fps, dev_fps), 1),
                "what": "image files on local disk -> PIL decode in a worker pool, frames handed over through a shared-memory ring (as DataLoader "
                        "workers hand tensors over); then, measured separately, pinned staging + async H2D of the uint8 frame + dvid_resize_u8_to_f32 on one "
                        "stream from one host thread"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def vidval(args, build, timed, barrier, device, rank, world, H, W, embedded=False):
    """BASELINE.json configs[4]: the VID-val-shaped set (555 videos, 176126 frames, lengths 24..3262) sharded over the ranks
    by frame-count-balanced whole videos, every rank running the reference's per-item loop over its videos,
    one gather of the predictions to rank 0 inside the timed region.  Strong scaling: the set is fixed, N ranks split it."""
    from diffusionvid_amd.data.samplers import balanced_video_partition, vid_val_shaped_lengths
    from diffusionvid_amd.data.synthetic_video import PooledVIDDataset
    lens = vid_val_shaped_lengths()
    if args.videos > 0:
        lens = lens[:args.videos]
    mine = balanced_video_partition(lens, world)[rank]
    cfg, model = build(args.arch, args.sample_step, args.lookahead)
    ds = PooledVIDDataset([lens[v] for v in mine], cfg, pool=128, height=H, width=W, device=device, emit_ref_ahead=False)
    for v in range(128):
        ds.frame(0, v)
    warm = PooledVIDDataset([min(lens), 304], cfg, pool=128, height=H, width=W, device=device, emit_ref_ahead=False)
    warm._cache = ds._cache
    with torch.no_grad():
        run_video(model, warm, device)             # repack + tuner (row buckets make the ragged tails hit cached winners)
    barrier()
    t0 = time.perf_counter()
    with torch.no_grad():
        res = run_video(model, ds, device)
    nfr = len(res)
    grouped = dist.is_available() and dist.is_initialized()
    merged = engine.gather_predictions({(rank << 32) + k: v.to("cpu") for k, v in res.items()}, device=device, always=True) if grouped else res
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    ff = torch.tensor([nfr], dtype=torch.float64, device=device)
    if grouped:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ff, op=dist.ReduceOp.SUM)
    record = None
    if rank == 0:
        total = int(sum(lens))
        assert int(ff.item()) == total and len(merged) == total
        record = ({
            "metric": "frames/sec (1000x600) DiffusionVID-%s x%d, VID-val-shaped set" % ("R101" if args.arch == "r101" else "SwinB", args.sample_step),
            "value": round(total / float(tt.item()), 2), "unit": "frames/sec", "n_gpus": world, "steps": 1, "warmup": 0,
            "ms_per_step": round(float(tt.item()) * 1e3, 1), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "%d synthetic videos / %d frames shaped like ImageNet-VID val (lengths %d..%d), 1000x600, 300 boxes, "
                                   "frame-count-balanced whole videos per rank; frames drawn from a pool of 128 resident frames"
                                   % (len(lens), total, min(lens), max(lens)),
                       "lookahead_batches": args.lookahead, "ranks": world, "process_group": ("nccl, %d rank(s)" % dist.get_world_size()) if grouped else "none",
                       "frames_of_heaviest_rank_over_mean": round(max(sum(lens[v] for v in p) for p in balanced_video_partition(lens, world))
                                                                  / (total / world), 5)}})
    if model._engine is not None:
        model._engine.close()
        model._engine = None
    del model, ds, warm
    torch.cuda.empty_cache()
    if embedded:
        return record
    if rank == 0:
        print(json.dumps(record), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    return 0


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
    if args.workload == "vidval":
        # BASELINE.json configs[4] without a GPU: the 555-video / 176126-frame VID-val-shaped set sharded over the ranks by whole videos
        # (data/samplers.balanced_video_partition -- what `--workload vidval` runs), each rank's load gathered to rank 0, and the host-thread
        # cap every rank would run with under the container's CPU quota (--assume-cpu-quota N injects one: the GPU boxes grant 16 of 256 CPUs)
        from diffusionvid_amd.data.samplers import balanced_video_partition, vid_val_shaped_lengths
        lens = vid_val_shaped_lengths()
        mine = balanced_video_partition(lens, world)[rank]
        quota = args.assume_cpu_quota if args.assume_cpu_quota > 0 else comm.cpu_quota()
        share = comm.rank_cpu_share(rank, world, allowed=range(args.assume_cpus if args.assume_cpus > 0 else (os.cpu_count() or 1)))
        rec = {"rank": rank, "videos": len(mine), "frames": int(sum(lens[v] for v in mine)), "cpus": len(share),
               "threads": comm.rank_thread_cap(len(share), world, quota)}
        recs = [None] * world
        if world > 1:
            dist.all_gather_object(recs, rec)
        else:
            recs = [rec]
        if rank == 0:
            loads = [r["frames"] for r in recs]
            print(json.dumps({"metric": "dry run (vidval partition + host-thread caps)", "n_gpus": world, "videos": sum(r["videos"] for r in recs),
                              "frames": sum(loads), "heaviest_over_mean": round(max(loads) / (sum(loads) / world), 5), "cpu_quota_cpus": quota,
                              "threads_per_rank": [r["threads"] for r in recs], "cpus_per_rank": [r["cpus"] for r in recs],
                              "note": "config #5 on real JPEGs is decode-bound at a 16-CPU quota: ~310 decoded frames/s per granted CPU "
                                      "(profiles/r05m_feed_groups.txt) against ~2300 frames/s per rank"}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0
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
                         "(1 = the reference's schedule; default 0 = 304 frames' worth: 38 for R101, 76 for Swin-B)")
    ap.add_argument("--arch", choices=("r101", "swinb"), default="r101",
                    help="r101 = the BASELINE.json headline configuration; swinb = configs/vid_Swin_B_DiffusionVID.yaml (INFER_BATCH 4)")
    ap.add_argument("--sample-step", type=int, default=1, help="MODEL.DiffusionDet.SAMPLE_STEP (4 = the x4 configuration)")
    ap.add_argument("--skip-unobservable", action="store_true",
                    help="MODEL.DiffusionDet.SKIP_UNOBSERVABLE for the main measurement (x4 only; SURVEY.md Appendix B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="set a library option (csrc/options.h; A/B runs -- tools/ab_bench.sh): the line's build.library_config echoes the table, "
                         "and a run with any non-default option says so in build.default_configuration")
    ap.add_argument("--dtype", choices=("float16", "float32"), default="float16",
                    help="the reference's DTYPE key for the HEADLINE configuration (BASELINE.json configs[1] is fp16; float32 = csrc/f32.hip)")
    ap.add_argument("--smooth-frames", action="store_true", help="low-frequency synthetic frames (the parity tests' content) instead of white noise: the chip runs at its "
                    "1.4 kW power limit on this workload, so its clock -- and the frame rate -- depends on how much the data toggles (A/B only; the default stays white noise)")
    ap.add_argument("--host-noise", action="store_true", help="draw the DDIM noise with the host generator and upload it (rounds 1-3) instead of on the device")
    ap.add_argument("--no-vidval", action="store_true", help="skip the VID-val-shaped measurement reported inside the line (other_configs.vidval)")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-fed (pinned frames, H2D in the timed region) measurement")
    ap.add_argument("--no-feed-rate", action="store_true", help="skip the real-data feed-rate measurement (image files -> decode workers -> uint8 H2D -> device resize)")
    ap.add_argument("--feed-workers", type=int, default=0, help="real-data feed: decode pool size (0 = sweep 16 / 32 / 64 / 128 and report the knee)")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the reference-protocol (look-ahead 1), x4 and Swin-B measurements reported inside the line")
    ap.add_argument("--workload", choices=("video", "vidval"), default="video",
                    help="video (default): one synthetic video per GPU per step, weak scaling (the BASELINE headline).  vidval: "
                         "BASELINE.json configs[4] -- a 555-video / 176126-frame VID-val-shaped set (data/samplers.vid_val_shaped_lengths) "
                         "sharded over the ranks by balanced_video_partition, strong scaling; one step = the whole set (or --videos K of it)")
    ap.add_argument("--videos", type=int, default=0, help="vidval: use only the first K videos of the set (0 = all 555)")
    ap.add_argument("--assume-cpu-quota", type=float, default=0, help="--dry --workload vidval: the container CPU quota to plan host threads for (0 = read cgroup cpu.max)")
    ap.add_argument("--assume-cpus", type=int, default=0, help="--dry --workload vidval: the host's CPU count to plan for (0 = os.cpu_count())")
    ap.add_argument("--force-dist", action="store_true",
                    help="build the process group even for one rank (env:// rendezvous on 127.0.0.1) so that a single GPU runs the RCCL "
                         "gather / all-reduce path of the N-rank job (tests/test_gpu_dist.py)")
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
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cpus = comm.bind_rank_to_cpus(local_rank, local_world)          # this rank's share of the host CPUs, next to its GPU's NUMA node
    # N ranks, ONE tile-tuner result: rank 0 times each GEMM shape in its set-up pass and appends the winners to a file every
    # other rank reads at its first launch (they keep timing shapes rank 0 did not see).  Must be in the environment before this
    # process's first library launch.  A user-provided DVID_IGEMM_TUNE_CACHE is left alone.
    shared_tune = None
    if world > 1 and "DVID_IGEMM_TUNE_CACHE" not in os.environ:
        import tempfile
        # one file per (library build, launch): the build's md5 keeps a stale file of another build from being trusted, the launcher's
        # pid (the parent of every local rank) and the rendezvous port keep concurrent runs of one user apart
        shared_tune = os.path.join(tempfile.gettempdir(), "dvid_tune_%s_%s_%s_%s.txt" % ((LIB_MD5 or "nolib")[:12], os.environ.get("MASTER_PORT", "0"),
                                                                                          os.getuid(), os.getppid()))
        os.environ["DVID_IGEMM_TUNE_CACHE"] = shared_tune
        os.environ.setdefault("DVID_IGEMM_TUNE", "1")
        if rank == 0 and os.path.exists(shared_tune):
            os.unlink(shared_tune)
    grouped = world > 1 or args.force_dist          # a process group exists: collectives run (through RCCL, also for one rank)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            sk.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # collectives are bounded: if one rank leaves a side measurement through an exception its peers' pending gather raises after
        # 15 minutes (and is reported as that side measurement's error) instead of hanging the run
        comm.init_dist("nccl", force=True, timeout_s=900)
        assert dist.get_world_size() == args.gpus and dist.get_backend() == "nccl"

    for kv in args.option:
        name, _, val = kv.partition("=")
        ops.set_option(name, int(val))
    headline = args.arch == "r101" and args.sample_step == 1 and args.dtype == "float16"
    if args.lookahead <= 0:
        args.lookahead = 38 if args.arch == "r101" else 76          # one launch group per 304-frame video (~60 GB of workspace on a 288 GB part)
    H, W, L = 600, 1000, args.frames

    def barrier():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def build(arch, sample_step, lookahead, skip_unobservable=False, extra=(), dtype="float16"):
        yaml = "configs/vid_R_101_DiffusionVID.yaml" if arch == "r101" else "configs/vid_Swin_B_DiffusionVID.yaml"
        cfg = get_cfg(os.path.join(ROOT, yaml), ["DTYPE", dtype, "INPUT.LOOKAHEAD_BATCHES", lookahead,
                                                 "MODEL.DiffusionDet.SAMPLE_STEP", sample_step,
                                                 "MODEL.DiffusionDet.SKIP_UNOBSERVABLE", bool(skip_unobservable)] + list(extra),
                      os.path.join(ROOT, "configs/BASE_RCNN_1gpu.yaml"))
        cfg.freeze()
        model = build_detection_model(cfg).to(device).eval()
        # every random draw of the DDIM loop is generated ON THE DEVICE (dvid_counter_normal; the reference draws with
        # torch.randn on its device, diffusion_det.py:449,:542,:587,:595) -- --host-noise restores the host generator + upload
        model.noise_fn = synthetic.noise_fn if args.host_noise else synthetic.DeviceNoise()
        model.results_on_host = True      # one D2H copy per batch group (results end up on the host either way)
        return cfg, model

    def timed(model, ds, steps, warmup, gather=False, before_step=None):
        """W warm-up passes, then exactly K passes between barriers; -> (seconds max over ranks, frames of all ranks)"""
        with torch.no_grad():
            for _ in range(warmup):
                if before_step:
                    before_step()
                run_video(model, ds, device)
            barrier()
            t0 = time.perf_counter()
            frames = 0
            results = {}
            wait0 = model.host_wait_s
            for s in range(steps):
                if before_step:
                    before_step()
                r = run_video(model, ds, device)
                frames += len(r)
                results.update({k + s * L + rank * steps * L: v.to("cpu") for k, v in r.items()})
            tg = time.perf_counter()
            if gather and grouped:
                engine.gather_predictions(results, device=device, always=True)
            gather_s = time.perf_counter() - tg
            barrier()
            dt = time.perf_counter() - t0
        # host view of this rank: the share of the timed region it spent blocked on its GPU (device->host result copies).  A rank
        # near 0 is host-bound (its Python loop, uploads, ... pace the GPU); the minimum over ranks is the rank to look at.
        wait_frac = (model.host_wait_s - wait0) / max(dt, 1e-9)
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        ff = torch.tensor([frames], dtype=torch.float64, device=device)
        wmin = torch.tensor([wait_frac], dtype=torch.float64, device=device)
        wmax = wmin.clone()
        gs = torch.tensor([gather_s], dtype=torch.float64, device=device)
        if grouped:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(ff, op=dist.ReduceOp.SUM)
            dist.all_reduce(wmin, op=dist.ReduceOp.MIN)
            dist.all_reduce(wmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(gs, op=dist.ReduceOp.MAX)
        timed.last = {"host_blocked_on_gpu_frac_min_over_ranks": round(float(wmin.item()), 4),
                      "host_blocked_on_gpu_frac_max_over_ranks": round(float(wmax.item()), 4),
                      "gather_ms_max_over_ranks": round(float(gs.item()) * 1e3, 3),
                      "omp_num_threads": int(os.environ.get("OMP_NUM_THREADS", "0") or 0), "host_cpus": os.cpu_count()}
        return float(tt.item()), float(ff.item())

    def release(model):
        if model._engine is not None:
            model._engine.close()
            model._engine = None
        torch.cuda.empty_cache()

    if args.workload == "vidval":
        return vidval(args, build, timed, barrier, device, rank, world, H, W)
    cfg, model = build(args.arch, args.sample_step, args.lookahead, args.skip_unobservable, dtype=args.dtype)
    ds = SyntheticVIDDataset([L], cfg, height=H, width=W, device=device, video_base=rank, emit_ref_ahead=False, smooth=args.smooth_frames)
    ds.preload()
    if shared_tune and rank != 0:
        barrier()                      # rank 0's set-up pass first: its tuner winners are what this rank starts from
    with torch.no_grad():
        # set-up, outside the step accounting: weight repack / upload and the per-shape tile tuner (it times every
        # configuration on the first launch of each GEMM shape; DVID_IGEMM_TUNE_CACHE makes that persistent)
        run_video(model, ds, device)
    if shared_tune and rank == 0:
        barrier()
    dt, total_frames = timed(model, ds, args.steps, args.warmup, gather=True)
    host_view = dict(timed.last)

    # ---- the same workload fed from the HOST: pinned fp32 frames, double-buffered H2D inside the timed region (what the
    # reference's loop times, mega_core/engine/inference.py:29-40) -------------------------------------------------
    host_fed = None
    if not args.no_host_fed:
        from diffusionvid_amd.data.prefetch import HostFedVideo
        hds = SyntheticVIDDataset([L], cfg, height=H, width=W, device="cpu", video_base=rank, emit_ref_ahead=False)
        # cyclic: the pass after the last group is the same video again, as in a stream of videos -- its first group is
        # staged under the previous pass's last group; every pass still copies every frame
        hf = HostFedVideo(hds, device, cfg.INPUT.INFER_BATCH * args.lookahead, cyclic=True).pin().attach(model)
        hsteps = max(1, min(args.steps, 3))
        hdt, hframes = timed(model, hf, hsteps, 1)
        per_pass = hf.h2d_bytes / (1 + hsteps + 1.0 / max(1, -(-L // (cfg.INPUT.INFER_BATCH * args.lookahead))))
        host_fed = {"value": round(hframes / hdt, 2), "unit": "frames/sec",
                    "h2d_gbytes_per_video": round(per_pass / 1e9, 3), "h2d_gbs": round(per_pass * hframes / L / world / hdt / 1e9, 2),
                    "what": "frames start in pinned host memory as fp32 [0,1] CHW (the reference's DataLoader output); per look-ahead group "
                            "one batch of async copies on a side stream into one of two HBM staging buffers, overlapped with the previous "
                            "group's kernels (issued by the detector right after its own small uploads and first kernels are queued -- a pageable "
                            "upload issued behind a 1-GB prefetch would wait for it and serialise copy and compute); copies are inside the "
                            "timed region"}
        model.after_first_launch = None
        del hf, hds

    # ---- roofline of the dominant kernel: instrumented repeat of one step -------------------------
    lib = _lib.load()

    def measure_roofline(model, ds, arch, sample_step, fps, frames_per_step, lookahead):
        f32 = model.dtype == "float32"
        f32_peak = PEAK_FP32_SPLIT_TFLOPS if ops.get_option("f32_split") else PEAK_FP32_TFLOPS
        """per-launch HIP events on the library's stream; sub-batch chains are switched off for this pass so that launches do
        not overlap and each event pair times one kernel alone (the same condition the rocprofv3 summaries in profiles/ are
        taken under: DVID_CHAINS=1)"""
        engine_model = model._get_engine()
        engine_model.set_chains(1)
        graphs, model.use_call_graph = model.use_call_graph, False          # per-launch events need kernel-by-kernel launches
        with torch.no_grad():
            run_video(model, ds, device)      # un-instrumented: lets the per-shape tile tuner see the chains=1 launch shapes first
        torch.cuda.synchronize()
        lib.dvid_profile_enable(1)
        # two instrumented passes, the second one kept: the first launch of the first pass (the stem) has come out at 1x-2.5x its steady
        # duration from run to run (9.0 / 14.2 / 22.9 ms in the float32 tables of round 6 while rocprofv3 shows the kernel at its steady time)
        for _ in range(2):
            lib.dvid_profile_reset()
            with torch.no_grad():
                run_video(model, ds, device)
            torch.cuda.synchronize()
        engine_model.set_chains(int(os.environ.get("DVID_CHAINS", "2")))
        model.use_call_graph = graphs
        ms, fl, nl = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.dvid_profile_read(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(nl)), "dvid_profile_read")
        ab = ctypes.c_double()
        _lib.check(lib.dvid_profile_read_bytes(ctypes.byref(ab)), "dvid_profile_read_bytes")
        lib.dvid_profile_enable(0)
        # per-launch table -> the five heaviest (kernel, shape) groups of this configuration, each against ITS OWN bound.  DVID_PROFILE_DUMP=
        # <path> keeps the headline configuration's table (committed as profiles/r05_layers_<cfg>.csv); <path> with a "{cfg}" placeholder
        # keeps every configuration's.
        top = None
        if rank == 0:
            import tempfile
            cfg_tag = "%s_x%d%s%s" % (arch, sample_step, "" if lookahead > 1 else "_lookahead1",
                                      ("_float32" if ops.get_option("f32_split") else "_float32_fp32mfma") if f32 else "")
            keep = os.environ.get("DVID_PROFILE_DUMP")
            if keep and "{cfg}" in keep:
                path = keep.replace("{cfg}", cfg_tag)
            elif keep and arch == args.arch and sample_step == args.sample_step and lookahead == args.lookahead and model.dtype == args.dtype:
                path = keep
            else:
                fd, path = tempfile.mkstemp(suffix=".csv")
                os.close(fd)
                keep = None
            try:
                _lib.check(lib.dvid_profile_dump(path.encode()), "dvid_profile_dump")
                top = top_kernels(path, mfma_peak=f32_peak if f32 else PEAK_FP16_TFLOPS)
            finally:
                if not keep:
                    os.unlink(path)
        lib.dvid_profile_reset()
        traffic = mfma_busy = stamp = None
        key = ((arch, sample_step, "float32" if ops.get_option("f32_split") else "float32_fp32mfma") if f32          # (no PMC profile of the fp32-MFMA variant is committed)
               else (arch, sample_step) if lookahead > 1 else (arch, sample_step, "lookahead1"))
        traffic_file = TRAFFIC_FILES.get(key, "")
        tpath = os.path.join(ROOT, "profiles", traffic_file)
        if traffic_file and os.path.exists(tpath):      # HBM bytes per launch from the committed rocprofv3 --pmc passes of THIS configuration (cannot be collected in-process)
            pmc = json.load(open(tpath))
            stamp = pmc.get("library_md5")
            if stamp == LIB_MD5:          # a profile of ANOTHER build is refused, not reported with a warning
                traffic = round(pmc["hbm_bytes_per_launch"])
                mfma_busy = pmc.get("mfma_busy_fraction")
        if ms.value <= 0:
            return None
        sec = ms.value * 1e-3
        tflops = fl.value / sec / 1e12
        gbs = ab.value / sec / 1e9
        gflop_frame = ALG_GFLOP_PER_FRAME.get((arch, sample_step), 0)
        # the path is MFMA-bound (249.3 GFLOP against 60-90 MB of ideal-fusion HBM traffic per frame, ~3-4
        # kFLOP/B); `achieved` = algorithmic FLOP of the launches / their summed durations.  The layer-by-layer byte model
        # (every layer's input + weights + output + residual once) and the measured traffic are printed next to it: their
        # ratio to the ideal-fusion figure is the activation round-trip traffic that fusion has yet to remove.
        peak = f32_peak if f32 else PEAK_FP16_TFLOPS
        r = {"bound": "mfma", "kernel": (("implicit-GEMM conv/linear kernels of the DTYPE float32 path with split operands (f32x3_igemm_kernel, csrc/f32.hip; f32x3_conv3x3_kernel for the 3x3 / stride-1 layers; "
                                          "f32x3_wstat_kernel for the short-K / wide-N 1x1 layers): fp32 storage, every product as three "
                                          "fp16-MFMA passes over (hi, lo) fp16 pairs with fp32 accumulation; `achieved` counts ALGORITHMIC FLOP (one per fp32-grade product), `peak` = the dense "
                                          "fp16 MFMA peak / 3" if ops.get_option("f32_split") else
                                          "implicit-GEMM conv/linear kernel of the DTYPE float32 path, fp32 MFMA v_mfma_f32_32x32x2_f32 (f32_igemm_kernel, csrc/f32.hip)") if f32 else
                                         "implicit-GEMM conv/linear kernels, fp16 MFMA (igemm2_kernel; conv3x3_* for the 3x3 / stride-1 layers; wstat_kernel for the short-K / wide-N 1x1 layers; bneck64 / bneck128_tail_kernel = a res2 / res3 block behind its conv1 as one launch)"),
             "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
             "traffic": traffic,
             "traffic_source": ("profile: profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh on this "
                                "configuration's workload and THIS build of the library, md5 %s; bytes per launch; not collected in this run)"
                                % (traffic_file, stamp)) if traffic is not None
                               else ("refused: profiles/%s was collected on another build of the library (md5 %s, this run %s) -- re-run tools/profile_round.sh"
                                     % (traffic_file, stamp, LIB_MD5)) if stamp is not None
                               else "no PMC profile of this configuration is committed (tools/profile_round.sh <tag> --arch ... --sample-step ...)",
             "mfma_busy_pmc": None if mfma_busy is None else round(mfma_busy, 4),
             "end_to_end_tflops": round(fps / max(world, 1) * gflop_frame / 1e3, 1),
             "end_to_end_frac": round(fps / max(world, 1) * gflop_frame / 1e3 / peak, 4),
             "layerwise_alg_gbs": round(gbs, 1), "layerwise_alg_hbm_frac": round(gbs / PEAK_HBM_GBS, 4),
             "layerwise_alg_mbytes_per_launch": round(ab.value / max(1, nl.value) / 1e6, 2),
             "ideal_fusion_mbytes_per_frame": "60-90 (SURVEY.md 8d)",
             "layerwise_alg_mbytes_per_frame": round(ab.value / 1e6 / (frames_per_step + 24), 1),
             "top_kernels": top,
             "alg_gflop_per_launch": round(fl.value / max(1, nl.value) / 1e9, 3),
             "launches_per_step": int(nl.value), "avg_launch_us": round(ms.value * 1e3 / max(1, nl.value), 2),
             "kernel_ms_per_step": round(ms.value, 2)}
        return r

    roofline = measure_roofline(model, ds, args.arch, args.sample_step, total_frames / dt, L, args.lookahead)

    # ---- the reference's own call protocol (no look-ahead hand-over from the dataset) and the other single-GPU
    # configurations of BASELINE.json, each with its own model; reported inside the same line ---------------------
    sd_cpu = {k: v.detach().cpu() for k, v in model.state_dict().items()} if (rank == 0 and headline) else None
    frames8 = [ds.frame(0, i).tensors.cpu() for i in range(8, 24)] if (rank == 0 and headline) else None
    infer_batch = cfg.INPUT.INFER_BATCH
    release(model)
    del model
    others = {}

    def side(name, arch, sample_step, lookahead, steps, skip_unobservable=False, extra=(), note=None, with_roofline=False, dtype="float16", options=None):
        try:
            for k, v in (options or {}).items():          # library options of this side measurement only (restored below)
                ops.set_option(k, v)
            c2, m2 = build(arch, sample_step, lookahead, skip_unobservable, extra, dtype=dtype)
            d2 = SyntheticVIDDataset([L], c2, height=H, width=W, device=device, video_base=rank, emit_ref_ahead=False)
            d2._cache = ds._cache                  # same frames, already resident
            with torch.no_grad():
                run_video(m2, d2, device)
            t2, f2 = timed(m2, d2, steps, 1)
            others[name] = {"value": round(f2 / t2, 2), "unit": "frames/sec", "ms_per_step": round(t2 / steps * 1e3, 2),
                            "lookahead_batches": lookahead, "infer_batch": c2.INPUT.INFER_BATCH, "sample_step": sample_step, "steps": steps, "DTYPE": dtype,
                            "host_blocked_on_gpu_frac": timed.last["host_blocked_on_gpu_frac_min_over_ranks"]}
            if lookahead == 1:
                others[name]["call_graph_replays"] = m2.graph_replays          # calls served by one hipGraph launch each (0: kernel by kernel)
            if with_roofline:
                others[name]["roofline"] = measure_roofline(m2, d2, arch, sample_step, f2 / t2, L, lookahead)
            if note:
                others[name]["ms_per_frame"] = round(t2 / max(f2, 1) * world * 1e3, 3)
                others[name]["what"] = note
            if options:
                others[name]["library_options"] = dict(options)
            release(m2)
        except Exception as e:                     # a side measurement must never take the headline line down
            others[name] = {"error": repr(e)[:300]}
        finally:
            if options:
                ops.reset_options()
                for kv in args.option:
                    ops.set_option(kv.partition("=")[0], int(kv.partition("=")[2]))

    if not args.no_side_configs:
        side("reference_protocol_lookahead_1", args.arch, args.sample_step, 1, 5, with_roofline=True)
        if headline and world == 1:
            side("r101_x4", "r101", 4, 38, 5, with_roofline=True)
            # 12 observable head passes per frame instead of the faithful 19 (same detections)
            side("r101_x4_observable_passes_only", "r101", 4, 38, 5, skip_unobservable=True)
            side("swinb_x1", "swinb", 1, 76, 5, with_roofline=True)
            # `DTYPE float32` (the reference's default precision; round 6): fp32 storage + fp32 MFMA end to end (csrc/f32.hip) -- the mode in
            # which the path meets SURVEY.md 8(d)'s tolerances against the fp32 oracle (tests/test_gpu_e2e.py); 1/16 of the fp16 MFMA rate
            f32_note = ("DTYPE float32: every weight and activation fp32; conv / linear and DynamicConv's per-box products as three fp16-MFMA passes over split (hi, lo) operands "
                        "with fp32 accumulation (library option f32_split = 1; 0 = the fp32 MFMA, 157.3 TFLOP/s); one group per 304-frame video")
            side("r101_x1_float32", "r101", 1, 38, 2, with_roofline=True, dtype="float32", note=f32_note)
            side("r101_x4_float32", "r101", 4, 38, 2, dtype="float32", note=f32_note)
            side("swinb_x1_float32", "swinb", 1, 76, 2, dtype="float32", note=f32_note)
            side("r101_x1_float32_fp32_mfma", "r101", 1, 38, 2, with_roofline=True, dtype="float32", options={"f32_split": 0},
                 note="DTYPE float32 with library option f32_split = 0: every product exact on the fp32 MFMA (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s dense peak)")
            # the latency-oriented variant of demo/demo.py:60-68 -- one frame per call, one new global frame per
            # call merged into the memory and pruned back (vid_mega.py:213-215)
            side("r101_x1_streaming", "r101", 1, 1, 3,
                 extra=["INPUT.INFER_BATCH", 1, "MODEL.VID.MEGA.MAX_OFFSET", 0, "MODEL.VID.MEGA.MIN_OFFSET", 0,
                        "MODEL.VID.MEGA.ALL_FRAME_INTERVAL", 1, "MODEL.VID.MEGA.KEY_FRAME_LOCATION", 0,
                        "MODEL.VID.MEGA.GLOBAL.STOP_UPDATE_AFTER_INIT_TEST", False],
                 note="INFER_BATCH 1, ALL_FRAME_INTERVAL 1, MAX_OFFSET 0, GLOBAL.STOP_UPDATE_AFTER_INIT_TEST False: every call runs the backbone + "
                      "extraction heads on its own frame and one new global frame, merges 75 / 25 rows into the 900 / 150-row memories, prunes them "
                      "by farthest-point sampling and finishes the frame; ms_per_frame is the per-call latency")

    # BASELINE.json configs[4] inside the same line: the VID-val-shaped set sharded over the ranks by whole videos (strong scaling).
    # N > 1: the whole 555-video set; one rank: its first 60 videos (so that the default run stays within minutes).
    if headline and not args.no_side_configs and not args.no_vidval:
        try:
            import copy
            a2 = copy.copy(args)
            a2.videos = args.videos if args.videos > 0 else (0 if world > 1 else 60)
            rec = vidval(a2, build, timed, barrier, device, rank, world, H, W, embedded=True)
            if rank == 0:
                others["vidval"] = {"value": rec["value"], "unit": rec["unit"], "seconds": round(rec["ms_per_step"] / 1e3, 2), "scaling": "strong",
                                    "config": rec["config"]}
        except Exception as e:
            others["vidval"] = {"error": repr(e)[:300]}

    feed = None
    if rank == 0 and headline and not grouped and not args.no_feed_rate:          # (forks decode workers: single-process runs only)
        try:
            feed = real_data_feed_rate(device, feed_workers=(args.feed_workers,) if args.feed_workers > 0 else None)
        except Exception as e:
            feed = {"error": repr(e)[:300]}

    if rank == 0:
        line = {
            "metric": "frames/sec (1000x600) DiffusionVID-%s x%d" % ("R101" if args.arch == "r101" else "SwinB", args.sample_step),
            "value": round(total_frames / dt, 2), "unit": "frames/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.dtype == "float32" else "f16", "data": "synthetic",
            "config": {"workload": "%s DiffusionVID x%d %s, 300 boxes, %d DDIM step(s); one step = one synthetic "
                                   "%d-frame 1000x600 video per GPU (24 global + %d local frames, %d batches of %d)"
                                   % ("ResNet-101" if args.arch == "r101" else "Swin-Base", args.sample_step, "fp32" if args.dtype == "float32" else "fp16",
                                      args.sample_step, L, L, -(-L // infer_batch), infer_batch),
                       "frames_per_step_per_gpu": L, "infer_batch": infer_batch, "lookahead_batches": args.lookahead,
                       "lookahead_note": "the dataset emits the reference's unchanged item dict (vid_mega.py:236-248); the engine loop reads the "
                                         "group's later items ahead and hands their frames to the detector with the group's first call "
                                         "(engine.lookahead_items).  other_configs.reference_protocol_lookahead_1 is the same loop without "
                                         "reading ahead",
                       "parallelism": "videos sharded across ranks (one process per GPU, %s)"
                                      % ("RCCL group of %d rank(s): one gather of the predictions to rank 0" % dist.get_world_size()
                                         if grouped else "single rank, no collective"),
                       "ranks": world},
            "roofline": roofline,
            "build": {"library_md5": LIB_MD5,
                      # every DVID_* variable the library or the host side reads decides which kernel / schedule ran: the ones set in
                      # this process's environment are listed; anything not listed ran on its default
                      "env_switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("DVID_")},
                      # the library's option table as it stood during the timed region (csrc/options.h; dvid_effective_config) and whether
                      # it is the default one -- tests/test_host_logic.py pins the default string
                      "tuner_timing_passes": int(_lib.load().dvid_igemm_tuning_passes()),          # shape buckets timed in this process (set-up + side configurations)
                      "library_config": ops.effective_config(), "default_configuration": ops.effective_config().split(" DVID_")[0] == DEFAULT_LIBRARY_CONFIG,
                      "noise": "host generator + upload" if args.host_noise else "device (dvid_counter_normal)",
                      "shared_tuner_cache": shared_tune, "rank0_cpus": None if cpus is None else "%d CPUs: %d..%d" % (len(cpus), cpus[0], cpus[-1])},
            "host_view": host_view,
            "host_fed": host_fed,
            "real_data_feed": feed,
            "other_configs": others,
        }
        if world == 1 and headline and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd_cpu, frames8, H, W)
            line["gpu_over_cpu"] = round(line["value"] / max(line["cpu_baseline"]["value"], 1e-9), 1)
        print(json.dumps(line), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
