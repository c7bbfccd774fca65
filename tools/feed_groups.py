#!/usr/bin/env python
"""Can ONE host feed EIGHT ranks with real frames?   (round 5, review item 7; CPU only -- no GPU is touched)

    python tools/feed_groups.py [--groups 8] [--workers 32] [--seconds 8]

BASELINE configs[4] (ImageNet-VID val sharded over 8 GPUs) needs 8 x ~2300 frames/s of decoded frames on one host.  bench.py's
`real_data_feed` measures ONE decode pool (4067 frames/s at 64 workers, knee at 32).  Here `groups` feeder groups run AT THE SAME TIME, each
what one rank would own: a pool of `workers` PIL decode workers pinned to that rank's share of the host CPUs (utils/comm.rank_cpu_share:
the CPUs next to the rank's GPU when the topology is known, contiguous slices otherwise), handing 1280x720 frames over through its own
shared-memory ring (data/prefetch.py's hand-over).  Reported: frames/s per group and in total, against the per-rank model rate.
"""
import argparse
import multiprocessing as mp
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_SHM = None


def _attach(name):
    global _SHM
    from multiprocessing import shared_memory
    _SHM = shared_memory.SharedMemory(name=name)


def _decode_into(job):
    import numpy as np
    from PIL import Image
    path, slot = job
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"))
    ring = np.ndarray((arr.size,), dtype=np.uint8, buffer=_SHM.buf, offset=slot * arr.size)
    ring[:] = arr.reshape(-1)
    return slot


def group_main(g, groups, workers, paths, seconds, start, out):
    from multiprocessing import shared_memory
    from diffusionvid_amd.utils import comm
    cpus = comm.rank_cpu_share(g, groups)
    if hasattr(os, "sched_setaffinity") and cpus:
        os.sched_setaffinity(0, cpus)                     # inherited by the pool's workers
    shm = shared_memory.SharedMemory(create=True, size=720 * 1280 * 3 * len(paths))
    try:
        jobs = [(p, i) for i, p in enumerate(paths)]
        with mp.get_context("fork").Pool(workers, initializer=_attach, initargs=(shm.name,)) as pool:
            pool.map(_decode_into, jobs[:min(workers, len(jobs))])
            start.wait()
            t0, n = time.perf_counter(), 0
            while time.perf_counter() - t0 < seconds:
                for _ in pool.imap_unordered(_decode_into, jobs, chunksize=1):
                    n += 1
            out.put((g, n / (time.perf_counter() - t0), len(cpus)))
    finally:
        shm.close()
        shm.unlink()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=8)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--files", type=int, default=48)
    ap.add_argument("--need", type=float, default=2300.0, help="frames/s one rank's model consumes (R101 x1 headline)")
    args = ap.parse_args()
    import numpy as np
    from PIL import Image
    d = tempfile.mkdtemp(prefix="dvid_feed_groups_")
    rng = np.random.RandomState(0)
    paths = []
    for i in range(args.files):
        yy, xx = np.mgrid[0:720, 0:1280]
        img = np.stack([(128 + 100 * np.sin(xx / (37.0 + i) + c) * np.cos(yy / (53.0 + 2 * i))) for c in range(3)], -1)
        img = np.clip(img + rng.randn(720, 1280, 3) * 6, 0, 255).astype(np.uint8)
        paths.append(os.path.join(d, "%06d.JPEG" % i))
        Image.fromarray(img).save(paths[-1], format="JPEG", quality=90)
    try:
        ctx = mp.get_context("fork")
        start, out = ctx.Barrier(args.groups), ctx.Queue()
        procs = [ctx.Process(target=group_main, args=(g, args.groups, args.workers, paths, args.seconds, start, out)) for g in range(args.groups)]
        for p in procs:
            p.start()
        res = sorted(out.get(timeout=args.seconds * 10 + 120) for _ in procs)
        for p in procs:
            p.join()
        total = sum(r[1] for r in res)
        from diffusionvid_amd.utils import comm
        print(f"container CPU quota (cgroup cpu.max): {comm.cpu_quota()} CPUs -- every figure below is what THAT many CPUs decode, however many workers share them")
        print(f"host CPUs {os.cpu_count()}; {args.groups} feeder groups x {args.workers} decode workers at the same time, 1280x720 JPEG q90 -> uint8 frames in shared memory")
        for g, fps, ncpu in res:
            print(f"  group {g}: {fps:8.1f} frames/s on {ncpu} CPUs")
        print(f"  total  : {total:8.1f} frames/s = {total / args.groups:.1f} per group against {args.need:.0f} the model consumes per rank "
              f"({100 * total / (args.groups * args.need):.0f} % of what {args.groups} ranks need)")
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
