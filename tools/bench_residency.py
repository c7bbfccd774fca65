#!/usr/bin/env python
"""Does a smaller activation working set pay?  (1) far-memory probe: read and write->read rates of torch streaming kernels
against buffer size (the 256 MB memory-side cache should show as a step); (2) the R101-FPN backbone on 104 frames of
1000x600 run as launch sequences of F frames each (Model.backbone(frames_per_launch=F)): whole-pass time, and igemm time
per stage from the per-launch event records (stage = the launch's output channel count / row count class)."""
import collections, csv, ctypes, os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionvid_amd import ops, _lib
from diffusionvid_amd.utils import synthetic


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def probe():
    print("## far-memory probe (torch streaming kernels, fp16)")
    for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
        n = mb * (1 << 20) // 2
        x = torch.ones(n, dtype=torch.float16, device="cuda")
        y = torch.empty_like(x)
        reps = max(3, min(50, 8192 // mb))
        t_rd = timed(lambda: torch.max(x), reps)               # read-only, same buffer every repeat
        t_cp = timed(lambda: y.copy_(x), reps)                # read x, write y
        t_pc = timed(lambda: (y.copy_(x), x.copy_(y)), reps)  # producer -> consumer ping-pong: each copy reads what the last wrote
        print("buf %5d MB: read %6.2f TB/s   copy %6.2f TB/s (r+w)   ping-pong %6.2f TB/s (r+w)" % (
            mb, mb / 1024 / 1024 / (t_rd * 1e-3) * 1.048576, 2 * mb / 1024 / 1024 / (t_cp * 1e-3) * 1.048576,
            4 * mb / 1024 / 1024 / (t_pc * 1e-3) * 1.048576))
        del x, y


def backbone():
    print("## R101-FPN, 104 frames 608x1024 (padded 1000x600), launch sequences of F frames")
    lib = _lib.load()
    sd = synthetic.make_state_dict(0)
    m = ops.Model(sd)
    n = 104
    m.reserve(n, 608, 1024, 300)
    m.set_chains(1)
    x = torch.rand(n, 3, 608, 1024, device="cuda")
    for f in (104, 52, 26, 13, 8, 4, 2):
        for _ in range(2): m.backbone(x, frames_per_launch=f)       # tunes the new row counts
        ms = timed(lambda: m.backbone(x, frames_per_launch=f), 3)
        lib.dvid_profile_reset(); lib.dvid_profile_enable(1)
        m.backbone(x, frames_per_launch=f)
        torch.cuda.synchronize()
        path = os.path.join(tempfile.gettempdir(), "residency_%d.csv" % f)
        lib.dvid_profile_dump(path.encode())
        lib.dvid_profile_enable(0); lib.dvid_profile_reset()
        stage = collections.OrderedDict()
        for r in csv.DictReader(open(path)):
            rows_per_frame = int(r["M"]) / f if int(r["M"]) % f == 0 else int(r["M"]) / (n % f or f)
            # output rows per frame: 77824 (stem) 19456 (res2) 4864 (res3) 1216 (res4) 304 (res5); FPN rows reuse those
            key = "rows/frame %6d" % round(rows_per_frame)
            a = stage.setdefault(key, [0.0, 0.0])
            a[0] += float(r["ms"]); a[1] += float(r["ms"]) * float(r["tflops"])
        line = "  ".join("%s: %6.2f ms" % (k.split()[-1], v[0]) for k, v in stage.items())
        print("F %3d: %7.2f ms per 104 frames (%5.0f frames/s) | igemm by output rows per frame  %s" % (f, ms, n / ms * 1e3, line))


if __name__ == "__main__":
    probe()
    backbone()
