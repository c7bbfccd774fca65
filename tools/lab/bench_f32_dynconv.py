"""Stand-alone timing of the DTYPE float32 DynamicConv kernels (csrc/f32.hip) at 104 frames x 300 boxes:
python tools/lab/bench_f32_dynconv.py [boxes]  ->  ms and TB/s of the 226 KB per box (128 KB parameters + 2 x 49 KB tiles), f32_split 1 and 0."""
import sys

import torch

sys.path.insert(0, ".")
from diffusionvid_amd import ops  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 31200
g = torch.Generator().manual_seed(0)
roi = torch.randn(R, 49, 256, generator=g).cuda()
params = (torch.randn(R, 32768, generator=g) / 8.0).cuda()
g1, b1, g2, b2 = torch.ones(64).cuda(), torch.zeros(64).cuda(), torch.ones(256).cuda(), torch.zeros(256).cuda()
for split in (1, 0):
    ops.set_option("f32_split", split)
    ops.dynconv_f32(roi, params, g1, b1, g2, b2)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ops.dynconv_f32(roi, params, g1, b1, g2, b2)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print("f32_split %d: %d boxes  %.3f ms  %.2f TB/s" % (split, R, ms, R * (32768 * 4 + 2 * 49 * 256 * 4) / ms / 1e9), flush=True)
