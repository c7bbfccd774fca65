"""Stand-alone timings of the DTYPE float32 implicit-GEMM kernel (csrc/f32.hip) on the layer shapes of R101 at 104 frames of 608 x 1024:
python tools/bench_f32_gemm.py [frames]  ->  one line per shape: ms, TFLOP/s, fraction of the 157.3 TFLOP/s fp32 MFMA peak."""
import sys

import torch

sys.path.insert(0, ".")
from diffusionvid_amd import ops  # noqa: E402

PEAK = 157.3


def run(n, h, w, cin, cout, k, stride, res, reps=5):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    wp, kpad, rs = ops.pack_conv_weight_f32(wt, scale_rows=True)
    ws = tuple(t.cuda() for t in ops.split_f16(wp))
    wp, rs = wp.cuda(), rs.cuda()
    bias = torch.zeros(cout).cuda()
    ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    r = torch.randn(n, ho, wo, cout, generator=g).cuda() if res else None
    f = lambda: ops.conv2d_nhwc_f32(x, wp, kpad, bias, cout, k, k, stride, k // 2, relu=1, residual=r, residual_mode=1 if res else 0, row_scale=rs, w_split=ws)
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    tf = 2.0 * n * ho * wo * cout * cin * k * k / ms / 1e9
    print("n %3d %3dx%-3d  %4d -> %4d  k%d s%d %s  %8.3f ms  %6.1f TFLOP/s  %.3f of peak" % (n, h, w, cin, cout, k, stride, "+res" if res else "    ", ms, tf, tf / PEAK), flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 104
    if len(sys.argv) > 2:          # library option f32_split: 1 = split fp16 operands (default), 0 = the fp32 MFMA
        ops.set_option("f32_split", int(sys.argv[2]))
        PEAK = 157.3
    if len(sys.argv) > 3:          # library option f32_wstat: 1 = weight-stationary kernel by its shape rule (default), 0 = tiled kernel everywhere
        ops.set_option("f32_wstat", int(sys.argv[3]))
    print("f32_split =", ops.get_option("f32_split"), "f32_wstat =", ops.get_option("f32_wstat"), "(fractions are of the 157.3 TFLOP/s fp32 MFMA peak either way)")
    for shape in [(152, 256, 64, 64, 1, 1, False), (152, 256, 64, 64, 3, 1, False), (152, 256, 64, 256, 1, 1, True), (152, 256, 256, 64, 1, 1, False),
                  (76, 128, 128, 128, 3, 1, False), (76, 128, 128, 512, 1, 1, True), (76, 128, 512, 128, 1, 1, False),
                  (38, 64, 1024, 256, 1, 1, False), (38, 64, 256, 256, 3, 1, False), (38, 64, 256, 1024, 1, 1, True),
                  (19, 32, 2048, 512, 1, 1, False), (19, 32, 512, 512, 3, 1, False), (19, 32, 512, 2048, 1, 1, True)]:
        run(n, *shape)
    # the decoder's linear layers at 104 frames x 300 boxes
    for rows, cin, cout in [(31200, 256, 32768), (31200, 12544, 256), (31200, 256, 2048), (31200, 2048, 256), (31200, 256, 768)]:
        run(1, 1, rows, cin, cout, 1, 1, False)
