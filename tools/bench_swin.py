import sys, time, torch
sys.path.insert(0, "/root/repo")
from diffusionvid_amd import ops
from diffusionvid_amd.utils import synthetic
sd = synthetic.make_state_dict(0, swin=dict(embed_dim=128, depths=(2,2,18,2), heads=(4,8,16,32), window=7))
m = ops.Model(sd, res_blocks=(0,0,0,0), backbone="swin")
n = 4
m.reserve(n, 608, 1024, 300)
x = torch.rand(n, 3, 608, 1024, device="cuda")
for _ in range(2): m.backbone(x)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): p = m.backbone(x)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print("Swin-B+FPN backbone: %.2f ms per %d frames -> %.1f frames/s; %.1f TFLOP/s (211.6 GMAC/frame)" % (ms, n, n / ms * 1e3, 211.6e9*2*n/ms/1e9))
print("finite:", all(torch.isfinite(t.float()).all().item() for t in p), "rms", [float(t.float().pow(2).mean().sqrt()) for t in p])
