import sys, torch
sys.path.insert(0, "/root/repo")
from diffusionvid_amd import _lib, ops
lib = _lib.load()
ncfg = lib.dvid_igemm_num_configs()
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for name, n, cin, cout, res in (("res4 conv1 304f", 304, 1024, 256, False), ("res4 conv1 104f", 104, 1024, 256, False), ("res4 conv1 24f", 24, 1024, 256, False),
                                ("res5 conv1 304f (19x32)", 76, 2048, 512, False), ("fpn lateral5-like K2048 N256", 76, 2048, 256, False), ("K512 N512 +res", 104, 512, 512, True)):
    x = torch.randn(n, 38, 64, cin, device="cuda").half()
    wp, kpad = ops.pack_conv_weight(torch.randn(cout, cin, 1, 1) * 0.05); wp = wp.cuda()
    b = torch.randn(cout, device="cuda")
    r = torch.randn(n, 38, 64, cout, device="cuda").half() if res else None
    f = lambda: ops.conv2d_nhwc(x, wp, kpad, b, cout, 1, 1, 1, 0, relu=True, residual=r, residual_mode=1 if res else 0)
    out = {}
    for cfg, tag in ((20, "256x256x32/5"), (ncfg - 1, "persistent")):
        lib.dvid_igemm_set_config(cfg)
        out[tag] = (timeit(f), f().clone())
    lib.dvid_igemm_set_config(-1)
    same = torch.equal(out["256x256x32/5"][1], out["persistent"][1])
    print(f"{name:32s} anti-phase {out['256x256x32/5'][0]*1e3:8.1f} us   persistent {out['persistent'][0]*1e3:8.1f} us   identical {same}")
