import torch
for (M, N, K) in [(252928, 256, 1024), (252928, 512, 1024), (63232, 512, 2048), (31200, 32768, 256), (252928, 1024, 256)]:
    x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
    o = torch.empty(M, N, dtype=torch.float16, device="cuda")
    for _ in range(3): torch.matmul(x, w.t(), out=o)
    torch.cuda.synchronize()
