#!/bin/bash
# Is the fp32 implicit GEMM (csrc/f32.hip) paced by the power budget / shader clock?   gpurun -- 'bash tools/lab/f32_power_probe.sh'
# Runs the K = 12544 linear layer (pure main loop) back to back for ~6 s while sampling the shader clock and socket power (rocm-smi).
cd "$(dirname "$0")/../.."
python - <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
from diffusionvid_amd import ops
g = torch.Generator().manual_seed(0)
x = torch.randn(31200, 12544, generator=g).cuda()
wp, kpad = ops.pack_conv_weight_f32(torch.randn(256, 12544, generator=g) / 112.0)
wp = wp.cuda(); b = torch.zeros(256).cuda()
ops.linear_f32(x, wp, kpad, b); torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < 6.0:
    for _ in range(50): ops.linear_f32(x, wp, kpad, b)
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
print("fp32 igemm 31200 x 256 x 12544: %.3f ms per launch, %.1f TFLOP/s sustained over %.1f s" % (dt / n * 1e3, 2.0 * 31200 * 256 * 12544 / (dt / n) / 1e12, dt))
PY
pid=$!
sleep 3
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower --csv 2>/dev/null | awk -F, 'NR==1{for(i=1;i<=NF;i++){if($i ~ /sclk clock speed/)s=i; if($i ~ /[Pp]ower/)p=i}} NR==2{print "sclk "$s"  power "$p}'
  sleep 0.5
done | sort | uniq -c | sort -rn | head -5
wait $pid
