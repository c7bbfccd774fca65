#!/bin/bash
# Is the res4 GEMM loop paced by the POWER budget?   (round 5)   gpurun -- 'bash tools/lab/power_probe.sh'
# Runs one variant of tools/lab/gemm_loader_waves on conv1's shape (739328 x 256 x 1024) back to back for a few seconds while sampling
# the shader clock and the socket power (rocm-smi) every 0.5 s: the loop with its DMA, the same loop without any DMA (matrix pipe + LDS
# only), and the DMA-only copy of the loop (tools/lab/strided_stream: fill path + HBM only).
cd "$(dirname "$0")/../.."
sample() {            # $1 = label; samples until the background job $2 ends
  while kill -0 $2 2>/dev/null; do
    rocm-smi --showclocks --showpower --csv 2>/dev/null | awk -F, -v l="$1" 'NR==1{for(i=1;i<=NF;i++){if($i ~ /sclk clock speed/)s=i; if($i ~ /[Pp]ower/)p=i}} NR==2{print l": sclk "$s"  power "$p}'
    sleep 0.5
  done
}
rocm-smi --showclocks --showpower --csv 2>/dev/null | head -3
for v in "0 4" "4 4" "6 4" "1 4" "0 5" "4 5" "6 5"; do
  set -- $v
  LW_SUSTAIN="$1 $2 7" tools/lab/gemm_loader_waves &
  pid=$!
  sleep 2.5
  sample "variant $1 shape $2" $pid | sort | uniq -c | sort -rn | head -4
  wait $pid
done
echo "DMA-only copy of the loop (strided_stream, all modes back to back):"
tools/lab/strided_stream > /tmp/ss.txt &
pid=$!
sleep 0.3
sample "strided_stream" $pid | sort | uniq -c | sort -rn | head -4
wait $pid
grep "weights 1" /tmp/ss.txt | head -2
