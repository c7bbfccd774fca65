#!/bin/bash
# PMC counters of the split mode's kernels, one representative layer each at 104 frames (separate --pmc passes, no other trace domain):
#   res4 conv2 (f32x3_conv3x3_kernel), res4 conv3 + residual (f32x3_wstat_kernel), res4 conv1 (f32x3_igemm_kernel), DynamicConv (f32x3_dynconv_kernel)
#   gpurun -- 'bash tools/lab/pmc_f32_kernels.sh'  ->  gpurun_out/pmc_f32_kernels.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/f32_layers.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from diffusionvid_amd import ops
g = torch.Generator().manual_seed(0)
def conv(n, h, w, cin, cout, k, res):
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    wp, kpad, rs = ops.pack_conv_weight_f32(wt, scale_rows=True)
    ws = tuple(t.cuda() for t in ops.split_f16(wp))
    r = torch.randn(n, h, w, cout, generator=g).cuda() if res else None
    for _ in range(3):
        ops.conv2d_nhwc_f32(x, wp.cuda(), kpad, torch.zeros(cout).cuda(), cout, k, k, 1, k // 2, relu=1, residual=r, residual_mode=1 if res else 0, row_scale=rs.cuda(), w_split=ws)
conv(104, 38, 64, 256, 256, 3, False)
conv(104, 38, 64, 256, 1024, 1, True)
conv(104, 38, 64, 1024, 256, 1, False)
R = 31200
roi = torch.randn(R, 49, 256, generator=g).cuda()
params = (torch.randn(R, 32768, generator=g) / 8.0).cuda()
for _ in range(3):
    ops.dynconv_f32(roi, params, torch.ones(64).cuda(), torch.zeros(64).cuda(), torch.ones(256).cuda(), torch.zeros(256).cuda())
torch.cuda.synchronize()
PY
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmck_$i -o p -- python /tmp/f32_layers.py > /tmp/pmck_$i.log 2>&1
  echo "pass $i rc=$?"
done
python - > $OUT/pmc_f32_kernels.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("/tmp/pmck_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"]
        for key in ("f32x3_conv3x3_kernel", "f32x3_wstat_kernel", "f32x3_igemm_kernel", "f32x3_dynconv_kernel"):
            if key in n:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print("== %s (mean over %d launches)" % (k, len(next(iter(cs.values())))))
    for c, v in sorted(m.items()):
        print("   %-28s %14.5g" % (c, v))
    g = m.get("GRBM_GUI_ACTIVE")
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        print("   -> MFMA pipe busy %.3f of the busy cycles (sum / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs)" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (g / 8.0)))
    if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]:
        print("   -> LDS bank-conflict cycles %.3f of the LDS-array cycles" % (m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]))
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
        print("   -> wave cycles: parked at s_waitcnt / barriers %.3f, issue stalls %.3f, issuing %.3f" % (m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"]))
PY
cat $OUT/pmc_f32_kernels.txt
