cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/one_gemm.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from diffusionvid_amd import ops
g = torch.Generator().manual_seed(0)
x = torch.randn(31200, 12544, generator=g).cuda()
wp, kpad = ops.pack_conv_weight_f32(torch.randn(256, 12544, generator=g) / 112.0)
wp = wp.cuda(); b = torch.zeros(256).cuda()
for _ in range(3): ops.linear_f32(x, wp, kpad, b)
torch.cuda.synchronize()
PY
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcf_$i -o p -- python /tmp/one_gemm.py > /tmp/pmcf_$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob("/tmp/pmcf_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "f32_igemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("%-40s %16.5g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
