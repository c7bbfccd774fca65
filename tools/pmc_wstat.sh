#!/bin/bash
# PMC view of the weight-stationary kernel on the bench_wstat shapes:  bash tools/pmc_wstat.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcw_$i -o p -- python $REPO/tools/bench_wstat.py --iters 3 > /tmp/pmcw_$i.log 2>&1
  echo "pass $i ($SET): rc=$?"
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("/tmp/pmcw_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "wstat_kernel" in r["Kernel_Name"] or "igemm2_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:80], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-40s %14.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
