#!/bin/bash
# PMC view of one GEMM shape under one forced igemm configuration:  bash tools/pmc_probe.sh <cfg index> M N K
CFG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export DVID_IGEMM_CFG=$CFG
i=0
for SET in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_avr TCC_EA0_WRREQ_STALL_sum" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcp_$i -o p -- python $REPO/tools/bench_gemm_probe.py "$@" > /tmp/pmcp_$i.log 2>&1
  echo "pass $i ($SET): rc=$?"
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("/tmp/pmcp_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "igemm2_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:70], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-40s %14.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
