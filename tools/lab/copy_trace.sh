#!/bin/bash
# Where do the __amd_rocclr_copyBuffer launches of a bench step come from?  Kernel trace of the headline command; keeps the copy rows
# plus the stem launches (markers of a sub-batch start) with start time, duration and grid size.
#   gpurun -- 'bash tools/lab/copy_trace.sh [extra bench args]'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DVID_IGEMM_TUNE_CACHE=/tmp/dvid_tune_cache.txt
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed --no-side-configs --no-feed-rate $*"
$CMD > /tmp/pre.log 2>&1
rm -rf /tmp/ct; rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o ct -- $CMD > /tmp/ct.log 2>&1
grep '^{"metric"' /tmp/ct.log | tail -1 > $OUT/copy_trace_bench.json
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob("/tmp/ct/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(out + "/copy_trace.txt", "w") as o:
    o.write("# t_ms dur_us grid wg name   (copies, fills and stem launches; %d kernel rows in all)\n" % len(rows))
    prev_end = None
    for i, r in enumerate(rows):
        n = r["Kernel_Name"]
        if "copyBuffer" in n or "fillBuffer" in n or "stem_pool" in n or "Memcpy" in n or "elementwise" in n or "CatArray" in n:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            nxt = rows[i + 1]["Kernel_Name"][:60] if i + 1 < len(rows) else ""
            prv = rows[i - 1]["Kernel_Name"][:60] if i else ""
            o.write("%10.3f %8.1f %9s %5s %-50s | after %-40s before %s\n" % ((s - t0) / 1e6, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), n[:50], prv[:40], nxt[:40]))
PY
ls -la $OUT/copy_trace.txt
