#!/bin/bash
# Kernel statistics of the reference call protocol (bench.py --lookahead 1: one 8-frame batch per call) -> gpurun_out/<tag>_kernel_stats_lookahead1.txt
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --lookahead 1 --steps 1 --warmup 1 --no-cpu-baseline --no-host-fed --no-side-configs --no-feed-rate"
export DVID_IGEMM_TUNE_CACHE=/tmp/dvid_tune_cache_l1.txt
rm -f $DVID_IGEMM_TUNE_CACHE
$CMD > /tmp/prof_pre.log 2>&1
rm -rf /tmp/prof_l1; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l1 -o st -- $CMD > /tmp/prof_l1.log 2>&1
grep '^{"metric"' /tmp/prof_l1.log | tail -1 > $OUT/${TAG}_bench_under_rocprof_lookahead1.json
python - "$TAG" "$OUT" <<'PY'
import csv, glob, sys
tag, out = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/prof_l1/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(f"{out}/{tag}_kernel_stats_lookahead1.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python bench.py --lookahead 1 --steps 1 --warmup 1 (one 8-frame batch per call; the process runs ~5 videos of 304 frames)\n")
    o.write("total kernel time %.1f ms over %d launches\n" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
    for r in rows[:45]:
        o.write("%-100s calls %7s total %9.2f ms avg %9.1f us %5.1f%%\n" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
