#!/bin/bash
# The reference call protocol (look-ahead 1: one 8-frame batch per working call): where a 304-frame video's wall time goes.
#   gpurun -- 'bash tools/profile_l1.sh r03b'
# Writes gpurun_out/<tag>_l1_kernel_stats.txt (rocprofv3 --kernel-trace --stats of bench.py --lookahead 1) and
# gpurun_out/<tag>_l1_host.txt (tools/profile_host.py 1: wall per call, host-side hot spots).
TAG=${1:-r03b}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DVID_IGEMM_TUNE_CACHE=/tmp/dvid_tune_cache_l1.txt
rm -f $DVID_IGEMM_TUNE_CACHE
CMD="python $REPO/bench.py --lookahead 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed --no-side-configs"
$CMD > /tmp/l1_pre.log 2>&1
grep '^{"metric"' /tmp/l1_pre.log | tail -1 > $OUT/${TAG}_l1_bench.json
python $REPO/tools/profile_host.py 1 > $OUT/${TAG}_l1_host.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l1 -o st -- $CMD > /tmp/prof_l1.log 2>&1
python - "$TAG" "$OUT" <<'PY'
import csv, glob, sys
tag, out = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/prof_l1/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(f"{out}/{tag}_l1_kernel_stats.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python bench.py --lookahead 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed --no-side-configs\n")
    o.write("# (6 videos of 304 frames: set-up, warm-up, 2 timed steps, chains=1 pass and instrumented pass; sub-batch chains on except in the last two)\n")
    o.write("total kernel time %.1f ms over 6 videos = %.1f ms per video\n" % (tot / 1e6, tot / 6e6))
    for r in rows[:45]:
        o.write("%-100s calls %7s total %9.2f ms avg %9.1f us %5.1f%%\n" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
