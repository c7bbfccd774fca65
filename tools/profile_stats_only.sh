#!/bin/bash
# Kernel statistics only (no PMC passes): rocprofv3 --kernel-trace --stats of the bench workload -> gpurun_out/<tag>_kernel_stats.txt
#   gpurun -- 'bash tools/profile_stats_only.sh r02e'
TAG=${1:-r02e}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DVID_CHAINS=1
CMD="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-fed --no-side-configs"
export DVID_IGEMM_TUNE_CACHE=/tmp/dvid_tune_cache.txt
rm -f $DVID_IGEMM_TUNE_CACHE
$CMD > /tmp/prof_pre.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o st -- $CMD > /tmp/prof_stats.log 2>&1
grep '^{"metric"' /tmp/prof_stats.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
python - "$TAG" "$OUT" <<'PY'
import csv, glob, sys
tag, out = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/prof_stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(f"{out}/{tag}_kernel_stats.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-fed --no-side-configs  (DVID_CHAINS=1; 5 videos of 304 frames:\n")
    o.write("# set-up, warm-up, timed step, chains=1 pass and instrumented pass)\n")
    o.write("total kernel time %.1f ms\n" % (tot / 1e6))
    ig = [r for r in rows if "igemm2_kernel" in r["Name"] or "conv3x3_" in r["Name"] or "wstat" in r["Name"] or "bneck64" in r["Name"]]
    igt = sum(float(r["TotalDurationNs"]) for r in ig); igc = sum(int(r["Calls"]) for r in ig)
    o.write("implicit-GEMM kernels (igemm2_kernel, conv3x3_halo_kernel, conv3x3_c64_kernel, wstat_kernel, wstat2_kernel, bneck64_tail_kernel; all instantiations): calls %d total %.2f ms avg %.2f us  %.1f%%\n" % (igc, igt / 1e6, igt / igc / 1e3, 100 * igt / tot))
    for r in rows[:40]:
        o.write("%-100s calls %7s total %9.2f ms avg %9.1f us %5.1f%%\n" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
