# Swin-B kernel statistics of a bench video: gpurun -- 'bash tools/prof_swin.sh' -> gpurun_out/swinb_kernel_stats.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
export DVID_CHAINS=1
export DVID_IGEMM_TUNE_CACHE=/tmp/dvid_tune_cache_swin.txt
CMD="python $REPO/bench.py --steps 1 --warmup 1 --arch swinb --no-cpu-baseline --no-host-fed --no-side-configs"
$CMD > /tmp/pre.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sw -o sw -- $CMD > /tmp/prof_sw.log 2>&1
f=$(find /tmp/prof_sw -name "*kernel_stats.csv" | head -1)
python - <<PY > $REPO/gpurun_out/swinb_kernel_stats.txt
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --arch swinb --no-cpu-baseline --no-host-fed --no-side-configs (DVID_CHAINS=1; 5 videos of 304 frames)")
print("total kernel time %.1f ms" % (tot / 1e6))
ig = [r for r in rows if "igemm2_kernel" in r["Name"] or "wstat" in r["Name"] or "bneck64" in r["Name"] or "conv3x3_" in r["Name"]]
print("implicit-GEMM kernels (all instantiations): calls %d total %.2f ms  %.1f%%" % (sum(int(r["Calls"]) for r in ig), sum(float(r["TotalDurationNs"]) for r in ig) / 1e6, 100 * sum(float(r["TotalDurationNs"]) for r in ig) / tot))
for r in rows[:30]:
    print("%-100s calls %7s total %9.2f ms avg %9.1f us %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
grep '^{"metric"' /tmp/prof_sw.log | tail -1 > $REPO/gpurun_out/swinb_bench_under_rocprof.json
