#!/bin/bash
# Round profile on the GPU box: per-kernel time (rocprofv3 --kernel-trace --stats) and HBM traffic of the igemm kernel
# (separate --pmc passes, MI355X_MICROARCH.md HBM section) for the bench.py workload.  Writes gpurun_out/<tag>_*.
#   gpurun -- 'bash tools/profile_round.sh r03 [r101|swinb] [sample_step]'
# The traffic file is per configuration: <tag>_pmc_igemm_traffic_<arch>_x<sample_step>.json (bench.py reads
# profiles/r03_pmc_igemm_traffic_<arch>_x<ss>.json for the matching line only).
TAG=${1:-r01}
ARCH=${2:-r101}
SS=${3:-1}
LA=${4:-0}                      # 0: the bench default (one group per video); 1: the reference call protocol (one batch per call)
DT=${5:-float16}                # the reference's DTYPE key: float32 = csrc/f32.hip (its implicit-GEMM kernel is f32_igemm_kernel)
SUF=${ARCH}_x${SS}
if [ "$LA" != "0" ]; then SUF=${SUF}_lookahead${LA}; fi
if [ "$DT" != "float16" ]; then SUF=${SUF}_${DT}; fi
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DVID_CHAINS=1            # sequential launches: per-kernel durations are not inflated by overlap
CMD="python $REPO/bench.py --arch $ARCH --sample-step $SS --lookahead $LA --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-host-fed --no-side-configs --no-feed-rate"
# tile-tuner timing launches would pollute the statistics: fill the tuning cache in an unprofiled run first
export DVID_IGEMM_TUNE_CACHE=/tmp/dvid_tune_cache.txt
rm -f $DVID_IGEMM_TUNE_CACHE
$CMD > /tmp/prof_pre.log 2>&1
rm -rf /tmp/prof_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o st -- $CMD > /tmp/prof_stats.log 2>&1
grep '^{"metric"' /tmp/prof_stats.log | tail -1 > $OUT/${TAG}_bench_under_rocprof_${SUF}.json
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rm -rf /tmp/prof_$C
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_$C -o pmc -- $CMD --warmup 0 > /tmp/prof_$C.log 2>&1
done
python - "$TAG" "$OUT" "$SUF" <<'PY'
import csv, glob, hashlib, json, os, sys
tag, out, suf = sys.argv[1], sys.argv[2], sys.argv[3]
lib_md5 = hashlib.md5(open(os.path.join(os.path.dirname(out), "diffusionvid_amd", "libdvid_hip.so"), "rb").read()).hexdigest()
f = glob.glob("/tmp/prof_stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(f"{out}/{tag}_kernel_stats_{suf}.txt", "w") as o:
    o.write("# configuration " + suf + "\n")
    o.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-fed --no-side-configs  (DVID_CHAINS=1; 5 videos of 304 frames:\n")
    o.write("# set-up, warm-up, timed step, chains=1 pass and instrumented pass)\n")
    o.write("total kernel time %.1f ms\n" % (tot / 1e6))
    ig = [r for r in rows if "igemm2_kernel" in r["Name"] or "f32_igemm_kernel" in r["Name"] or "f32x3_igemm_kernel" in r["Name"] or "conv3x3_" in r["Name"] or "conv4x4_" in r["Name"] or "stem_pool" in r["Name"] or "wstat" in r["Name"] or "bneck" in r["Name"]]
    igt = sum(float(r["TotalDurationNs"]) for r in ig); igc = sum(int(r["Calls"]) for r in ig)
    o.write("implicit-GEMM kernels (igemm2_kernel / f32_igemm_kernel, conv3x3_halo_kernel, conv3x3_c64_kernel, conv4x4_s2d / stem_pool_kernel, wstat_kernel, wstat2_kernel, bneck64 / bneck128_tail_kernel; all instantiations): calls %d total %.2f ms avg %.2f us  %.1f%%\n" % (igc, igt / 1e6, igt / igc / 1e3, 100 * igt / tot))
    for r in rows[:40]:
        o.write("%-100s calls %7s total %9.2f ms avg %9.1f us %5.1f%%\n" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"/tmp/prof_{c}/**/*counter_collection.csv", recursive=True)
    s = n = 0.0
    for fn in fs:
        for r in csv.DictReader(open(fn)):
            if ("igemm2_kernel" in r["Kernel_Name"] or "f32_igemm_kernel" in r["Kernel_Name"] or "f32x3_igemm_kernel" in r["Kernel_Name"] or "conv3x3_" in r["Kernel_Name"] or "conv4x4_" in r["Kernel_Name"] or "stem_pool" in r["Kernel_Name"] or "wstat" in r["Kernel_Name"] or "bneck" in r["Kernel_Name"]) and r["Counter_Name"] == c:
                s += float(r["Counter_Value"]); n += 1
    res[c] = (s, n)
def per_kernel(c):
    s = n = 0.0
    for fn in glob.glob(f"/tmp/prof_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if ("igemm2_kernel" in r["Kernel_Name"] or "f32_igemm_kernel" in r["Kernel_Name"] or "f32x3_igemm_kernel" in r["Kernel_Name"] or "conv3x3_" in r["Kernel_Name"] or "conv4x4_" in r["Kernel_Name"] or "stem_pool" in r["Kernel_Name"] or "wstat" in r["Kernel_Name"] or "bneck" in r["Kernel_Name"]) and r["Counter_Name"] == c:
                s += float(r["Counter_Value"]); n += 1
    return s, n
mb, _ = per_kernel("SQ_VALU_MFMA_BUSY_CYCLES")
ga, _ = per_kernel("GRBM_GUI_ACTIVE")
mfma_busy = None
if mb and ga:
    # MFMA_BUSY_CYCLES sums over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md counter notes)
    mfma_busy = mb / 1024.0 / (ga / 8.0)
if res["FETCH_SIZE"][1] and res["WRITE_SIZE"][1]:
    fetch = res["FETCH_SIZE"][0] / res["FETCH_SIZE"][1] * 1024 * 2       # KiB units; gfx950 counts 128-B requests at 64 B
    write = res["WRITE_SIZE"][0] / res["WRITE_SIZE"][1] * 1024
    same = json.loads([l for l in open("/tmp/prof_FETCH_SIZE.log") if l.startswith('{"metric"')][-1])["roofline"]
    json.dump({"kernel": "implicit-GEMM kernels (igemm2_kernel + conv3x3_* + stem_pool_kernel + wstat*_kernel + bneck*_tail_kernel; all instantiations)", "launches": int(res["FETCH_SIZE"][1]),
               "configuration": suf, "library_md5": lib_md5,
               "alg_bytes_per_launch_same_run": same["layerwise_alg_mbytes_per_launch"] * 1e6,
               "mfma_busy_fraction": mfma_busy,
               "mfma_busy_method": "sum(SQ_VALU_MFMA_BUSY_CYCLES) / 1024 SIMDs over sum(GRBM_GUI_ACTIVE) / 8 XCDs, implicit-GEMM launches only",
               "fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
               "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 1 "
                         "--warmup 0 --no-cpu-baseline --no-host-fed --no-side-configs (the bench workload itself, 304-frame videos), DVID_CHAINS=1; KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                         "(gfx950 counts 128-B requests at 64 B); WRITE_SIZE uncorrected", "round": int(tag[1:3]) if tag[1:3].isdigit() else None},
              open(f"{out}/{tag}_pmc_igemm_traffic_{suf}.json", "w"), indent=1)
else:
    open(f"{out}/{tag}_pmc_error_{suf}.txt", "w").write(repr(res) + "\n" + open("/tmp/prof_FETCH_SIZE.log").read()[-3000:])
PY
