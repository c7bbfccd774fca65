#!/bin/bash
# Where does the GPU idle during one bench video?  rocprofv3 kernel trace -> union of busy intervals, gaps by neighbour kernels.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export DVID_IGEMM_TUNE_CACHE=/tmp/dvid_tune_cache_gap.txt
python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > /tmp/gap_pre.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_prof -o g -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > /tmp/gap.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/gap_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the timed step = second quarter of the run (warm-up, timed, tuner warm-up, instrumented): take the window between 25% and 50% of kernels
n = len(rows)
seg = rows[n // 4: n // 2]
def short(k):
    k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return k[:46]
busy = 0; idle = 0; cur_end = seg[0][0]; gaps = collections.Counter(); gapn = collections.Counter(); last = None
hist = collections.Counter()
for s, e, k in seg:
    if s > cur_end:
        g = s - cur_end
        idle += g
        key = (short(last) if last else "-", short(k))
        gaps[key] += g; gapn[key] += 1
        hist["<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"] += g
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e; last = k
span = seg[-1][1] - seg[0][0]
print("window %.1f ms: busy %.1f ms, idle %.1f ms (%.1f%%), %d kernels" % (span / 1e6, busy / 1e6, idle / 1e6, 100.0 * idle / span, len(seg)))
print("idle time by gap length:", {k: round(v / 1e6, 2) for k, v in hist.items()})
for key, g in gaps.most_common(25):
    print("%8.2f ms in %5d gaps (avg %7.1f us)  after %-46s before %s" % (g / 1e6, gapn[key], g / gapn[key] / 1e3, key[0], key[1]))
PY
