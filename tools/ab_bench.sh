#!/bin/bash
# A/B of one library option (csrc/options.h) on ONE box:  tools/ab_bench.sh stem_pool "0 1 0 1" [extra bench.py args]
# prints value / ms_per_step / family frac and the five heaviest kernel groups per run
var=$1; vals=$2; shift 2
for v in $vals; do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-fed --no-side-configs --option $var=$v "$@" 2>/dev/null > /tmp/ab_line.json
  python - "$var" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
top = d["roofline"].get("top_kernels") or {"kernels": []}
print(sys.argv[1], sys.argv[2], "frames/s", d["value"], "ms/step", d["ms_per_step"], "family frac", d["roofline"]["frac"],
      "| " + "; ".join("%s %.2f ms %s %.3f" % (k["name"][:44], k["ms"], k["bound"], k["frac"]) for k in top["kernels"]))
PY
done
