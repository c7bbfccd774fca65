#!/usr/bin/env python
"""Static check of the counted-wait contract of the LDS-DMA kernels (csrc/bneck.hip, csrc/wstat.hip; DESIGN.md section 5,
"an ordinary load and an LDS-DMA piece do not retire in issue order relative to each other").

    python tools/check_dma_waits.py [file.hip ...]          # default: bneck.hip wstat.hip

Compiles each source to gfx950 assembly (hipcc -S --cuda-device-only, seconds per file) and walks every kernel's instruction
stream in program order with a queue of the vector-memory operations in flight:
  * `global_load_lds_*` / `buffer_load_* ... lds`      -> a DMA piece,
  * any other `global_load_*` / `buffer_load_*` / `flat_load_*` / `scratch_load_*` -> an ordinary load,
  * stores are ignored (the kernels never count them: a wait that counts too few operations only waits longer).
A counted wait `s_waitcnt vmcnt(N)`, N > 0, relies on operations retiring in issue order: "all but the youngest N are done".  Data
returns in issue order, but a piece decrements the counter only after its LDS write, so a YOUNGER ordinary load can retire ahead of
it; pieces stay in order among themselves.  With the queue split into the youngest N operations Y and the rest R (which the wait
is there to cover), the counter can reach N with a piece of R still in flight only if an operation of Y has retired in its place,
and only an ordinary load can do that.  Rule checked at every counted wait: NOT (R holds a piece AND Y holds an ordinary load).
(Ordinary loads older than the pieces, or pieces older than loads that the wait also covers, are harmless.)  After the wait Y stays queued;
`vmcnt(0)` empties the queue.  Branches are followed linearly (the loops of these kernels are straight-line bodies; a loop-carried
mix shows up in the body's second trip: at every backward branch the walk goes through the loop body once more with the queue it has).
Exit status 1 and one line per violation (kernel, asm line, the mixed queue) when the rule is broken.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffusionvid_amd", "csrc")

_LOAD = re.compile(r"^\s*(global_load|buffer_load|flat_load|scratch_load)_")
_WAIT = re.compile(r"^\s*s_waitcnt\b(.*)")
_KERNEL = re.compile(r"^(_Z\w+):")


def assemble(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), src, "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(out) as f:
        text = f.read().splitlines()
    os.unlink(out)
    return text


def kind_of(line):
    if not _LOAD.match(line):
        return None
    code = line.split(";")[0]
    return "piece" if ("_lds_" in code or re.search(r"\blds\b", code)) else "load"


_LABEL = re.compile(r"^(\.LBB\w+):")
_BRANCH = re.compile(r"^\s*s_c?branch\w*\s+(\.LBB\w+)")


def check_kernel(name, lines):
    """lines: [(asm line number, text)] of one kernel.  -> list of violations"""
    bad, seen = [], set()
    label_at = {m.group(1): i for i, (_, t) in enumerate(lines) for m in [_LABEL.match(t)] if m}
    replayed = set()

    def walk(lo, hi, queue):
        i = lo
        while i < hi:
            no, text = lines[i]
            i += 1
            k = kind_of(text)
            if k:
                queue.append((k, no))
                continue
            b = _BRANCH.match(text)
            if b and label_at.get(b.group(1), len(lines)) < i - 1 and (i - 1) not in replayed:
                replayed.add(i - 1)                     # a backward branch: one more trip through the loop body with what is in flight
                queue = walk(label_at[b.group(1)], i - 1, queue)
                continue
            m = _WAIT.match(text)
            v = re.search(r"vmcnt\((\d+)\)", m.group(1)) if m else None
            if not v:
                continue
            n = int(v.group(1))
            if n > 0:
                rest, young = queue[: max(0, len(queue) - n)], queue[max(0, len(queue) - n):]
                if any(k == "piece" for k, _ in rest) and any(k == "load" for k, _ in young) and no not in seen:
                    seen.add(no)
                    bad.append((name, no, n, list(queue)[-12:]))
            queue = queue[len(queue) - n:] if n > 0 else []
        return queue

    walk(0, len(lines), [])
    return bad


def check_source(src):
    text = assemble(src)
    kernels, cur = {}, None
    for no, line in enumerate(text, 1):
        m = _KERNEL.match(line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            kernels[cur].append((no, line))
    stats, bad = {}, []
    for name, lines in kernels.items():
        pieces = sum(1 for _, t in lines if kind_of(t) == "piece")
        counted = sum(1 for _, t in lines if re.search(r"s_waitcnt.*vmcnt\(([1-9]\d*)\)", t))
        if pieces:
            stats[name] = (pieces, counted)
            bad += check_kernel(name, lines)
    return stats, bad


def main(argv):
    files = argv or ["bneck.hip", "wstat.hip"]
    rc = 0
    for f in files:
        src = f if os.path.isabs(f) else os.path.join(CSRC, f)
        stats, bad = check_source(src)
        print("%s: %d kernels with DMA pieces, %d counted waits checked" % (os.path.basename(src), len(stats), sum(c for _, c in stats.values())))
        for name, no, n, q in bad:
            rc = 1
            print("  VIOLATION %s: asm line %d: vmcnt(%d) may return with a covered piece in flight (an ordinary load among its youngest N): %s" % (name, no, n, q))
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
