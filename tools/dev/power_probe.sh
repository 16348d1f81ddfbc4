#!/bin/bash
# Samples the board power / clock (rocm-smi) every 0.5 s while bench.py runs with the bf16-piece switch off and on.
# usage (through gpurun): bash tools/dev/power_probe.sh <tag>
TAG=${1:-power}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for MODE in 0 2; do
  ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|Average Graphics Package Power|sclk clock level" | tr '\n' ' '; echo; sleep 0.5; done ) > $OUT/smi_$MODE.log &
  SMI=$!
  CONTEXTFLOW_BF16_SPLIT=$MODE timeout -k 10 300 python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_$MODE.json 2> $OUT/bench_$MODE.err
  kill $SMI; wait $SMI 2>/dev/null
done
python3 - <<PY
import json, re, statistics
for m in (0, 2):
    j = json.loads(open("$OUT/bench_%d.json" % m).read().strip().splitlines()[-1])
    pw, ck = [], []
    for line in open("$OUT/smi_%d.log" % m):
        p = re.search(r"Power \(W\): ([0-9.]+)", line); c = re.search(r"\(([0-9]+)Mhz\)", line)
        if p: pw.append(float(p.group(1)))
        if c: ck.append(float(c.group(1)))
    busy = sorted(pw)[len(pw) // 2:] if pw else []
    print("bf16 switch %d: %.3f M samples/s, %.1f ms/step; power samples %d, median of the upper half %.0f W, max %.0f W; sclk median %s MHz"
          % (m, j["value"] / 1e6, j["ms_per_step"], len(pw), statistics.median(busy) if busy else float("nan"), max(pw) if pw else float("nan"),
             statistics.median(sorted(ck)[len(ck) // 2:]) if ck else "n/a"))
PY
