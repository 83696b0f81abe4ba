#!/bin/bash
# Per-kernel average durations of one bench.py run under rocprofv3 --kernel-trace --stats.
# usage (on the GPU box): tools/kernel_times.sh <tag> <bench.py arguments...>   -> gpurun_out/kt_<tag>/stats.txt
set -u
TAG=$1; shift
OUT=gpurun_out/kt_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k -- python3 bench.py --no-cpu-baseline --no-rm-leg "$@" > "$OUT/trace.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
grep "^{" "$OUT/trace.log" > "$OUT/bench.json"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
f = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(os.path.join(out, "stats.txt"), "w") as w:
    for r in rows:
        line = f"{r['Name'][:70]:70s} calls {int(r['Calls']):6d} avg_us {float(r['AverageNs'])/1e3:10.1f} min_us {float(r['MinNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} pct {float(r['Percentage']):6.2f}"
        print(line); w.write(line + "\n")
PY
rm -rf "$OUT/trace"
