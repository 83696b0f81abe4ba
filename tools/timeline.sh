#!/bin/bash
# Device timeline of one bench.py run: per kernel busy time, time with nothing running, overlap.
# usage (on the GPU box): tools/timeline.sh <tag> <bench.py arguments...>   -> gpurun_out/tl_<tag>/timeline.txt
set -u
TAG=$1; shift
OUT=gpurun_out/tl_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o k -- python3 bench.py --no-cpu-baseline --no-rm-leg "$@" > "$OUT/trace.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
grep "^{" "$OUT/trace.log" > "$OUT/bench.json"
python3 - "$OUT" <<'PY' | tee "$OUT/timeline.txt"
import csv, glob, sys, os, json
out = sys.argv[1]
f = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in csv.DictReader(open(f))]
rows.sort()
j = json.load(open(os.path.join(out, "bench.json")))
# the timed region: the last steps * launches_per_step closest-hit launches of the <false> instantiation, and what runs between them
tr = [r for r in rows if "k_trace_closest<false" in r[2]]
per_step = len(tr) // (j["steps"] + j["warmup"])
first = tr[j["warmup"] * per_step]
prim = [r for r in rows if ("k_packet_prep" in r[2] or "k_primary_search" in r[2]) and r[0] <= first[0]]
t0 = prim[-1][0]
t1 = tr[(j["steps"] + j["warmup"]) * per_step - 1][1]
# the view ends with k_resolve after the last trace
res = [r for r in rows if "k_resolve" in r[2] and r[0] >= t1]
t1 = res[0][1] if res else t1
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
span = (t1 - t0) / 1e6
busy = {}
for s, e, n in sel:
    busy[n] = busy.get(n, 0) + (e - s) / 1e6
# union of intervals
ev = sorted(sel)
covered, cur_s, cur_e = 0, None, None
for s, e, n in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: covered += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
covered += cur_e - cur_s
print(f"timed region {span:.2f} ms over {j['steps']} steps = {span / j['steps']:.2f} ms per step (bench: {j['ms_per_step']} ms)")
print(f"device idle (no kernel running) {span - covered / 1e6:.2f} ms = {100 * (1 - covered / 1e6 / span):.1f} %")
for n, b in sorted(busy.items(), key=lambda x: -x[1]):
    print(f"  {n:42s} {b:9.2f} ms  {100 * b / span:5.1f} % of the region")
tt = sum(b for n, b in busy.items() if "k_trace_closest<false" in n)
print(f"not covered by the closest-hit kernel: {span - tt:.2f} ms = {100 * (1 - tt / span):.1f} %")
# the kernels of the last timed step in start order (ms from the step's first kernel)
steps_prim = [r for r in rows if "k_packet_prep" in r[2] and t0 <= r[0] <= t1]
if steps_prim:
    b = steps_prim[-1][0]
    print("last step, kernel by kernel (start .. end, ms):")
    for s_, e_, n in sel:
        if s_ >= b and e_ - s_ > 20000:
            print(f"  {(s_ - b) / 1e6:8.3f} .. {(e_ - b) / 1e6:8.3f}  {n}")
PY
rm -rf "$OUT/trace"
