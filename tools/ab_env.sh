#!/bin/bash
# A/B runs of bench.py under sets of environment knobs (on a GPU box).
# usage: tools/ab_env.sh <tag> "<bench args>" "VAR=1 VAR2=2" "VAR=3" ...   -> gpurun_out/ab_<tag>.txt
set -u
TAG=$1; ARGS=$2; shift 2
OUT=gpurun_out/ab_$TAG.txt
mkdir -p gpurun_out
: > "$OUT"
for cfg in "$@"; do
  line=$(env $cfg timeout -k 10 240 python3 -u bench.py --no-cpu-baseline --no-rm-leg $ARGS 2>/dev/null | grep '^{' | tail -1) || { echo "$cfg: failed" >> "$OUT"; exit 1; }
  python3 - "$cfg" "$line" >> "$OUT" <<'PY'
import json, sys
j = json.loads(sys.argv[2])
print("%-60s value %8.1f  ms/step %7.2f  kernel %8.1f" % (sys.argv[1], j["value"], j["ms_per_step"], j["roofline"].get("kernel_mrays_per_s") or 0))
PY
done
cat "$OUT"
