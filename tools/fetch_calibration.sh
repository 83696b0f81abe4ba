#!/bin/bash
# What FETCH_SIZE reports for loads whose memory-side traffic is known (tools/ta_rate --calibrate), per the guide's unit and gfx950 note:
# one rocprofv3 --pmc run per counter set and case.   -> gpurun_out/fetch_calibration.txt
set -u
OUT=gpurun_out/fetch_cal; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in divergent stream; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${c}_fetch -o p -- tools/ta_rate --calibrate $c > $OUT/${c}_fetch.log 2>&1 || { echo "$c fetch failed"; tail -3 $OUT/${c}_fetch.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/${c}_rdreq -o p -- tools/ta_rate --calibrate $c > $OUT/${c}_rdreq.log 2>&1 || { echo "$c rdreq failed"; tail -3 $OUT/${c}_rdreq.log; exit 1; }
done
python3 - $OUT <<'PY' | tee gpurun_out/fetch_calibration.txt
import csv, glob, json, os, sys
out = sys.argv[1]
for c, kern in (("divergent", "k_gather"), ("stream", "k_stream")):
    info = json.loads([l for l in open(f"{out}/{c}_fetch.log") if l.startswith("{")][-1])
    vals = {}
    for sub in ("fetch", "rdreq"):
        for f in glob.glob(f"{out}/{c}_{sub}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    vals[r["Counter_Name"]] = float(r["Counter_Value"])
    print(c, info, vals, "FETCH_SIZE at face value (KiB x 1024):", vals.get("FETCH_SIZE", 0) * 1024)
PY
rm -rf $OUT
