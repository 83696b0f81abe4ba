#!/usr/bin/env python3
"""Run bench.py for a few workloads and print the key numbers (development aid)."""
import json
import os
import subprocess
import sys

args = sys.argv[1:] or ["car", "atrium"]
for w in args:
    out = subprocess.run([sys.executable, "bench.py", "--workload", w, "--steps", "3", "--warmup", "1", "--spp", "64",
                          "--no-cpu-baseline --no-rm-leg"] + os.environ.get("QB_ARGS", "").split(), capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(w, "FAILED", out.stdout[-2000:], out.stderr[-2000:])
        continue
    j = json.loads(line[-1])
    r = j["roofline"]
    print(f"{w}: value {j['value']} Mrays/s, kernel {r['kernel_mrays_per_s']} Mrays/s, ms/launch {r['ms_per_launch']}, "
          f"nodes/ray {r['nodes_per_ray']}, tris/ray {r['tris_per_ray']}, frac {r['frac']}, ms/step {j['ms_per_step']}, "
          f"tree {j['config']['device_tree']['dev_nodes']}n/{j['config']['device_tree']['dev_refs']}r depth {j['config']['device_tree']['dev_depth']} setup {j['scene_setup_s']}s")
