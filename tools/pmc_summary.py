#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes: per kernel, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_trace_closest<false"
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in agg:
    if want not in k:
        continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"  {c:40s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
