#!/usr/bin/env python3
"""Print the per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print(f"{'kernel':70s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}")
for name, calls, total, avg, pct in rows:
    print(f"{name[:70]:70s} {calls:6d} {total / 1e6:10.3f} {avg / 1e3:10.1f} {pct:6.2f}")
