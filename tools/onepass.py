import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:34]) for r in csv.DictReader(open(f))]
rows.sort()
tr = [i for i, r in enumerate(rows) if "k_trace_closest<false" in r[2]]
a = tr[len(tr) // 2]            # a launch in the middle of the run
t0 = rows[a][0]
for s, e, n in rows[a - 3: a + 12]:
    print(f"{(s - t0) / 1e6:9.3f} ms  +{(e - s) / 1e6:8.3f} ms  {n}")
