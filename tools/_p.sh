cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/prof_p -o p -- python3 tools/_p.py > gpurun_out/prof_p.log 2>&1 || exit 1
python3 - <<'PY'
import sqlite3, glob
db = sqlite3.connect(sorted(glob.glob("gpurun_out/prof_p/*.db"))[-1])
for row in db.execute("select name, start, end from kernels order by start"):
    if "exact" in row[0]: print(row[0][:40], round((row[2] - row[1]) / 1e3, 1), "us")
PY
