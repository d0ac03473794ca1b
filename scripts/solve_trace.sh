#!/bin/bash
# one fd_tridiag_solve_async at N = 10^7 (CSC layout) kernel by kernel, in launch order: duration and the gap to the kernel before
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${1:-solve_trace}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/tr -o run -- python $REPO/scripts/solve_probe.py > $OUT/trace.log 2>&1
DB=$(find $OUT/tr -name '*.db' | head -1)
python - "$DB" ${2:-5} > $OUT/solve_trace.md <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
sc = "start" if "start" in cols else [c for c in cols if "start" in c][0]
ec = "end" if "end" in cols else [c for c in cols if "end" in c][0]
rows = cur.execute("select name, %s, %s from kernels order by %s" % (sc, ec, sc)).fetchall()
rows = [r for r in rows if "tri_" in r[0]]
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 5
last = rows[-(NL + 2):]
print("| kernel | us | gap before us |")
print("|---|---|---|")
prev = None
for n, s, e in last:
    print("| %s | %.2f | %s |" % (n.split("(")[0].replace("fdjac::", "")[:90], (e - s) / 1e3, "%.2f" % ((s - prev) / 1e3) if prev else "-"))
    prev = e
print("span of the last solve: %.1f us" % ((last[-1][2] - last[-NL][1]) / 1e3))
PY
tail -3 $OUT/trace.log >> $OUT/solve_trace.md
rm -rf $OUT/tr
cat $OUT/solve_trace.md
