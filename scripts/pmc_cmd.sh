#!/bin/bash
# one rocprofv3 PMC pass of an arbitrary command: pmc_cmd.sh NAME KERNEL_LIKE "COUNTERS" -- command...   -> gpurun_out/NAME.md
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; KERN=$2; GROUP=$3; shift 4
OUT=$REPO/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 ${PMC_PASS_TIMEOUT:-200} rocprofv3 --pmc $GROUP --kernel-trace -d $OUT/pmcx -o run -- "$@" > $OUT/log.txt 2>&1
DB=$(find $OUT/pmcx -name '*.db' | head -1)
python - "$DB" "$KERN" > $REPO/gpurun_out/$NAME.md <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
            "where kernel_name like ? group by kernel_name, counter_name", ("%" + sys.argv[2] + "%",))
for k, c, n, v, d in cur.fetchall():
    print("| %s | %s | %d | %.4g | %.1f |" % (k.split("(")[0].replace("void fdjac::", "")[:60], c, n, v, d / 1e3))
PY
rm -rf $OUT/pmcx
cat $REPO/gpurun_out/$NAME.md
