#!/bin/bash
# Extra rocprofv3 PMC passes (one counter group per run, --kernel-trace only) for a bench config.
# usage: pmc_passes.sh NAME KERNEL_LIKE "GROUP1" "GROUP2" ... -- bench args     -> gpurun_out/NAME/pmc_extra.md
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; KERN=$2; shift 2
GROUPS_=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift || true
OUT=$REPO/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-plain-handover $*"
: > $OUT/pmc_extra.md
i=0
for g in "${GROUPS_[@]}"; do
  i=$((i+1))
  rm -rf $OUT/pmcx
  # a counter group the hardware cannot schedule makes rocprofv3 abort and then hang in its signal handler: bound every pass
  timeout -k 5 ${PMC_PASS_TIMEOUT:-120} rocprofv3 --pmc $g --kernel-trace -d $OUT/pmcx -o run -- $CMD > $OUT/pmcx_$i.log 2>&1
  DB=$(find $OUT/pmcx -name '*.db' | head -1)
  if [ -n "$DB" ]; then
    python - "$DB" "$KERN" >> $OUT/pmc_extra.md <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
            "where kernel_name like ? group by kernel_name, counter_name", ("%" + sys.argv[2] + "%",))
for k, c, n, v, d in cur.fetchall():
    print("| %s | %s | %d | %.4g | %.1f |" % (k.split("(")[0].replace("void fdjac::", "")[:60], c, n, v, d / 1e3))
PY
  else
    echo "| group $g failed |" >> $OUT/pmc_extra.md; tail -3 $OUT/pmcx_$i.log >> $OUT/pmc_extra.md
  fi
done
rm -rf $OUT/pmcx
cat $OUT/pmc_extra.md
