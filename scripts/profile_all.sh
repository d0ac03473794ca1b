#!/bin/bash
# One gpurun call: rocprofv3 stats + PMC passes of bench.py for a config, summarised on the box.
# usage: profile_all.sh NAME CONFIG N KERNEL [bench args...]   -> gpurun_out/NAME/{summary.md,pmc.json,bench.json}
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; CFG=$2; N=$3; KERN=$4; shift 4
OUT=$REPO/gpurun_out/$NAME
bash $REPO/scripts/profile.sh $NAME --config $CFG "$@"
S=$(find $OUT/stats -name '*.db' | head -1); F=$(find $OUT/pmc_fetch -name '*.db' | head -1); W=$(find $OUT/pmc_write -name '*.db' | head -1)
python $REPO/scripts/rocpd_summary.py $S $F $W > $OUT/summary.md 2> $OUT/summary.err
# (forward / central configs hand differences over unless FDJAC_LAZY_DIFF=0; the complex step never does)
DIFF=1; [ "${FDJAC_LAZY_DIFF:-1}" = "0" ] && DIFF=0; [ "$CFG" = "c5" ] && DIFF=0
# (a storing f! launch is the graded kernel of the exact-band configs unless FDJAC_LAZY_STORE=0)
STORE=0; case "$KERN" in *store*|*fused*) STORE=1;; esac
python $REPO/scripts/make_pmc_json.py $F $W $N 1 $KERN $DIFF $STORE > $OUT/pmc.json 2>> $OUT/summary.err
cd $REPO && python bench.py --config $CFG "$@" > $OUT/bench.json 2> $OUT/bench.err
python - $OUT >> $OUT/summary.md <<'PY'
import re, sys
d = sys.argv[1]
def ev(path):
    m = re.search(r'"avg_launch_ms": ([0-9.]+)', open(path).read())
    return float(m.group(1)) * 1e3 if m else float("nan")
print("\n## HIP events vs rocprofv3 (same kernel)\n")
print("HIP-event time of the graded kernel measured by `bench.py` inside the profiled process (the `--kernel-trace --stats` "
      "pass above): **%.1f us**; in the separate, unprofiled `bench.py` run of the same command: **%.1f us**. The event "
      "bracket contains the launch gaps the profiler adds; different processes place their buffers differently "
      "(DESIGN section 8: +-5 %% per placement)." % (ev(d + "/stats.log"), ev(d + "/bench.json")))
PY
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
tail -3 $OUT/summary.err; head -12 $OUT/summary.md; cat $OUT/pmc.json | head -30
