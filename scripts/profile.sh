#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box via gpurun). usage: profile.sh [outdir-name] [bench args...]
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
NAME=${1:-prof}; shift || true
OUT=$REPO/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --soak-seconds 0 --no-plain-handover $*"
timeout -k 5 ${PROF_PASS_TIMEOUT:-150} rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
timeout -k 5 ${PROF_PASS_TIMEOUT:-150} rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o run -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout -k 5 ${PROF_PASS_TIMEOUT:-150} rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o run -- $CMD > $OUT/pmc_write.log 2>&1
