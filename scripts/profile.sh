#!/bin/bash
# rocprofv3 passes for the headline bench (run on the GPU box via gpurun). Outputs under gpurun_out/prof/.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o run -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o run -- $CMD > $OUT/pmc_write.log 2>&1
find $OUT -type f | head -50
