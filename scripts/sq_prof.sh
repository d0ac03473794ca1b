#!/bin/bash
# SQ counters of one bench configuration's kernels: where the wave cycles go.  usage (GPU box): sq_prof.sh OUTDIR CONFIG [bench args]
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${1:-sq_prof}
CFG=${2:-c3}
shift 2 || true
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-plain-handover --soak-seconds 0 $*"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/sq -o run -- $CMD > $OUT/sq.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d $OUT/sq2 -o run -- $CMD > $OUT/sq2.log 2>&1
S=$(find $OUT/stats -name '*.db' | head -1)
DBS=""
for d in sq sq2; do F=$(find $OUT/$d -name '*.db' | head -1); [ -n "$F" ] && DBS="$DBS $F"; done
python $REPO/scripts/rocpd_summary.py $S $DBS 2>&1 | grep -E "^\| kernel|^\|---|k_f_|k_eps|^## " > $OUT/summary.md
rm -rf $OUT/stats $OUT/sq $OUT/sq2
cat $OUT/summary.md
