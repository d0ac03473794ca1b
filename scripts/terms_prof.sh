#!/bin/bash
# rocprofv3 passes of the random band through the row-wise store of a SEPARABLE user functor (scripts/terms_probe.py): kernel times, SQ counters, FETCH / WRITE.
# counters of the storing kernel (where the wave cycles go), then FETCH / WRITE.  usage (GPU box): terms_prof.sh OUTDIR [terms|builtin|cols]
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${1:-band_prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/scripts/terms_probe.py --only ${2:-terms} --reps 8"
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
timeout -k 5 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/sq -o run -- $CMD > $OUT/sq.log 2>&1
timeout -k 5 240 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d $OUT/sq2 -o run -- $CMD > $OUT/sq2.log 2>&1
timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o run -- $CMD > $OUT/fetch.log 2>&1
timeout -k 5 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o run -- $CMD > $OUT/write.log 2>&1
S=$(find $OUT/stats -name '*.db' | head -1)
DBS=""
for d in sq sq2 fetch write; do F=$(find $OUT/$d -name '*.db' | head -1); [ -n "$F" ] && DBS="$DBS $F"; done
python $REPO/scripts/rocpd_summary.py $S $DBS 2>&1 | grep -E "^\| kernel|^\|---|store_rows|store_sorted|store_cols|^## " > $OUT/summary.md
grep -E "^\| " $OUT/stats.log >> $OUT/summary.md
tail -3 $OUT/sq.log $OUT/sq2.log >> $OUT/summary.md
rm -rf $OUT/stats $OUT/sq $OUT/sq2 $OUT/fetch $OUT/write
cat $OUT/summary.md
