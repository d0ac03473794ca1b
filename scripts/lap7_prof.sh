set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05_y_lap7
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/scripts/pattern_probe_store.py --only stencil --store-only --reps 5"
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
timeout -k 5 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/sq -o run -- $CMD > $OUT/sq.log 2>&1
timeout -k 5 240 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/sq2 -o run -- $CMD > $OUT/sq2.log 2>&1
S=$(find $OUT/stats -name '*.db' | head -1)
DBS=""
for d in sq sq2; do F=$(find $OUT/$d -name '*.db' | head -1); [ -n "$F" ] && DBS="$DBS $F"; done
python $REPO/scripts/rocpd_summary.py $S $DBS 2>&1 | grep -E "^\| kernel|^\|---|lap7_store" > $OUT/summary.md
rm -rf $OUT/stats $OUT/sq $OUT/sq2
cat $OUT/summary.md
