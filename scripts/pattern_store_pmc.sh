#!/bin/bash
# rocprofv3 passes of scripts/pattern_probe_store.py, one pattern per run: kernel time + FETCH_SIZE / WRITE_SIZE of the storing
# kernels, calibrated on k_stream_copy of the same run.  usage (on the GPU box): pattern_store_pmc.sh OUTDIR [extra probe args]
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${1:-pattern_store_pmc}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pat in stencil band; do
  CMD="python $REPO/scripts/pattern_probe_store.py --only $pat --reps 5 $*"
  timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/${pat}_stats -o run -- $CMD > $OUT/${pat}_stats.log 2>&1
  timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${pat}_fetch -o run -- $CMD > $OUT/${pat}_fetch.log 2>&1
  timeout -k 5 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${pat}_write -o run -- $CMD > $OUT/${pat}_write.log 2>&1
  S=$(find $OUT/${pat}_stats -name '*.db' | head -1); F=$(find $OUT/${pat}_fetch -name '*.db' | head -1); W=$(find $OUT/${pat}_write -name '*.db' | head -1)
  if [ -n "$S" ] && [ -n "$F" ] && [ -n "$W" ]; then
    python $REPO/scripts/rocpd_summary.py $S $F $W 2>&1 | grep -E "^\| kernel|^\|---|store|k_decompress|k_stream_copy|k_f_|k_perturb|^## " > $OUT/${pat}_summary.md
  fi
  grep -E "^\| " $OUT/${pat}_stats.log >> $OUT/${pat}_summary.md
  rm -rf $OUT/${pat}_stats $OUT/${pat}_fetch $OUT/${pat}_write
done
cat $OUT/stencil_summary.md $OUT/band_summary.md
