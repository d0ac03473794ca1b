#!/bin/bash
# the 7-point column kernel taken apart: variants built by scripts/build_variant.sh (-DLAP7_NOSTORE, ...), store path only
cd "$(dirname "$0")/.."
run() { python scripts/pattern_probe_store.py --only stencil --store-only --reps 10 2>/dev/null | grep "7-point" | awk -F'|' -v v="$1" '{printf "%-12s %-8s call %s us  store kernel %s  f! %s\n", v, $3, $8, $9, $11}'; }
run base
for v in "$@"; do
  lib=finitediff.jl_amd/lib
  cp $lib/libfdjac.so $lib/libfdjac.so.keep; cp $lib/variants/libfdjac_$v.so $lib/libfdjac.so
  run $v
  mv $lib/libfdjac.so.keep $lib/libfdjac.so
done
