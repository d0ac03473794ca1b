#!/bin/bash
# the generic column kernel (fd_csc_store_cols with a functor), store path only: variant NAME runs the random band (`sparse` family),
# variant NAMEG (built with -DLAP7_GENERIC) the 7-point functor routed through the generic kernel
cd "$(dirname "$0")/.."
run() { python scripts/pattern_probe_store.py --reps 8 --store-only $2 2>/dev/null | grep -E "7-point|random band" | awk -F'|' -v v="$1" '{printf "%-6s %-34s %-8s call %s us  store %s\n", v, $2, $3, $8, $9}'; }
lib=finitediff.jl_amd/lib
run base "--only band"
for v in "$@"; do
  cp $lib/libfdjac.so $lib/libfdjac.so.keep; cp $lib/variants/libfdjac_$v.so $lib/libfdjac.so
  run $v "--only band"
  cp $lib/variants/libfdjac_${v}G.so $lib/libfdjac.so
  run ${v}G "--only stencil"
  mv $lib/libfdjac.so.keep $lib/libfdjac.so
done
