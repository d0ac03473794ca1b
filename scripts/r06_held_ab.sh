#!/bin/bash
# A/B of the held step's variants on config 4 (variants: scripts/build_variant.sh)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06_held
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-14s ms/step %.4f median %.4f kernel %s avg %.4f ms frac %.3f check %s" % (sys.argv[1], d["ms_per_step"], d.get("median_call_ms") or -1, r["kernel"][:34], r["avg_launch_ms"], r["frac"], d.get("result_check", {}).get("ok")))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
export FDJAC_TEST_SWITCHES=1
for rep in 1 2; do
  FDJAC_FUSED_HELD=0 python bench.py --config c4 $BARGS > gpurun_out/r06_held/ab_two.json 2>/dev/null; show two_launch gpurun_out/r06_held/ab_two.json
  python bench.py --config c4 $BARGS > gpurun_out/r06_held/ab_held.json 2>/dev/null; show held gpurun_out/r06_held/ab_held.json
  for v in "$@"; do
    case $v in occ*) export FDJAC_FUSED_HELD=0;; *) unset FDJAC_FUSED_HELD;; esac
    scripts/with_variant.sh $v python bench.py --config c4 $BARGS > gpurun_out/r06_held/ab_$v.json 2>/dev/null; show $v gpurun_out/r06_held/ab_$v.json
    unset FDJAC_FUSED_HELD
  done
done
