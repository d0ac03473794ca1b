#!/bin/bash
# A/B of one library variant on the graded kernels of c3 / c4 / c5 and the Float32 forms of c3 / c5 (one box)
cd "$(dirname "$0")/.."
v=$1
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-14s ms/step %.4f  kernel %s avg %.4f ms median %.4f ms  frac %.3f  check %s" % (sys.argv[1], d["ms_per_step"], r["kernel"][:36], r["avg_launch_ms"], r["median_launch_ms"], r["frac"], d.get("result_check", {}).get("ok")))
PY
}
for cfg in "c3" "c3 --dtype f32" "c5" "c5 --dtype f32" "c4" "c4 --dtype f32"; do
for rep in 1 2; do
python bench.py --config $cfg --soak-seconds 0 --no-cpu-baseline --no-plain-handover > gpurun_out/ab_base.json 2>/dev/null; show "base $cfg" gpurun_out/ab_base.json
scripts/with_variant.sh $v python bench.py --config $cfg --soak-seconds 0 --no-cpu-baseline --no-plain-handover > gpurun_out/ab_$v.json 2>/dev/null; show "$v $cfg" gpurun_out/ab_$v.json
done
done
