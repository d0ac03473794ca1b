#!/bin/bash
# config 5 in both element types at the head (events; the kernel's name, average and median launch)
cd "$(dirname "$0")/.."
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-10s ms/step %.4f  kernel %s avg %.4f ms median %.4f ms  frac %.3f  check %s" % (sys.argv[1], d["ms_per_step"], r["kernel"][:40], r["avg_launch_ms"], r["median_launch_ms"], r["frac"], d.get("result_check", {}).get("ok")))
PY
}
for rep in 1 2; do
python bench.py --config c5 --dtype f32 --soak-seconds 0 --no-cpu-baseline --no-plain-handover > gpurun_out/ab_c5f32.json 2>/dev/null; show c5_f32 gpurun_out/ab_c5f32.json
python bench.py --config c5 --soak-seconds 0 --no-cpu-baseline --no-plain-handover > gpurun_out/ab_c5.json 2>/dev/null; show c5 gpurun_out/ab_c5.json
done
