#!/bin/bash
# A/B of the Float32 5-point storing kernel against library variants (scripts/build_variant.sh), with the new kernel's tests on each variant
cd "$(dirname "$0")/.."
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-10s ms/step %.4f  kernel %s avg %.4f ms median %.4f ms  frac %.3f  check %s" % (sys.argv[1], d["ms_per_step"], r["kernel"][:40], r["avg_launch_ms"], r["median_launch_ms"], r["frac"], d.get("result_check", {}).get("ok")))
PY
}
for rep in 1 2; do
python bench.py --config c3 --dtype f32 --soak-seconds 0 --no-cpu-baseline --no-plain-handover > gpurun_out/ab_base.json 2>/dev/null; show base gpurun_out/ab_base.json
for v in "$@"; do
  scripts/with_variant.sh $v python bench.py --config c3 --dtype f32 --soak-seconds 0 --no-cpu-baseline --no-plain-handover > gpurun_out/ab_$v.json 2>/dev/null; show $v gpurun_out/ab_$v.json
done
done
for v in "$@"; do scripts/with_variant.sh $v python -m pytest tests/test_gpu_float32.py -q -m gpu -k "stencil5 or lap5" 2>&1 | tail -2; done
