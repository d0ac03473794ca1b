#!/bin/bash
# A/B of the block-coupled storing kernel's workgroup shape on config 5 (variants built by scripts/build_variant.sh)
cd "$(dirname "$0")/.."
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-10s ms/step %.4f  kernel %s avg %.4f ms median %.4f ms  frac %.3f  check %s" % (sys.argv[1], d["ms_per_step"], r["kernel"][:40], r["avg_launch_ms"], r["median_launch_ms"], r["frac"], d.get("result_check", {}).get("ok")))
PY
}
for rep in 1 2; do
python bench.py --config c5 $BENCH_EXTRA --soak-seconds 0 --no-cpu-baseline --no-plain-handover > gpurun_out/ab_base.json 2>/dev/null; show base gpurun_out/ab_base.json
for v in "$@"; do
  scripts/with_variant.sh $v python bench.py --config c5 $BENCH_EXTRA --soak-seconds 0 --no-cpu-baseline --no-plain-handover > gpurun_out/ab_$v.json 2>/dev/null; show $v gpurun_out/ab_$v.json
done
done
