#!/bin/bash
# the reduction's own launch: tickets (k_eps_partial_reg) against flags (k_eps_flags), configs 3, 4, 5
mkdir -p gpurun_out/r06_flags
timeout 1200 python -m pytest tests -x -q -m gpu -k "step_size or eps or fused or sharded" 2>&1 | tail -3
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = d["roofline"]; st = d.get("stage_ms") or {}
    print("%-12s ms/step %.4f eps %s kernel %s avg %.4f ms check %s" % (sys.argv[1], d["ms_per_step"], (d.get("stages_us") or d.get("stage_us") or st), r["kernel"][:30], r["avg_launch_ms"], d.get("result_check", {}).get("ok")))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
export FDJAC_TEST_SWITCHES=1
for rep in 1 2; do
for c in c4 c3 c5; do
  for v in 0 1; do
    FDJAC_EPS_FLAGS=$v python bench.py --config $c > gpurun_out/r06_flags/${c}_flags$v.json 2>gpurun_out/r06_flags/${c}_flags$v.err; show ${c}_flags$v gpurun_out/r06_flags/${c}_flags$v.json
  done
done
done
