#!/bin/bash
for c in "$@"; do
  t0=$(date +%s.%N)
  python bench.py --config $c > gpurun_out/live_$c.json 2> gpurun_out/live_$c.err
  t1=$(date +%s.%N)
  python - $c $t0 $t1 <<'PY'
import json, sys
c, t0, t1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
d = json.loads(open("gpurun_out/live_%s.json" % c).read().strip().splitlines()[-1]); r = d["roofline"]
print(c, "wall %.1f s" % (t1 - t0), d["ms_per_step"], r["kernel"], r["traffic"], "%.3f" % r["frac"], r["traffic_source"][:150], r.get("traffic_live"))
PY
done
