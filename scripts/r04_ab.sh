#!/bin/bash
# round-4 A/B runs inside ONE gpurun call: every variant in its own process, the summary line of each appended to gpurun_out/$1.txt
OUT=gpurun_out/$1.txt; : > $OUT
run() {  # label, env..., -- bench args
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --soak-seconds 0 --no-side-runs --steps 40 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j['roofline']
print('%-34s ms/step %.4f  median %.4f  kernel avg %.2f us median %.2f us  stages %s  check %s' % ('$label', j['ms_per_step'], j['median_ms_per_step'] or 0, r['avg_launch_ms']*1e3, r['median_launch_ms']*1e3, {k: round(v*1e3,1) for k,v in j['stages_ms'].items()}, j['result_check']['ok']))" >> $OUT 2>&1
}
run "c3 TPW=1"                FDJAC_STORE_TPW=1 -- --config c3
run "c3 TPW=2"                FDJAC_STORE_TPW=2 -- --config c3
run "c3 TPW=4"                FDJAC_STORE_TPW=4 -- --config c3
run "c3 TPW=8"                FDJAC_STORE_TPW=8 -- --config c3
run "c3 TPW=32"               FDJAC_STORE_TPW=32 -- --config c3
run "c3 TPW=4 WAVES=5"        FDJAC_STORE_TPW=4 FDJAC_STORE_WAVES=5 -- --config c3
run "c3 TPW=4 WAVES=1"        FDJAC_STORE_TPW=4 FDJAC_STORE_WAVES=1 -- --config c3
run "c3 TPW=1 again"          FDJAC_STORE_TPW=1 -- --config c3
run "c4 materialized"         X=1 -- --config c4 --f-mode materialized
run "c3 materialized"         X=1 -- --config c3 --f-mode materialized
cat $OUT
