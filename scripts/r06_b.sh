#!/bin/bash
# fused-step iteration: parity + c2 / c4 timings (tag = $1)
cd "$(dirname "$0")/.."
O=gpurun_out/r06_${1:-b}; mkdir -p $O
export FDJAC_TEST_SWITCHES=1
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q > $O/pytest_fused.log 2>&1; echo "fused rc=$?" >> $O/pytest_fused.log
tail -4 $O/pytest_fused.log
for rep in 1 2; do
for cfg in c2; do
  FDJAC_FUSED_MAX_N=0 timeout 300 python bench.py --config $cfg --no-side-runs --no-cpu-baseline --soak-seconds 0 > $O/bench_${cfg}_unfused_$rep.json 2> $O/bench_${cfg}_unfused_$rep.err
  timeout 300 python bench.py --config $cfg --no-side-runs --no-cpu-baseline --soak-seconds 0 > $O/bench_${cfg}_fused_$rep.json 2> $O/bench_${cfg}_fused_$rep.err
done
done
FDJAC_FUSED_MAX_N=100000000 timeout 300 python bench.py --config c4 --no-side-runs --no-cpu-baseline --soak-seconds 0 > $O/bench_c4_fused.json 2> $O/bench_c4_fused.err
timeout 300 python bench.py --config c2 --dtype f32 --no-side-runs --no-cpu-baseline --soak-seconds 0 > $O/bench_c2_f32_fused.json 2> $O/bench_c2_f32_fused.err
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("ms_per_step","median_ms_per_step")}, {k:round(v*1e3,2) for k,v in d.get("stages_ms").items()})
except Exception as e: print("ERR",e)
PY
done
