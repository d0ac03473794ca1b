#!/bin/bash
# round-6 first measurement: fused step parity + timings, rank-share baseline
cd "$(dirname "$0")/.."
O=gpurun_out/r06_a; mkdir -p $O
export FDJAC_TEST_SWITCHES=1
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q > $O/pytest_fused.log 2>&1; echo "fused rc=$?" >> $O/pytest_fused.log
tail -15 $O/pytest_fused.log
for cfg in c2 c4; do
  FDJAC_FUSED_MAX_N=0 timeout 300 python bench.py --config $cfg --no-side-runs --no-cpu-baseline --soak-seconds 0 > $O/bench_${cfg}_unfused.json 2> $O/bench_${cfg}_unfused.err
  FDJAC_FUSED_MAX_N=0 FDJAC_EPS_FORM=1 timeout 300 python bench.py --config $cfg --no-side-runs --no-cpu-baseline --soak-seconds 0 > $O/bench_${cfg}_unfused_form1.json 2> $O/bench_${cfg}_unfused_form1.err
  FDJAC_FUSED_MAX_N=100000000 timeout 300 python bench.py --config $cfg --no-side-runs --no-cpu-baseline --soak-seconds 0 > $O/bench_${cfg}_fused.json 2> $O/bench_${cfg}_fused.err
done
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step","median_ms_per_step","stages_ms")})
except Exception as e: print("ERR",e)
PY
done
timeout 600 python scripts/bench_sides.py --ranks 2 4 8 > $O/rank_share_c4.json 2> $O/rank_share_c4.err; cat $O/rank_share_c4.json; tail -3 $O/rank_share_c4.err
timeout 900 python -m pytest tests/test_gpu_multigpu.py tests/test_gpu_float32.py tests/test_gpu_fullsize.py -x -q > $O/pytest_more.log 2>&1; tail -5 $O/pytest_more.log
