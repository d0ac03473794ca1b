#!/bin/bash
# the held step: parity, then c4 with and without it (two-launch form)
mkdir -p gpurun_out/r06_held
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -k "held" > gpurun_out/r06_held/pytest.log 2>&1
tail -5 gpurun_out/r06_held/pytest.log
for v in 1 0; do
  FDJAC_TEST_SWITCHES=1 FDJAC_FUSED_HELD=$v timeout 600 python bench.py --config c4 --steps 200 --warmup 20 > gpurun_out/r06_held/bench_c4_held$v.json 2> gpurun_out/r06_held/bench_c4_held$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r06_held/bench_c4_held$v.json').read().strip().splitlines()[-1])
print('held=$v', d.get('ms_per_step'), d.get('value'), d['roofline'].get('kernel'), d['roofline'].get('kernel_us'), d['roofline'].get('frac'), d.get('median_call_ms'))
PY
done
FDJAC_TEST_SWITCHES=1 FDJAC_FUSED_TRACE=1 timeout 300 python scripts/fused_trace.py 10000000 > gpurun_out/r06_held/trace.txt 2>&1
tail -20 gpurun_out/r06_held/trace.txt
