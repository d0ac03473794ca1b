mkdir -p gpurun_out/r06_s3
python -m pytest tests/test_gpu_float32.py -x -q -m gpu 2>&1 | tail -8
export FDJAC_TEST_SWITCHES=1
for rep in 1 2; do
for v in 1 0; do
FDJAC_S5_WAVE4=$v python bench.py --config c3 --dtype f32 --soak-seconds 0 --no-cpu-baseline --no-plain-handover > gpurun_out/r06_s3/c3f32_w$v.json 2>gpurun_out/r06_s3/c3f32_w$v.err
python - gpurun_out/r06_s3/c3f32_w$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("wave4=%s ms/step %.4f kernel %s avg %.4f ms median %.4f frac %.3f check %s" % (sys.argv[2], d["ms_per_step"], r["kernel"][:40], r["avg_launch_ms"], r["median_launch_ms"], r["frac"], d.get("result_check", {}).get("ok")))
PY
done
done
