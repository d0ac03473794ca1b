#!/bin/bash
# A/B of row-window kernel variants on the headline config (separate processes: placement noise is +-3 %; run twice)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in "" "FDJAC_WIN_TILE=1024" "FDJAC_WIN_TILE=512" "FDJAC_WIN_PERIODIC=0"; do
  echo -n "rep $rep [$v] "
  env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --soak-seconds 0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms/step  dec %.1f us  f %.1f  eps %.1f  frac %.3f' % (r['ms_per_step'], r['roofline']['avg_launch_ms']*1e3, r['stages_ms']['f']*1e3, r['stages_ms']['eps']*1e3, r['roofline']['frac']))"
done; done
