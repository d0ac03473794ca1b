#!/bin/bash
# A/B of the reduction's load form on one box: configs 4, 3, 2 (ms per Jacobian; the eps stage by events)
cd "$(dirname "$0")/.."
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-3s ms/step %.4f  eps stage %.4f  store %.4f' % ('$1', '$2', d['ms_per_step'], d['stages_ms'].get('eps',0), d['stages_ms'].get('decompress',0)))
"; }
for rep in 1 2 3; do
for c in c4 c3 c2; do
  python bench.py --config $c --soak-seconds 0 --no-cpu-baseline --no-plain-handover 2>/dev/null | show new $c
  scripts/with_variant.sh oldload python bench.py --config $c --soak-seconds 0 --no-cpu-baseline --no-plain-handover 2>/dev/null | show old $c
done
done
